#include "../gemm_impl.h"
namespace dvla_gemm { template void launch_ring_one<RCfgM64, true, true, 5>(const GemmKArgs&, int, hipStream_t); }
