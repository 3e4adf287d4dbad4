#include "../gemm_impl.h"
namespace dvla_gemm { template void launch_ring_one<RCfgM64, false, true, 3>(const GemmKArgs&, int, hipStream_t); }
