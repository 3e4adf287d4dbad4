#include "../gemm_impl.h"
namespace dvla_gemm { template void launch_ring_one<RCfgS, false, false, 6>(const GemmKArgs&, int, hipStream_t); }
