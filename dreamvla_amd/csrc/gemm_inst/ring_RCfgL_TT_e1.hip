#include "../gemm_impl.h"
namespace dvla_gemm { template void launch_ring_one<RCfgL, true, true, 1>(const GemmKArgs&, int, hipStream_t); }
