#include "../gemm_impl.h"
namespace dvla_gemm { template void launch_ring_one<RCfgL64, false, false>(const GemmKArgs&, int, hipStream_t); }
