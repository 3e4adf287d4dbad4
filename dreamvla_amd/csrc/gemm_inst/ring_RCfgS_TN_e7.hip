#include "../gemm_impl.h"
namespace dvla_gemm { template void launch_ring_one<RCfgS, true, false, 7>(const GemmKArgs&, int, hipStream_t); }
