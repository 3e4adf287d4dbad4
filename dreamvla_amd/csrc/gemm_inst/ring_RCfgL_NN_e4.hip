#include "../gemm_impl.h"
namespace dvla_gemm { template void launch_ring_one<RCfgL, false, false, 4>(const GemmKArgs&, int, hipStream_t); }
