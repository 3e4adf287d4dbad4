#include "../gemm_phase.h"
namespace dvla_gemm { template void launch_phase_one<true, false, 7, 0>(const GemmKArgs&, int, hipStream_t); }
