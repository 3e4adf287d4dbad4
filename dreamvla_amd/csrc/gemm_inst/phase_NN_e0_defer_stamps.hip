#include "../gemm_phase.h"
namespace dvla_gemm { template void launch_phase_one<false, false, 0, 16448>(const GemmKArgs&, int, hipStream_t); }
