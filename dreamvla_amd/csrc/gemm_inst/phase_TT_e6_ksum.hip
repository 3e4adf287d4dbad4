// the phase kernel's fp32 class WITH the k-sum code (gemm_phase.h DBG & 8192): TT layout = the weight gradients that also
// produce a bias gradient; 8320 = the same with a partial last K-tile (DBG & 128)
#include "../gemm_phase.h"
namespace dvla_gemm {
template void launch_phase_one<true, true, 6, 8192>(const GemmKArgs&, int, hipStream_t);
template void launch_phase_one<true, true, 6, 8320>(const GemmKArgs&, int, hipStream_t);
}
