// measurement variants of the phase kernel (gemm.hip variants 40..43: the round-4 main loop for A/B, placements; timing / stamps only)
#include "../gemm_phase.h"
namespace dvla_gemm {
template void launch_phase_one<false, false, 0, 7424>(const GemmKArgs&, int, hipStream_t);
template void launch_phase_one<false, false, 0, 7488>(const GemmKArgs&, int, hipStream_t);
template void launch_phase_one<false, false, 0, 3328>(const GemmKArgs&, int, hipStream_t);
template void launch_phase_one<false, false, 0, 256>(const GemmKArgs&, int, hipStream_t);
}
