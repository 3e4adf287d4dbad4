#include "../gemm_phase.h"
// round-6 measurement builds (gemm.hip variants 44 / 45 / 46 / 47 / 88; NN layout, plain epilogue)
namespace dvla_gemm {
template void launch_phase_one<false, false, 0, 16384>(const GemmKArgs&, int, hipStream_t);
template void launch_phase_one<false, false, 0, 32768>(const GemmKArgs&, int, hipStream_t);
template void launch_phase_one<false, false, 0, 81920>(const GemmKArgs&, int, hipStream_t);
template void launch_phase_one<false, false, 0, 212992>(const GemmKArgs&, int, hipStream_t);
}
