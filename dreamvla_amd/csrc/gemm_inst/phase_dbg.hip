// ablation builds of the phase kernel (NT layout, plain epilogue; variants 81..86 -> DBG bits), timing only
#include "../gemm_phase.h"
namespace dvla_gemm {
template void launch_phase_one<false, false, 0, 1>(const GemmKArgs&, int, hipStream_t);
template void launch_phase_one<false, false, 0, 3>(const GemmKArgs&, int, hipStream_t);
template void launch_phase_one<false, false, 0, 4>(const GemmKArgs&, int, hipStream_t);
template void launch_phase_one<false, false, 0, 16>(const GemmKArgs&, int, hipStream_t);
template void launch_phase_one<false, false, 0, 32>(const GemmKArgs&, int, hipStream_t);
template void launch_phase_one<false, false, 0, 64>(const GemmKArgs&, int, hipStream_t);
}
