// the phase kernel with a partial last K-tile (gemm_phase.h DBG & 128): TT layout, fp32 class -- the weight gradients whose
// contraction length is not a multiple of 64 (the trunk's 20 832 = 651 x 32 tokens)
#include "../gemm_phase.h"
namespace dvla_gemm { template void launch_phase_one<true, true, 6, 128>(const GemmKArgs&, int, hipStream_t); }
