/* dvla_cmp.h -- C ABI of libdvla_cmp.so: the hipBLASLt COMPARATOR.  Not part of the product: libdvla_hip.so (include/dvla.h)
 * links no vendor GEMM library and nothing in dreamvla_amd's forward / backward / optimizer path loads this library.  It
 * exists so that tests/gpu_perf.py and `bench.py --library-yardstick` can time the vendor library on the very same
 * parameter block, on the same box, next to the hand-written kernels (the per-shape gap list in profiles/).
 */
#ifndef DVLA_CMP_H_
#define DVLA_CMP_H_
#include "dvla.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Library GEMM (hipBLASLt) on the same parameter block: C = A . B^T (+ bias[n], the library's own bias epilogue)
 * (C += when accumulate, fp32 C, no bias).  A *comparator* for the plain GEMMs of the step -- weight
 * gradients dW = X^T dY and input gradients dX = dY W of nn.Linear / Conv1D (autograd of models/gpt2.py:160,172-173,
 * 296-301, timm Block Linear layers) and the bias-only projections (qkv: timm Attention, gpt2.py c_attn) -- never for a
 * GEMM fused with an activation / act' / dropout / residual / pre-activation store: any act / preact / dact / dropout /
 * residual / split_k > 1 in `p` returns DVLA_ERR_UNSUPPORTED (-3), as does a problem the library has no kernel for within
 * `workspace_bytes`.  `workspace` is caller-owned device memory the library may use for its own split-K (may be NULL
 * with 0 bytes). */
int dvla_gemm_library_bf16(const dvla_gemm_params* p, void* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
