// gemm_library.hip -- plain (optionally + bias vector) bf16 GEMM through hipBLASLt, behind the same dvla_gemm_params as the hand-written
// kernels (include/dvla.h).  It exists as a *tuner candidate and comparator* for the plain GEMMs of the step (weight
// gradients dW = X^T dY, the plain input gradients dX = dY W): dreamvla_amd.ops.GemmTuner times it in turn with the
// hand-written configurations on the real calls and keeps whichever is fastest per problem key, and bench.py reports
// how much of the step's GEMM time each side won -- the per-shape gap list for the next kernel round.  The library's
// own bias epilogue (C = A.B^T + bias[n]) is accepted too -- a library GEMM as plain as they come: the qkv projections.
// Every GEMM with an activation / act' / dropout / residual / pre-activation store / split-K is only ever run by gemm.hip.
//
// Row-major C[M,N] = A[M,K] . B[N,K]^T is handed to the column-major library as C^T[N,M] = op(B) . op(A).
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>

#include <cstdint>
#include <mutex>
#include <unordered_map>

#include "../../../include/dvla_cmp.h"
#include "../common.h"

namespace {

struct LibKey {
  int64_t M, N, K, lda, ldb, ldc;
  int a_trans, b_trans, c_f32;
  int bias;   // 0 none, 1 bf16 vector, 2 fp32 vector
  bool operator==(const LibKey& o) const {
    return M == o.M && N == o.N && K == o.K && lda == o.lda && ldb == o.ldb && ldc == o.ldc && a_trans == o.a_trans &&
           b_trans == o.b_trans && c_f32 == o.c_f32 && bias == o.bias;
  }
};
struct LibKeyHash {
  size_t operator()(const LibKey& k) const {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h ^= v; h *= 1099511628211ull; };
    mix((uint64_t)k.M); mix((uint64_t)k.N); mix((uint64_t)k.K); mix((uint64_t)k.lda); mix((uint64_t)k.ldb);
    mix((uint64_t)k.ldc); mix((uint64_t)(k.a_trans * 4 + k.b_trans * 2 + k.c_f32 + 8 * k.bias));
    return (size_t)h;
  }
};
struct LibPlan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t first = nullptr, second = nullptr, out = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t workspace = 0;
  bool ok = false;
};

std::mutex g_mu;
hipblasLtHandle_t g_handle = nullptr;
bool g_handle_failed = false;
std::unordered_map<LibKey, LibPlan, LibKeyHash> g_plans;

bool make_plan(const LibKey& k, size_t max_ws, const void* bias, LibPlan& p) {
  const hipblasOperation_t op_first = k.b_trans ? HIPBLAS_OP_N : HIPBLAS_OP_T;    // our B, stored (N,K) or (K,N) row-major
  const hipblasOperation_t op_second = k.a_trans ? HIPBLAS_OP_T : HIPBLAS_OP_N;   // our A, stored (M,K) or (K,M) row-major
  if (hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return false;
  int32_t ta = (int32_t)op_first, tb = (int32_t)op_second;
  if (hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)) != HIPBLAS_STATUS_SUCCESS) return false;
  if (hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)) != HIPBLAS_STATUS_SUCCESS) return false;
  if (k.bias) {   // bias[n] along the rows of the column-major D = C^T: exactly the library's EPILOGUE_BIAS
    uint32_t epi = (uint32_t)HIPBLASLT_EPILOGUE_BIAS;
    int32_t bdt = (int32_t)(k.bias == 2 ? HIP_R_32F : HIP_R_16BF);
    if (hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)) != HIPBLAS_STATUS_SUCCESS) return false;
    if (hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bdt, sizeof(bdt)) != HIPBLAS_STATUS_SUCCESS) return false;
    if (hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) != HIPBLAS_STATUS_SUCCESS) return false;
  }
  // column-major view of a row-major (r, c) array with row stride ld is a (c, r) matrix with leading dimension ld
  const uint64_t f_rows = k.b_trans ? (uint64_t)k.N : (uint64_t)k.K, f_cols = k.b_trans ? (uint64_t)k.K : (uint64_t)k.N;
  const uint64_t s_rows = k.a_trans ? (uint64_t)k.M : (uint64_t)k.K, s_cols = k.a_trans ? (uint64_t)k.K : (uint64_t)k.M;
  if (hipblasLtMatrixLayoutCreate(&p.first, HIP_R_16BF, f_rows, f_cols, k.ldb) != HIPBLAS_STATUS_SUCCESS) return false;
  if (hipblasLtMatrixLayoutCreate(&p.second, HIP_R_16BF, s_rows, s_cols, k.lda) != HIPBLAS_STATUS_SUCCESS) return false;
  if (hipblasLtMatrixLayoutCreate(&p.out, k.c_f32 ? HIP_R_32F : HIP_R_16BF, (uint64_t)k.N, (uint64_t)k.M, k.ldc) !=
      HIPBLAS_STATUS_SUCCESS)
    return false;
  hipblasLtMatmulPreference_t pref = nullptr;
  if (hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS) return false;
  uint64_t ws = (uint64_t)max_ws;
  (void)hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws));
  hipblasLtMatmulHeuristicResult_t res[1];
  int found = 0;
  hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(g_handle, p.desc, p.first, p.second, p.out, p.out, pref, 1, res, &found);
  hipblasLtMatmulPreferenceDestroy(pref);
  if (st != HIPBLAS_STATUS_SUCCESS || found < 1 || res[0].workspaceSize > max_ws) return false;
  p.algo = res[0].algo;
  p.workspace = res[0].workspaceSize;
  p.ok = true;
  return true;
}

}  // namespace

extern "C" int dvla_gemm_library_bf16(const dvla_gemm_params* q, void* workspace, int64_t workspace_bytes, void* stream_) {
  if (!q || !q->A || !q->B || !q->C) return DVLA_ERR_ARG;
  if (q->M <= 0 || q->N <= 0 || q->K <= 0) return DVLA_ERR_ARG;
  if (q->act || q->preact || q->dact_aux || q->residual || q->dropout_p > 0.f || q->split_k > 1)
    return DVLA_ERR_UNSUPPORTED;   // fused epilogues and split-K belong to dvla_gemm_bf16
  if (q->bias && q->accumulate) return DVLA_ERR_UNSUPPORTED;
  if (q->accumulate && q->c_dtype != DVLA_DT_F32) return DVLA_ERR_ARG;
  if (workspace_bytes < 0 || (workspace_bytes > 0 && !workspace)) return DVLA_ERR_ARG;
  LibKey key{q->M, q->N, q->K, q->lda, q->ldb, q->ldc, q->a_trans ? 1 : 0, q->b_trans ? 1 : 0,
             q->c_dtype == DVLA_DT_F32 ? 1 : 0, q->bias ? (q->bias_dtype == DVLA_DT_F32 ? 2 : 1) : 0};
  LibPlan plan;
  std::lock_guard<std::mutex> lock(g_mu);   // held through the enqueue: the cached descriptor carries this call's bias pointer
  {
    if (!g_handle) {
      if (g_handle_failed) return DVLA_ERR_UNSUPPORTED;
      if (hipblasLtCreate(&g_handle) != HIPBLAS_STATUS_SUCCESS) {
        g_handle = nullptr;
        g_handle_failed = true;
        return DVLA_ERR_UNSUPPORTED;
      }
    }
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
      LibPlan p;
      make_plan(key, (size_t)workspace_bytes, q->bias, p);   // a failed plan is cached too (ok = false)
      it = g_plans.emplace(key, p).first;
    }
    plan = it->second;
  }
  if (!plan.ok || plan.workspace > (size_t)workspace_bytes) return DVLA_ERR_UNSUPPORTED;
  if (q->bias) {
    const void* bp = q->bias;
    if (hipblasLtMatmulDescSetAttribute(plan.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bp, sizeof(bp)) != HIPBLAS_STATUS_SUCCESS)
      return DVLA_ERR_LAUNCH;
  }
  const float alpha = 1.0f, beta = q->accumulate ? 1.0f : 0.0f;
  hipblasStatus_t st = hipblasLtMatmul(g_handle, plan.desc, &alpha, q->B, plan.first, q->A, plan.second, &beta, q->C,
                                       plan.out, q->C, plan.out, &plan.algo, workspace, plan.workspace,
                                       reinterpret_cast<hipStream_t>(stream_));
  return st == HIPBLAS_STATUS_SUCCESS ? DVLA_OK : DVLA_ERR_LAUNCH;
}
