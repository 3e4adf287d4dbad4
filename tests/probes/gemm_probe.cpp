// gemm_probe.cpp -- standalone (no torch) timing + correctness probe of dvla_gemm_bf16 through the C ABI.
// Measurement infrastructure, not product.  Build (tests/probes/build_probes.sh):
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tests/probes/gemm_probe.cpp -o build/gemm_probe -Ldreamvla_amd -ldvla_hip \
//         -Wl,-rpath,'$ORIGIN/../dreamvla_amd'
// Usage: gemm_probe [--variants 0,2,4,...] [--iters N] [--check-only] [--cases model|small|big]
// For every case: a small-shape correctness pass of each variant against a naive fp32 device reference that restates the
// epilogue of include/dvla.h (exact erf / tanh), then paired timing rounds (variants interleaved in one process, median
// and min reported -- cdna guide section 5.4 rule 24).  One JSON object per line on stdout.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/dvla.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef unsigned short bf16_t;
__host__ __device__ static inline float bf2f(bf16_t h) { union { uint32_t u; float f; } v; v.u = ((uint32_t)h) << 16; return v.f; }
__host__ __device__ static inline bf16_t f2bf(float f) {
  union { uint32_t u; float f; } v; v.f = f;
  if ((v.u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((v.u >> 16) | 0x40);
  const uint32_t r = 0x7fffu + ((v.u >> 16) & 1u);
  return (bf16_t)((v.u + r) >> 16);
}
__host__ __device__ static inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void fill_kernel(bf16_t* p, int64_t n, uint32_t seed, float scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t h = hash32((uint32_t)i * 2654435761u + seed);
  const float u = ((h >> 8) * (1.0f / 16777216.0f)) * 2.f - 1.f;   // uniform [-1, 1): full-range random data (rule 25)
  p[i] = f2bf(u * scale);
}

__device__ float ref_act(float x, int act) {
  switch (act) {
    case 1: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
    case 2: return 0.5f * x * (1.f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
    case 3: return x > 0.f ? x : 0.f;
    case 4: return x / (1.f + expf(-x));
    case 5: return x / (1.f + expf(-1.702f * x));
    default: return x;
  }
}
__device__ float ref_dact(float x, int act) {
  switch (act) {
    case 1: return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
    case 2: {
      const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x), t = tanhf(u);
      return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * 0.7978845608028654f * (1.f + 3.f * 0.044715f * x * x);
    }
    case 3: return x > 0.f ? 1.f : 0.f;
    default: return 1.f;
  }
}

// naive reference: one thread per output element, fp32 accumulate in k order, epilogue as include/dvla.h states it
__global__ void ref_kernel(dvla_gemm_params p, float drop_scale, uint32_t drop_thr, float* out_f32, float* pre_f32) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= p.M * p.N) return;
  const int64_t m = idx / p.N, n = idx % p.N;
  const bf16_t* A = (const bf16_t*)p.A; const bf16_t* B = (const bf16_t*)p.B;
  float s = 0.f;
  for (int64_t k = 0; k < p.K; ++k) {
    const float a = bf2f(p.a_trans ? A[k * p.lda + m] : A[m * p.lda + k]);
    const float b = bf2f(p.b_trans ? B[k * p.ldb + n] : B[n * p.ldb + k]);
    s = fmaf(a, b, s);
  }
  if (p.bias) s += bf2f(((const bf16_t*)p.bias)[n]);
  if (p.preact) { pre_f32[idx] = s; s = bf2f(f2bf(s)); }
  s = ref_act(s, p.act);
  if (p.dropout_p > 0.f) {
    const uint32_t rowkey = hash32((uint32_t)m ^ p.seed_hi) + p.seed_lo;
    const uint32_t h = hash32(rowkey + (uint32_t)n * 0x9E3779B9u);
    s = (h >= drop_thr) ? s * drop_scale : 0.f;
  }
  if (p.c_dtype == DVLA_DT_BF16 && (p.dact_aux || p.residual)) s = bf2f(f2bf(s));
  if (p.dact_aux) s *= ref_dact(bf2f(((const bf16_t*)p.dact_aux)[m * p.ld_dact + n]), p.dact);
  if (p.residual) {
    const int64_t rr = p.res_rows > 0 ? m % p.res_rows : m;
    s += bf2f(((const bf16_t*)p.residual)[rr * p.ld_res + n]);
  }
  out_f32[idx] = s;
}

struct Case {
  std::string name; int64_t M, N, K; int at, bt; std::string epi; int split_k;
};

struct Buf { void* p = nullptr; size_t bytes = 0; };
static Buf dalloc(size_t bytes) { Buf b; b.bytes = bytes; CK(hipMalloc(&b.p, bytes ? bytes : 16)); return b; }
static void fill(Buf& b, uint32_t seed, float scale) {
  const int64_t n = (int64_t)(b.bytes / 2);
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (bf16_t*)b.p, n, seed, scale);
}

struct Problem {
  dvla_gemm_params p; Buf A, B, C, bias, preact, dact, res, ws, ksum, ksum_ws;
  void release() { for (Buf* b : {&A, &B, &C, &bias, &preact, &dact, &res, &ws, &ksum, &ksum_ws}) if (b->p) { (void)hipFree(b->p); b->p = nullptr; } }
};

// DVLA_PROBE_PAD_C / DVLA_PROBE_PAD_AB (elements): leading dimensions of C (and the epilogue operands) / of A and B are padded
// by that much -- does a power-of-two row stride cost anything (channel camping)?  Timing only: the checks assume ld = width.
static int64_t env_pad(const char* name) { const char* e = getenv(name); return e ? atoll(e) : 0; }
static Problem make_problem(const Case& c, int64_t M) {
  Problem q; memset(&q.p, 0, sizeof(q.p));
  const int64_t K = c.K, padC = env_pad("DVLA_PROBE_PAD_C"), padAB = env_pad("DVLA_PROBE_PAD_AB");
  const int64_t N = c.N + padC;   // allocation / leading-dimension width of the C-shaped operands (p.N stays c.N)
  const int64_t lda = (c.at ? M : K) + padAB, ldb = (c.bt ? c.N : K) + padAB;
  q.A = dalloc((size_t)(c.at ? K : M) * lda * 2); q.B = dalloc((size_t)(c.bt ? K : c.N) * ldb * 2);
  fill(q.A, 1u, 1.0f); fill(q.B, 2u, 0.06f);
  dvla_gemm_params& p = q.p;
  p.A = q.A.p; p.lda = lda; p.a_trans = c.at;
  p.B = q.B.p; p.ldb = ldb; p.b_trans = c.bt;
  p.M = M; p.N = c.N; p.K = K; p.split_k = c.split_k;
  const bool f32 = c.epi == "f32";
  q.C = dalloc((size_t)M * N * (f32 ? 4 : 2));
  p.C = q.C.p; p.ldc = N; p.c_dtype = f32 ? DVLA_DT_F32 : DVLA_DT_BF16;
  auto need_bias = [&]() { q.bias = dalloc((size_t)N * 2); fill(q.bias, 3u, 0.5f); p.bias = q.bias.p; p.bias_dtype = DVLA_DT_BF16; };
  if (c.epi == "bias") need_bias();
  if (c.epi == "gelu_erf") { need_bias(); p.act = 1; }
  if (c.epi == "relu") { need_bias(); p.act = 3; }
  if (c.epi == "silu_res") { need_bias(); p.act = 4; q.res = dalloc((size_t)M * N * 2); fill(q.res, 4u, 1.0f); p.residual = q.res.p; p.ld_res = N; }
  if (c.epi == "gelu_tanh_preact" || c.epi == "gelu_erf_preact" || c.epi == "none_preact" || c.epi == "relu_preact") {
    need_bias(); p.act = c.epi == "gelu_tanh_preact" ? 2 : (c.epi == "gelu_erf_preact" ? 1 : (c.epi == "relu_preact" ? 3 : 0));
    q.preact = dalloc((size_t)M * N * 2); p.preact = q.preact.p; p.ld_preact = N;
  }
  if (c.epi == "drop_res" || c.epi == "res") {
    need_bias();
    q.res = dalloc((size_t)M * N * 2); fill(q.res, 4u, 1.0f); p.residual = q.res.p; p.ld_res = N;
    if (c.epi == "drop_res") { p.dropout_p = 0.1f; p.seed_lo = 12345u; p.seed_hi = 777u; }
  }
  if (c.epi == "dact_tanh" || c.epi == "dact_erf") {
    q.dact = dalloc((size_t)M * N * 2); fill(q.dact, 5u, 2.0f); p.dact_aux = q.dact.p; p.ld_dact = N; p.dact = c.epi == "dact_tanh" ? 2 : 1;
  }
  if (c.split_k > 1) { q.ws = dalloc((size_t)c.split_k * M * N * 4); p.workspace = q.ws.p; }
  // DVLA_PROBE_KSUM=a|b: the launch also produces the k-sums of that operand (timing of the fused bias gradient; fp32 class only)
  if (const char* ks = getenv("DVLA_PROBE_KSUM")) {
    if (f32 && (ks[0] == 'a' || ks[0] == 'b')) {
      const int64_t len = ks[0] == 'a' ? M : c.N;
      q.ksum = dalloc((size_t)len * 4); q.ksum_ws = dalloc((size_t)dvla_gemm_ksum_partial_rows(c.split_k) * len * 4);
      p.ksum = q.ksum.p; p.ksum_dtype = DVLA_DT_F32; p.ksum_operand = ks[0] == 'a' ? 1 : 2; p.ksum_workspace = (float*)q.ksum_ws.p;
    }
  }
  return q;
}

static std::vector<int> parse_list(const char* s) {
  std::vector<int> v; std::string t(s); size_t pos = 0;
  while (pos < t.size()) { size_t e = t.find(',', pos); if (e == std::string::npos) e = t.size(); v.push_back(atoi(t.substr(pos, e - pos).c_str())); pos = e + 1; }
  return v;
}

// returns {max rel err (vs |ref| + atol scale), fraction of elements off by > tol}
// --hog N: N workgroups that each take a whole CU's LDS and spin for `cycles` -- what a communication kernel on another
// stream does to the persistent GEMM schedules (a CU it sits on cannot host a 160-KiB workgroup)
__global__ void hog_kernel(long long cycles, int* sink) {
  extern __shared__ char hog_smem[];
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  while ((long long)__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(16);
  if (cycles < 0) sink[0] = hog_smem[threadIdx.x];
}

static void check_variant(const Case& c, int variant, int64_t Mchk) {
  Problem q = make_problem(c, Mchk);
  const int64_t MN = Mchk * c.N;
  Buf ref = dalloc((size_t)MN * 4), pre = dalloc((size_t)MN * 4);
  const float drop_scale = q.p.dropout_p > 0.f ? 1.f / (1.f - q.p.dropout_p) : 1.f;
  const double thr = (double)q.p.dropout_p * 4294967296.0;
  hipLaunchKernelGGL(ref_kernel, dim3((unsigned)((MN + 255) / 256)), dim3(256), 0, 0, q.p, drop_scale, (uint32_t)thr, (float*)ref.p, (float*)pre.p);
  CK(hipMemset(q.C.p, 0xff, q.C.bytes));
  if (q.preact.p) CK(hipMemset(q.preact.p, 0xff, q.preact.bytes));
  dvla_set_gemm_variant(variant);
  const int rc = dvla_gemm_bf16(&q.p, nullptr);
  dvla_set_gemm_variant(0);
  CK(hipDeviceSynchronize());
  std::vector<float> h_ref(MN), h_pre(MN);
  CK(hipMemcpy(h_ref.data(), ref.p, MN * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h_pre.data(), pre.p, MN * 4, hipMemcpyDeviceToHost));
  const bool f32 = q.p.c_dtype == DVLA_DT_F32;
  std::vector<char> h_c(q.C.bytes);
  CK(hipMemcpy(h_c.data(), q.C.p, q.C.bytes, hipMemcpyDeviceToHost));
  std::vector<bf16_t> h_p;
  if (q.preact.p) { h_p.resize(MN); CK(hipMemcpy(h_p.data(), q.preact.p, MN * 2, hipMemcpyDeviceToHost)); }
  double num = 0, den = 0, max_abs = 0, ref_max = 0; int64_t bad = 0, badp = 0, first_bad = -1;
  for (int64_t i = 0; i < MN; ++i) {
    const float got = f32 ? ((const float*)h_c.data())[i] : bf2f(((const bf16_t*)h_c.data())[i]);
    const float want = f32 ? h_ref[i] : bf2f(f2bf(h_ref[i]));
    const double d = fabs((double)got - (double)want);
    num += d * d; den += (double)want * want; max_abs = std::max(max_abs, d); ref_max = std::max(ref_max, (double)fabs(want));
    // element-wise criterion: 2 bf16 ulps of the value (one rounding flip of an intermediate + the final rounding) plus an
    // absolute floor for cancelling sums (branch + residual); a misplaced element is off by O(1)
    const double tol = (f32 ? 2e-4 * fabs(want) + 2e-4 : 0.01 * fabs(want) + 0.012);
    if (!(d <= tol)) { ++bad; if (first_bad < 0) first_bad = i; }
    if (!h_p.empty()) {
      const float gp = bf2f(h_p[i]), wp = bf2f(f2bf(h_pre[i]));
      if (!(fabs(gp - wp) <= 0.0157 * fabs(wp) + 1e-3)) ++badp;
    }
  }
  printf("{\"check\": \"%s\", \"epi\": \"%s\", \"M\": %lld, \"N\": %lld, \"K\": %lld, \"at\": %d, \"bt\": %d, \"split_k\": %d, \"variant\": %d, \"rc\": %d, "
         "\"rel_l2\": %.3e, \"max_abs\": %.3e, \"ref_absmax\": %.3e, \"bad\": %lld, \"bad_preact\": %lld, \"first_bad\": %lld, \"ok\": %s}\n",
         c.name.c_str(), c.epi.c_str(), (long long)Mchk, (long long)c.N, (long long)c.K, c.at, c.bt, c.split_k, variant, rc,
         sqrt(num / (den > 0 ? den : 1)), max_abs, ref_max, (long long)bad, (long long)badp, (long long)first_bad,
         (rc == 0 && bad == 0 && badp == 0) ? "true" : "false");
  fflush(stdout);
  (void)hipFree(ref.p); (void)hipFree(pre.p); q.release();
}

int main(int argc, char** argv) {
  std::vector<int> variants = {0, 2, 4, 5, 6};
  int iters = 7, rounds = 3; bool check_only = false, no_check = false, full_check = false, stamps = false; int64_t stamps_k = 1024; std::string which = "model";
  std::vector<int> stamp_variants = {89};
  int hog = 0, only = -1;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--variants") && i + 1 < argc) variants = parse_list(argv[++i]);
    else if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--rounds") && i + 1 < argc) rounds = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--check-only")) check_only = true;
    else if (!strcmp(argv[i], "--no-check")) no_check = true;
    else if (!strcmp(argv[i], "--full-check")) full_check = true;   // additionally: every variant at the case's FULL size, twice
    else if (!strcmp(argv[i], "--stamps")) { stamps = true; if (i + 1 < argc && argv[i + 1][0] != '-') stamps_k = atoll(argv[++i]); }
    else if (!strcmp(argv[i], "--stamp-variants") && i + 1 < argc) stamp_variants = parse_list(argv[++i]);
    else if (!strcmp(argv[i], "--hog") && i + 1 < argc) hog = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--cases") && i + 1 < argc) which = argv[++i];
    else if (!strcmp(argv[i], "--only") && i + 1 < argc) only = atoi(argv[++i]);     // run case number `only` of the group alone
  }
  if (stamps) {   // timeline of the phase kernel (variants 89 / 97 / 98): s_memtime at the 8 segment edges of the first 32 K-tiles, waves 0 and 4
    Case c{"stamps", 20832, 4096, stamps_k, 0, 0, "plain", 1};
    Problem q = make_problem(c, c.M);
    q.ws = dalloc((2 * 256 + 2 * 64 + 32) * 8); q.p.workspace = q.ws.p;
    printf("# s_memtime ticks (shader cycles, MI355X_MICROARCH.md); segments of K-tiles 3..6: LOAD1 | wait+barrier | MFMA1 | barrier | LOAD2 | wait+barrier | MFMA2(+vmcnt) ; period\n");
    for (int v : stamp_variants) {
      CK(hipMemset(q.ws.p, 0, q.ws.bytes));
      dvla_set_gemm_variant(v);
      for (int it = 0; it < 3; ++it) (void)dvla_gemm_bf16(&q.p, nullptr);
      if (hog > 0) {   // --hog N --stamps: the stamped launch shares the chip with N LDS-hogging workgroups, i.e. runs on 256 - N CUs
        hipStream_t hs; int* sink; CK(hipStreamCreateWithFlags(&hs, hipStreamNonBlocking)); CK(hipMalloc(&sink, 64));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(hog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        CK(hipDeviceSynchronize());
        CK(hipMemset(q.ws.p, 0, q.ws.bytes));
        hipLaunchKernelGGL(hog_kernel, dim3(hog), dim3(256), 96 * 1024, hs, 12000000LL, sink);
        usleep(500);
        (void)dvla_gemm_bf16(&q.p, nullptr);
      }
      dvla_set_gemm_variant(0);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> h(512 + 128 + 32);
      CK(hipMemcpy(h.data(), q.ws.p, (512 + 128 + 32) * 8, hipMemcpyDeviceToHost));
      if (getenv("DVLA_STAMPS_ALL")) {
        for (int g = 0; g < 2; ++g)
          for (int tile = 0; tile < 4; ++tile) {
            const unsigned long long* e = &h[512 + g * 64 + tile * 4];
            printf("variant %d group %d tile %d boundary: re-align %5lld | epilogue code (conversion + store issue) %6lld | to next K loop %5lld   total %6lld\n", v, g, tile,
                   (long long)(e[1] - e[0]), (long long)(e[2] - e[1]), (long long)(e[3] - e[2]), (long long)(e[3] - e[0]));
          }
      }
      if (getenv("DVLA_STAMPS_ALL")) {   // inside tile 1's epilogue: per 32-row slab, conversion + transposition | store issue
        for (int g = 0; g < 2; ++g) {
          const unsigned long long* e = &h[640 + g * 16];
          printf("variant %d group %d tile 1 epilogue slabs (convert | stores):", v, g);
          for (int j = 0; j < 4; ++j) printf("  %5lld | %5lld", (long long)(e[1 + 2 * j] - e[2 * j]), (long long)(e[2 + 2 * j] - e[1 + 2 * j]));
          printf("   (entry at +%lld after re-align)\n", (long long)(e[0] - h[512 + g * 64 + 4 + 1]));
        }
      }
      if (getenv("DVLA_STAMPS_ALL")) {   // every recorded K-tile: start (relative to K-tile 0), length, gap to the next one (tile boundaries show up as gaps)
        for (int g = 0; g < 2; ++g)
          for (int kt = 0; kt < 31; ++kt) {
            const unsigned long long* e = &h[g * 256 + kt * 8];
            printf("variant %d group %d K-tile %2d: start %7lld  length %5lld  gap-to-next %6lld\n", v, g, kt, (long long)(e[0] - h[g * 256]),
                   (long long)(e[7] - e[0]), (long long)(e[8] - e[7]));
          }
        continue;
      }
      for (int g = 0; g < 2; ++g)
        for (int kt = 3; kt < 7; ++kt) {
          const unsigned long long* e = &h[g * 256 + kt * 8];
          printf("variant %d group %d K-tile %d: LOAD1 %4lld | wait+barrier %4lld | MFMA1 %4lld | barrier %4lld | LOAD2 %4lld | wait+barrier %4lld | MFMA2 %4lld   period %lld\n",
                 v, g, kt, (long long)(e[1] - e[0]), (long long)(e[2] - e[1]), (long long)(e[3] - e[2]), (long long)(e[4] - e[3]),
                 (long long)(e[5] - e[4]), (long long)(e[6] - e[5]), (long long)(e[7] - e[6]), (long long)(e[8] - e[0]));
        }
    }
    q.release();
    return 0;
  }
  std::vector<Case> cases;
  const int sk_override = getenv("DVLA_PROBE_SK") ? atoi(getenv("DVLA_PROBE_SK")) : 0;   // split-K cases: another slice count
  if (which == "model" || which == "all") {
    cases = {
      {"trunk fc1 fwd", 20832, 4096, 1024, 0, 1, "gelu_tanh_preact", 1},
      {"trunk fc1 dX", 20832, 4096, 1024, 0, 0, "dact_tanh", 1},          // dY[M,1024]... stated as M x N=4096 x K=1024 in the breakdown
      {"trunk fc2 fwd", 20832, 1024, 4096, 0, 1, "drop_res", 1},
      {"trunk fc2 dX", 20832, 1024, 4096, 0, 0, "plain", 1},
      {"trunk c_attn fwd", 20832, 3072, 1024, 0, 1, "bias", 1},
      {"trunk c_proj fwd", 20832, 1024, 1024, 0, 1, "drop_res", 1},
      {"trunk c_attn dX", 20832, 1024, 3072, 0, 0, "plain", 1},
      {"vit fc1", 88256, 3072, 768, 0, 0, "gelu_erf", 1},
      {"vit fc2", 88256, 768, 3072, 0, 0, "res", 1},
      {"vit qkv", 88256, 2304, 768, 0, 0, "bias", 1},
      {"vit proj", 88256, 768, 768, 0, 0, "res", 1},
      {"dec fc1 fwd", 91840, 4096, 1024, 0, 0, "gelu_erf_preact", 1},
      {"dec fc1 dX", 91840, 4096, 1024, 0, 1, "dact_erf", 1},
      {"dW fc1", 1024, 4096, 20832, 1, 1, "f32", 4},
      {"dW fc2", 4096, 1024, 20832, 1, 1, "f32", 4},
      {"dW c_attn", 1024, 3072, 20832, 1, 1, "f32", 6},
      {"dW c_proj", 1024, 1024, 20832, 1, 1, "f32", 10},
      {"square", 8192, 8192, 8192, 0, 0, "plain", 1},
      {"generic relu", 20832, 1024, 1024, 0, 0, "relu", 1},
      {"generic silu+res", 20832, 1024, 1024, 0, 1, "silu_res", 1},
    };
  } else if (which == "dw") {   // the weight-gradient problems (fp32 class, split-K): DVLA_PROBE_KSUM adds the bias gradient
    cases = {
      {"dW fc1", 1024, 4096, 20832, 1, 1, "f32", 4},
      {"dW fc2", 4096, 1024, 20832, 1, 1, "f32", 4},
      {"dW c_attn", 1024, 3072, 20832, 1, 1, "f32", 5},
      {"dW c_proj", 1024, 1024, 20832, 1, 1, "f32", 16},
      {"dW dec fc1", 4096, 1024, 91840, 1, 1, "f32", 4},
      {"dW dec fc2", 1024, 4096, 91840, 1, 1, "f32", 4},
    };
  } else if (which == "plain") {
    cases = {
      {"NT plain", 20832, 4096, 1024, 0, 0, "plain", 1},
      {"NT plain", 20832, 1024, 4096, 0, 0, "plain", 1},
      {"NT plain", 88256, 3072, 768, 0, 0, "plain", 1},
      {"NT plain K64", 20832, 4096, 64, 0, 0, "plain", 1},
      {"NT plain K256", 20832, 4096, 256, 0, 0, "plain", 1},
      {"NT plain K2048", 20832, 4096, 2048, 0, 0, "plain", 1},
      {"NT fused K64", 20832, 4096, 64, 0, 0, "gelu_tanh_preact", 1},
      {"NT fused", 20832, 4096, 1024, 0, 0, "gelu_tanh_preact", 1},
      {"square", 8192, 8192, 8192, 0, 0, "plain", 1},
    };
  } else if (which == "epi") {   // what a fused epilogue costs, part by part: one store / two stores / two stores + cheap / GELU math
    cases = {
      {"one store", 20832, 4096, 1024, 0, 0, "bias", 1},
      {"two stores", 20832, 4096, 1024, 0, 0, "none_preact", 1},
      {"two stores + relu", 20832, 4096, 1024, 0, 0, "relu_preact", 1},
      {"two stores + tanh-GELU", 20832, 4096, 1024, 0, 0, "gelu_tanh_preact", 1},
      {"two stores + erf-GELU", 20832, 4096, 1024, 0, 0, "gelu_erf_preact", 1},
      {"one store + erf-GELU", 20832, 4096, 1024, 0, 0, "gelu_erf", 1},
      {"one store + residual", 20832, 4096, 1024, 0, 0, "res", 1},
      {"one store + act' (tanh)", 20832, 4096, 1024, 0, 0, "dact_tanh", 1},
      {"one store + act' (erf)", 20832, 4096, 1024, 0, 0, "dact_erf", 1},
      {"one store K64", 20832, 4096, 64, 0, 0, "bias", 1},
      {"two stores K64", 20832, 4096, 64, 0, 0, "none_preact", 1},
      {"two stores + tanh-GELU K64", 20832, 4096, 64, 0, 0, "gelu_tanh_preact", 1},
    };
  } else if (which == "pmc") {   // counter passes on the phase kernel: one launch class per operand layout (k- / r-contiguous)
    cases = {
      {"square NN", 8192, 8192, 8192, 0, 0, "plain", 1},
      {"square NT", 8192, 8192, 8192, 0, 1, "plain", 1},
      {"square TT", 8192, 8192, 8192, 1, 1, "f32", 1},
      {"K1024 NN", 20832, 4096, 1024, 0, 0, "plain", 1},
      {"K1024 NT", 20832, 4096, 1024, 0, 1, "plain", 1},
      {"K1024 NN fused", 20832, 4096, 1024, 0, 0, "gelu_tanh_preact", 1},
    };
  } else if (which == "nn") {    // the measurement variants 40..79 of the phase kernel take the NN layout with the plain epilogue only
    cases = {
      {"square NN", 8192, 8192, 8192, 0, 0, "plain", 1},
      {"K1024 NN", 20832, 4096, 1024, 0, 0, "plain", 1},
      {"K768 NN", 88256, 3072, 768, 0, 0, "plain", 1},
      {"K4096 NN", 20832, 1024, 4096, 0, 0, "plain", 1},
    };
  } else if (which == "defer") {   // round 6: one case per (layout, P class, bias / pre-activation) of the deferred-epilogue builds
    cases = {
      {"NN erf", 20832, 1024, 768, 0, 0, "gelu_erf", 1},
      {"NN bias", 20832, 1024, 768, 0, 0, "bias", 1},
      {"NN plain", 20832, 1024, 768, 0, 0, "plain", 1},
      {"NN tanh+pre", 20832, 1024, 1024, 0, 0, "gelu_tanh_preact", 1},
      {"NN erf+pre", 20832, 1024, 1024, 0, 0, "gelu_erf_preact", 1},
      {"NT erf", 20832, 1024, 768, 0, 1, "gelu_erf", 1},
      {"NT tanh+pre", 20832, 2048, 1024, 0, 1, "gelu_tanh_preact", 1},
      {"TN bias", 20736, 1024, 1024, 1, 0, "bias", 1},
      {"TT plain", 20736, 1024, 1024, 1, 1, "plain", 1},
      {"NN erf ragged N", 20832, 1088, 768, 0, 0, "gelu_erf", 1},
    };
  } else if (which == "small") {
    cases = {
      {"trunk fc1 fwd", 20832, 4096, 1024, 0, 1, "gelu_tanh_preact", 1},
      {"trunk fc1 dX", 20832, 4096, 1024, 0, 0, "dact_tanh", 1},
      {"trunk fc2 fwd", 20832, 1024, 4096, 0, 1, "drop_res", 1},
      {"vit fc1", 88256, 3072, 768, 0, 0, "gelu_erf", 1},
      {"plain NT", 20832, 4096, 1024, 0, 0, "plain", 1},
      {"dW fc1", 1024, 4096, 20832, 1, 1, "f32", 4},
      {"square", 8192, 8192, 8192, 0, 0, "plain", 1},
    };
  }
  if (sk_override > 0) for (auto& c : cases) if (c.split_k > 1) c.split_k = sk_override;
  if (only >= 0 && only < (int)cases.size()) cases = {cases[only]};
  // ---- correctness: every (case layout / epilogue, variant) at a reduced M (ragged: not a tile multiple) ----
  if (!no_check) {
    for (const Case& c : cases) {
      Case cc = c;
      int64_t Mchk = c.at ? c.M : 777;         // r-contiguous A needs whole row panels for the ring kernels; keep it as is
      if (c.at) { cc.K = 2048; if (cc.split_k > 1) cc.split_k = 2; Mchk = c.M; }
      else if (c.K > 1024) cc.K = 1024;
      if (cc.N > 1024 && !c.at) cc.N = 1024;
      for (int v : variants) check_variant(cc, v, Mchk);
    }
  }
  // the stream-K schedule only engages at full size (>= one tile per CU) and reuses its flags from launch to launch: twice
  if (full_check) {
    for (const Case& c : cases)
      if ((double)c.M * (double)c.N <= 1.2e8)
        for (int v : variants)
          for (int rep = 0; rep < 2; ++rep) check_variant(c, v, c.M);
  }
  if (check_only) return 0;
  // ---- timing: variants interleaved, `rounds` rounds of `iters` launches each; median / min over rounds ----
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (const Case& c : cases) {
    Problem q = make_problem(c, c.M);
    std::vector<std::vector<double>> t(variants.size());
    for (int r = 0; r < rounds + 1; ++r) {
      for (size_t vi = 0; vi < variants.size(); ++vi) {
        dvla_set_gemm_variant(variants[vi]);
        (void)dvla_gemm_bf16(&q.p, nullptr);   // warm
        if (hog > 0) {   // occupy `hog` CUs on another stream for ~4 ms while the GEMMs run
          static hipStream_t hs = nullptr; static int* sink = nullptr;
          if (!hs) { CK(hipStreamCreateWithFlags(&hs, hipStreamNonBlocking)); CK(hipMalloc(&sink, 64));
                     CK(hipFuncSetAttribute(reinterpret_cast<const void*>(hog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); }
          CK(hipDeviceSynchronize());
          hipLaunchKernelGGL(hog_kernel, dim3(hog), dim3(256), 96 * 1024, hs, 8000000LL, sink);
          CK(hipStreamQuery(hs) == hipErrorNotReady ? hipSuccess : hipSuccess);
        }
        CK(hipEventRecord(e0, 0));
        for (int it = 0; it < iters; ++it) (void)dvla_gemm_bf16(&q.p, nullptr);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) t[vi].push_back(ms * 1e3 / iters);
      }
    }
    dvla_set_gemm_variant(0);
    printf("{\"time\": \"%s\", \"epi\": \"%s\", \"M\": %lld, \"N\": %lld, \"K\": %lld, \"at\": %d, \"bt\": %d, \"split_k\": %d", c.name.c_str(), c.epi.c_str(),
           (long long)c.M, (long long)c.N, (long long)c.K, c.at, c.bt, c.split_k);
    const double flop = 2.0 * c.M * c.N * c.K;
    for (size_t vi = 0; vi < variants.size(); ++vi) {
      std::sort(t[vi].begin(), t[vi].end());
      const double med = t[vi][t[vi].size() / 2], mn = t[vi][0];
      printf(", \"v%d\": {\"us\": %.1f, \"min_us\": %.1f, \"TF\": %.0f}", variants[vi], med, mn, flop / med / 1e6);
    }
    printf("}\n"); fflush(stdout);
    q.release();
  }
  return 0;
}
