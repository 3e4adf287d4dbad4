// sync_probe.cpp -- what does a dependent exchange between workgroups cost INSIDE one kernel?  (measurement infrastructure)
//
// The evaluation-time DiT head is a chain of ~50 dependent few-row GEMMs per sampler step (12 rows at one episode); under hipGraph
// replay each launch boundary costs ~5 us and each kernel lives for two memory round trips.  A persistent kernel replaces the
// launch boundary by a barrier between its workgroups.  This probe measures the pieces such a kernel is made of:
//   barrier  : G workgroups; per iteration every workgroup writes 64 B, passes a counter barrier, reads the 64 B another workgroup
//              wrote and passes a second barrier -- us per barrier; two ways of making the data visible: (a) agent-scope release
//              add, relaxed polling, acquire fence (buffer_wbl2 / buffer_inv: what grid.sync() does), (b) the data itself moves
//              with device-scope (sc1) accesses and nothing is flushed or invalidated; for
//                 all   : 256 workgroups, one per CU, all eight XCDs (the exchange crosses the fabric)
//                 xcd0  : the 32 workgroups that run on XCC 0 only (the exchange stays in one L2)
//   stream   : the same teams read a 170-MB buffer (the DiT-B weights) once, 12 x 16 B per lane in flight -- GB/s, first and
//              second pass (HBM / Infinity Cache)
// Build: hipcc --offload-arch=gfx950 -O3 tests/probes/sync_probe.cpp -o build/sync_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}

// counter barrier among `team` workgroups: *ctr counts arrivals since the launch, `target` = team * (barriers passed + 1).
// Returns false when the spin limit was hit (a missing workgroup must not hang the box).
__device__ __forceinline__ bool team_barrier(unsigned* ctr, unsigned target, unsigned* err) {
  __shared__ int ok_s;
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    bool ok = true;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1u << 21)) { ok = false; atomicAdd(err, 1u); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    ok_s = ok;
  }
  __syncthreads();
  return ok_s != 0;
}

// The same exchange without cache maintenance: the data moves with device-scope (sc1) 16-byte accesses -- performed at the
// L2 / fabric level by themselves -- and the counter with relaxed device-scope atomics; no buffer_wbl2 / buffer_inv.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ void store16_dev(uint4* p, uint4 v) {
  const u32x4 w = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ uint4 load16_dev(const uint4* p) {
  u32x4 w;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
  return make_uint4(w.x, w.y, w.z, w.w);
}
__device__ __forceinline__ bool team_barrier_nofence(unsigned* ctr, unsigned target, unsigned* err) {
  __shared__ int ok2_s;
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    bool ok = true;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1u << 21)) { ok = false; atomicAdd(err, 1u); break; }
    }
    ok2_s = ok;
  }
  __syncthreads();
  return ok2_s != 0;
}

// mode 0: all workgroups; mode 1: only the workgroups whose XCC_ID is 0 (the team size is found at run time: every workgroup
// registers under its XCC first, behind one chip-wide barrier)
template <int FLAVOUR>
__global__ __launch_bounds__(512) void barrier_kernel(unsigned* ctrs, uint4* slots, unsigned* xcc_of, int iters, int mode,
                                                      unsigned long long* cycles, unsigned* bad) {
  __shared__ int rank_s, team_s;
  const unsigned xcc = xcc_id();
  unsigned* reg_ctr = ctrs;              // [0] chip-wide registration barrier
  unsigned* team_cnt = ctrs + 16;        // [16 + xcc] members registered per XCC
  unsigned* bar = ctrs + 64;             // the measured barrier's counter
  unsigned* err = ctrs + 96;
  if (threadIdx.x == 0) {
    xcc_of[blockIdx.x] = xcc;
    rank_s = (int)atomicAdd(team_cnt + xcc, 1u);
  }
  if (!team_barrier(reg_ctr, gridDim.x, err)) return;
  if (threadIdx.x == 0) team_s = mode == 0 ? (int)gridDim.x : (int)__hip_atomic_load(team_cnt + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int team = team_s;
  int me;
  if (mode == 0) me = (int)blockIdx.x;
  else { if (xcc != 0) return; me = rank_s; }
  const unsigned long long t0 = wall_clock64();
  unsigned wrong = 0;
  for (int it = 0; it < iters; ++it) {
    const uint4 mine = make_uint4((unsigned)it, (unsigned)me, threadIdx.x, 0x5a5au);
    const int nb = (me + team / 2 + 1) % team;          // (a workgroup of another XCD when the team spans the chip)
    if (FLAVOUR == 0) {
      if (threadIdx.x < 4) slots[(size_t)me * 4 + threadIdx.x] = mine;
      if (!team_barrier(bar, (unsigned)team * (unsigned)(it + 1), err)) return;
      if (threadIdx.x < 4) {
        const uint4 v = slots[(size_t)nb * 4 + threadIdx.x];
        if (v.x != (unsigned)it || v.y != (unsigned)nb) ++wrong;
      }
      // (the next iteration overwrites slot `me` while the neighbour may still be reading the previous value: a second barrier)
      if (!team_barrier(bar + 1, (unsigned)team * (unsigned)(it + 1), err)) return;
    } else {
      if (threadIdx.x < 4) store16_dev(slots + (size_t)me * 4 + threadIdx.x, mine);
      if (!team_barrier_nofence(bar, (unsigned)team * (unsigned)(it + 1), err)) return;
      if (threadIdx.x < 4) {
        const uint4 v = load16_dev(slots + (size_t)nb * 4 + threadIdx.x);
        if (v.x != (unsigned)it || v.y != (unsigned)nb) ++wrong;
      }
      if (!team_barrier_nofence(bar + 1, (unsigned)team * (unsigned)(it + 1), err)) return;
    }
  }
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0 && me == 0) { cycles[0] = t1 - t0; cycles[1] = (unsigned long long)team; }
  if (wrong) atomicAdd(bad, wrong);
}

__global__ __launch_bounds__(512) void stream_kernel(const uint4* __restrict__ buf, size_t n16, int mode, unsigned* sink) {
  const unsigned xcc = xcc_id();
  int team, me;
  if (mode == 0) { team = (int)gridDim.x; me = (int)blockIdx.x; }
  else { if (xcc != 0) return; team = (int)gridDim.x / 8; me = (int)blockIdx.x / 8; }      // (round-robin placement, checked by barrier_kernel's table)
  const size_t per = n16 / (size_t)team;
  const uint4* p = buf + (size_t)me * per;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = threadIdx.x; i + 11 * 512 < per; i += 12 * 512) {
    uint4 v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] = p[i + (size_t)k * 512];
#pragma unroll
    for (int k = 0; k < 12; ++k) { acc.x ^= v[k].x; acc.y ^= v[k].y; acc.z ^= v[k].z; acc.w ^= v[k].w; }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 400;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int G = prop.multiProcessorCount;
  unsigned *ctrs, *xcc_of, *bad;
  uint4* slots;
  unsigned long long* cycles;
  CK(hipMalloc(&ctrs, 4096));
  CK(hipMalloc(&xcc_of, 4096 * 4));
  CK(hipMalloc(&bad, 4));
  CK(hipMalloc(&slots, 4096 * 64));
  CK(hipMalloc(&cycles, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // 80 KB of dynamic LDS per workgroup: one workgroup per CU, so that `G` workgroups are co-resident on G CUs
  CK(hipFuncSetAttribute((const void*)barrier_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
  CK(hipFuncSetAttribute((const void*)barrier_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
  CK(hipFuncSetAttribute((const void*)stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
  for (int flavour = 0; flavour < 2; ++flavour)
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemset(ctrs, 0, 4096));
      CK(hipMemset(bad, 0, 4));
      CK(hipMemset(cycles, 0, 64));
      CK(hipEventRecord(e0));
      if (flavour == 0) hipLaunchKernelGGL(barrier_kernel<0>, dim3(G), dim3(512), 81920, 0, ctrs, slots, xcc_of, iters, mode, cycles, bad);
      else hipLaunchKernelGGL(barrier_kernel<1>, dim3(G), dim3(512), 81920, 0, ctrs, slots, xcc_of, iters, mode, cycles, bad);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long cyc[2];
      unsigned h_ctrs[128], h_bad;
      CK(hipMemcpy(cyc, cycles, 16, hipMemcpyDeviceToHost));
      CK(hipMemcpy(h_ctrs, ctrs, 512, hipMemcpyDeviceToHost));
      CK(hipMemcpy(&h_bad, bad, 4, hipMemcpyDeviceToHost));
      // wall_clock64 ticks at 100 MHz
      printf("{\"probe\": \"barrier\", \"sync\": \"%s\", \"team\": \"%s\", \"workgroups\": %llu, \"iters\": %d, \"us_per_exchange\": %.3f, \"event_ms\": %.3f, "
             "\"stale_reads\": %u, \"spin_limit_hits\": %u}\n", flavour == 0 ? "release add / acquire fence" : "sc1 accesses, relaxed counter, no cache maintenance",
             mode == 0 ? "all" : "xcd0", cyc[1], iters,
             (double)cyc[0] / 100.0 / (2.0 * iters), ms, h_bad, h_ctrs[96]);
      if (flavour == 0 && mode == 0 && rep == 0) {
        unsigned h_x[4096];
        CK(hipMemcpy(h_x, xcc_of, G * 4, hipMemcpyDeviceToHost));
        int rr = 0;
        for (int i = 0; i < G; ++i) rr += (h_x[i] == (unsigned)(i % 8));
        printf("{\"probe\": \"placement\", \"workgroups\": %d, \"on_xcc_blockidx_mod_8\": %d, \"per_xcc\": [%u, %u, %u, %u, %u, %u, %u, %u]}\n", G, rr,
               h_ctrs[16], h_ctrs[17], h_ctrs[18], h_ctrs[19], h_ctrs[20], h_ctrs[21], h_ctrs[22], h_ctrs[23]);
      }
    }
  }
  const size_t bytes = (size_t)170 << 20;
  uint4* buf;
  CK(hipMalloc(&buf, bytes));
  CK(hipMemset(buf, 1, bytes));
  uint4* flush;
  CK(hipMalloc(&flush, (size_t)1 << 30));
  for (int mode = 0; mode < 2; ++mode) {
    CK(hipMemset(flush, 2, (size_t)1 << 30));       // push the buffer out of the Infinity Cache
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(stream_kernel, dim3(G), dim3(512), 81920, 0, buf, bytes / 16, mode, bad);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("{\"probe\": \"stream\", \"team\": \"%s\", \"pass\": %d, \"MB\": %zu, \"us\": %.1f, \"GBps\": %.0f}\n", mode == 0 ? "all" : "xcd0", rep,
             bytes >> 20, ms * 1e3, (double)bytes / (ms * 1e-3) / 1e9);
    }
  }
  return 0;
}
