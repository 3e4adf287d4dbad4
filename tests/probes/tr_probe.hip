// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (which LDS elements does each lane receive?).
// LDS is filled with the element index (u16 value = index); every lane reads 8 bytes at a per-lane address.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(const int* addr_bytes, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  uint32_t a = (uint32_t)(uintptr_t)(&lds[0]) + (uint32_t)addr_bytes[threadIdx.x];
  uint2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
  out[threadIdx.x * 4 + 0] = r.x & 0xffff; out[threadIdx.x * 4 + 1] = r.x >> 16;
  out[threadIdx.x * 4 + 2] = r.y & 0xffff; out[threadIdx.x * 4 + 3] = r.y >> 16;
}
int main() {
  int h_addr[64]; unsigned short h_out[256];
  int* d_addr; unsigned short* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int pat = 0; pat < 3; ++pat) {
    for (int l = 0; l < 64; ++l) {
      if (pat == 0) h_addr[l] = l * 8;                                   // linear: lane l -> elements 4l..4l+3
      if (pat == 1) h_addr[l] = (l & 15) * 8 + (l >> 4) * 1024;          // each 16-lane group linear inside its own 1 KiB
      if (pat == 2) h_addr[l] = ((l & 3) * 8) + ((l >> 2) & 3) * 256 + (l >> 4) * 32;  // 4 k-rows of 128 elems: chunk c -> row c>>2, col 4(c&3); groups side by side
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("pattern %d\n", pat);
    for (int l = 0; l < 64; ++l) printf("lane %2d addr_elem %4d -> %4d %4d %4d %4d\n", l, h_addr[l] / 2, h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3]);
  }
  return 0;
}
