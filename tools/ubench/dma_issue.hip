// Microbenchmark: what does one LDS-DMA piece (global_load_lds_dwordx4, 1 KiB per wave instruction) cost the issuing
// wave, by address pattern and by how many waves of the CU issue at once?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ void glds16_off(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// seg: bytes contiguous per row piece (64, 128, 256, 1024); rows are `stride` bytes apart, picked pseudo-randomly in a
// window of `win` rows per workgroup; nwaves: how many waves of the workgroup issue; NP pieces back to back, R rounds
template <int NP>
__global__ __launch_bounds__(512, 2) void k(const float* x, int seg, int stride, int win, int nwaves, int R,
                                            unsigned long long* out, int spread, int cold) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[8 * NP * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lanes_per_row = seg / 16;
  unsigned voff[NP];
  for (int p = 0; p < NP; ++p) {
    const int rowslot = (lane / lanes_per_row) + (64 / lanes_per_row) * (p + NP * wave);
    const unsigned row = (unsigned)((rowslot * 2654435761u) >> 8) % (unsigned)win;
    voff[p] = row * (unsigned)stride + (unsigned)(lane % lanes_per_row) * 16u;
  }
  const float* base = x + (size_t)blockIdx.x * ((size_t)4096 * 4 * stride / 4);
  unsigned long long t_issue = 0, t_total = 0;
  for (int r = 0; r < R; ++r) {
    __syncthreads();
    unsigned long long t0, t1, t2;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (wave < nwaves) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        glds16_off(base, voff[p] + (cold ? (unsigned)r * 64u * (unsigned)stride * 0u + (unsigned)(r % 13) * 147456u : 0u), __builtin_amdgcn_readfirstlane((unsigned)((wave * NP + p) * 1024)));
        if (spread) __builtin_amdgcn_s_sleep(4);
      }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    if (r > 0) { t_issue += t1 - t0; t_total += t2 - t0; }
  }
  if (lane == 0 && blockIdx.x == 40) { out[wave * 2] = t_issue / (R - 1); out[wave * 2 + 1] = t_total / (R - 1); }
  if (smem[tid] == 77 && out[100] == 1) out[101] = 1;
}

int main() {
  const size_t bytes = (size_t)256 * 4096 * 256 * 4;  // 256 blocks x 4096 rows x 256 B = 256 MB
  float* x; unsigned long long* out;
  hipMalloc(&x, bytes); hipMemset(x, 0, bytes);
  hipMalloc(&out, 1024); hipMemset(out, 0, 1024);
  const int segs[] = {64, 128, 256, 1024};
  const int wins[] = {576};  // rows per workgroup window: 147 KB (L2-ish) and 1 MB
  for (int win : wins)
    for (int seg : segs)
      for (int nw : {1, 8})
        for (int spread : {0})
        for (int cold : {0, 1}) {
          k<8><<<256, 512>>>(x, seg, 256, win, nw, 40, out, spread, cold);
          hipDeviceSynchronize();
          unsigned long long h[16];
          hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
          printf("cold %d win %4d rows seg %4d B waves %d sleep %d: issue %5llu cyc / 8 pieces (wave0), total %5llu; last wave issue %5llu total %5llu\n",
                 cold, win, seg, nw, spread, h[0], h[1], h[(nw - 1) * 2], h[(nw - 1) * 2 + 1]);
        }
  return 0;
}
