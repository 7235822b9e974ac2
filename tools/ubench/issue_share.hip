// Microbenchmark (round 5): what W waves on one SIMD get TOGETHER, every wave timed, W = 1, 2, 3, 4 (one workgroup per CU)
// and 6, 8 (two workgroups per CU).  Answers the three questions the strip kernel's redesign hangs on:
//   (1) how many waves a SIMD needs before its vector pipe runs at the 2 cycles per wave64 instruction of a SIMD-32
//       (one wave alone issues one every 4-5), for plain, DPP and mixed multiply-adds;
//   (2) what a matrix-only wave and vector-only waves get beside each other (roles by pipe);
//   (3) what waves that carry BOTH get at 2, 3, 4 per SIMD (roles by level, as the strip kernel has them).
// Every wave stamps s_memtime at its start and end and records HW_ID / XCC_ID; the host groups the waves by (XCC, SE, CU,
// SIMD) and prints, per mode and W, the median over SIMDs of: span (first start .. last end), vector instructions per
// cycle-pair, MFMAs per 32 cycles.
//   hipcc --offload-arch=gfx950 -O3 -o issue_share issue_share.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <vector>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));

#define DPPL " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define DPPR " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define ROWL " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"

struct Rec { unsigned long long t0, t1; unsigned hwid, xcc, valu, mfma; };

// modes
enum { M_PLAIN = 0, M_DPP, M_QUARTER, M_ROWDPP, M_ROLE_PLAIN, M_ROLE_QUARTER, M_SAME_32, M_SAME_16, M_SAME_32_LDS, M_ROLE_QUARTER_LDS, M_HALFDPP, M_QUAD, M_QUAD_CL, M_QUAD_MFMA, M_QUAD_MFMA_LDS, M_COUNT };

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, Rec* recs, int groups, int lds_pad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float a[16], s[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x + i; s[i] = threadIdx.x * 0.5f + i; }
  float w = 1.0001f, u = 0.5f, v = 0.25f;
  asm volatile("" : "+v"(w), "+v"(u), "+v"(v));
  float cq[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { cq[i] = 0.01f * i + threadIdx.x; asm volatile("" : "+v"(cq[i])); }
  f4v c2 = {0}, c3 = {0};
  f16v acc0 = {0}, acc1 = {0};
  f4v c0 = {0}, c1 = {0};
  s8v fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 1, 1, 1, 1, 1, 1, 1}, fc = fb, fd = fb;
  asm volatile("" : "+v"(fa), "+v"(fb));
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 1.f;
  const unsigned laddr = (threadIdx.x & 63) * 16u;
  __syncthreads();
  const bool role_mode = MODE == M_ROLE_PLAIN || MODE == M_ROLE_QUARTER || MODE == M_ROLE_QUARTER_LDS;
  const bool mfma_wave = role_mode && wave < 4;
  unsigned nvalu = 0, nmfma = 0;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#define QUARTER(i)                                                                                                        \
  asm volatile(                                                                                                           \
      "v_fmac_f32_e32 %0, %4, %9\n\tv_fmac_f32_e32 %1, %5, %9\n\tv_fmac_f32_e32 %2, %6, %9\n\tv_fmac_f32_e32 %3, %7, %9\n\t" \
      "v_fmac_f32_dpp %0, %4, %8" DPPL "\n\tv_fmac_f32_dpp %1, %5, %8" DPPL "\n\tv_fmac_f32_dpp %2, %6, %8" DPPL             \
      "\n\tv_fmac_f32_dpp %3, %7, %8" DPPL "\n\t"                                                                         \
      "v_fmac_f32_dpp %0, %4, %10" DPPR "\n\tv_fmac_f32_dpp %1, %5, %10" DPPR "\n\tv_fmac_f32_dpp %2, %6, %10" DPPR          \
      "\n\tv_fmac_f32_dpp %3, %7, %10" DPPR                                                                               \
      : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3])                                                        \
      : "v"(s[i]), "v"(s[i + 1]), "v"(s[i + 2]), "v"(s[i + 3]), "v"(w), "v"(u), "v"(v))
// the interleaved-pixel form of a quarter: of the eight side terms four need no lane shift
#define QUARTER_HALF(i)                                                                                                   \
  asm volatile(                                                                                                           \
      "v_fmac_f32_e32 %0, %4, %9\n\tv_fmac_f32_e32 %1, %5, %9\n\tv_fmac_f32_e32 %2, %6, %9\n\tv_fmac_f32_e32 %3, %7, %9\n\t" \
      "v_fmac_f32_e32 %0, %5, %8\n\tv_fmac_f32_dpp %1, %4, %8" ROWL "\n\tv_fmac_f32_e32 %2, %7, %8"                       \
      "\n\tv_fmac_f32_dpp %3, %6, %8" ROWL "\n\t"                                                                         \
      "v_fmac_f32_e32 %1, %4, %10\n\tv_fmac_f32_dpp %0, %5, %10" ROWL "\n\tv_fmac_f32_e32 %3, %6, %10"                    \
      "\n\tv_fmac_f32_dpp %2, %7, %10" ROWL                                                                               \
      : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3])                                                        \
      : "v"(s[i]), "v"(s[i + 1]), "v"(s[i + 2]), "v"(s[i + 3]), "v"(w), "v"(u), "v"(v))

// the quad layout (round 5): a lane holds four pixels 4p + t of a 64-pixel strip row in four accumulator tiles (16x16x32 MFMA
// tiles t = 0..3, register e = one of four output channels); per source row and register: 4 centre + 3 + 3 side terms plain,
// the west term of tile 0 and the east term of tile 3 through DPP row shifts.  a[4 t + e], s[4 t + e]; cw/cc/ce per tile.
#define QROW_PLAIN(e)                                                                                                     \
  asm volatile(                                                                                                           \
      "v_fmac_f32_e32 %0, %4, %8\n\tv_fmac_f32_e32 %1, %5, %9\n\tv_fmac_f32_e32 %2, %6, %10\n\tv_fmac_f32_e32 %3, %7, %11\n\t" \
      "v_fmac_f32_e32 %1, %4, %12\n\tv_fmac_f32_e32 %2, %5, %13\n\tv_fmac_f32_e32 %3, %6, %14\n\t"                        \
      "v_fmac_f32_e32 %0, %5, %12\n\tv_fmac_f32_e32 %1, %6, %13\n\tv_fmac_f32_e32 %2, %7, %14"                            \
      : "+v"(a[e]), "+v"(a[4 + e]), "+v"(a[8 + e]), "+v"(a[12 + e])                                                       \
      : "v"(s[e]), "v"(s[4 + e]), "v"(s[8 + e]), "v"(s[12 + e]), "v"(cq[0]), "v"(cq[1]), "v"(cq[2]), "v"(cq[3]), "v"(cq[4]), \
        "v"(cq[5]), "v"(cq[6]))
#define QROW_DPP(e)                                                                                                       \
  asm volatile("v_fmac_f32_dpp %0, %2, %4" ROWL "\n\tv_fmac_f32_dpp %1, %3, %5 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" \
               : "+v"(a[e]), "+v"(a[12 + e])                                                                              \
               : "v"(s[12 + e]), "v"(s[e]), "v"(cq[7]), "v"(cq[8]))
#define MFMA32(ACC, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(fa), "v"(B))
#define MFMA16(ACC) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(fa), "v"(fb))
  for (int g = 0; g < groups; ++g) {
    if (MODE == M_PLAIN) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(s[i]), "v"(u));
      nvalu += 64;
    } else if (MODE == M_DPP) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_dpp %0, %1, %2" DPPL : "+v"(a[i]) : "v"(s[i]), "v"(u));
      nvalu += 64;
    } else if (MODE == M_ROWDPP) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_dpp %0, %1, %2" ROWL : "+v"(a[i]) : "v"(s[i]), "v"(u));
      nvalu += 64;
    } else if (MODE == M_QUARTER) {
#pragma unroll
      for (int r = 0; r < 2; ++r) { QUARTER(0); QUARTER(4); QUARTER(8); QUARTER(12); }
      nvalu += 96;
    } else if (MODE == M_HALFDPP) {
#pragma unroll
      for (int r = 0; r < 2; ++r) { QUARTER_HALF(0); QUARTER_HALF(4); QUARTER_HALF(8); QUARTER_HALF(12); }
      nvalu += 96;
    } else if (MODE == M_QUAD) {
      // one source row of a plane row: 48 multiply-adds, the two DPP ones behind each register's ten plain ones
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { QROW_PLAIN(e); QROW_DPP(e); }
      }
      nvalu += 96;
    } else if (MODE == M_QUAD_CL) {
      // the same with the eight DPP ones of a source row in one cluster
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int e = 0; e < 4; ++e) QROW_PLAIN(e);
#pragma unroll
        for (int e = 0; e < 4; ++e) QROW_DPP(e);
      }
      nvalu += 96;
    } else if (MODE == M_QUAD_MFMA || MODE == M_QUAD_MFMA_LDS) {
      // the quad strip kernel's mix: per source row (48 multiply-adds) eight 16x16x32 MFMAs (24 per 144), with the B
      // fragments of those MFMAs read from LDS (16 fragments per 24 MFMAs) in the _LDS variant
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (MODE == M_QUAD_MFMA_LDS && (e & 1) == 0) {
            fb = *reinterpret_cast<const s8v*>(smem + laddr + (e * 2 + r * 8) * 1024 % 16384);
            fc = *reinterpret_cast<const s8v*>(smem + laddr + (e * 2 + 1 + r * 8) * 1024 % 16384);
          }
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(fa), "v"(fb));
          QROW_PLAIN(e);
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(fa), "v"(fc));
          QROW_DPP(e);
        }
      }
      nvalu += 96; nmfma += 16;
    } else if (role_mode) {
      if (mfma_wave) {
        if (MODE == M_ROLE_QUARTER_LDS) {
          // 4 MFMAs fed by 4 fragment reads (two ahead)
          fb = *reinterpret_cast<const s8v*>(smem + laddr);
          fc = *reinterpret_cast<const s8v*>(smem + laddr + 1024);
          MFMA32(acc0, fb); MFMA32(acc1, fc); MFMA32(acc0, fb); MFMA32(acc1, fc);
          fb = *reinterpret_cast<const s8v*>(smem + laddr + 2048);
          fc = *reinterpret_cast<const s8v*>(smem + laddr + 3072);
          MFMA32(acc0, fb); MFMA32(acc1, fc); MFMA32(acc0, fb); MFMA32(acc1, fc);
        } else {
          MFMA32(acc0, fb); MFMA32(acc1, fb); MFMA32(acc0, fb); MFMA32(acc1, fb);
          MFMA32(acc0, fb); MFMA32(acc1, fb); MFMA32(acc0, fb); MFMA32(acc1, fb);
        }
        nmfma += 8;
      } else {
        if (MODE == M_ROLE_PLAIN) {
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(s[i]), "v"(u));
        } else {
          QUARTER(0); QUARTER(4); QUARTER(8); QUARTER(12);
        }
        nvalu += 48;
      }
    } else if (MODE == M_SAME_32 || MODE == M_SAME_32_LDS) {
      // the strip kernel's mix: one 32x32x16 MFMA per 12 multiply-adds (a quarter); here 4 + 48 per group
      if (MODE == M_SAME_32_LDS) {
        fb = *reinterpret_cast<const s8v*>(smem + laddr);
        fc = *reinterpret_cast<const s8v*>(smem + laddr + 1024);
      }
      MFMA32(acc0, fb); QUARTER(0);
      MFMA32(acc0, fc); QUARTER(4);
      if (MODE == M_SAME_32_LDS) {
        fd = *reinterpret_cast<const s8v*>(smem + laddr + 2048);
        fb = *reinterpret_cast<const s8v*>(smem + laddr + 3072);
      }
      MFMA32(acc0, fd); QUARTER(8);
      MFMA32(acc0, fb); QUARTER(12);
      nvalu += 48; nmfma += 4;
    } else if (MODE == M_SAME_16) {
      // the same arithmetic on 16x16x32 MFMAs: two of them per quarter
      MFMA16(c0); MFMA16(c1); QUARTER(0);
      MFMA16(c0); MFMA16(c1); QUARTER(4);
      MFMA16(c0); MFMA16(c1); QUARTER(8);
      MFMA16(c0); MFMA16(c1); QUARTER(12);
      nvalu += 48; nmfma += 8;
    }
  }
  asm volatile("s_nop 0" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float q = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) q += a[i] + acc0[i] + acc1[i];
  q += c2[0] + c3[0] + c0[0] + c1[0] + c0[1] + c1[1] + c0[2] + c1[2] + c0[3] + c1[3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = q + (float)fb[0] + (float)fc[0] + (float)fd[0];
  if ((threadIdx.x & 63) == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    Rec r;
    r.t0 = t0; r.t1 = t1; r.hwid = hwid; r.xcc = xcc & 0xf; r.valu = nvalu; r.mfma = nmfma;
    recs[blockIdx.x * (blockDim.x >> 6) + wave] = r;
  }
}

template <int MODE> static void run(const char* name, float* out, Rec* d_recs, int W) {
  const int groups = 400;
  const int blocks_per_cu = W > 4 ? 2 : 1;
  const int threads = 256 * (W / blocks_per_cu);
  const int grid = 256 * blocks_per_cu;
  const int lds = blocks_per_cu == 2 ? 64 * 1024 : 100 * 1024;  // residency: exactly blocks_per_cu per CU
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int nw = grid * threads / 64;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), lds, 0, out, d_recs, groups, 0);
    hipDeviceSynchronize();
  }
  std::vector<Rec> h(nw);
  hipMemcpy(h.data(), d_recs, nw * sizeof(Rec), hipMemcpyDeviceToHost);
  struct Agg { unsigned long long lo = ~0ull, hi = 0; unsigned long long valu = 0, mfma = 0; int n = 0; double wv = 0, wm = 0; int nv = 0, nm = 0; };
  std::map<unsigned long long, Agg> by_simd;
  for (const Rec& r : h) {
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...; keep everything but the wave id
    const unsigned long long key = ((unsigned long long)r.xcc << 32) | (r.hwid & 0xff30u);
    Agg& a = by_simd[key];
    a.lo = std::min(a.lo, r.t0); a.hi = std::max(a.hi, r.t1);
    a.valu += r.valu; a.mfma += r.mfma; a.n++;
    if (r.valu && !r.mfma) { a.wv += (double)(r.t1 - r.t0) / r.valu; a.nv++; }
    if (r.mfma && !r.valu) { a.wm += (double)(r.t1 - r.t0) / r.mfma; a.nm++; }
  }
  std::vector<double> span, cpv, cpm, wv, wm;
  int full = 0;
  for (auto& kv : by_simd) {
    const Agg& a = kv.second;
    if (a.n != W) continue;  // (a SIMD that got another count of waves: dispatch did not spread as assumed)
    ++full;
    const double sp = (double)(a.hi - a.lo);
    span.push_back(sp);
    if (a.valu) cpv.push_back(sp / a.valu);
    if (a.mfma) cpm.push_back(sp / a.mfma);
    if (a.nv) wv.push_back(a.wv / a.nv);
    if (a.nm) wm.push_back(a.wm / a.nm);
  }
  auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  printf("%-44s W=%d: SIMDs %4d/%4zu  span %8.0f  cycles per VALU instr (SIMD) %5.2f  per MFMA (SIMD) %6.1f", name, W, full, by_simd.size(), med(span),
         med(cpv), med(cpm));
  if (!wv.empty() || !wm.empty()) printf("  | own-span: VALU-only wave %5.2f cyc/instr, MFMA-only wave %5.1f cyc/MFMA", med(wv), med(wm));
  printf("\n");
}

int main() {
  float* out; Rec* recs;
  hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&recs, 512 * 16 * sizeof(Rec));
  for (int W : {1, 2, 4}) {
    run<M_QUAD>("quad rows (10 plain + 2 row-dpp) x 8", out, recs, W);
    run<M_QUAD_CL>("quad rows, dpp clustered", out, recs, W);
    run<M_QUAD_MFMA>("quad rows + 16 MFMA 16x16x32", out, recs, W);
    run<M_QUAD_MFMA_LDS>("quad rows + 16 MFMA 16x16x32 + frag reads", out, recs, W);
  }
  if (getenv("QUAD_ONLY")) return 0;
  for (int W : {1, 2, 3, 4, 6, 8}) {
    run<M_PLAIN>("64 v_fmac_f32", out, recs, W);
    run<M_DPP>("64 v_fmac_f32_dpp wave_shr", out, recs, W);
    run<M_ROWDPP>("64 v_fmac_f32_dpp row_shr", out, recs, W);
    run<M_QUARTER>("quarters (4 plain + 8 wave-dpp)", out, recs, W);
    run<M_HALFDPP>("interleaved quarters (8 plain + 4 row-dpp)", out, recs, W);
  }
  for (int W : {2, 3, 4, 6, 8}) {
    run<M_ROLE_PLAIN>("roles: 1 MFMA wave + (W-1) x 48 v_fmac", out, recs, W);
    run<M_ROLE_QUARTER>("roles: 1 MFMA wave + (W-1) x 4 quarters", out, recs, W);
    run<M_ROLE_QUARTER_LDS>("roles: 1 MFMA wave (LDS frags) + quarters", out, recs, W);
  }
  for (int W : {1, 2, 3, 4, 6, 8}) {
    run<M_SAME_32>("same wave: 4 x (MFMA 32x32x16 + quarter)", out, recs, W);
    run<M_SAME_32_LDS>("same wave: ... + fragment reads", out, recs, W);
    run<M_SAME_16>("same wave: 4 x (2 MFMA 16x16x32 + quarter)", out, recs, W);
  }
  return 0;
}
