// Can one of the three LDS exchanges of the 4096-point row transform (512 threads x 8 complex points) be done in registers?
// (review item, round 4: gfx950 has v_permlane32_swap / v_permlane16_swap; DPP reaches the lower lane bits)
// An exchange whose partners differ only in lane bits is an 8 x 8 transpose between the 3 bits of the register index and 3 bits
// of the lane number: one step per bit pair, 4 register pairs x 4 dwords per step.  Steps on lane bits 5 and 4 are ONE swap
// instruction per dword pair (v_permlane32_swap, v_permlane16_swap); a step on lane bit 3 (or lower) has no swap form: two DPP
// moves (row_ror:8 brings lane ^ 8) and two v_cndmask per dword pair.
// Measured here, one 512-thread workgroup per CU (2 waves per SIMD, like the row kernels), all 256 CUs busy:
//   lds     : 8 x ds_write_b128 (stride-64 scatter, swizzled like WgFft) + barrier + 8 x ds_read_b128 + barrier
//   swap2   : the two swap steps only (lane bits 5, 4)            -- a lower bound for any register exchange
//   swap3   : swap steps + the DPP step on lane bit 3              -- a full 8 x 8 transpose on lane bits 3..5
//   dpp3    : three DPP steps (lane bits 0..2: quad_perm, quad_perm, row_ror-style) -- what the FIRST wave-local exchange needs
// each with and without 64 independent fp64 FMAs per thread between two exchanges (does the exchange hide under arithmetic?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double2 cd;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void barrier_lds() { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); }

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) { auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false); a = r[0]; b = r[1]; }
__device__ __forceinline__ void swap16(unsigned& a, unsigned& b) { auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false); a = r[0]; b = r[1]; }
#else
__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {}
__device__ __forceinline__ void swap16(unsigned& a, unsigned& b) {}
#endif
// transpose step through a "bring the partner's value" DPP move: lanes with the bit clear keep a and take the partner's a into b's
// place, lanes with the bit set the other way round
template <int CTRL>
__device__ __forceinline__ void dppstep(unsigned& a, unsigned& b, bool upper) {
  const unsigned pa = __builtin_amdgcn_mov_dpp(a, CTRL, 0xf, 0xf, true), pb = __builtin_amdgcn_mov_dpp(b, CTRL, 0xf, 0xf, true);
  const unsigned na = upper ? pb : a, nb = upper ? b : pa;
  a = na;
  b = nb;
}
union U { cd c; unsigned w[4]; };

template <int MODE, int FMAS>
__global__ void __launch_bounds__(512) k_bench(cd* out, int reps) {
  extern __shared__ cd lds[];
  const int j = threadIdx.x, lane = j & 63;
  U v[8];
  for (int u = 0; u < 8; ++u) v[u].c = make_double2(1.0 + 1e-3 * (j + u), 1e-3 * (j - u));
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) {
      const int pos = (j / 64) * 512 + (j % 64);
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int p = pos + u * 64; lds[p ^ ((p >> 3) & 7)] = v[u].c; }
      barrier_lds();
#pragma unroll
      for (int t = 0; t < 8; ++t) { const int p = j + t * 512; v[t].c = lds[p ^ ((p >> 3) & 7)]; }
      barrier_lds();
    } else {
      if (MODE == 1 || MODE == 2) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int w = 0; w < 4; ++w) swap32(v[u].w[w], v[u + 4].w[w]);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if ((u & 2) == 0) {
#pragma unroll
            for (int w = 0; w < 4; ++w) swap16(v[u].w[w], v[u + 2].w[w]);
          }
      }
      if (MODE == 2) {
        const bool up = lane & 8;
#pragma unroll
        for (int u = 0; u < 8; u += 2)
#pragma unroll
          for (int w = 0; w < 4; ++w) dppstep<0x128>(v[u].w[w], v[u + 1].w[w], up);     // row_ror:8
      }
      if (MODE == 3) {
        const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int w = 0; w < 4; ++w) dppstep<0xB1>(v[u].w[w], v[u + 4].w[w], b0);       // quad_perm [1,0,3,2]: lane ^ 1
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if ((u & 2) == 0) {
#pragma unroll
            for (int w = 0; w < 4; ++w) dppstep<0x4E>(v[u].w[w], v[u + 2].w[w], b1);     // quad_perm [2,3,0,1]: lane ^ 2
          }
#pragma unroll
        for (int u = 0; u < 8; u += 2)
#pragma unroll
          for (int w = 0; w < 4; ++w) dppstep<0x124>(v[u].w[w], v[u + 1].w[w], b2);      // row_ror:4 (stands in for lane ^ 4: same cost)
      }
    }
#pragma unroll
    for (int f = 0; f < FMAS; ++f) acc[f & 7] = __builtin_fma(v[f & 7].c.x, 1.0000001, acc[f & 7] * 0.5 + v[(f + 3) & 7].c.y);
  }
  cd o = make_double2(0, 0);
  for (int u = 0; u < 8; ++u) { o.x += v[u].c.x + acc[u]; o.y += v[u].c.y; }
  out[(size_t)blockIdx.x * 512 + j] = o;
}

template <int MODE, int FMAS>
static double run(const char* name, cd* out, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f;
  for (int it = 0; it < 4; ++it) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_bench<MODE, FMAS>), dim3(256), dim3(512), 4096 * sizeof(cd), 0, out, reps);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  const double us = best * 1e3 / reps;
  printf("%-64s %8.3f us per iteration per workgroup\n", name, us);
  return us;
}

int main() {
  cd* out;
  CK(hipMalloc(&out, sizeof(cd) * 256 * 512));
  const int reps = 20000;
  const double f0 = run<4, 64>("64 fp64 FMA chains only (no exchange)", out, reps);
  const double l0 = run<0, 0>("LDS exchange alone", out, reps), l1 = run<0, 64>("LDS exchange + 64 FMAs", out, reps);
  const double s0 = run<1, 0>("two swap steps (lane bits 5, 4) alone", out, reps), s1 = run<1, 64>("two swap steps + 64 FMAs", out, reps);
  const double t0 = run<2, 0>("8x8 transpose on lane bits 3..5 (2 swaps + 1 DPP step) alone", out, reps), t1 = run<2, 64>("... + 64 FMAs", out, reps);
  const double d0 = run<3, 0>("8x8 transpose on lane bits 0..2 (3 DPP steps) alone", out, reps), d1 = run<3, 64>("... + 64 FMAs", out, reps);
  printf("\nexposed cost next to the FMAs (time with - FMAs alone): LDS %.3f  swaps(5,4) %.3f  transpose(3..5) %.3f  transpose(0..2) %.3f us\n",
         l1 - f0, s1 - f0, t1 - f0, d1 - f0);
  (void)l0; (void)s0; (void)t0; (void)d0;
  return 0;
}
