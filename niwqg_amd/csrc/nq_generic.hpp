// Generic 2-D transform building blocks (not the fused hot path): row FFTs along x and the two
// coalesced sub-passes of the y transform.  Used by the FFT seam (nq_fft2 ...), set_q/set_phi and
// field downloads; the fused step kernels in nq_step.hpp reuse the same index conventions.
//
// y transform of length N = S1*S2 (y = y1 + S1*y2, l = l1 + S2*l2):
//   forward  A: for fixed y1, FFT_{S2} over y2 -> l1, times w_N^(y1*l1), stored at row y1 + S1*l1
//            B: for fixed l1, FFT_{S1} over y1 (rows S1*l1 + y1, contiguous) -> l2, stored at the
//               NATURAL row l = l1 + S2*l2
//   inverse  B^-1 then A^-1, conjugated.
// Both sub-passes touch memory only in segments of CL*16 bytes (CL = 32 columns -> 512 B), which is
// what makes them run at streaming bandwidth (tools/access_bench.hip).
#pragma once
#include "nq_fft.hpp"

namespace nq {

constexpr int CL = 32;   // columns per workgroup in the y sub-passes

// CLX = CL: the tiles of the two-pass y transform (S = one radix of N = S1 * S2).  CLX = CLS: SINGLE-PASS columns for small
// grids (S = N <= 512, S2 = 1): a tile of CLS whole columns fits a workgroup with room for the fused spectral kernels' four
// arrays (8 points per thread) and the A sub-pass disappears: 12 instead of 20 launches per QGModel step.  These grids are
// cache-resident and their steps are chains of dependent 5-10 microsecond kernels, so narrow tiles (more workgroups) beat
// wide segments: CLS = 8 / 4 / 2 measured 3584 / 3855 / 3470 steps/s on UnCoupledModel 512^2 and 8805 / 8820 / 8440 on
// QGModel 256^2, against 3450 / 8380 with the two-pass tiles (profiles/r03_small_grids_single_pass_columns.txt).
#ifndef NQ_CLS
#define NQ_CLS 4
#endif
constexpr int CLS = NQ_CLS;
template <int S, int CLX = CL> struct YPlanT {
  static constexpr int CLW = CLX;
  static constexpr int P = (CLX != CL) ? 8 : ((S >= 128) ? 16 : 8);
  static constexpr int T = S / P;
  static constexpr int THREADS = CLX * T;
  typedef WgFft<S, P, CLX, false> F;
  static constexpr size_t LDS_BYTES = (size_t)F::LDS_ELEMS * sizeof(cd) + 512;   // + reduction scratch
};
template <int S> using YPlan = YPlanT<S, CL>;

#ifndef NQ_XP
#define NQ_XP 8      // points per thread in k_x_products: 8 -> 512 threads per 4096-point row, no spills
#endif
#ifndef NQ_XP1
#define NQ_XP1 16    // points per thread in k_x_wavepv (fewer live fields: 16 points, two workgroups per CU)
#endif
// Row-kernel plan: PP points per thread, WG workgroups per CU the kernel is compiled for.
#ifndef NQ_XP_BIG
#define NQ_XP_BIG 8   // points per thread for rows of 8192 points (one workgroup per CU: the exchange alone is 128 KB)
#endif
template <int N, int PP = NQ_XP, int WG_ = 1, int PBIG = NQ_XP_BIG> struct XPlanT {
  __host__ __device__ static constexpr int pts(int n) { return n >= 8192 ? PBIG : (n >= 128 ? PP : 8); }
  static constexpr int P = pts(N);
  static constexpr int WG = (N >= 8192) ? 1 : WG_;
  static constexpr int T = N / P;
  static constexpr int C = (T >= 64) ? 1 : 64 / T;     // rows per workgroup (>= one wave)
  static constexpr int THREADS = C * T;
  static constexpr int MIN_WAVES = (THREADS * WG + 255) / 256;
  typedef WgFft<N, P, C, true> F;
  // [exchange][stage twiddle table][per-row scratch words][reduction scratch]
  static constexpr size_t LDS_BYTES = (size_t)(F::LDS_ELEMS + F::TW_LDS_ELEMS) * sizeof(cd) + 16 * C + 512;
  static_assert(LDS_BYTES <= 160 * 1024, "row plan exceeds the 160 KB of LDS of a CU");
};
template <int N> using XPlan = XPlanT<N, NQ_XP, 1, NQ_XP_BIG>;
template <int N> using XPlan1 = XPlanT<N, NQ_XP1, 2, 16>;

extern __shared__ __attribute__((aligned(16))) unsigned char nq_smem[];

// Sum NV per-thread values over the workgroup and let thread 0 store them at dst[0..NV) (one slot per
// workgroup: deterministic, no atomics; a later kernel adds the slots up).  `scratch` = 512 B of LDS.
template <int NV>
__device__ __forceinline__ void block_sum_thread0(double (&vals)[NV], double* scratch, int tid) {
  static_assert(NV <= 4, "scratch holds 16 waves x 4 values");
  const int lane = tid & 63, wave = tid >> 6, nw = (blockDim.x + 63) >> 6;
  const int live = (int)blockDim.x - (wave << 6);        // lanes of this wave that exist (blocks of 32)
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double x = vals[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double y = __shfl_down(x, off, 64);
      if (lane + off < live) x += y;
    }
    vals[i] = x;
  }
  wg_barrier();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) scratch[wave * NV + i] = vals[i];
  }
  wg_barrier();
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double x = 0.0;
      for (int w = 0; w < nw; ++w) x += scratch[w * NV + i];
      vals[i] = x;                                        // totals: valid in thread 0 only
    }
  }
}
template <int NV>
__device__ __forceinline__ void block_sum_thread0(double (&vals)[NV], double* scratch) {
  block_sum_thread0<NV>(vals, scratch, (int)threadIdx.x);
}
// tid: the caller's copy of threadIdx.x (the even/odd row kernels pass a laundered one, so that the wave's scratch address is
// formed where it is used instead of being held -- or spilled -- across the whole row loop)
template <int NV>
__device__ __forceinline__ void block_sum_store(double (&vals)[NV], double* scratch, double* __restrict__ dst, int tid) {
  block_sum_thread0<NV>(vals, scratch, tid);
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) dst[i] = vals[i];
  }
}
template <int NV>
__device__ __forceinline__ void block_sum_store(double (&vals)[NV], double* scratch, double* __restrict__ dst) {
  block_sum_store<NV>(vals, scratch, dst, (int)threadIdx.x);
}

// The same for workgroups made of FULL waves (every row plan), with the in-wave sum on DPP lane permutations instead of
// ds_bpermute shuffles: a shuffle needs its source lane's address in a VGPR, six of them per reduction, all loop-invariant --
// in the persistent row kernels they were hoisted out of the row loop and spilled (the only scratch k_x_products<4096> had).
template <int CTRL>
__device__ __forceinline__ double dpp_lanes(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_full(double x) {
  x += dpp_lanes<0xB1>(x);          // quad_perm [1,0,3,2]
  x += dpp_lanes<0x4E>(x);          // quad_perm [2,3,0,1]
  x += dpp_lanes<0x141>(x);         // row_half_mirror
  x += dpp_lanes<0x140>(x);         // row_mirror: every lane of a 16-lane row now holds the row's sum
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    s += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 16 * r), __builtin_amdgcn_readlane(__double2loint(x), 16 * r));
  return s;                          // the wave's sum, in every lane
}
template <int NV>
__device__ __forceinline__ void block_sum_store_full_waves(double (&vals)[NV], double* scratch, double* __restrict__ dst, int tid) {
  static_assert(NV <= 4, "scratch holds 16 waves x 4 values");
  const int wave = tid >> 6, nw = (int)blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) vals[i] = wave_sum_full(vals[i]);
  wg_barrier();
  if ((tid & 63) == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) scratch[wave * NV + i] = vals[i];
  }
  wg_barrier();
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double x = 0.0;
      for (int w = 0; w < nw; ++w) x += scratch[w * NV + i];
      dst[i] = x;
    }
  }
}

// ---------------------------------------------------------------- x direction, generic
// mode 0: complex in (pitch_in) -> complex out; mode 1: multiply input by i*kk[kx] first
template <int N, bool INV>
__global__ void __launch_bounds__(XPlan<N>::THREADS)
k_x_c2c(const cd* __restrict__ in, cd* __restrict__ out, int pitch_in, int pitch_out, int nrows, double scale,
        const cd* __restrict__ tw, const double* __restrict__ kk, int mul_ik) {
  typedef XPlan<N> X;
  constexpr int P = X::P, T = X::T;
  const int j = threadIdx.x % T, c = threadIdx.x / T;
  const int row = blockIdx.x * X::C + c;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  typename XPlan<N>::F::Tw twr;
  XPlan<N>::F::load_tw(twr, j, tw, 1);
  cd r[P];
  const bool ok = row < nrows;
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const int kx = j + t * T;
    cd v = ok ? in[(size_t)row * pitch_in + kx] : cmake(0, 0);
    if (mul_ik) v = cscale(cmul_i(v), kk[kx]);
    r[t] = v;
  }
  X::F::template run<INV>(r, j, c, lds, twr);
  if (ok) {
#pragma unroll
    for (int t = 0; t < P; ++t) out[(size_t)row * pitch_out + j + t * T] = cscale(r[t], scale);
  }
}

// real rows -> half spectrum (N/2+1 entries per row)
template <int N>
__global__ void __launch_bounds__(XPlan<N>::THREADS)
k_x_r2c(const double* __restrict__ in, cd* __restrict__ out, int pitch_in, int pitch_out, int nrows,
        const cd* __restrict__ tw) {
  typedef XPlan<N> X;
  constexpr int P = X::P, T = X::T;
  const int j = threadIdx.x % T, c = threadIdx.x / T;
  const int row = blockIdx.x * X::C + c;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  typename XPlan<N>::F::Tw twr;
  XPlan<N>::F::load_tw(twr, j, tw, 1);
  cd r[P];
  const bool ok = row < nrows;
#pragma unroll
  for (int t = 0; t < P; ++t) r[t] = cmake(ok ? in[(size_t)row * pitch_in + j + t * T] : 0.0, 0.0);
  X::F::template run<false>(r, j, c, lds, twr);
  if (ok) {
#pragma unroll
    for (int t = 0; t <= P / 2; ++t) {
      const int kx = j + t * T;
      if (kx <= N / 2) out[(size_t)row * pitch_out + kx] = r[t];
    }
  }
}

// half-spectrum rows (mixed space) -> real rows; imaginary parts of kx = 0 and N/2 are ignored,
// exactly like numpy.fft.irfft.
template <int N>
__global__ void __launch_bounds__(XPlan<N>::THREADS)
k_x_c2r(const cd* __restrict__ in, double* __restrict__ out, int pitch_in, int pitch_out, int nrows, double scale,
        const cd* __restrict__ tw) {
  typedef XPlan<N> X;
  constexpr int P = X::P, T = X::T;
  const int j = threadIdx.x % T, c = threadIdx.x / T;
  const int row = blockIdx.x * X::C + c;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  typename XPlan<N>::F::Tw twr;
  XPlan<N>::F::load_tw(twr, j, tw, 1);
  cd r[P];
  const bool ok = row < nrows;
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const int kx = j + t * T;
    cd v = cmake(0, 0);
    if (ok) {
      if (kx <= N / 2) {
        v = in[(size_t)row * pitch_in + kx];
        if (kx == 0 || kx == N / 2) v.y = 0.0;
      } else {
        v = cconj(in[(size_t)row * pitch_in + (N - kx)]);
      }
    }
    r[t] = v;
  }
  X::F::template run<true>(r, j, c, lds, twr);
  if (ok) {
#pragma unroll
    for (int t = 0; t < P; ++t) out[(size_t)row * pitch_out + j + t * T] = r[t].x * scale;
  }
}

// ---------------------------------------------------------------- y direction, generic
struct ArrayList {        // up to 6 arrays processed by one launch (blockIdx.z selects)
  cd* ptr[6];
  int width[6];
  int pitch[6];
  static constexpr bool REDIR = false;
};
// Slab ranks, inside a step: the rows [own0, own1) of an exchange-group array are this rank's OWN block, which sits at the same
// offset on the X side (alt) as on the Y side (ptr).  The forward sub-pass, first consumer after an x -> y exchange, reads those
// rows from the X side; the inverse sub-pass, last producer before a y -> x exchange, writes them to the X side: the device copy
// of the own block that every exchange used to make is not needed.
struct ArrayListR : ArrayList {
  cd* alt[6];
  int own0, own1;
  static constexpr bool REDIR = true;
};

// "A" sub-pass, in place.  grid = (col tiles, S1, narrays); S2 = transform length.
template <int S2, bool INV, typename AL = ArrayList>
__global__ void __launch_bounds__(YPlan<S2>::THREADS)
k_y_A(AL al, int S1, const cd* __restrict__ tw, int tw_step_N /* NT/N */) {
  typedef YPlan<S2> Y;
  constexpr int P = Y::P, T = Y::T;
  const int c = threadIdx.x % CL, j = threadIdx.x / CL;
  cd* data = al.ptr[blockIdx.z];
  const int width = al.width[blockIdx.z], pitch = al.pitch[blockIdx.z];
  const int col = blockIdx.x * CL + c;
  if (blockIdx.x * CL >= width) return;            // whole tile outside this array (uniform)
  const int y1 = blockIdx.y;
  const bool ok = col < width;
  const int N = S1 * S2;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  typename Y::F::Tw twr;
  Y::F::load_tw(twr, j, tw, tw_step_N * (N / S2));
  cd r[P];
  cd* other = data;                                 // where the rank's own rows are read from (forward) / written to (inverse)
  int own0 = 0, own1 = 0;
  if constexpr (AL::REDIR) {
    other = al.alt[blockIdx.z];
    own0 = al.own0;
    own1 = al.own1;
  }
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const int i2 = j + t * T;                       // y2 (forward) or l1 (inverse)
    const int row = y1 + S1 * i2;
    const cd* src = data;
    if constexpr (AL::REDIR && !INV) src = (row >= own0 && row < own1) ? other : data;
    cd v = ok ? src[(size_t)row * pitch + col] : cmake(0, 0);
    if (INV) v = cmulc(v, tw[(size_t)(y1 * i2) * tw_step_N]);
    r[t] = v;
  }
  Y::F::template run<INV>(r, j, c, lds, twr);
  if (ok) {
#pragma unroll
    for (int t = 0; t < P; ++t) {
      const int o2 = j + t * T;                     // l1 (forward) or y2 (inverse)
      const int row = y1 + S1 * o2;
      cd v = r[t];
      if (!INV) v = cmul(v, tw[(size_t)(y1 * o2) * tw_step_N]);
      cd* dst = data;
      if constexpr (AL::REDIR && INV) dst = (row >= own0 && row < own1) ? other : data;
      dst[(size_t)row * pitch + col] = v;
    }
  }
}

// "B" sub-pass, out of place.  grid = (col tiles, S2, 1); S1 = transform length.
// forward: in = half-transformed rows S1*l1 + y1, out = natural rows l1 + S2*l2
// inverse: in = natural rows, out = half-transformed rows; `scale` applied on output.
template <int S1, bool INV, int CLX = CL>
__global__ void __launch_bounds__((YPlanT<S1, CLX>::THREADS))
k_y_B(const cd* __restrict__ in, cd* __restrict__ out, int width, int pitch_in, int pitch_out, int S2, double scale,
      const cd* __restrict__ tw, int tw_step_N) {
  typedef YPlanT<S1, CLX> Y;
  constexpr int P = Y::P, T = Y::T;
  const int c = threadIdx.x % CLX, j = threadIdx.x / CLX;
  const int col = blockIdx.x * CLX + c;
  const int l1 = blockIdx.y;
  const bool ok = col < width;
  const int N = S1 * S2;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  typename Y::F::Tw twr;
  Y::F::load_tw(twr, j, tw, tw_step_N * (N / S1));
  cd r[P];
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const int i = j + t * T;
    const size_t row = INV ? (size_t)(l1 + S2 * i) : (size_t)(l1 * S1 + i);
    r[t] = ok ? in[row * pitch_in + col] : cmake(0, 0);
  }
  Y::F::template run<INV>(r, j, c, lds, twr);
  if (ok) {
#pragma unroll
    for (int t = 0; t < P; ++t) {
      const int i = j + t * T;
      const size_t row = INV ? (size_t)(l1 * S1 + i) : (size_t)(l1 + S2 * i);
      out[row * pitch_out + col] = cscale(r[t], scale);
    }
  }
}

}  // namespace nq
