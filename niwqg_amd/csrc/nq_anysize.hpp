// niwqg_amd: the any-size engine -- grids the fused kernels have no plan for.
//
// The reference takes any nx (ref niwqg/Kernel.py:100-103; numpy.fft transforms any length, :562-566).  The fused step of
// nq_step.hpp exists for powers of two in [64, 8192]; for every other EVEN nx in [4, 4096] the model classes run the reference's
// own sequence of whole-plane operations (niwqg_amd/_anysize.py) on device planes through this engine:
//   * 1-D transforms of ANY length n along either axis of a (rows, cols) complex128 plane, numpy.fft conventions, by
//     Bluestein's chirp-z identity on top of the power-of-two row engine (WgFft / k_x_c2c of length M >= 2n - 1):
//         X[k] = w[k] sum_j (x[j] w[j]) conj(w[k - j]),   w[j] = exp(-i pi j^2 / n)
//     pack (chirp multiply, zero padding, transposing for axis 0) -> FFT_M -> multiply by the transformed chirp -> IFFT_M ->
//     unpack (chirp multiply); the inverse transform is conj(fft(conj x)) / n.  The chirp's angles are reduced exactly
//     (j^2 mod 2n in integers) and evaluated in long double on the host.
//   * element-wise operations and deterministic reductions on whole planes (every block writes its partial, one block adds them).
// This path trades speed for generality (a 2-D transform is ~10 plane passes over rows of length M ~ 2-4 n); it is HBM-bound
// streaming work like everything else here, no MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <string>
#include <vector>

#include "nq_generic.hpp"

namespace nq {

enum {  // element-wise operations (include/niwqg_amd.h: NQ_EW_*)
  EW_COPY = 0, EW_MUL = 1, EW_MULCONJ = 2, EW_AXPBY = 3, EW_AXPBYPCZ = 4, EW_REAL = 5, EW_ABS2 = 6, EW_SCALE = 7, EW_CONJ = 8,
  EW_ADDS = 9, EW_IMAG = 10, EW_MULADD = 11, EW_FILL = 12
};
enum { RD_SUM = 0, RD_SUMABS2 = 1, RD_DOT = 2, RD_DOTC = 3, RD_MAXABS = 4, RD_WSUMABS2 = 5, RD_MAXABSRE = 6 };

// scalars: s[0..1] = s0, s[2..3] = s1, s[4..5] = s2 (complex)
struct EwScalars { double s[6]; };

__global__ void __launch_bounds__(256) k_any_ew(int op, cd* __restrict__ d, const cd* __restrict__ a, const cd* __restrict__ b,
                                                const cd* __restrict__ c, size_t n, EwScalars sc) {
  const cd s0 = cmake(sc.s[0], sc.s[1]), s1 = cmake(sc.s[2], sc.s[3]), s2 = cmake(sc.s[4], sc.s[5]);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    cd r;
    switch (op) {
      case EW_COPY: r = a[i]; break;
      case EW_MUL: r = cmul(s0, cmul(a[i], b[i])); break;
      case EW_MULCONJ: r = cmul(s0, cmul(cconj(a[i]), b[i])); break;
      case EW_AXPBY: r = cadd(cmul(s0, a[i]), cmul(s1, b[i])); break;
      case EW_AXPBYPCZ: r = cadd(cadd(cmul(s0, a[i]), cmul(s1, b[i])), cmul(s2, c[i])); break;
      case EW_REAL: r = cmake(a[i].x, 0.0); break;
      case EW_IMAG: r = cmake(a[i].y, 0.0); break;
      case EW_ABS2: r = cmake(a[i].x * a[i].x + a[i].y * a[i].y, 0.0); break;
      case EW_SCALE: r = cmul(s0, a[i]); break;
      case EW_CONJ: r = cconj(a[i]); break;
      case EW_ADDS: r = cadd(a[i], s0); break;
      case EW_MULADD: r = cadd(cmul(s0, cmul(a[i], b[i])), cmul(s1, c[i])); break;      // s0 a b + s1 c
      case EW_FILL: r = s0; break;                                                       // (a is not read: no 0 * NaN)
      default: r = cmake(0, 0);
    }
    d[i] = r;
  }
}

// deterministic two-stage reduction: stage 1, every block its partial (2 doubles)
__global__ void __launch_bounds__(256) k_any_reduce1(int op, const cd* __restrict__ a, const cd* __restrict__ b, size_t n,
                                                     double* __restrict__ part) {
  double x = 0.0, y = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const cd v = a[i];
    switch (op) {
      case RD_SUM: x += v.x; y += v.y; break;
      case RD_SUMABS2: x += v.x * v.x + v.y * v.y; break;
      case RD_DOT: { const cd p = cmul(v, b[i]); x += p.x; y += p.y; } break;
      case RD_DOTC: { const cd p = cmul(cconj(v), b[i]); x += p.x; y += p.y; } break;
      case RD_MAXABS: x = fmax(x, sqrt(v.x * v.x + v.y * v.y)); break;
      case RD_MAXABSRE: x = fmax(x, fabs(v.x)); break;
      case RD_WSUMABS2: x += b[i].x * (v.x * v.x + v.y * v.y); break;
    }
  }
  __shared__ double sx[256], sy[256];
  sx[threadIdx.x] = x;
  sy[threadIdx.x] = y;
  __syncthreads();
  const bool mx = (op == RD_MAXABS || op == RD_MAXABSRE);
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sx[threadIdx.x] = mx ? fmax(sx[threadIdx.x], sx[threadIdx.x + s]) : sx[threadIdx.x] + sx[threadIdx.x + s];
      sy[threadIdx.x] += sy[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = sx[0];
    part[2 * blockIdx.x + 1] = sy[0];
  }
}
// stage 2: one workgroup, a fixed tree (thread t takes partials t, t + 256, ... in order, then a power-of-two tree in LDS): the
// same summation order on every run, a few microseconds instead of a 1024-long serial chain of dependent loads
__global__ void __launch_bounds__(256) k_any_reduce2(int op, const double* __restrict__ part, int nblocks, double* __restrict__ out) {
  const bool mx = (op == RD_MAXABS || op == RD_MAXABSRE);
  double x = 0.0, y = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) {
    x = mx ? fmax(x, part[2 * i]) : x + part[2 * i];
    y += part[2 * i + 1];
  }
  __shared__ double sx[256], sy[256];
  sx[threadIdx.x] = x;
  sy[threadIdx.x] = y;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sx[threadIdx.x] = mx ? fmax(sx[threadIdx.x], sx[threadIdx.x + s]) : sx[threadIdx.x] + sx[threadIdx.x + s];
      sy[threadIdx.x] += sy[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = sx[0];
    out[1] = sy[0];
  }
}

// ---- Bluestein: pack / unpack between a (rows, cols) plane and the work rows tmp[line][M] ------------------------------
// axis 1: line = row, element j at src[line * cols + j];  axis 0: line = column, element j at src[j * cols + line] -- moved in
// 16 x 16 tiles through LDS so that both sides are read and written along their contiguous index.
template <bool PACK>
__global__ void __launch_bounds__(256) k_any_lines(cd* __restrict__ plane, cd* __restrict__ tmp, int rows, int cols, int axis,
                                                   int M, const cd* __restrict__ chirp, int conj_io, double scale) {
  const int n = axis == 1 ? cols : rows, nlines = axis == 1 ? rows : cols;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int j0 = blockIdx.x * 16, l0 = blockIdx.y * 16;          // tile: elements j0.., lines l0..
  __shared__ cd tile[16][17];
  if (PACK) {
    if (axis == 1) {
      const int line = l0 + ty, j = j0 + tx;
      if (line < nlines && j < M) {
        cd v = cmake(0, 0);
        if (j < n) {
          v = plane[(size_t)line * cols + j];
          if (conj_io) v = cconj(v);
          if (chirp) v = cmul(v, chirp[j]);
        }
        tmp[(size_t)line * M + j] = v;
      }
    } else {
      const int j = j0 + ty, line = l0 + tx;
      cd v = cmake(0, 0);
      if (line < nlines && j < n) {
        v = plane[(size_t)j * cols + line];
        if (conj_io) v = cconj(v);
        if (chirp) v = cmul(v, chirp[j]);
      }
      tile[ty][tx] = v;
      __syncthreads();
      const int jo = j0 + tx, lo = l0 + ty;
      if (lo < nlines && jo < M) tmp[(size_t)lo * M + jo] = tile[tx][ty];
    }
  } else {
    if (axis == 1) {
      const int line = l0 + ty, j = j0 + tx;
      if (line < nlines && j < n) {
        cd v = tmp[(size_t)line * M + j];
        if (chirp) v = cmul(v, chirp[j]);
        v = cscale(v, scale);
        if (conj_io) v = cconj(v);
        plane[(size_t)line * cols + j] = v;
      }
    } else {
      const int ji = j0 + tx, li = l0 + ty;
      cd v = cmake(0, 0);
      if (li < nlines && ji < n) {
        v = tmp[(size_t)li * M + ji];
        if (chirp) v = cmul(v, chirp[ji]);
        v = cscale(v, scale);
      }
      tile[ty][tx] = v;
      __syncthreads();
      const int j = j0 + ty, line = l0 + tx;
      if (line < nlines && j < n) {
        cd w = tile[tx][ty];
        if (conj_io) w = cconj(w);
        plane[(size_t)j * cols + line] = w;
      }
    }
  }
}
__global__ void __launch_bounds__(256) k_any_mul_rows(cd* __restrict__ tmp, const cd* __restrict__ bhat, int nlines, int M) {
  const size_t n = (size_t)nlines * M;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    tmp[i] = cmul(tmp[i], bhat[i % M]);
}
// half spectrum (rows, n/2+1) -> full (rows, n) Hermitian extension: full[l, n-k] = conj(half[(rows-l) % rows, k]); `project` = 1
// takes the Hermitian part (in l) of the two self-mirrored columns first (what numpy.fft.irfft2 sees of them); `project` = 2 is the
// one-dimensional rule, row by row (the y transform already done): full[y, n-k] = conj(half[y, k]), imaginary parts of columns 0
// and n/2 dropped -- numpy.fft.irfft along x
__global__ void k_any_expand_half(const cd* __restrict__ half, cd* __restrict__ full, int rows, int n, int project) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y;
  if (k >= n) return;
  const int nh = n / 2 + 1, lm = (project == 2) ? l : (rows - l) % rows;
  cd v;
  if (k < nh) {
    v = half[(size_t)l * nh + k];
    if (project && (k == 0 || k == n / 2)) {
      const cd m = half[(size_t)lm * nh + k];
      v = cmake(0.5 * (v.x + m.x), 0.5 * (v.y - m.y));
    }
  } else {
    v = cconj(half[(size_t)lm * nh + (n - k)]);
  }
  full[(size_t)l * n + k] = v;
}
__global__ void k_any_take_cols(const cd* __restrict__ src, cd* __restrict__ dst, int rows, int scols, int dcols) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y;
  if (k < dcols) dst[(size_t)l * dcols + k] = src[(size_t)l * scols + k];
}
__global__ void k_any_set_elem(cd* p, size_t idx, double re, double im) { p[idx] = cmake(re, im); }

// Work rows longer than the row engine's longest plan (M = 16384 = 128 x 128): the four-step transform on the rows of tmp, every
// step a batched [R][C] -> [C][R] transpose per line through LDS (16 x 16 tiles) or a pass of the 128-point row engine:
//   x[a + A b]  --transpose-->  [a][b]  --FFT_B over b-->  Y[a][kb]  --x w_M^(a kb), transpose-->  [kb][a]  --FFT_A over a-->  Z[kb][ka]
//   --transpose-->  X[kb + B ka]
// tw: exp(-2 pi i m / M), m < M (a kb < M for a < A, kb < B); inv conjugates it.
__global__ void __launch_bounds__(256) k_any_btranspose(const cd* __restrict__ src, cd* __restrict__ dst, int R, int C,
                                                        const cd* __restrict__ tw, int use_tw, int inv) {
  __shared__ cd tile[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c0 = blockIdx.x * 16, r0 = blockIdx.y * 16;
  const size_t base = (size_t)blockIdx.z * R * C;
  const int r = r0 + ty, c = c0 + tx;
  cd v = cmake(0, 0);
  if (r < R && c < C) {
    v = src[base + (size_t)r * C + c];
    if (use_tw) {
      cd w = tw[r * c];
      if (inv) w.y = -w.y;
      v = cmul(v, w);
    }
  }
  tile[ty][tx] = v;
  __syncthreads();
  const int ro = r0 + tx, co = c0 + ty;
  if (ro < R && co < C) dst[base + (size_t)co * R + ro] = tile[tx][ty];
}

// Bluestein in ONE kernel for rows whose work length M the row engine takes in registers (M <= 8192): chirp multiply and zero
// padding on load, FFT_M, multiply by the transformed chirp, inverse FFT_M, chirp multiply on store -- the line never leaves the
// workgroup between the two transforms (five launches and five passes over rows of length M otherwise).  Rows are contiguous
// (axis 1); columns go through a transpose of the plane (nq_any_fft).
template <int M>
__global__ void __launch_bounds__(XPlan<M>::THREADS)
k_any_bluestein_rows(const cd* src, cd* dst, int nrows, int n, int pitch, const cd* __restrict__ chirp,
                     const cd* __restrict__ bhat, const cd* __restrict__ tw, int conj_io, double scale) {
  typedef XPlan<M> X;
  constexpr int P = X::P, T = X::T;
  const int j = threadIdx.x % T, c = threadIdx.x / T;
  const int row = blockIdx.x * X::C + c;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  typename X::F::Tw twr;
  X::F::load_tw(twr, j, tw, 1);
  const bool ok = row < nrows;
  cd r[P];
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const int k = j + t * T;
    cd v = cmake(0, 0);
    if (ok && k < n) {
      v = src[(size_t)row * pitch + k];
      if (conj_io) v = cconj(v);
      v = cmul(v, chirp[k]);
    }
    r[t] = v;
  }
  X::F::template run<false>(r, j, c, lds, twr);
#pragma unroll
  for (int t = 0; t < P; ++t) r[t] = cmul(r[t], bhat[j + t * T]);
  X::F::template run<true>(r, j, c, lds, twr);
  if (ok) {
#pragma unroll
    for (int t = 0; t < P; ++t) {
      const int k = j + t * T;
      if (k < n) {
        cd v = cscale(cmul(r[t], chirp[k]), scale);
        if (conj_io) v = cconj(v);
        dst[(size_t)row * pitch + k] = v;
      }
    }
  }
}

// Lengths n = R m with a small odd R (3 or 5) and a power of two m the row engine takes (64 <= m <= 2048) need no chirp: the R
// decimated sub-sequences x[R j + r] are transformed as R m-point problems held in registers side by side, and one radix-R
// butterfly with the twiddles w_n^(r k) combines them,
//     X[k + m s] = sum_r w_R^(r s) w_n^(r k) Y_r[k],     k < m, s < R
// (3 x FFT_1024 instead of 2 x FFT_8192 for n = 3072).  One kernel per line; the inverse is conj(fft(conj x)) / n.
// twn: exp(-2 pi i q / n), q < n;  twm: the m-point plan's table.
template <int M, int R>
__global__ void __launch_bounds__(XPlan<M>::THREADS)
k_any_split_rows(const cd* src, cd* dst, int nrows, int pitch, const cd* __restrict__ twn, const cd* __restrict__ twm, int conj_io,
                 double scale) {
  typedef XPlan<M> X;
  constexpr int P = X::P, T = X::T;
  const int j = threadIdx.x % T, c = threadIdx.x / T;
  const int row = blockIdx.x * X::C + c;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  typename X::F::Tw twr;
  X::F::load_tw(twr, j, twm, 1);
  const bool ok = row < nrows;
  cd y[R][P];
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int t = 0; t < P; ++t) {
      cd v = cmake(0, 0);
      if (ok) {
        v = src[(size_t)row * pitch + (size_t)R * (j + t * T) + r];
        if (conj_io) v = cconj(v);
      }
      y[r][t] = v;
    }
    X::F::template run<false>(y[r], j, c, lds, twr);
  }
  if (!ok) return;
  // w_R^q, q < R (forward sign)
  cd wr[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    double sn, cs;
    sincospi(-2.0 * (double)q / (double)R, &sn, &cs);
    wr[q] = cmake(cs, sn);
  }
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const int k = j + t * T;
    cd z[R];
    z[0] = y[0][t];
#pragma unroll
    for (int r = 1; r < R; ++r) z[r] = cmul(y[r][t], twn[(size_t)r * k]);        // r k < R m = n
#pragma unroll
    for (int sidx = 0; sidx < R; ++sidx) {
      cd acc = z[0];
#pragma unroll
      for (int r = 1; r < R; ++r) acc = cadd(acc, cmul(z[r], wr[(r * sidx) % R]));
      acc = cscale(acc, scale);
      if (conj_io) acc = cconj(acc);
      dst[(size_t)row * pitch + k + (size_t)M * sidx] = acc;
    }
  }
}

}  // namespace nq

struct nq_any {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  std::vector<void*> allocs;
  long long bytes = 0;
  // Bluestein plans by transform length n
  struct Plan {
    int n = 0, M = 0;
    bool direct = false;                 // n itself is a power of two the row engine (or its four-step form) takes: no chirp
    int split = 0;                       // n = split * M with split = 3 or 5 and M a power of two: k_any_split_rows, no chirp
    nq::cd *chirp = nullptr, *bhat = nullptr, *tw = nullptr;
    nq::cd* tw_small = nullptr;          // M = 16384: the twiddles of the 128-point passes of the four-step transform
  };
  std::vector<Plan> plans;
  nq::cd *tmp = nullptr, *tmp2 = nullptr;      // work rows [line][M]; tmp2 only for the four-step transform
  size_t tmp_elems = 0, tmp2_elems = 0;
  double *part = nullptr, *red = nullptr;
  double* red_host = nullptr;
};
