// Fused kernels of one ETDRK4 stage.  See DESIGN.md for the data flow; in short, per stage
//
//   X2  rows : Mu,Mp,Mq,Mqw,Mphi,Mphiy --x-ifft--> u,v,q,qw,phi,phix,phiy --products--x-fft--> Muq,Mvq,Mw
//   A   (generic, in place) on Muq,Mvq,Mw
//   Sq  tiles: B-fft of Muq,Mvq -> N_q -> ETDRK4 stage update of qh               (half spectrum)
//   Sw  tiles: B-fft of Mw      -> N_phi -> ETDRK4 stage update of phih -> B-ifft -> Hphi,Hphiy
//   A^-1 on Hphi,Hphiy
//   X1  rows : Mphi,Mphiy --x-ifft--> phi,phix,phiy --|phi|^2, J(phi*,phi)--x-fft--> Ma,Mb   (Coupled)
//   A   on Ma,Mb
//   Si  tiles: B-fft of Ma,Mb -> qwh, ph -> B-ifft -> Hu,Hp,Hq,Hqw                 (inversion)
//   A^-1 on Hu,Hp,Hq,Hqw
//
// "M*" arrays live in mixed space [y][kx] (x spectral, y physical); inverse-direction ones carry the
// full 1/(nx*ny) factor.  Spectra of real fields are half spectra: kx = 0..N/2, pitch Ph.
#pragma once
#include "nq_generic.hpp"

namespace nq {

enum { MODE_COUPLED = 0, MODE_UNCOUPLED = 1, MODE_QG = 2,
       MODE_QGC = 3 /* QGModel with its passive scalar c (ref QGModel.py:345-404, :483-495): c rides in the qw slots */ };

// ---- slab decomposition: layouts --------------------------------------------
// With P ranks, physical / mixed-space rows are split by y (N/P local rows, "X side") and spectral /
// mixed-space columns by kx (W columns per rank, "Y side").  Arrays that travel together between the two
// sides form an exchange GROUP: one buffer per side whose rows interleave the group's arrays
// (row = [array0 segment | array1 segment | ...], `pitch` elements).  On the X side the buffer is cut into P
// blocks, block d holding the columns owned by rank d for every local row -- exactly the send (or receive)
// layout of all_to_all_single; on the Y side the P received blocks are simply rows [0, N) of a column slab.
// So element kx of local row y of one array sits at
//     X side: xs + (kx / W) * blk + y * pitch + kx % W          Y side: ys + y_global * pitch + k_local
// and nothing is ever packed or unpacked.  P = 1 is the same code with W = full width (block 0 only).
struct MArr {
  cd* xs;            // X-side base, column offset of this array inside the group's rows included
  cd* ys;            // Y-side base, idem (== xs when P == 1)
  int pitch;         // row pitch of the group (elements)
  int W;             // columns of this array owned by one rank
  int shift;         // kx / W = kx >> shift when >= 0 (power-of-two W, or 30 when P == 1)
  unsigned magic;    // else kx / W = (kx * magic) >> 24, magic = ceil(2^24 / W); exact while kx * W < 2^24 (nq_create checks every column)
  long long blk;     // X-side block stride = local rows * pitch
};
// One local row of an MArr on the X side.  SLAB = false (one rank: a single block) makes at() a plain `p + kx`: the
// block arithmetic costs eight integer instructions per access, three of them quarter-rate 32-bit multiplies, and the
// row kernels make ~90 accesses per row -- a quarter of their VALU time (profiles/r02 notes in DESIGN.md section 4).
template <bool SLAB>
struct XRowT {
  cd* p;
  int W, shift;
  unsigned magic;
  long long blk;
  __device__ __forceinline__ cd* at(int kx) const {
    if constexpr (!SLAB) return p + kx;
    const int b = shift >= 0 ? (kx >> shift) : (int)(((unsigned)kx * magic) >> 24);
    return p + (long long)b * blk + (kx - b * W);
  }
};
template <bool SLAB>
__device__ __forceinline__ XRowT<SLAB> xrow(const MArr& a, size_t row) {
  XRowT<SLAB> r;
  r.p = a.xs + row * (size_t)a.pitch;
  if constexpr (SLAB) {
    r.W = a.W;
    r.shift = a.shift;
    r.magic = a.magic;
    r.blk = a.blk;
  }
  return r;
}
// The same, with the block geometry (W, shift, magic) read from another array of the same width class: a kernel that walks
// ten arrays then holds two copies of it in SGPRs instead of ten (the slab instantiation of k_x_products_eo spilled 41 SGPRs).
template <bool SLAB>
__device__ __forceinline__ XRowT<SLAB> xrow(const MArr& a, const MArr& geom, size_t row) {
  XRowT<SLAB> r;
  r.p = a.xs + row * (size_t)a.pitch;
  if constexpr (SLAB) {
    r.W = geom.W;
    r.shift = geom.shift;
    r.magic = geom.magic;
    r.blk = a.blk;
  }
  return r;
}
// Column-slab geometry of the spectral (Y-side) kernels.
struct YGeom {
  int k0;            // global column index of local column 0
  int width;         // valid local columns
  int pitch_s;       // row pitch of the local spectral planes (state, coefficients, filter)
  int S2;            // N = S1 * S2
  int kernel_family; // 1: Kernel family (c2c semantics of the reference), 0: QGModel (rfft semantics)
  int cmirror;       // 1: the ETDRK4 coefficient planes hold rows l = 0..N/2 only (row N-l is bit-identical: c, the filter and the
                     //    contour patches depend on l through l^2), indexed through crow(); 0: all N rows
};
// Row of a coefficient plane that holds the values of spectral row l.
__device__ __forceinline__ int crow(const YGeom& g, int l, int N) { return (g.cmirror && l > N / 2) ? N - l : l; }
// Order in which the B-sub-pass workgroups take the S2 residues l1 = l mod S2: 0, S2/2, then the pairs (m, S2 - m).  The rows of
// residue S2 - m are the mirrors N - l of the rows of residue m, i.e. the SAME coefficient rows: launched back to back (consecutive
// blockIdx.y: linear workgroup ids that differ by gridDim.x, a multiple of 8, hence the same XCD and its L2) the second read of a
// coefficient line is served on chip instead of from HBM.
__device__ __forceinline__ int pair_order(int y, int S2) {
  if (S2 < 4) return y;
  if (y < 2) return y == 0 ? 0 : S2 / 2;
  const int m = y >> 1;
  return (y & 1) ? S2 - m : m;
}


// Keeps hipcc from hoisting the next phase's global loads (and interleaving independent FFTs) across a
// phase boundary of the fused row kernels: that inflates the live set past 256 VGPRs and spills.
#define NQ_PHASE_FENCE()                      \
  do {                                        \
    __builtin_amdgcn_sched_barrier(0);        \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);        \
  } while (0)

// ---- helpers for the row kernels ----------------------------------------------------------
// Half-spectrum row pair held in registers between its (prefetched) load and its use.
template <int P> struct HsRegs {
  cd a[P / 2], b[P / 2], an, bn;      // elements m = j + t*T (t < P/2) and, for thread j = 0, m = N/2
};
template <int N, int P, int T, bool PAIR, typename Row>
__device__ __forceinline__ void hs_load(HsRegs<P>& r, const Row& rowA, const Row& rowB, int j) {
#pragma unroll
  for (int t = 0; t < P / 2; ++t) {
    r.a[t] = *rowA.at(j + t * T);
    r.b[t] = PAIR ? *rowB.at(j + t * T) : cmake(0, 0);
  }
  r.an = cmake(0, 0);
  r.bn = cmake(0, 0);
  if (j == 0) {
    r.an = *rowA.at(N / 2);
    if (PAIR) r.bn = *rowB.at(N / 2);
  }
}
// Build Z = A + i*B over the full row from the registers: every half-spectrum element was fetched from
// global memory ONCE; thread (j, t < P/2) keeps Z[m] and hands conj(A[m]) + i*conj(B[m]) = Z[N-m] to the
// owner of position N-m through LDS (the double fetch was 0.5 GB per launch, profiles/r01_c_pmc.txt).
template <int N, int P, int T, typename F, bool PAIR>
__device__ __forceinline__ void hs_pack(cd (&w)[P], const HsRegs<P>& r, int j, int c, cd* lds,
                                        const double* __restrict__ kk, bool b_mul_ik, bool b_zero_nyq) {
#pragma unroll
  for (int t = 0; t < P / 2; ++t) {
    const int m = j + t * T;
    cd a = r.a[t], b = r.b[t];
    if (PAIR && b_mul_ik) b = cscale(cmul_i(b), kk[m]);
    if (m == 0) {
      a.y = 0.0;
      b.y = 0.0;
    }
    w[t] = cmake(a.x - b.y, a.y + b.x);
    if (m != 0) lds[F::lds_index(m, c)] = cmake(a.x + b.y, b.x - a.y);      // conj(a) + i conj(b)
  }
  if (j == 0) {                                                             // self-mirrored m = N/2
    cd a = r.an, b = r.bn;
    if (PAIR && b_mul_ik) b = cscale(cmul_i(b), kk[N / 2]);
    a.y = 0.0;
    b.y = 0.0;
    if (b_zero_nyq) b.x = 0.0;
    w[P / 2] = cmake(a.x - b.y, a.y + b.x);
  }
  wg_barrier();
#pragma unroll
  for (int t = P / 2; t < P; ++t) {
    const int kx = j + t * T;
    if (kx > N / 2) w[t] = lds[F::lds_index(N - kx, c)];
  }
  wg_barrier();
}

// max of two non-negative doubles over one ROW (T threads), through the row's two LDS words mx[0..1] (zeroed, barrier
// before).  Rows of at least one wave reduce inside the wave first: 512 lanes hitting ONE LDS address serialise
// (9.3 % of the LDS cycles of k_x_wavepv2 were bank conflicts from exactly that, profiles/r01_pmc_summary.json).
template <int T>
__device__ __forceinline__ void row_atomic_max(unsigned long long* mx, double ma, double mb) {
  if constexpr (T >= 64) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      ma = fmax(ma, __shfl_xor(ma, off, 64));
      mb = fmax(mb, __shfl_xor(mb, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMax(&mx[0], (unsigned long long)__double_as_longlong(ma));
      atomicMax(&mx[1], (unsigned long long)__double_as_longlong(mb));
    }
  } else {
    atomicMax(&mx[0], (unsigned long long)__double_as_longlong(ma));
    atomicMax(&mx[1], (unsigned long long)__double_as_longlong(mb));
  }
}

// After a forward row FFT of z = a + i*b (a, b real), split into the two half spectra and store
// kx = 0..N/2.  Needs the mirrored element Z[N-kx], fetched through LDS.
template <int N, int P, int T, typename F, typename Row>
__device__ __forceinline__ void unpack_pair_store(cd (&r)[P], int j, int c, cd* lds, const Row& rowA,
                                                  const Row& rowB, double scaleB = 1.0) {
  wg_barrier();
#pragma unroll
  for (int t = 0; t < P; ++t) lds[F::lds_index(j + t * T, c)] = r[t];
  wg_barrier();
#pragma unroll
  for (int t = 0; t <= P / 2; ++t) {
    const int kx = j + t * T;
    if (kx <= N / 2) {
      const cd z = r[t];
      const cd zm = lds[F::lds_index((N - kx) % N, c)];
      // A = (Z + conj Zm)/2 ; B = (Z - conj Zm)/(2i)
      *rowA.at(kx) = cmake(0.5 * (z.x + zm.x), 0.5 * (z.y - zm.y));
      *rowB.at(kx) = cmake(scaleB * 0.5 * (z.y + zm.y), scaleB * 0.5 * (zm.x - z.x));
    }
  }
  wg_barrier();
}

// ---- X1: wave potential-vorticity sources (CoupledModel._invert, ref CoupledModel.py:59-88) ------
// The spectral row of phi is fetched once and kept (sp) for the second transform (phix = ifft(ik phi)).
template <int N, bool SLAB>
__global__ void __launch_bounds__(XPlan1<N>::THREADS, XPlan1<N>::MIN_WAVES)
k_x_wavepv(MArr Mphi, MArr Mphiy, MArr Ma, MArr Mb, const cd* __restrict__ tw, const double* __restrict__ kk) {
  typedef XPlan1<N> X;
  constexpr int P = X::P, T = X::T;
  const int j = threadIdx.x % T, c = threadIdx.x / T;
  const size_t row = (size_t)blockIdx.x * X::C + c;
  const XRowT<SLAB> rphi = xrow<SLAB>(Mphi, row), rphiy = xrow<SLAB>(Mphiy, row);
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  // stage twiddles: table in LDS behind the exchange area (`tw` = host-built stage table here)
  cd* twl = lds + XPlan1<N>::F::LDS_ELEMS;
  for (int i = threadIdx.x; i < XPlan1<N>::F::TW_LDS_ELEMS; i += XPlan1<N>::THREADS) twl[i] = tw[i];
  typename XPlan1<N>::F::TwLds twr;
  twr.base = twl;
  wg_barrier_all();
  cd w[P], gx[P], py[P];
  double a[P];
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const int kx = j + t * T;
    w[t] = *rphi.at(kx);
    gx[t] = cscale(cmul_i(w[t]), kk[kx]);
  }
  constexpr bool PREFETCH = (P <= 8);       // with 16 points/thread the extra 64 VGPRs would spill
  if (PREFETCH) {
#pragma unroll
    for (int t = 0; t < P; ++t) py[t] = *rphiy.at(j + t * T);      // in flight during the next two transforms
  }
  NQ_PHASE_FENCE();
  X::F::template run<true>(w, j, c, lds, twr);
  double ma = 0.0, mb = 0.0;
#pragma unroll
  for (int t = 0; t < P; ++t) {
    a[t] = w[t].x * w[t].x + w[t].y * w[t].y;
    ma = fmax(ma, a[t]);
  }
  NQ_PHASE_FENCE();
  X::F::template run<true>(gx, j, c, lds, twr);
#pragma unroll
  for (int t = 0; t < P; ++t) w[t] = PREFETCH ? py[t] : *rphiy.at(j + t * T);
  NQ_PHASE_FENCE();
  X::F::template run<true>(w, j, c, lds, twr);
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const double b = -2.0 * (gx[t].x * w[t].y - gx[t].y * w[t].x);
    mb = fmax(mb, fabs(b));
    w[t] = cmake(a[t], b);
  }
  // The two real fields share one complex transform; |phi|^2 is typically 1e8 times larger than
  // J(phi*,phi), so the second is rescaled per row by a power of two (exactly undone after the
  // split) to keep the roundoff of one from swamping the other.
  unsigned long long* mx = reinterpret_cast<unsigned long long*>(lds + X::F::LDS_ELEMS + X::F::TW_LDS_ELEMS) + 2 * c;
  if (j == 0) {
    mx[0] = 0ull;
    mx[1] = 0ull;
  }
  wg_barrier();
  row_atomic_max<T>(mx, ma, mb);
  wg_barrier();
  ma = __longlong_as_double((long long)mx[0]);
  mb = __longlong_as_double((long long)mx[1]);
  int e = 0;
  if (ma > 0.0 && mb > 0.0) e = ilogb(ma) - ilogb(mb);
  e = e > 900 ? 900 : (e < -900 ? -900 : e);
  const double sb = ldexp(1.0, e), isb = ldexp(1.0, -e);
#pragma unroll
  for (int t = 0; t < P; ++t) w[t].y *= sb;
  NQ_PHASE_FENCE();
  X::F::template run<false>(w, j, c, lds, twr);
  NQ_PHASE_FENCE();
  unpack_pair_store<N, P, T, typename X::F>(w, j, c, lds, xrow<SLAB>(Ma, row), xrow<SLAB>(Mb, row), isb);
}

// Variant for long rows: 8 points per thread, ONE workgroup per CU, the two transforms that share an input (phi and
// phix = ifft(ik phi)) in flight together (WgFft::run2: the LDS stores of one drain under the butterflies of the
// other), phiy prefetched under them.  No spills (k_x_wavepv at 16 points per thread spills 32 VGPRs at 4096:
// +0.33 GB of scratch traffic per launch, profiles/r01_pmc_summary.json).
template <int N, bool SLAB>
__global__ void __launch_bounds__(XPlan<N>::THREADS, XPlan<N>::MIN_WAVES)
k_x_wavepv2(MArr Mphi, MArr Mphiy, MArr Ma, MArr Mb, const cd* __restrict__ tw, const double* __restrict__ kk,
            int nblocks) {
  typedef XPlan<N> X;
  typedef typename X::F F;
  constexpr int P = X::P, T = X::T;
  const int j_tid = threadIdx.x % T, c_tid = threadIdx.x / T;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  cd* lds_b = lds + F::LDS_ELEMS;                       // second exchange area
  cd* twl = lds_b + F::LDS_ELEMS;
  for (int i = threadIdx.x; i < F::TW_LDS_ELEMS; i += X::THREADS) twl[i] = tw[i];
  typename F::TwLds twr;
  twr.base = twl;
  wg_barrier_all();
  // persistent: row blocks rb = blockIdx.x, +gridDim.x, ...; the next block's phi row is requested before the
  // forward transform of this one, so its latency hides behind that transform and the stores
  cd nxt[P];
  {
    const XRowT<SLAB> r0 = xrow<SLAB>(Mphi, (size_t)blockIdx.x * X::C + c_tid);
#pragma unroll
    for (int t = 0; t < P; ++t) nxt[t] = *r0.at(j_tid + t * T);
  }
  for (int rb = blockIdx.x; rb < nblocks; rb += gridDim.x) {
    // per-iteration copies the compiler cannot see through: otherwise every LDS / row address of the transforms is
    // loop-invariant, gets hoisted out of the loop and spills
    int j = j_tid, c = c_tid;
    asm volatile("" : "+v"(j), "+v"(c));
    const size_t row = (size_t)rb * X::C + c;
    const bool more = rb + (int)gridDim.x < nblocks;
    const XRowT<SLAB> rphiy = xrow<SLAB>(Mphiy, row);
    cd w[P], gx[P], py[P];
    double a[P];
#pragma unroll
    for (int t = 0; t < P; ++t) {
      const int kx = j + t * T;
      w[t] = nxt[t];
      gx[t] = cscale(cmul_i(w[t]), kk[kx]);
    }
#pragma unroll
    for (int t = 0; t < P; ++t) py[t] = *rphiy.at(j + t * T);
    NQ_PHASE_FENCE();
    F::template run2<true>(w, gx, j, c, lds, lds_b, twr);
    double ma = 0.0, mb = 0.0;
#pragma unroll
    for (int t = 0; t < P; ++t) {
      a[t] = w[t].x * w[t].x + w[t].y * w[t].y;
      ma = fmax(ma, a[t]);
    }
    NQ_PHASE_FENCE();
    F::template run<true>(py, j, c, lds, twr);
#pragma unroll
    for (int t = 0; t < P; ++t) {
      const double b = -2.0 * (gx[t].x * py[t].y - gx[t].y * py[t].x);
      mb = fmax(mb, fabs(b));
      w[t] = cmake(a[t], b);
    }
    if (more) {
      const XRowT<SLAB> rn = xrow<SLAB>(Mphi, row + (size_t)gridDim.x * X::C);
#pragma unroll
      for (int t = 0; t < P; ++t) nxt[t] = *rn.at(j + t * T);
    }
    unsigned long long* mx = reinterpret_cast<unsigned long long*>(twl + F::TW_LDS_ELEMS) + 2 * c;
    if (j == 0) {
      unsigned long long zero = 0ull;                    // formed here: held across the loop it is four VGPRs of zeros, spilled
      asm volatile("" : "+v"(zero));
      mx[0] = zero;
      mx[1] = zero;
    }
    wg_barrier();
    row_atomic_max<T>(mx, ma, mb);
    wg_barrier();
    ma = __longlong_as_double((long long)mx[0]);
    mb = __longlong_as_double((long long)mx[1]);
    int e = 0;
    if (ma > 0.0 && mb > 0.0) e = ilogb(ma) - ilogb(mb);
    e = e > 900 ? 900 : (e < -900 ? -900 : e);
    const double sb = ldexp(1.0, e), isb = ldexp(1.0, -e);
#pragma unroll
    for (int t = 0; t < P; ++t) w[t].y *= sb;
    NQ_PHASE_FENCE();
    F::template run<false>(w, j, c, lds, twr);
    NQ_PHASE_FENCE();
    unpack_pair_store<N, P, T, F>(w, j, c, lds, xrow<SLAB>(Ma, row), xrow<SLAB>(Mb, row), isb);
  }
}

// ---- X2: all nonlinear products of one stage ------------------------------------------------
// ref Kernel.py:471-486 (jacobian_psi_q), :457-469 (jacobian_psi_phi), :332 (refraction).  The budget terms
// gamma1+gamma2 and xi1+xi2 (Kernel.py:691-700) are Parseval sums against the transformed phi tendency in k_s_phi:
// this kernel has no budget work and lap(phi) never comes to physical space.
// MODE_QG: only Muq, Mvq.  MODE_QGC: the passive scalar c arrives paired with q (Mqw slot) and the products u c, v c leave
// through Mgx, Mgy (= the Muc, Mvc half-spectrum arrays).  MODE_UNCOUPLED: q_psi = q, phix/phiy from the (possibly stale) Mgx/Mgy.
// Register plan: q, q_psi, u, v are reals (32 VGPRs each); complex working sets are 64.
template <int N, int MODE, bool SLAB>
__global__ void __launch_bounds__(XPlan<N>::THREADS, XPlan<N>::MIN_WAVES)
k_x_products(MArr Mu, MArr Mp, MArr Mq, MArr Mqw, MArr Mphi, MArr Mgx, MArr Mgy, MArr Muq, MArr Mvq, MArr Mw,
             const cd* __restrict__ tw, const double* __restrict__ kk, int v_zero_nyq, double cj, double cr,
             int nblocks) {
  typedef XPlan<N> X;
  typedef typename X::F F;
  constexpr int P = X::P, T = X::T;
  const int j_tid = threadIdx.x % T, c_tid = threadIdx.x / T;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  // stage twiddles: table in LDS behind the exchange area (`tw` = host-built stage table here)
  cd* twl = lds + XPlan<N>::F::LDS_ELEMS;
  for (int i = threadIdx.x; i < XPlan<N>::F::TW_LDS_ELEMS; i += XPlan<N>::THREADS) twl[i] = tw[i];
  typename XPlan<N>::F::TwLds twr;
  twr.base = twl;
  wg_barrier_all();
  // Software pipeline: the rows of the NEXT phase are requested before each transform and consumed after
  // it (barriers inside the transforms no longer wait for global memory), so with one workgroup per CU
  // the HBM latency hides behind the FFTs.  Every mixed-space input is fetched exactly once.
  // The workgroup is persistent over row blocks rb = blockIdx.x, +gridDim.x, ...; the first inputs of its next block
  // are requested before the last transform of the current one.
  constexpr bool PAIRQ = (MODE == MODE_COUPLED || MODE == MODE_QGC);     // q travels paired with qw (or with c)
  constexpr bool ONLYQ = (MODE == MODE_QG || MODE == MODE_QGC);           // no wave field
  HsRegs<P> h1;
  {
    const size_t row0 = (size_t)blockIdx.x * X::C + c_tid;
    hs_load<N, P, T, PAIRQ>(h1, xrow<SLAB>(Mq, row0), xrow<SLAB>(PAIRQ ? Mqw : Mq, row0), j_tid);
  }
  for (int rb = blockIdx.x; rb < nblocks; rb += gridDim.x) {
  // per-iteration copies the compiler cannot see through: otherwise every LDS / row address of the transforms is
  // loop-invariant, gets hoisted out of the loop and spills
  int j = j_tid, c = c_tid;
  asm volatile("" : "+v"(j), "+v"(c));
  const size_t row = (size_t)rb * X::C + c;
  const bool more = rb + (int)gridDim.x < nblocks;
  const size_t row_next = more ? row + (size_t)gridDim.x * X::C : row;
  cd w[P];
  double q[P], qpsi[P], u[P], v[P];
  HsRegs<P> h2;
  hs_load<N, P, T, true>(h2, xrow<SLAB>(Mu, row), xrow<SLAB>(Mp, row), j);
  NQ_PHASE_FENCE();
  double c_unscale = 1.0;
  if constexpr (MODE == MODE_QGC) {
    // q and the passive scalar share one complex transform, but c has arbitrary units (|c| ~ 1 against |q| ~ 1e-5 in
    // the reference's examples): the roundoff of the larger would swamp the smaller (2e-11 in q after 20 steps).
    // Rescale c per row by a power of two, exactly undone after the transform (as k_x_wavepv does for its pair).
    double ma = 0.0, mb = 0.0;
#pragma unroll
    for (int t = 0; t < P / 2; ++t) {
      ma = fmax(ma, fmax(fabs(h1.a[t].x), fabs(h1.a[t].y)));
      mb = fmax(mb, fmax(fabs(h1.b[t].x), fabs(h1.b[t].y)));
    }
    unsigned long long* mx = reinterpret_cast<unsigned long long*>(nq_smem + X::LDS_BYTES - 512) + 2 * c;
    if (j == 0) {
      unsigned long long zero = 0ull;                    // formed here: held across the loop it is four VGPRs of zeros, spilled
      asm volatile("" : "+v"(zero));
      mx[0] = zero;
      mx[1] = zero;
    }
    wg_barrier();
    row_atomic_max<T>(mx, ma, mb);
    wg_barrier();
    ma = __longlong_as_double((long long)mx[0]);
    mb = __longlong_as_double((long long)mx[1]);
    int e = 0;
    if (ma > 0.0 && mb > 0.0) e = ilogb(ma) - ilogb(mb);
    e = e > 900 ? 900 : (e < -900 ? -900 : e);
    const double sb = ldexp(1.0, e);
    c_unscale = ldexp(1.0, -e);
#pragma unroll
    for (int t = 0; t < P / 2; ++t) h1.b[t] = cscale(h1.b[t], sb);
    h1.bn = cscale(h1.bn, sb);
    wg_barrier();
  }
  hs_pack<N, P, T, F, PAIRQ>(w, h1, j, c, lds, kk, false, false);
  NQ_PHASE_FENCE();
  F::template run<true>(w, j, c, lds, twr);
#pragma unroll
  for (int t = 0; t < P; ++t) {
    q[t] = w[t].x;
    qpsi[t] = (MODE == MODE_COUPLED) ? w[t].x - w[t].y : (MODE == MODE_QGC ? w[t].y * c_unscale : w[t].x);   // QGC: c
  }
  // (u, v) = ifft of (-il psi, ik psi): Mu already holds T_y^-1[-il psi], Mp holds T_y^-1[psi]
  NQ_PHASE_FENCE();
  hs_pack<N, P, T, F, true>(w, h2, j, c, lds, kk, true, v_zero_nyq != 0);
  cd sp[P];          // spectral row of phi: kept for phix (Coupled)
  if (!ONLYQ) {
    const XRowT<SLAB> rp = xrow<SLAB>(Mphi, row);
#pragma unroll
    for (int t = 0; t < P; ++t) sp[t] = *rp.at(j + t * T);
  }
  NQ_PHASE_FENCE();
  F::template run<true>(w, j, c, lds, twr);
#pragma unroll
  for (int t = 0; t < P; ++t) {
    u[t] = w[t].x;
    v[t] = w[t].y;
    w[t] = cmake(u[t] * q[t], v[t] * q[t]);                  // u q + i v q
  }
  NQ_PHASE_FENCE();
  F::template run<false>(w, j, c, lds, twr);
  NQ_PHASE_FENCE();
  unpack_pair_store<N, P, T, F>(w, j, c, lds, xrow<SLAB>(Muq, row), xrow<SLAB>(Mvq, row));
  if (ONLYQ) {
    if (MODE == MODE_QGC) {          // ik F[u c] + il F[v c] needs F[u c], F[v c]  (ref QGModel.py:483-495)
#pragma unroll
      for (int t = 0; t < P; ++t) w[t] = cmake(u[t] * qpsi[t], v[t] * qpsi[t]);
      NQ_PHASE_FENCE();
      F::template run<false>(w, j, c, lds, twr);
      NQ_PHASE_FENCE();
      unpack_pair_store<N, P, T, F>(w, j, c, lds, xrow<SLAB>(Mgx, row), xrow<SLAB>(Mgy, row));
    }
    if (more) hs_load<N, P, T, PAIRQ>(h1, xrow<SLAB>(Mq, row_next), xrow<SLAB>(PAIRQ ? Mqw : Mq, row_next), j);
    continue;
  }
  // phi tendency source in ONE array: W = cj (u phix + v phiy) + i cr phi q_psi  (cj = -1, cr = -1/2 in a step:
  // N_phi = F[W] except at [0,0], where the reference zeroes the Jacobian part only, ref Kernel.py:468 vs :332;
  // the row sums of the Jacobian part travel in a padding column of Muq and are added back in k_s_phi)
  cd pre[P];         // prefetch buffer: Mgy, in flight during the next two transforms
  {
    const XRowT<SLAB> rgy = xrow<SLAB>(Mgy, row);
#pragma unroll
    for (int t = 0; t < P; ++t) {
      w[t] = sp[t];
      pre[t] = *rgy.at(j + t * T);
    }
  }
  NQ_PHASE_FENCE();
  F::template run<true>(w, j, c, lds, twr);
  cd acc[P];
#pragma unroll
  for (int t = 0; t < P; ++t) acc[t] = cmake(-cr * qpsi[t] * w[t].y, cr * qpsi[t] * w[t].x);   // i cr phi q_psi
  {
    const XRowT<SLAB> rgx = xrow<SLAB>(Mgx, row);
#pragma unroll
    for (int t = 0; t < P; ++t) {
      const int kx = j + t * T;
      const cd g = (MODE == MODE_COUPLED) ? sp[t] : *rgx.at(kx);
      w[t] = cscale(cmul_i(g), kk[kx]);
    }
  }
  NQ_PHASE_FENCE();
  F::template run<true>(w, j, c, lds, twr);
#pragma unroll
  for (int t = 0; t < P; ++t) w[t] = cscale(w[t], u[t]);
  NQ_PHASE_FENCE();
  F::template run<true>(pre, j, c, lds, twr);
  double js[2] = {0.0, 0.0};
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const cd J = cmake(w[t].x + pre[t].x * v[t], w[t].y + pre[t].y * v[t]);
    js[0] += J.x;
    js[1] += J.y;
    acc[t] = cmake(acc[t].x + cj * J.x, acc[t].y + cj * J.y);
  }
  NQ_PHASE_FENCE();
  F::template run<false>(acc, j, c, lds, twr);
  NQ_PHASE_FENCE();
  // first inputs of the next row block: requested before the stores so that they are not queued behind them
  if (more) hs_load<N, P, T, PAIRQ>(h1, xrow<SLAB>(Mq, row_next), xrow<SLAB>(PAIRQ ? Mqw : Mq, row_next), j);
  {
    const XRowT<SLAB> rp = xrow<SLAB>(Mw, row);
#pragma unroll
    for (int t = 0; t < P; ++t) *rp.at(j + t * T) = acc[t];
  }
  // sum of the Jacobian part over this workgroup's rows -> passenger slot of its first row (the slots of the
  // other rows of a multi-row workgroup stay zero); column Muq.W of block 0 is padding of the half-spectrum row
  NQ_PHASE_FENCE();
  double* red = reinterpret_cast<double*>(nq_smem + X::LDS_BYTES - 512);
  static_assert(X::THREADS % 64 == 0, "full waves");
  block_sum_store_full_waves<2>(js, red, reinterpret_cast<double*>(Muq.xs + (size_t)rb * X::C * Muq.pitch + Muq.W), c * T + j);
  }   // row blocks
}

// ---- row kernels for rows of 2M points that do not fit the register budget as one transform -----------------------
// An 8192-point row with 8 points per thread needs 1024 threads, i.e. 128 VGPRs: k_x_products then spills 700 B per
// lane.  But every physical-space operation of the row kernels is pointwise, so the even and the odd samples of a row
// are two independent problems of M = 4096 points with the 4096 plan (512 threads, 256 VGPRs):
//     x[2m]   = ifft_M( X[k] + X[k+M] )[m],          x[2m+1] = ifft_M( (X[k] - X[k+M]) w^k )[m],   w = exp(2 pi i / 2M)
//     Y[k]    = E[k] + conj(w^k) O[k],               Y[k+M]  = E[k] - conj(w^k) O[k],   E, O = fft_M(y_even), fft_M(y_odd)
// Round 3 (profiles/r03_pmc_summary_8192_before.json): the first version ran the even problem of a row to the end, parked
// its output spectra in a global scratch row, then ran the odd problem -- every input was read TWICE and the scratch made a
// round trip: 15.4 GB per launch for 6.4 GB of algorithmic bytes, at 4.8 TB/s of fabric traffic, 70 % of the wave cycles
// waiting for memory.  Now BOTH parities go through every phase together: a raw input row is loaded once, folded into its
// even and its odd problem in registers, the two transforms run back to back, and the two output spectra are combined in
// registers (cache-warming "touch" loads a transform ahead of the real ones were tried on top, as LDS-DMA loads without a VGPR
// destination: products unchanged, wave-PV 13 % slower, profiles/r03_tried_8192_cache_touches.txt -- the kernels are not
// latency-bound any more).  What that costs is live state for two problems; the real fields that have to survive the longest (q_psi of
// both parities; the odd (ik phi) transform in the wave-PV kernel) are parked in thread-private LDS slots (no barrier:
// every thread reads back only what it wrote).  w^k = w^j w^(512 t) with w^(512 t) = exp(2 pi i t / 16) a compile-time
// constant: no table.  LDS: exchange 64 KB + stage twiddles of the M plan 18.7 KB + park 64 KB.
template <int T>
__device__ __forceinline__ cd eo_omega(cd wj, int t) { return tw16<true>(wj, t); }      // w^(j + T t), T = M / 8

// half-spectrum pair (a, b real-field spectra, k = 0..M) -> even and odd folds of Z = a + i b over the full row
template <int M, int P, int T, bool PAIR, typename Row>
__device__ __forceinline__ void eo_fold_pair2(cd (&we)[P], cd (&wo)[P], const Row& rowA, const Row& rowB, int j, cd wj,
                                              const double* __restrict__ kk, bool b_mul_ik, bool b_zero_nyq, double b_scale) {
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const int k = j + t * T, km = (k == 0) ? M : M - k;
    cd a1 = *rowA.at(k), a2 = *rowA.at(km);
    cd b1 = PAIR ? *rowB.at(k) : cmake(0, 0), b2 = PAIR ? *rowB.at(km) : cmake(0, 0);
    if (PAIR && b_mul_ik) {
      b1 = cscale(cmul_i(b1), kk[k]);
      b2 = cscale(cmul_i(b2), kk[km]);
    }
    if (PAIR) {
      b1 = cscale(b1, b_scale);
      b2 = cscale(b2, b_scale);
    }
    cd z1, z2;
    if (k == 0) {                 // the two self-mirrored entries Z[0], Z[M]: imaginary parts of a, b dropped (hs_pack)
      if (b_zero_nyq) b2.x = 0.0;
      z1 = cmake(a1.x, b1.x);
      z2 = cmake(a2.x, b2.x);
    } else {
      z1 = cmake(a1.x - b1.y, a1.y + b1.x);             // Z[k]   = a + i b
      z2 = cmake(a2.x + b2.y, b2.x - a2.y);             // Z[k+M] = conj(a[M-k]) + i conj(b[M-k])
    }
    we[t] = cadd(z1, z2);
    wo[t] = cmul(csub(z1, z2), eo_omega<T>(wj, t));
  }
}
// full-width row: raw X[k], X[k+M] in registers -> one parity's fold (of X or of ik X)
template <int M, int P, int T, typename Row>
__device__ __forceinline__ void eo_load_full(cd (&x1)[P], cd (&x2)[P], const Row& row, int j) {
#pragma unroll
  for (int t = 0; t < P; ++t) {
    x1[t] = *row.at(j + t * T);
    x2[t] = *row.at(j + t * T + M);
  }
}
template <int M, int P, int T>
__device__ __forceinline__ void eo_fold_regs(cd (&w)[P], const cd (&x1)[P], const cd (&x2)[P], int j, int par, cd wj,
                                             const double* __restrict__ kk, bool mul_ik) {
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const int k = j + t * T;
    cd a = x1[t], b = x2[t];
    if (mul_ik) {
      a = cscale(cmul_i(a), kk[k]);
      b = cscale(cmul_i(b), kk[k + M]);
    }
    w[t] = (par == 0) ? cadd(a, b) : cmul(csub(a, b), eo_omega<T>(wj, t));
  }
}
// E, O = spectra of the even / odd problem of a PACKED pair y = a + i b (a, b real): combine to Y and split into the two
// half spectra A, B (k = 0..M).  The mirrored entry Y[2M - k] = E[M-k] - conj(w^(M-k)) O[M-k] comes from the thread that
// owns M - k, through the exchange area (one LDS round trip).  O is destroyed.
template <int M, int P, int T, typename F, typename Row>
__device__ __forceinline__ void eo_unpack_pair_store2(const cd (&e)[P], cd (&o)[P], int j, int c, cd* lds, cd wj,
                                                      const Row& rowA, const Row& rowB, double scaleB = 1.0) {
  wg_barrier();
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const cd w = eo_omega<T>(wj, t);
    o[t] = cmul(cmake(w.x, -w.y), o[t]);                                // conj(w^k) O[k]
    lds[F::lds_index(j + t * T, c)] = csub(e[t], o[t]);                 // D[k] = Y[k + M]
  }
  wg_barrier();
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const int k = j + t * T;
    const cd y = cadd(e[t], o[t]);
    cd ym;
    if (k == 0) {
      ym = y;                                                    // Y[2M] = Y[0]
      const cd yM = csub(e[t], o[t]);                            // Y[M] = E[0] - O[0], its own mirror
      *rowA.at(M) = cmake(yM.x, 0.0);
      *rowB.at(M) = cmake(scaleB * yM.y, 0.0);
    } else {
      ym = lds[F::lds_index(M - k, c)];                          // Y[2M - k]
    }
    *rowA.at(k) = cmake(0.5 * (y.x + ym.x), 0.5 * (y.y - ym.y));
    *rowB.at(k) = cmake(scaleB * 0.5 * (y.y + ym.y), scaleB * 0.5 * (ym.x - y.x));
  }
  wg_barrier();
}

// Phase boundary of the even/odd kernels: NQ_PHASE_FENCE plus a laundering of j, c and w^j.  Everything derived from
// them (the eight w^k, kk[k], kk[k + M], row addresses) would otherwise be computed once and kept live across the whole
// row by common-subexpression elimination: ~60 VGPRs that the two-problem live state cannot spare (500 B/lane of scratch).
#define NQ_EO_FENCE()                                                  \
  do {                                                                 \
    NQ_PHASE_FENCE();                                                  \
    asm volatile("" : "+v"(j), "+v"(c), "+v"(wj.x), "+v"(wj.y));      \
  } while (0)

template <int N2, int MODE, bool SLAB>
__global__ void __launch_bounds__(XPlan<N2 / 2>::THREADS, XPlan<N2 / 2>::MIN_WAVES)
k_x_products_eo(MArr Mu, MArr Mp, MArr Mq, MArr Mqw, MArr Mphi, MArr Mgx, MArr Mgy, MArr Muq, MArr Mvq, MArr Mw,
                const cd* __restrict__ tw, const cd* __restrict__ wglob, const double* __restrict__ kk, int v_zero_nyq,
                double cj, double cr, int nblocks) {
  constexpr int M = N2 / 2;
  typedef XPlan<M> X;
  typedef typename X::F F;
  constexpr int P = X::P, T = X::T;
  static_assert(X::C == 1 && P == 8, "one row per workgroup, 8 points per thread (w^(T t) = 16th roots of unity)");
  cd* lds = reinterpret_cast<cd*>(nq_smem);                             // [exchange][stage twiddles][park][red]
  cd* twl = lds + F::LDS_ELEMS;
  double* park = reinterpret_cast<double*>(twl + F::TW_LDS_ELEMS);      // [2][M] doubles, thread-private slots
  double* red = park + 2 * M;                                            // 512 B of reduction scratch
  for (int i = threadIdx.x; i < F::TW_LDS_ELEMS; i += X::THREADS) twl[i] = tw[i];
  typename F::TwLds twr;
  twr.base = twl;
  wg_barrier_all();
  constexpr bool PAIRQ = (MODE == MODE_COUPLED || MODE == MODE_QGC);
  constexpr bool ONLYQ = (MODE == MODE_QG || MODE == MODE_QGC);
  for (int rb = blockIdx.x; rb < nblocks; rb += gridDim.x) {
    // per-iteration copies the compiler cannot see through (loop-invariant addresses would be hoisted and spill)
    int j = threadIdx.x, c = 0;
    asm volatile("" : "+v"(j), "+v"(c));
    cd wj;                                               // w^j, re-read per row (a cache hit) instead of held across the loop:
    {                                                    // four loop-invariant VGPRs less is what keeps the kernel out of scratch
      const cd z = wglob[j];                             // global table holds exp(-2 pi i m / N2)
      wj = cmake(z.x, -z.y);
    }
    asm volatile("" : "+v"(wj.x), "+v"(wj.y));
    const size_t row = (size_t)rb;
    cd we[P], wo[P];
    double qe[P], qo[P];
    double c_unscale = 1.0, c_scale = 1.0;
    if constexpr (MODE == MODE_QGC) {
      // q and the passive scalar share one complex transform; c has arbitrary units: rescale it per row by a power of two
      // (k_x_products does the same).  One extra sweep over the two rows (they are re-read from cache by the fold).
      const XRowT<SLAB> ra = xrow<SLAB>(Mq, Mq, row), rbw = xrow<SLAB>(Mqw, Mq, row);
      double ma = 0.0, mb = 0.0;
#pragma unroll
      for (int t = 0; t < P; ++t) {
        const cd a = *ra.at(j + t * T), b = *rbw.at(j + t * T);
        ma = fmax(ma, fmax(fabs(a.x), fabs(a.y)));
        mb = fmax(mb, fmax(fabs(b.x), fabs(b.y)));
      }
      if (j == 0) {
        const cd a = *ra.at(M), b = *rbw.at(M);
        ma = fmax(ma, fabs(a.x));
        mb = fmax(mb, fabs(b.x));
      }
      unsigned long long* mx = reinterpret_cast<unsigned long long*>(red + 40);
      if (j == 0) {
        mx[0] = 0ull;
        mx[1] = 0ull;
      }
      wg_barrier();
      row_atomic_max<T>(mx, ma, mb);
      wg_barrier();
      ma = __longlong_as_double((long long)mx[0]);
      mb = __longlong_as_double((long long)mx[1]);
      int e = 0;
      if (ma > 0.0 && mb > 0.0) e = ilogb(ma) - ilogb(mb);
      e = e > 900 ? 900 : (e < -900 ? -900 : e);
      c_scale = ldexp(1.0, e);
      c_unscale = ldexp(1.0, -e);
      wg_barrier();
    }
    // ---- phase 1: (q, qw | c) -> q, q_psi of both parities
    eo_fold_pair2<M, P, T, PAIRQ>(we, wo, xrow<SLAB>(Mq, Mq, row), xrow<SLAB>(PAIRQ ? Mqw : Mq, Mq, row), j, wj, kk, false, false, c_scale);
    NQ_EO_FENCE();
    F::template run<true>(we, j, c, lds, twr);
#pragma unroll
    for (int t = 0; t < P; ++t) {
      qe[t] = we[t].x;
      const double qp = (MODE == MODE_COUPLED) ? we[t].x - we[t].y : (MODE == MODE_QGC ? we[t].y * c_unscale : we[t].x);
      if constexpr (MODE != MODE_QG) park[j + t * T] = qp;              // q_psi, or QGModel's passive scalar
    }
    NQ_EO_FENCE();
    F::template run<true>(wo, j, c, lds, twr);
#pragma unroll
    for (int t = 0; t < P; ++t) {
      qo[t] = wo[t].x;
      const double qp = (MODE == MODE_COUPLED) ? wo[t].x - wo[t].y : (MODE == MODE_QGC ? wo[t].y * c_unscale : wo[t].x);
      if constexpr (MODE != MODE_QG) park[M + j + t * T] = qp;
    }
    // ---- phase 2: (u, v) = ifft of (-il psi, ik psi); u q + i v q of both parities -> Muq, Mvq
    NQ_EO_FENCE();
    eo_fold_pair2<M, P, T, true>(we, wo, xrow<SLAB>(Mu, Mq, row), xrow<SLAB>(Mp, Mq, row), j, wj, kk, true, v_zero_nyq != 0, 1.0);
    NQ_EO_FENCE();
    F::template run<true>(we, j, c, lds, twr);
    double ue[P], ve[P], uo[P], vo[P];
#pragma unroll
    for (int t = 0; t < P; ++t) {
      ue[t] = we[t].x;
      ve[t] = we[t].y;
      we[t] = cmake(ue[t] * qe[t], ve[t] * qe[t]);
    }
    NQ_EO_FENCE();
    F::template run<true>(wo, j, c, lds, twr);
#pragma unroll
    for (int t = 0; t < P; ++t) {
      uo[t] = wo[t].x;
      vo[t] = wo[t].y;
      wo[t] = cmake(uo[t] * qo[t], vo[t] * qo[t]);
    }
    NQ_EO_FENCE();
    F::template run<false>(we, j, c, lds, twr);
    NQ_EO_FENCE();
    F::template run<false>(wo, j, c, lds, twr);
    NQ_EO_FENCE();
    eo_unpack_pair_store2<M, P, T, F>(we, wo, j, c, lds, wj, xrow<SLAB>(Muq, Mq, row), xrow<SLAB>(Mvq, Mq, row));
    if constexpr (ONLYQ) {
      if constexpr (MODE == MODE_QGC) {          // second packed pair (u c, v c) -> Mgx, Mgy (= the Muc, Mvc arrays)
#pragma unroll
        for (int t = 0; t < P; ++t) {
          const double ce = park[j + t * T], co = park[M + j + t * T];
          we[t] = cmake(ue[t] * ce, ve[t] * ce);
          wo[t] = cmake(uo[t] * co, vo[t] * co);
        }
        NQ_EO_FENCE();
        F::template run<false>(we, j, c, lds, twr);
        NQ_EO_FENCE();
        F::template run<false>(wo, j, c, lds, twr);
        NQ_EO_FENCE();
        eo_unpack_pair_store2<M, P, T, F>(we, wo, j, c, lds, wj, xrow<SLAB>(Mgx, Mq, row), xrow<SLAB>(Mgy, Mq, row));
      }
      continue;
    } else {
      // The three wave phases in the order that ends the longest-lived real fields first: phiy x v (v dies), phix x u (u
      // dies), phi x q_psi (q_psi comes back from its LDS slots).  je, jo accumulate cj (u phix + v phiy) + i cr phi q_psi.
      cd x1[P], x2[P], je[P], jo[P];
      double js[2] = {0.0, 0.0};
      // ---- phase 3: phiy, times v
      eo_load_full<M, P, T>(x1, x2, xrow<SLAB>(Mgy, Mgy, row), j);
      NQ_EO_FENCE();
      eo_fold_regs<M, P, T>(je, x1, x2, j, 0, wj, kk, false);
      eo_fold_regs<M, P, T>(jo, x1, x2, j, 1, wj, kk, false);
      NQ_EO_FENCE();
      F::template run<true>(je, j, c, lds, twr);
#pragma unroll
      for (int t = 0; t < P; ++t) je[t] = cscale(je[t], ve[t]);
      NQ_EO_FENCE();
      F::template run<true>(jo, j, c, lds, twr);
#pragma unroll
      for (int t = 0; t < P; ++t) jo[t] = cscale(jo[t], vo[t]);
      // ---- phase 4: phix = ifft(ik g), times u.  Coupled: g is the phi row itself.
      NQ_EO_FENCE();
      eo_load_full<M, P, T>(x1, x2, xrow<SLAB>(MODE == MODE_COUPLED ? Mphi : Mgx, Mgy, row), j);
      NQ_EO_FENCE();
      eo_fold_regs<M, P, T>(we, x1, x2, j, 0, wj, kk, true);
      NQ_EO_FENCE();
      F::template run<true>(we, j, c, lds, twr);
#pragma unroll
      for (int t = 0; t < P; ++t) {
        je[t] = cmake(je[t].x + we[t].x * ue[t], je[t].y + we[t].y * ue[t]);
        js[0] += je[t].x;
        js[1] += je[t].y;
        je[t] = cscale(je[t], cj);
      }
      NQ_EO_FENCE();
      eo_fold_regs<M, P, T>(wo, x1, x2, j, 1, wj, kk, true);
      NQ_EO_FENCE();
      F::template run<true>(wo, j, c, lds, twr);
#pragma unroll
      for (int t = 0; t < P; ++t) {
        jo[t] = cmake(jo[t].x + wo[t].x * uo[t], jo[t].y + wo[t].y * uo[t]);
        js[0] += jo[t].x;
        js[1] += jo[t].y;
        jo[t] = cscale(jo[t], cj);
      }
      // ---- phase 5: phi, the refraction factor: + i cr phi q_psi
      NQ_EO_FENCE();
      // Coupled: phi's raw row is still in x1, x2 (one read serves phix and phi; folding one parity at a time from the held row
      // needs 36 B/lane of scratch, re-reading the row and folding both parities at once 84-296)
      if constexpr (MODE != MODE_COUPLED) eo_load_full<M, P, T>(x1, x2, xrow<SLAB>(Mphi, Mgy, row), j);
      NQ_EO_FENCE();
      eo_fold_regs<M, P, T>(we, x1, x2, j, 0, wj, kk, false);
      eo_fold_regs<M, P, T>(wo, x1, x2, j, 1, wj, kk, false);
      NQ_EO_FENCE();
      F::template run<true>(we, j, c, lds, twr);
#pragma unroll
      for (int t = 0; t < P; ++t) {
        const double qp = cr * park[j + t * T];
        je[t] = cmake(je[t].x - qp * we[t].y, je[t].y + qp * we[t].x);
      }
      NQ_EO_FENCE();
      F::template run<true>(wo, j, c, lds, twr);
#pragma unroll
      for (int t = 0; t < P; ++t) {
        const double qp = cr * park[M + j + t * T];
        jo[t] = cmake(jo[t].x - qp * wo[t].y, jo[t].y + qp * wo[t].x);
      }
      // ---- forward: W = fft of the phi tendency source, both parities, combined in registers
      NQ_EO_FENCE();
      F::template run<false>(je, j, c, lds, twr);
      NQ_EO_FENCE();
      F::template run<false>(jo, j, c, lds, twr);
      NQ_EO_FENCE();
      {
        const XRowT<SLAB> rp = xrow<SLAB>(Mw, Mgy, row);
#pragma unroll
        for (int t = 0; t < P; ++t) {
          const int k = j + t * T;
          const cd w = eo_omega<T>(wj, t);
          const cd o = cmul(cmake(w.x, -w.y), jo[t]);
          *rp.at(k) = cadd(je[t], o);
          *rp.at(k + M) = csub(je[t], o);
        }
      }
      // sum of the Jacobian part over this row -> passenger slot of the row (padding column Muq.W of block 0)
      NQ_EO_FENCE();
      block_sum_store<2>(js, red, reinterpret_cast<double*>(Muq.xs + (size_t)rb * Muq.pitch + Muq.W), j);    // (the DPP variant tips this kernel into 4 dwords of scratch)
    }
  }
}

// wave-PV sources for rows of 2M points: both parities through every phase, inputs read once (see k_x_products_eo)
template <int N2, bool SLAB>
__global__ void __launch_bounds__(XPlan<N2 / 2>::THREADS, XPlan<N2 / 2>::MIN_WAVES)
k_x_wavepv_eo(MArr Mphi, MArr Mphiy, MArr Ma, MArr Mb, const cd* __restrict__ tw, const cd* __restrict__ wglob,
              const double* __restrict__ kk, int nblocks) {
  constexpr int M = N2 / 2;
  typedef XPlan<M> X;
  typedef typename X::F F;
  constexpr int P = X::P, T = X::T;
  static_assert(X::C == 1 && P == 8, "one row per workgroup, 8 points per thread");
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  cd* twl = lds + F::LDS_ELEMS;
  cd* park = twl + F::TW_LDS_ELEMS;                        // [M] complex, thread-private slots: phix of the odd parity
  unsigned long long* mx = reinterpret_cast<unsigned long long*>(park + M);
  for (int i = threadIdx.x; i < F::TW_LDS_ELEMS; i += X::THREADS) twl[i] = tw[i];
  typename F::TwLds twr;
  twr.base = twl;
  wg_barrier_all();
  for (int rb = blockIdx.x; rb < nblocks; rb += gridDim.x) {
    int j = threadIdx.x, c = 0;
    asm volatile("" : "+v"(j), "+v"(c));
    cd wj;                                               // w^j, re-read per row (a cache hit) instead of held across the loop:
    {                                                    // four loop-invariant VGPRs less is what keeps the kernel out of scratch
      const cd z = wglob[j];                             // global table holds exp(-2 pi i m / N2)
      wj = cmake(z.x, -z.y);
    }
    asm volatile("" : "+v"(wj.x), "+v"(wj.y));
    const size_t row = (size_t)rb;
    cd we[P], wo[P], ge[P];
    double ae[P], ao[P];
    // (requesting the NEXT row block's phi row before the forward transforms of this one keeps 64 VGPRs live across the loop's
    // back edge, which the register allocator answers with 143 dwords of scratch: the row is loaded where it is used)
    cd x1[P], x2[P];
    eo_load_full<M, P, T>(x1, x2, xrow<SLAB>(Mphi, row), j);
    NQ_EO_FENCE();
    // phi and phix = ifft(ik phi) of both parities from ONE read of the row
    eo_fold_regs<M, P, T>(we, x1, x2, j, 0, wj, kk, false);
    NQ_EO_FENCE();
    F::template run<true>(we, j, c, lds, twr);
    double ma = 0.0, mb = 0.0;
#pragma unroll
    for (int t = 0; t < P; ++t) {
      ae[t] = we[t].x * we[t].x + we[t].y * we[t].y;
      ma = fmax(ma, ae[t]);
    }
    NQ_EO_FENCE();
    eo_fold_regs<M, P, T>(wo, x1, x2, j, 1, wj, kk, false);
    NQ_EO_FENCE();
    F::template run<true>(wo, j, c, lds, twr);
#pragma unroll
    for (int t = 0; t < P; ++t) {
      ao[t] = wo[t].x * wo[t].x + wo[t].y * wo[t].y;
      ma = fmax(ma, ao[t]);
    }
    NQ_EO_FENCE();
    eo_fold_regs<M, P, T>(wo, x1, x2, j, 1, wj, kk, true);
    NQ_EO_FENCE();
    F::template run<true>(wo, j, c, lds, twr);
#pragma unroll
    for (int t = 0; t < P; ++t) park[j + t * T] = wo[t];
    NQ_EO_FENCE();
    eo_fold_regs<M, P, T>(ge, x1, x2, j, 0, wj, kk, true);
    // phi_y's row: in flight during the next transform
    eo_load_full<M, P, T>(x1, x2, xrow<SLAB>(Mphiy, row), j);
    NQ_EO_FENCE();
    F::template run<true>(ge, j, c, lds, twr);
    NQ_EO_FENCE();
    eo_fold_regs<M, P, T>(we, x1, x2, j, 0, wj, kk, false);
    eo_fold_regs<M, P, T>(wo, x1, x2, j, 1, wj, kk, false);
    NQ_EO_FENCE();
    F::template run<true>(we, j, c, lds, twr);
#pragma unroll
    for (int t = 0; t < P; ++t) {
      const double b = -2.0 * (ge[t].x * we[t].y - ge[t].y * we[t].x);
      mb = fmax(mb, fabs(b));
      we[t] = cmake(ae[t], b);
    }
    NQ_EO_FENCE();
    F::template run<true>(wo, j, c, lds, twr);
#pragma unroll
    for (int t = 0; t < P; ++t) {
      const cd g = park[j + t * T];
      const double b = -2.0 * (g.x * wo[t].y - g.y * wo[t].x);
      mb = fmax(mb, fabs(b));
      wo[t] = cmake(ao[t], b);
    }
    // |phi|^2 is typically 1e8 times J(phi*, phi): the second field is rescaled per row by a power of two (k_x_wavepv)
    if (j == 0) {
      unsigned long long zero = 0ull;                    // formed here: held across the loop it is four VGPRs of zeros, spilled
      asm volatile("" : "+v"(zero));
      mx[0] = zero;
      mx[1] = zero;
    }
    wg_barrier();
    row_atomic_max<T>(mx, ma, mb);
    wg_barrier();
    ma = __longlong_as_double((long long)mx[0]);
    mb = __longlong_as_double((long long)mx[1]);
    int e = 0;
    if (ma > 0.0 && mb > 0.0) e = ilogb(ma) - ilogb(mb);
    e = e > 900 ? 900 : (e < -900 ? -900 : e);
    const double sb = ldexp(1.0, e);
#pragma unroll
    for (int t = 0; t < P; ++t) {
      we[t].y *= sb;
      wo[t].y *= sb;
    }
    NQ_EO_FENCE();
    F::template run<false>(we, j, c, lds, twr);
    NQ_EO_FENCE();
    F::template run<false>(wo, j, c, lds, twr);
    NQ_EO_FENCE();
    eo_unpack_pair_store2<M, P, T, F>(we, wo, j, c, lds, wj, xrow<SLAB>(Ma, row), xrow<SLAB>(Mb, row), ldexp(1.0, -e));
  }
}

// ---- diagnostics tick: physical-space statistics (ref Kernel.py:613-623 conc, skew; :701 pi) ----------------
// One row block: q, q_psi from the (q, qw) half-spectrum pair, phi from its mixed-space row; eight sums per
// workgroup -> part[workgroup][8]:
//   sum q^2, sum q_psi^2, sum q_psi^3, sum (q_psi - qbar)^2, sum ups^2, sum ups q_psi, sum q_psi Re phi, sum q_psi Im phi
// with ups = |phi|^2 - abar (abar = mean |phi|^2 and qbar = mean q_psi come from the spectral sums: centring BEFORE
// squaring, as the reference does, keeps std(ups) meaningful for nearly uniform waves).
template <int N, int MODE, bool SLAB>
__global__ void __launch_bounds__(XPlan<N>::THREADS, XPlan<N>::MIN_WAVES)
k_x_diag(MArr Mq, MArr Mqw, MArr Mphi, const cd* __restrict__ tw, const double* __restrict__ kk, double qbar,
         double abar, double* __restrict__ part) {
  typedef XPlan<N> X;
  typedef typename X::F F;
  constexpr int P = X::P, T = X::T;
  const int j = threadIdx.x % T, c = threadIdx.x / T;
  const size_t row = (size_t)blockIdx.x * X::C + c;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  cd* twl = lds + F::LDS_ELEMS;
  for (int i = threadIdx.x; i < F::TW_LDS_ELEMS; i += X::THREADS) twl[i] = tw[i];
  typename F::TwLds twr;
  twr.base = twl;
  wg_barrier_all();
  double* red = reinterpret_cast<double*>(nq_smem + X::LDS_BYTES - 512);
  cd w[P];
  HsRegs<P> h1;
  hs_load<N, P, T, MODE == MODE_COUPLED>(h1, xrow<SLAB>(Mq, row), xrow<SLAB>(MODE == MODE_COUPLED ? Mqw : Mq, row), j);
  NQ_PHASE_FENCE();
  hs_pack<N, P, T, F, MODE == MODE_COUPLED>(w, h1, j, c, lds, kk, false, false);
  NQ_PHASE_FENCE();
  F::template run<true>(w, j, c, lds, twr);
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  double qpsi[P];
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const double q = w[t].x;
    qpsi[t] = (MODE == MODE_COUPLED) ? w[t].x - w[t].y : w[t].x;
    const double qc = qpsi[t] - qbar;
    s1[0] += q * q;
    s1[1] += qpsi[t] * qpsi[t];
    s1[2] += qpsi[t] * qpsi[t] * qpsi[t];
    s1[3] += qc * qc;
  }
  {
    const XRowT<SLAB> rp = xrow<SLAB>(Mphi, row);
#pragma unroll
    for (int t = 0; t < P; ++t) w[t] = *rp.at(j + t * T);
  }
  NQ_PHASE_FENCE();
  F::template run<true>(w, j, c, lds, twr);
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const double ups = w[t].x * w[t].x + w[t].y * w[t].y - abar;
    s2[0] += ups * ups;
    s2[1] += ups * qpsi[t];
    s2[2] += qpsi[t] * w[t].x;
    s2[3] += qpsi[t] * w[t].y;
  }
  NQ_PHASE_FENCE();
  block_sum_store<4>(s1, red, part + 8 * (size_t)blockIdx.x);
  block_sum_store<4>(s2, red, part + 8 * (size_t)blockIdx.x + 4);
}

// ---- diagnostics tick: projections of one transformed mixed-space plane on lap_h and diss_h ------------
// a = B-fft of Hw (Hw already went through the A sub-pass).  part[workgroup][4] =
//   sum Re(conj(lap_h) a), sum Im(conj(lap_h) a), sum Re(conj(diss_h) a), sum Im(conj(diss_h) a)
// with lap_h = -wv2 phih, diss_h = -(nu4w wv4 + nuw wv2 + muw) phih: gamma1, gamma2, xi1, xi2 of ref Kernel.py:691-700.
template <int S1, int CLX = CL>
__global__ void __launch_bounds__((YPlanT<S1, CLX>::THREADS))
k_s_project(MArr Hw, const cd* __restrict__ phih, YGeom g, const double* __restrict__ kk, const double* __restrict__ ll,
            const cd* __restrict__ tw, int tw_step_N, double nu4w, double nuw, double muw, double* __restrict__ part) {
  typedef YPlanT<S1, CLX> Y;
  constexpr int P = Y::P, T = Y::T;
  const int c = threadIdx.x % CLX, j = threadIdx.x / CLX;
  const int k = blockIdx.x * CLX + c, l1 = blockIdx.y;
  const int kg = g.k0 + k, S2 = g.S2;
  const int N = S1 * S2;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  typename Y::F::Tw twr;
  Y::F::load_tw(twr, j, tw, tw_step_N * (N / S1));
  double* red = reinterpret_cast<double*>(nq_smem + Y::LDS_BYTES - 512);
  cd a[P];
#pragma unroll
  for (int t = 0; t < P; ++t) a[t] = Hw.ys[(size_t)(l1 * S1 + j + t * T) * Hw.pitch + k];
  Y::F::template run<false>(a, j, c, lds, twr);
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  const double kx = kk[kg];
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const int l = l1 + S2 * (j + t * T);
    const cd ys = phih[(size_t)l * g.pitch_s + k];
    const double ly = ll[l];
    const double wv2 = kx * kx + ly * ly;
    const double d = nu4w * wv2 * wv2 + nuw * wv2 + muw;
    const double re = ys.x * a[t].x + ys.y * a[t].y, im = ys.x * a[t].y - ys.y * a[t].x;   // conj(ys) * a
    s[0] += -wv2 * re;
    s[1] += -wv2 * im;
    s[2] += -d * re;
    s[3] += -d * im;
  }
  block_sum_store<4>(s, red, part + 4 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x));
}

// ---- diagnostics tick of QGModel's passive scalar: Gamma_c = 2 mean(lap c * J(psi, c)) (ref QGModel.py:727-731) by Parseval ----
// part[workgroup] = sum over the tile of w * Re(conj(-wv2 c-hat) * (i k F[u c] + i l F[v c])), w = 1 on the two self-mirrored
// columns, 2 elsewhere; Huc, Hvc already went through the A sub-pass.
template <int S1, int CLX = CL>
__global__ void __launch_bounds__((YPlanT<S1, CLX>::THREADS))
k_s_project_c(MArr Huc, MArr Hvc, const cd* __restrict__ ch, YGeom g, const double* __restrict__ kk,
              const double* __restrict__ ll, const cd* __restrict__ tw, int tw_step_N, double* __restrict__ part) {
  typedef YPlanT<S1, CLX> Y;
  constexpr int P = Y::P, T = Y::T;
  const int c = threadIdx.x % CLX, j = threadIdx.x / CLX;
  const int k = blockIdx.x * CLX + c, l1 = blockIdx.y;
  const int kg = g.k0 + k, S2 = g.S2;
  const bool ok = k < g.width;
  const int N = S1 * S2;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  typename Y::F::Tw twr;
  Y::F::load_tw(twr, j, tw, tw_step_N * (N / S1));
  double* red = reinterpret_cast<double*>(nq_smem + Y::LDS_BYTES - 512);
  cd f1[P], f2[P];
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const size_t at = (size_t)(l1 * S1 + j + t * T) * Huc.pitch + k;
    f1[t] = ok ? Huc.ys[at] : cmake(0, 0);
    f2[t] = ok ? Hvc.ys[at] : cmake(0, 0);
  }
  Y::F::template run<false>(f1, j, c, lds, twr);
  Y::F::template run<false>(f2, j, c, lds, twr);
  double s[1] = {0.0};
  if (ok) {
    const double kx = kk[kg], wt = (kg == 0 || kg == N / 2) ? 1.0 : 2.0;
#pragma unroll
    for (int t = 0; t < P; ++t) {
      const int l = l1 + S2 * (j + t * T);
      const double ly = ll[l], wv2 = kx * kx + ly * ly;
      cd cc = ch[(size_t)l * g.pitch_s + k];
      if (wt == 1.0) {      // lap c in physical space sees only the Hermitian part (in l) of the self-mirrored columns
        const cd cm = ch[(size_t)((N - l) % N) * g.pitch_s + k];
        cc = cmake(0.5 * (cc.x + cm.x), 0.5 * (cc.y - cm.y));
      }
      const double jx = -(kx * f1[t].y + ly * f2[t].y), jy = kx * f1[t].x + ly * f2[t].x;      // i k F1 + i l F2
      s[0] += wt * (-wv2) * (cc.x * jx + cc.y * jy);
    }
  }
  block_sum_store<1>(s, red, part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x));
}

// ---- ETDRK4 stage update of one spectral element -----------------------------------------------
// ref Kernel.py:327,:347,:364,:381-382 (q) and :333,:351,:368,:386-387 (phi).  The filter is folded
// into the coefficient planes (Ef = E*filtr ...), which is the same arithmetic up to rounding.
struct EtdArrays {
  const cd* y_in;      // stage 0,1,3: y(t_n) ; stage 2: y after stage 0
  cd* y_out;
  cd* fn0;
  cd* fna;             // Na, then Na+Nb
  const cd* Eh;        // exp(c dt/2) * filtr
  const cd* Q;
  const cd* E;
  const cd* f0;
  const cd* fab;
  const cd* fc;
};

// idx: element of the state / tendency planes; ci: element of the coefficient planes (crow(): mirrored rows share one)
__device__ __forceinline__ cd etd_update(const EtdArrays& a, size_t idx, size_t ci, cd Nl, int stage) {
  cd y;
  if (stage == 0) {
    y = cadd(cmul(a.Eh[ci], a.y_in[idx]), cmul(a.Q[ci], Nl));
    a.fn0[idx] = Nl;
  } else if (stage == 1) {
    y = cadd(cmul(a.Eh[ci], a.y_in[idx]), cmul(a.Q[ci], Nl));
    a.fna[idx] = Nl;
  } else if (stage == 2) {
    const cd n0 = a.fn0[idx];
    const cd comb = cmake(2.0 * Nl.x - n0.x, 2.0 * Nl.y - n0.y);
    y = cadd(cmul(a.Eh[ci], a.y_in[idx]), cmul(a.Q[ci], comb));
    const cd na = a.fna[idx];
    a.fna[idx] = cadd(na, Nl);
  } else {
    const cd n0 = a.fn0[idx], nab = a.fna[idx];
    y = cadd(cadd(cmul(a.E[ci], a.y_in[idx]), cmul(a.f0[ci], n0)),
             cadd(cscale(cmul(a.fab[ci], nab), 2.0), cmul(a.fc[ci], Nl)));
  }
  a.y_out[idx] = y;
  return y;
}

// Same update with an explicit filter factor (coefficient planes WITHOUT the filter folded in); used by the
// dual-copy q equation where the two copies see different (mirrored) filter planes.
__device__ __forceinline__ cd etd_update_f(const EtdArrays& a, size_t idx, size_t ci, cd Nl, int stage, double fl) {
  cd y;
  if (stage == 0) {
    y = cadd(cmul(a.Eh[ci], a.y_in[idx]), cmul(a.Q[ci], Nl));
    a.fn0[idx] = Nl;
  } else if (stage == 1) {
    y = cadd(cmul(a.Eh[ci], a.y_in[idx]), cmul(a.Q[ci], Nl));
    a.fna[idx] = Nl;
  } else if (stage == 2) {
    const cd n0 = a.fn0[idx];
    const cd comb = cmake(2.0 * Nl.x - n0.x, 2.0 * Nl.y - n0.y);
    y = cadd(cmul(a.Eh[ci], a.y_in[idx]), cmul(a.Q[ci], comb));
    const cd na = a.fna[idx];
    a.fna[idx] = cadd(na, Nl);
  } else {
    const cd n0 = a.fn0[idx], nab = a.fna[idx];
    y = cadd(cadd(cmul(a.E[ci], a.y_in[idx]), cmul(a.f0[ci], n0)),
             cadd(cscale(cmul(a.fab[ci], nab), 2.0), cmul(a.fc[ci], Nl)));
  }
  y = cscale(y, fl);
  a.y_out[idx] = y;
  return y;
}

// Dual-copy q equation (DESIGN.md "dual copy"): X+ = qh(l,k), X- = conj(qh(-l,-k)), k = 0..N/2.  The
// reference's full-plane q-hat is not Hermitian when its filter is not mirror-symmetric (the 2/3 mask,
// ref Kernel.py:277-281) and always carries an anti-Hermitian passenger on row l = N/2; both copies obey the
// same ETDRK4 recursion with the filter taken at (l,k) and at (-l,-k), and N- = conj(N(-l,-k)) differs from
// N+ only by the sign of the il term on row N/2.  Physical space sees (X+ + X-)/2.
struct DualQ {
  EtdArrays minus;          // state of X- (y_in == nullptr: single-copy mode)
  const double* filt_p;     // filter at (l, k)
  const double* filt_m;     // filter at (-l, -k)
};

// ---- Sq: nonlinear term + stage update of q-hat on the half spectrum --------------------------------
template <int S1, bool DUAL, int CLX = CL>
__global__ void __launch_bounds__((YPlanT<S1, CLX>::THREADS))
k_s_q(MArr Huq, MArr Hvq, EtdArrays ea, int stage, YGeom g, const double* __restrict__ kk,
      const double* __restrict__ ll, const cd* __restrict__ tw, int tw_step_N, DualQ dq, EtdArrays ep) {
  typedef YPlanT<S1, CLX> Y;
  constexpr int P = Y::P, T = Y::T;
  const int c = threadIdx.x % CLX, j = threadIdx.x / CLX;
  const int k = blockIdx.x * CLX + c, l1 = pair_order(blockIdx.y, g.S2);
  const int kg = g.k0 + k, S2 = g.S2, kernel_family = g.kernel_family;
  const bool ok = k < g.width;
  const int N = S1 * S2;
  constexpr bool dual = DUAL;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  typename Y::F::Tw twr;
  Y::F::load_tw(twr, j, tw, tw_step_N * (N / S1));
  cd f1[P], f2[P];
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const size_t at = (size_t)(l1 * S1 + j + t * T) * Huq.pitch + k;
    f1[t] = ok ? Huq.ys[at] : cmake(0, 0);
    f2[t] = ok ? Hvq.ys[at] : cmake(0, 0);
  }
  Y::F::template run<false>(f1, j, c, lds, twr);
  Y::F::template run<false>(f2, j, c, lds, twr);
  if (!ok) return;
  const double kx = kk[kg];
  const bool interior = (kg > 0) && (kg < N / 2);
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const int l = l1 + S2 * (j + t * T);
    const size_t idx = (size_t)l * g.pitch_s + k, ci = (size_t)crow(g, l, N) * g.pitch_s + k;
    const double ly = ll[l];
    const bool pass_row = kernel_family && interior && l == N / 2;      // see DESIGN.md "Nyquist lines"
    if constexpr (!dual) {
      const double lz = pass_row ? 0.0 : ly;
      // N_q = -(ik F1 + il F2)
      cd Nl = cmake(kx * f1[t].y + lz * f2[t].y, -(kx * f1[t].x + lz * f2[t].x));
      if (kernel_family && l == 0 && kg == 0) Nl = cmake(0, 0);   // jach[0,0] = 0 (QGModel does not)
      etd_update(ea, idx, ci, Nl, stage);
    } else {
      const double lm = pass_row ? -ly : ly;
      cd Np = cmake(kx * f1[t].y + ly * f2[t].y, -(kx * f1[t].x + ly * f2[t].x));
      cd Nm = cmake(kx * f1[t].y + lm * f2[t].y, -(kx * f1[t].x + lm * f2[t].x));
      if (l == 0 && kg == 0) {
        Np = cmake(0, 0);
        Nm = cmake(0, 0);
      }
      etd_update_f(ea, idx, ci, Np, stage, dq.filt_p[idx]);
      if (interior) etd_update_f(dq.minus, idx, ci, Nm, stage, dq.filt_m[idx]);
    }
  }
  // The passenger (DESIGN.md "Nyquist lines"): on row l = N/2 the il term of N_q is ANTI-Hermitian in the reference's full
  // plane (numpy's ll[N/2] is not odd), never reaches physical space, but is part of the reference's qh
  // (ref Kernel.py:471-486, :327).  It obeys the same ETDRK4 recursion with the coefficients of that row, so it is carried
  // as ONE row of N/2+1 values (ep: state and tendency rows indexed by the local column, coefficient pointers at row N/2):
  // qh[N/2, k] = X+ + A_k, qh[N/2, -k] = conj(X+) - conj(A_k).  The thread that holds l = N/2 is (l1 = 0, j = 0, t = P/2).
  if constexpr (!dual) {
    if (ep.y_out != nullptr && kernel_family && interior && l1 == 0 && j == 0) {
      constexpr int tp = P / 2;
      const double ly = ll[N / 2];
      etd_update(ep, (size_t)k, (size_t)k, cmake(ly * f2[tp].y, -ly * f2[tp].x), stage);      // -i l F[v q]
    }
  }
}

// ---- Sw: nonlinear term + stage update of phi-hat, then first half of the inverse y transform -----
// With budgets (bw.part != null) it also emits the
// Parseval sums of ref Kernel.py:629-633, :646-652, :698-699 (see oracle/reduced_pipeline.py).
struct BudgetW {
  double* part;        // [workgroup][6]: S0..S3 of the NEW phih, then SG, SX of this stage; null = off
  const cd* y_start;   // phih at the start of this stage (what the tendency was computed from)
  double nu4w, nuw, muw;
};
constexpr int NQ_PARTW = 6;

// a[], b[] hold y/M and i*l*y/M on return; budgets: sums over the new y
template <int P, int T>
__device__ __forceinline__ void phi_outputs(const cd (&y)[P], cd (&a)[P], cd (&b)[P], int l1, int S2,
                                            int j, int kg, double invM, const double* __restrict__ kk,
                                            const double* __restrict__ ll, bool bud, double (&s)[4]) {
  const double kx = kk[kg];
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const int l = l1 + S2 * (j + t * T);
    const double ly = ll[l];
    a[t] = cscale(y[t], invM);
    b[t] = cscale(cmul_i(y[t]), ly * invM);
    if (bud) {
      const double wv2 = kx * kx + ly * ly;
      const double m2 = y[t].x * y[t].x + y[t].y * y[t].y;
      s[0] += m2;
      s[1] += wv2 * m2;
      s[2] += wv2 * wv2 * m2;
      s[3] += wv2 * wv2 * wv2 * m2;
    }
  }
}

template <int S1, int CLX = CL>
__global__ void __launch_bounds__((YPlanT<S1, CLX>::THREADS))
k_s_phi(MArr Hw, const cd* __restrict__ jpass, const cd* __restrict__ jpass_own, int own0, int own1, int jpitch, EtdArrays ea, int stage, YGeom g, MArr Hphi, MArr Hphiy,
        double invM, const double* __restrict__ kk, const double* __restrict__ ll, const cd* __restrict__ tw,
        int tw_step_N, BudgetW bw) {
  typedef YPlanT<S1, CLX> Y;
  constexpr int P = Y::P, T = Y::T;
  const int c = threadIdx.x % CLX, j = threadIdx.x / CLX;
  const int k = blockIdx.x * CLX + c, l1 = pair_order(blockIdx.y, g.S2);
  const int kg = g.k0 + k, S2 = g.S2;
  const int N = S1 * S2;
  const bool bud = bw.part != nullptr;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  typename Y::F::Tw twr;
  Y::F::load_tw(twr, j, tw, tw_step_N * (N / S1));
  double* red = reinterpret_cast<double*>(nq_smem + Y::LDS_BYTES - 512);
  double* part = bud ? bw.part + NQ_PARTW * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) : nullptr;
  cd a[P], b[P], y[P];
#pragma unroll
  for (int t = 0; t < P; ++t) a[t] = Hw.ys[(size_t)(l1 * S1 + j + t * T) * Hw.pitch + k];
  // [0,0]: the Jacobian part of the tendency is zeroed there (ref Kernel.py:468) -- add its domain sum back
  double jfix[2] = {0.0, 0.0};
  if (jpass != nullptr && g.k0 == 0 && blockIdx.x == 0 && l1 == 0) {   // jpass == null: YBJModel keeps [0,0]
    for (int yy = threadIdx.x; yy < N; yy += Y::THREADS) {
      // (rows [own0, own1): this rank's own block, still on the X side when the step skips its copy -- ArrayListR)
      const cd z = ((yy >= own0 && yy < own1) ? jpass_own : jpass)[(size_t)yy * jpitch];
      jfix[0] += z.x;
      jfix[1] += z.y;
    }
    block_sum_thread0<2>(jfix, red);
  }
  Y::F::template run<false>(a, j, c, lds, twr);
  double sj[2] = {0.0, 0.0};
  const double kx = kk[kg];
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const int l = l1 + S2 * (j + t * T);
    const size_t idx = (size_t)l * g.pitch_s + k;
    if (bud) {
      // With W = F[-J - (i/2) phi q_psi] (un-zeroed), lap_h = -wv2 ys, diss_h = -d ys (ref Kernel.py:691-700):
      //   gamma1 + gamma2 = -hslash/2 sum Re(conj(lap_h) W) / (M^2 f),   xi1 + xi2 = -sum Im(conj(diss_h) W) / (M^2 f)
      const cd ys = bw.y_start[idx];
      const double ly = ll[l];
      const double wv2 = kx * kx + ly * ly;
      const double d = bw.nu4w * wv2 * wv2 + bw.nuw * wv2 + bw.muw;
      sj[0] += -wv2 * (ys.x * a[t].x + ys.y * a[t].y);
      sj[1] += -d * (ys.x * a[t].y - ys.y * a[t].x);
    }
    cd Nl = a[t];                                       // N_phi = -J - 0.5 i R
    if (l == 0 && kg == 0) Nl = cmake(Nl.x + jfix[0], Nl.y + jfix[1]);
    y[t] = etd_update(ea, idx, (size_t)crow(g, l, N) * g.pitch_s + k, Nl, stage);
  }
  double s4[4] = {0.0, 0.0, 0.0, 0.0};
  phi_outputs<P, T>(y, a, b, l1, S2, j, kg, invM, kk, ll, bud, s4);
  Y::F::template run<true>(a, j, c, lds, twr);
  Y::F::template run<true>(b, j, c, lds, twr);
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const size_t at = (size_t)(l1 * S1 + j + t * T) * Hphi.pitch + k;
    Hphi.ys[at] = a[t];
    Hphiy.ys[at] = b[t];
  }
  if (bud) {
    block_sum_store<4>(s4, red, part);
    block_sum_store<2>(sj, red, part + 4);
  }
}

// emit-only variant (set_phi): phih -> Hphi, Hphiy (+ sums with budgets)
template <int S1, int CLX = CL>
__global__ void __launch_bounds__((YPlanT<S1, CLX>::THREADS))
k_s_emit_phi(const cd* __restrict__ phih, YGeom g, MArr Hphi, MArr Hphiy, double invM,
             const double* __restrict__ kk, const double* __restrict__ ll, const cd* __restrict__ tw, int tw_step_N,
             BudgetW bw) {
  typedef YPlanT<S1, CLX> Y;
  constexpr int P = Y::P, T = Y::T;
  const int c = threadIdx.x % CLX, j = threadIdx.x / CLX;
  const int k = blockIdx.x * CLX + c, l1 = blockIdx.y;
  const int kg = g.k0 + k, S2 = g.S2;
  const int N = S1 * S2;
  const bool bud = bw.part != nullptr;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  typename Y::F::Tw twr;
  Y::F::load_tw(twr, j, tw, tw_step_N * (N / S1));
  double* red = reinterpret_cast<double*>(nq_smem + Y::LDS_BYTES - 512);
  cd a[P], b[P], y[P];
#pragma unroll
  for (int t = 0; t < P; ++t) y[t] = phih[(size_t)(l1 + S2 * (j + t * T)) * g.pitch_s + k];
  double s4[4] = {0.0, 0.0, 0.0, 0.0};
  phi_outputs<P, T>(y, a, b, l1, S2, j, kg, invM, kk, ll, bud, s4);
  Y::F::template run<true>(a, j, c, lds, twr);
  Y::F::template run<true>(b, j, c, lds, twr);
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const size_t at = (size_t)(l1 * S1 + j + t * T) * Hphi.pitch + k;
    Hphi.ys[at] = a[t];
    Hphiy.ys[at] = b[t];
  }
  if (bud) {
    double* part = bw.part + NQ_PARTW * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
    double z2[2] = {0.0, 0.0};
    block_sum_store<4>(s4, red, part);
    block_sum_store<2>(z2, red, part + 4);
  }
}

// ---- Si: the psi inversion on the half spectrum + first half of the four inverse y transforms -----
// MODE_COUPLED: ref CoupledModel.py:75-97 with ph = wv2i*(qwh - qh) instead of fft(ifft(.).real)
// (oracle/reduced_pipeline.py proves the equivalence).  Other modes: ph = -wv2i*qh (Ha, Hb unused).
template <int S1, int MODE, int CLX = CL>
__global__ void __launch_bounds__((YPlanT<S1, CLX>::THREADS))
k_s_invert(MArr Ha, MArr Hb, const cd* __restrict__ qh, const double* __restrict__ filt, MArr Hu, MArr Hp, MArr Hq,
           MArr Hqw, cd* __restrict__ qwh_out, cd* __restrict__ ph_out, YGeom g, double invM, double f,
           const double* __restrict__ kk, const double* __restrict__ ll, const cd* __restrict__ tw, int tw_step_N,
           double* __restrict__ bud_part, const cd* __restrict__ q_bud, const cd* __restrict__ qh_minus,
           const double* __restrict__ filt_m, const cd* __restrict__ c_hat) {
  // MODE_QGC: c_hat (the passive scalar's spectrum) goes out through Hqw, and bud_part is [workgroup][6]: the three ep_psi
  // sums, then sum w |c|^2 (without [0,0]), sum w wv2 |c|^2, sum w wv4 |c|^2 for ep_c (ref QGModel.py:595-598).
  // bud_part: [workgroup][3] Parseval sums for ep_psi (ref Kernel.py:635-640 / QGModel.py:588-593):
  //   sum w*wv4*Re(qb conj psi), sum w*wv2*Re(q conj psi), sum w*Re(qb conj psi); qb = q_bud (QGModel's
  //   stale q, QGModel.py:401) or q; w = 1 on the self-mirrored columns, 2 elsewhere.
  typedef YPlanT<S1, CLX> Y;
  constexpr int P = Y::P, T = Y::T;
  const int c = threadIdx.x % CLX, j = threadIdx.x / CLX;
  const int k = blockIdx.x * CLX + c, l1 = pair_order(blockIdx.y, g.S2);
  const int kg = g.k0 + k, S2 = g.S2, kernel_family = g.kernel_family;
  const bool ok = k < g.width;
  const int N = S1 * S2;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  typename Y::F::Tw twr;
  Y::F::load_tw(twr, j, tw, tw_step_N * (N / S1));
  double* red = reinterpret_cast<double*>(nq_smem + Y::LDS_BYTES - 512);
  double s3[3] = {0.0, 0.0, 0.0}, sc[3] = {0.0, 0.0, 0.0};
  constexpr bool FOURTH = (MODE == MODE_COUPLED || MODE == MODE_QGC);
  cd a[P], b[P], u[P], q[P];
  if (MODE == MODE_COUPLED) {
#pragma unroll
    for (int t = 0; t < P; ++t) {
      const size_t at = (size_t)(l1 * S1 + j + t * T) * Ha.pitch + k;
      a[t] = ok ? Ha.ys[at] : cmake(0, 0);
      b[t] = ok ? Hb.ys[at] : cmake(0, 0);
    }
    Y::F::template run<false>(a, j, c, lds, twr);
    Y::F::template run<false>(b, j, c, lds, twr);
  }
  const double kx = ok ? kk[kg] : 0.0;
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const int l = l1 + S2 * (j + t * T);
    const size_t idx = (size_t)l * g.pitch_s + k;
    const double ly = ll[l];
    const double wv2 = kx * kx + ly * ly;
    const double wv2i = (wv2 != 0.0) ? 1.0 / wv2 : 0.0;
    cd qv = ok ? qh[idx] : cmake(0, 0);
    if (qh_minus && ok && kg > 0 && kg < N / 2) {          // dual copy: physical space sees the mean
      const cd qm = qh_minus[idx];
      qv = cmake(0.5 * (qv.x + qm.x), 0.5 * (qv.y + qm.y));
    }
    cd qw = cmake(0, 0), psi;
    if (MODE == MODE_COUPLED) {
      cd B = b[t];
      if (l == 0 && kg == 0) B = cmake(0, 0);
      const double g = 0.5 * (-wv2);
      double fl = ok ? filt[idx] : 0.0;
      if (filt_m && ok) fl = 0.5 * (fl + filt_m[idx]);     // Hermitian part of filtr * (Hermitian field)
      qw = cmake(0.5 * (g * a[t].x + B.x) / f * fl, 0.5 * (g * a[t].y + B.y) / f * fl);
      psi = cmake(wv2i * (qw.x - qv.x), wv2i * (qw.y - qv.y));
    } else {
      psi = cmake(-wv2i * qv.x, -wv2i * qv.y);
      if (MODE == MODE_QGC) {
        qw = ok ? c_hat[idx] : cmake(0, 0);
        if (bud_part && ok) {
          // ep_c = -2 nu4c mean(lap c ^2) - 2 nu gradC2 - 2 muc C2 (ref QGModel.py:595-598): C2, gradC2 are spec_var sums of
          // c-hat as it is; mean(lap c ^2) is a physical-space mean and sees only the Hermitian part (in l) of the two
          // self-mirrored columns -- column nx/2 of c-hat picks up an anti-Hermitian part from the ik term of its Jacobian
          const bool special = (kg == 0 || kg == N / 2);
          const double wgt = special ? 1.0 : 2.0;
          const double m2 = wgt * (qw.x * qw.x + qw.y * qw.y);
          double m2h = m2;
          if (special) {
            const cd cm = c_hat[(size_t)((N - l) % N) * g.pitch_s + k];
            const double hx = 0.5 * (qw.x + cm.x), hy = 0.5 * (qw.y - cm.y);
            m2h = hx * hx + hy * hy;
          }
          sc[0] += (l == 0 && kg == 0) ? 0.0 : m2;
          sc[1] += wv2 * m2;
          sc[2] += wv2 * wv2 * m2h;
        }
      }
    }
    if (ok && ph_out) {
      ph_out[idx] = psi;
      if (MODE == MODE_COUPLED) qwh_out[idx] = qw;
    }
    if (bud_part && ok) {
      const bool special = (kg == 0 || kg == N / 2);
      cd qB = q_bud ? q_bud[idx] : qv, qQ = qv, ps = psi;
      if (special) {
        // mean(a*b) of REAL fields uses the Hermitian part (in l) of the self-mirrored columns -- what
        // irfft2 / `.real` keep.  q-hat(-l) comes from memory; psi(-l) follows from it because qwh is
        // Hermitian there: psi(-l) = wv2i (conj(qwh(l)) - q(-l)).
        const size_t im = (size_t)((N - l) % N) * g.pitch_s + k;
        const cd qm = qh[im];
        const cd qbm = q_bud ? q_bud[im] : qm;
        const cd psm = (MODE == MODE_COUPLED) ? cmake(wv2i * (qw.x - qm.x), wv2i * (-qw.y - qm.y))
                                              : cmake(-wv2i * qm.x, -wv2i * qm.y);
        qQ = cmake(0.5 * (qv.x + qm.x), 0.5 * (qv.y - qm.y));
        qB = cmake(0.5 * (qB.x + qbm.x), 0.5 * (qB.y - qbm.y));
        ps = cmake(0.5 * (psi.x + psm.x), 0.5 * (psi.y - psm.y));
      }
      const double wgt = special ? 1.0 : 2.0;
      const double rb = wgt * (qB.x * ps.x + qB.y * ps.y), rq = wgt * (qQ.x * ps.x + qQ.y * ps.y);
      s3[0] += wv2 * wv2 * rb;
      s3[1] += wv2 * rq;
      s3[2] += rb;
    }
    // u = Re ifft(-il psi) with psi Hermitian (ref Kernel.py:481): the whole Nyquist row drops out.
    // On the two self-mirrored columns psi is kept un-projected here and the row kernels take the
    // real part after the y transform; that commutes with -il everywhere except at l = N/2, where
    // numpy's ll is not odd -- so the row is zeroed on every column (DESIGN.md "Nyquist lines").
    const double lz = (kernel_family && l == N / 2) ? 0.0 : ly;
    u[t] = cmake(lz * psi.y * invM, -lz * psi.x * invM);        // -i l psi
    a[t] = cscale(psi, invM);
    q[t] = cscale(qv, invM);
    b[t] = cscale(qw, invM);
  }
  Y::F::template run<true>(u, j, c, lds, twr);
  Y::F::template run<true>(a, j, c, lds, twr);
  Y::F::template run<true>(q, j, c, lds, twr);
  if (FOURTH) Y::F::template run<true>(b, j, c, lds, twr);
  if (ok) {
#pragma unroll
    for (int t = 0; t < P; ++t) {
      const size_t at = (size_t)(l1 * S1 + j + t * T) * Hu.pitch + k;
      Hu.ys[at] = u[t];
      Hp.ys[at] = a[t];
      Hq.ys[at] = q[t];
      if (FOURTH) Hqw.ys[at] = b[t];
    }
  }
  constexpr int NB = (MODE == MODE_QGC) ? 6 : 3;
  if (bud_part) block_sum_store<3>(s3, red, bud_part + NB * ((size_t)blockIdx.y * gridDim.x + blockIdx.x));
  if (bud_part && MODE == MODE_QGC) block_sum_store<3>(sc, red, bud_part + NB * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + 3);
}


// ---- QGModel on small grids (one rank, N <= 512): the spectral side of a stage as ONE array-parallel kernel ---------------
// A step of these grids is a chain of dependent kernels of 7-14 us each, bound by the latency inside every kernel: load ->
// transform -> transform -> load state -> pointwise -> transform -> transform -> transform -> store, one after the other in each
// workgroup (profiles/r03_small_grids_single_pass_columns.txt).  Here k_s_q and k_s_invert<MODE_UNCOUPLED> of the QG stage
// graph are one kernel whose workgroup is three wave groups working side by side on a tile of whole columns:
//   group 0: F[uq] = fft_y(Huq) ........................... later  u-hat = -i l psi -> ifft_y -> Hu
//   group 1: F[vq] = fft_y(Hvq) ........................... later  psi-hat            -> ifft_y -> Hp
//   group 2: requests the ETDRK4 operands of its points WHILE the other two transform, then does the pointwise work of both
//            kernels (N_q, stage update, psi = -q/wv2, the ep_psi sums) ... q-hat -> ifft_y -> Hq
// Values cross between the groups through LDS (same (row, column) slot in every group: no conflicts); the new q-hat never
// makes the round trip through memory that separates k_s_q from k_s_invert.  Arithmetic: ref QGModel.py:328-407 (stage
// updates), :469-481 (N_q, [0,0] kept), :497-505 (psi = -wv2i q), :588-593 (ep_psi with the stale q of :401) -- the same
// expressions, in the same order, as k_s_q / etd_update / k_s_invert.
template <int N, int CW>
struct SmallColPlan {
  static constexpr int P = 8, T = N / P, GROUP = CW * T, THREADS = 3 * GROUP;
  typedef WgFft<N, P, CW, false> F;
  // [3 exchange areas][3 hand-over planes of N x CW][reduction scratch]
  static constexpr size_t LDS_BYTES = (size_t)(3 * F::LDS_ELEMS + 3 * N * CW) * sizeof(cd) + 512;
};

template <int N, int CW>
__global__ void __launch_bounds__((SmallColPlan<N, CW>::THREADS))
k_c_qg(MArr Huq, MArr Hvq, EtdArrays ea, int stage, YGeom g, MArr Hu, MArr Hp, MArr Hq, cd* __restrict__ ph_out, double invM,
       const double* __restrict__ kk, const double* __restrict__ ll, const cd* __restrict__ tw, double* __restrict__ bud_part,
       const cd* __restrict__ q_bud) {
  typedef SmallColPlan<N, CW> Y;
  typedef typename Y::F F;
  constexpr int P = Y::P, T = Y::T;
  const int grp = threadIdx.x / Y::GROUP, tig = threadIdx.x % Y::GROUP;
  const int c = tig % CW, j = tig / CW;
  const int k = blockIdx.x * CW + c;
  const bool ok = k < g.width;
  cd* lds = reinterpret_cast<cd*>(nq_smem) + (size_t)grp * F::LDS_ELEMS;
  cd* X1 = reinterpret_cast<cd*>(nq_smem) + 3 * (size_t)F::LDS_ELEMS;      // F[uq], then u-hat
  cd* X2 = X1 + N * CW;                                                     // F[vq], then psi-hat
  cd* X3 = X2 + N * CW;                                                     // the new q-hat (mirrored rows of the budget sums)
  double* red = reinterpret_cast<double*>(X3 + N * CW);
  typename F::Tw twr;
  F::load_tw(twr, j, tw, 1);
  const double kx = ok ? kk[k] : 0.0;
  cd r[P];
  double s3[3] = {0.0, 0.0, 0.0};
  // ---- phase A: forward transforms (groups 0, 1) beside the operand requests (group 2)
  cd o1[P], o2[P], o3[P], o4[P], o5[P];        // (fab, fc of the last stage are fetched where they are used: 64 VGPRs less)
  constexpr bool QB_EARLY = Y::THREADS <= 192;   // (two waves per SIMD at 384 threads: 256 VGPRs, no room for it)
  cd qb[P];                                    // the stale q of the ep_psi sums (stages 0..2)
  double lyv[P];
  if (grp < 2) {
    const MArr& H = grp == 0 ? Huq : Hvq;
#pragma unroll
    for (int t = 0; t < P; ++t) r[t] = ok ? H.ys[(size_t)(j + t * T) * H.pitch + k] : cmake(0, 0);
    F::template run<false>(r, j, c, lds, twr);
#pragma unroll
    for (int t = 0; t < P; ++t) (grp == 0 ? X1 : X2)[(j + t * T) * CW + c] = r[t];
  } else {
#pragma unroll
    for (int t = 0; t < P; ++t) lyv[t] = ll[j + t * T];
    if (ok) {
#pragma unroll
      for (int t = 0; t < P; ++t) {
        const size_t idx = (size_t)(j + t * T) * g.pitch_s + k, ci = (size_t)crow(g, j + t * T, N) * g.pitch_s + k;
        if (QB_EARLY && bud_part && q_bud) qb[t] = q_bud[idx];
        o2[t] = ea.y_in[idx];
        if (stage < 3) {
          o1[t] = ea.Eh[ci];
          o3[t] = ea.Q[ci];
        } else {
          o1[t] = ea.E[ci];
          o3[t] = ea.f0[ci];
        }
        if (stage >= 2) {
          o4[t] = ea.fn0[idx];
          o5[t] = ea.fna[idx];
        }
      }
    }
    F::idle();
  }
  wg_barrier();
  // ---- phase B: pointwise (group 2)
  cd y[P];
  if (grp == 2) {
#pragma unroll
    for (int t = 0; t < P; ++t) {
      const int l = j + t * T;
      const size_t idx = (size_t)l * g.pitch_s + k;
      const double ly = lyv[t];
      const cd f1 = X1[l * CW + c], f2 = X2[l * CW + c];
      const cd Nl = cmake(kx * f1.y + ly * f2.y, -(kx * f1.x + ly * f2.x));          // N_q = -(ik F1 + il F2), [0,0] kept
      cd yy = cmake(0, 0);
      if (ok) {
        if (stage == 0) {
          yy = cadd(cmul(o1[t], o2[t]), cmul(o3[t], Nl));
          ea.fn0[idx] = Nl;
        } else if (stage == 1) {
          yy = cadd(cmul(o1[t], o2[t]), cmul(o3[t], Nl));
          ea.fna[idx] = Nl;
        } else if (stage == 2) {
          const cd comb = cmake(2.0 * Nl.x - o4[t].x, 2.0 * Nl.y - o4[t].y);
          yy = cadd(cmul(o1[t], o2[t]), cmul(o3[t], comb));
          ea.fna[idx] = cadd(o5[t], Nl);
        } else {
          const size_t ci = (size_t)crow(g, l, N) * g.pitch_s + k;
          yy = cadd(cadd(cmul(o1[t], o2[t]), cmul(o3[t], o4[t])), cadd(cscale(cmul(ea.fab[ci], o5[t]), 2.0), cmul(ea.fc[ci], Nl)));
        }
        ea.y_out[idx] = yy;
      }
      y[t] = yy;
      X3[l * CW + c] = yy;
    }
  }
  wg_barrier();
  if (grp == 2) {
    const bool special = ok && (k == 0 || k == N / 2);
#pragma unroll
    for (int t = 0; t < P; ++t) {
      const int l = j + t * T;
      const size_t idx = (size_t)l * g.pitch_s + k;
      const double ly = lyv[t];
      const double wv2 = kx * kx + ly * ly;
      const double wv2i = (wv2 != 0.0) ? 1.0 / wv2 : 0.0;
      const cd qv = y[t];
      const cd psi = cmake(-wv2i * qv.x, -wv2i * qv.y);
      if (ok && ph_out) ph_out[idx] = psi;
      if (bud_part && ok) {
        cd qB = q_bud ? (QB_EARLY ? qb[t] : q_bud[idx]) : qv, qQ = qv, ps = psi;
        if (special) {      // Hermitian part (in l) of the self-mirrored columns: what irfft2 keeps (k_s_invert)
          const int lm = (N - l) % N;
          const cd qm = X3[lm * CW + c];
          const cd qbm = q_bud ? q_bud[(size_t)lm * g.pitch_s + k] : qm;
          const cd psm = cmake(-wv2i * qm.x, -wv2i * qm.y);
          qQ = cmake(0.5 * (qv.x + qm.x), 0.5 * (qv.y - qm.y));
          qB = cmake(0.5 * (qB.x + qbm.x), 0.5 * (qB.y - qbm.y));
          ps = cmake(0.5 * (psi.x + psm.x), 0.5 * (psi.y - psm.y));
        }
        const double wgt = special ? 1.0 : 2.0;
        const double rb = wgt * (qB.x * ps.x + qB.y * ps.y), rq = wgt * (qQ.x * ps.x + qQ.y * ps.y);
        s3[0] += wv2 * wv2 * rb;
        s3[1] += wv2 * rq;
        s3[2] += rb;
      }
      X1[l * CW + c] = cmake(ly * psi.y * invM, -ly * psi.x * invM);       // -i l psi (QGModel: literal irfft2 arithmetic)
      X2[l * CW + c] = cscale(psi, invM);
      r[t] = cscale(qv, invM);
    }
  }
  wg_barrier();
  // ---- phase C: the three inverse transforms side by side
  if (grp < 2) {
#pragma unroll
    for (int t = 0; t < P; ++t) r[t] = (grp == 0 ? X1 : X2)[(j + t * T) * CW + c];
  }
  F::template run<true>(r, j, c, lds, twr);
  if (ok) {
    const MArr& O = grp == 0 ? Hu : (grp == 1 ? Hp : Hq);
#pragma unroll
    for (int t = 0; t < P; ++t) O.ys[(size_t)(j + t * T) * O.pitch + k] = r[t];
  }
  if (bud_part) block_sum_store<3>(s3, red, bud_part + 3 * (size_t)blockIdx.x);
}

// ---- physical rows <-> mixed-space rows of the slab layout (set_q / set_phi / field reads of a slab context) --------
// Same transforms as k_x_r2c / k_x_c2c / k_x_c2r (nq_generic.hpp), addressed through MArr so that the rows land in / come
// from the x side of an exchange group.  `rows` are this rank's local rows, contiguous (nrows, N).
template <int N, bool SLAB>
__global__ void __launch_bounds__(XPlan<N>::THREADS)
k_x_put_real(const double* __restrict__ rows, MArr out, int nrows, const cd* __restrict__ tw) {
  typedef XPlan<N> X;
  constexpr int P = X::P, T = X::T;
  const int j = threadIdx.x % T, c = threadIdx.x / T;
  const int row = blockIdx.x * X::C + c;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  typename X::F::Tw twr;
  X::F::load_tw(twr, j, tw, 1);
  cd r[P];
  const bool ok = row < nrows;
#pragma unroll
  for (int t = 0; t < P; ++t) r[t] = cmake(ok ? rows[(size_t)row * N + j + t * T] : 0.0, 0.0);
  X::F::template run<false>(r, j, c, lds, twr);
  if (ok) {
    const XRowT<SLAB> o = xrow<SLAB>(out, (size_t)row);
#pragma unroll
    for (int t = 0; t <= P / 2; ++t) {
      const int kx = j + t * T;
      if (kx <= N / 2) *o.at(kx) = r[t];
    }
  }
}
template <int N, bool SLAB>
__global__ void __launch_bounds__(XPlan<N>::THREADS)
k_x_put_cplx(const cd* __restrict__ rows, MArr out, int nrows, const cd* __restrict__ tw) {
  typedef XPlan<N> X;
  constexpr int P = X::P, T = X::T;
  const int j = threadIdx.x % T, c = threadIdx.x / T;
  const int row = blockIdx.x * X::C + c;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  typename X::F::Tw twr;
  X::F::load_tw(twr, j, tw, 1);
  cd r[P];
  const bool ok = row < nrows;
#pragma unroll
  for (int t = 0; t < P; ++t) r[t] = ok ? rows[(size_t)row * N + j + t * T] : cmake(0, 0);
  X::F::template run<false>(r, j, c, lds, twr);
  if (ok) {
    const XRowT<SLAB> o = xrow<SLAB>(out, (size_t)row);
#pragma unroll
    for (int t = 0; t < P; ++t) *o.at(j + t * T) = r[t];
  }
}
// half-spectrum rows -> real rows (numpy.fft.irfft along x: imaginary parts of kx = 0 and N/2 ignored); mode 1: the rows are
// multiplied by i*kk first (v = Re ifft(ik psi)), zero_nyq drops column N/2 (Kernel family, DESIGN.md "Nyquist lines")
template <int N, bool SLAB>
__global__ void __launch_bounds__(XPlan<N>::THREADS)
k_x_get_real(MArr in, double* __restrict__ rows, int nrows, const cd* __restrict__ tw, const double* __restrict__ kk,
             int mode, int zero_nyq) {
  typedef XPlan<N> X;
  constexpr int P = X::P, T = X::T;
  const int j = threadIdx.x % T, c = threadIdx.x / T;
  const int row = blockIdx.x * X::C + c;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  typename X::F::Tw twr;
  X::F::load_tw(twr, j, tw, 1);
  cd r[P];
  const bool ok = row < nrows;
  const XRowT<SLAB> src = xrow<SLAB>(in, (size_t)(ok ? row : 0));
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const int kx = j + t * T;
    cd v = cmake(0, 0);
    if (ok) {
      const int m = kx <= N / 2 ? kx : N - kx;
      v = *src.at(m);
      if (mode == 1) v = cscale(cmul_i(v), kk[m]);
      if (m == 0 || m == N / 2) v.y = 0.0;
      if (zero_nyq && m == N / 2) v.x = 0.0;
      if (kx > N / 2) v = cconj(v);
    }
    r[t] = v;
  }
  X::F::template run<true>(r, j, c, lds, twr);
  if (ok) {
#pragma unroll
    for (int t = 0; t < P; ++t) rows[(size_t)row * N + j + t * T] = r[t].x;
  }
}
template <int N, bool SLAB>
__global__ void __launch_bounds__(XPlan<N>::THREADS)
k_x_get_cplx(MArr in, cd* __restrict__ rows, int nrows, const cd* __restrict__ tw, const double* __restrict__ kk, int mul_ik) {
  typedef XPlan<N> X;
  constexpr int P = X::P, T = X::T;
  const int j = threadIdx.x % T, c = threadIdx.x / T;
  const int row = blockIdx.x * X::C + c;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  typename X::F::Tw twr;
  X::F::load_tw(twr, j, tw, 1);
  cd r[P];
  const bool ok = row < nrows;
  const XRowT<SLAB> src = xrow<SLAB>(in, (size_t)(ok ? row : 0));
#pragma unroll
  for (int t = 0; t < P; ++t) {
    const int kx = j + t * T;
    cd v = ok ? *src.at(kx) : cmake(0, 0);
    if (mul_ik) v = cscale(cmul_i(v), kk[kx]);
    r[t] = v;
  }
  X::F::template run<true>(r, j, c, lds, twr);
  if (ok) {
#pragma unroll
    for (int t = 0; t < P; ++t) rows[(size_t)row * N + j + t * T] = r[t];
  }
}

}  // namespace nq
