// Workgroup-cooperative complex128 FFT for gfx950 (MI355X), data resident in registers.
//
// WgFft<N, P, C, LINE_MAJOR>::run<INV>() transforms C independent lines of length N with
// T = N/P threads per line (block = C*T threads).  Thread (j, c) holds, before and after,
//        r[t] = x_c[j + t*T],   t = 0..P-1          ("canonical distribution")
// so consecutive j are consecutive elements: global loads/stores of a contiguous line coalesce
// without any staging, and the same holds for column tiles (consecutive c = adjacent columns).
//
// Algorithm: Stockham autosort, radices <= 16 chosen at compile time.  With the canonical
// distribution every stage's butterfly inputs are already thread-local (butterfly jj = j + b*T
// reads x[jj + u*N/R] = r[b + u*P/R]); only the outputs have to be re-distributed, through LDS,
// between stages -- (stages-1) exchanges of 16 B/point, the last stage lands canonical again.
// LDS rows are padded by one element per 16 so the strided stage-1 scatter is conflict-free.
//
// 64-wide wavefronts; no MFMA (there is no contraction here); fp64 throughout.
#pragma once
#include <hip/hip_runtime.h>

namespace nq {

typedef double2 cd;

__device__ __forceinline__ cd cmul(cd a, cd b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cd cmulc(cd a, cd b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a*conj(b)
__device__ __forceinline__ cd cadd(cd a, cd b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cd csub(cd a, cd b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cd cscale(cd a, double s) { return make_double2(a.x * s, a.y * s); }
__device__ __forceinline__ cd cconj(cd a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ cd cmul_i(cd a) { return make_double2(-a.y, a.x); }    // a * i
__device__ __forceinline__ cd cmul_mi(cd a) { return make_double2(a.y, -a.x); }   // a * (-i)
__device__ __forceinline__ cd cmake(double x, double y) { return make_double2(x, y); }

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding
// global load/store (s_waitcnt vmcnt(0)), which would serialise the prefetched rows behind each of the
// FFT's exchanges.  0xC07F = lgkmcnt(0) with vmcnt/expcnt left alone.
__device__ __forceinline__ void wg_barrier() {
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ void wg_barrier_all() { __syncthreads(); }   // also waits for global memory

// ---- small DFTs on register arrays, natural order in and out ------------------------------
// forward: X[k] = sum x[n] exp(-2 pi i n k / R);  INV: conjugate kernel, no scaling.
template <bool INV> __device__ __forceinline__ cd rot90(cd a) { return INV ? cmul_i(a) : cmul_mi(a); }  // * exp(-+ i pi/2)

template <bool INV> __device__ __forceinline__ void dft2(cd& a, cd& b) {
  cd t = csub(a, b);
  a = cadd(a, b);
  b = t;
}

template <bool INV> __device__ __forceinline__ void dft4(cd& a0, cd& a1, cd& a2, cd& a3) {
  cd t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = rot90<INV>(csub(a1, a3));
  a0 = cadd(t0, t2);
  a1 = cadd(t1, t3);
  a2 = csub(t0, t2);
  a3 = csub(t1, t3);
}

// exp(-+ 2 pi i m/16), m = 0..15 (only the ones used)
#define NQ_C1 0.92387953251128673848   // cos(pi/8)
#define NQ_S1 0.38268343236508978178   // sin(pi/8)
#define NQ_R2 0.70710678118654752440   // sqrt(1/2)

template <bool INV> __device__ __forceinline__ cd tw16(cd a, int m) {
  // multiply by w16^m (forward: exp(-2 pi i m/16)); m is a compile-time constant after unrolling
  switch (m & 15) {
    case 0: return a;
    case 1: return INV ? cmul(a, cmake(NQ_C1, NQ_S1)) : cmul(a, cmake(NQ_C1, -NQ_S1));
    case 2: return INV ? cscale(cmake(a.x - a.y, a.x + a.y), NQ_R2) : cscale(cmake(a.x + a.y, a.y - a.x), NQ_R2);
    case 3: return INV ? cmul(a, cmake(NQ_S1, NQ_C1)) : cmul(a, cmake(NQ_S1, -NQ_C1));
    case 4: return rot90<INV>(a);
    case 5: return INV ? cmul(a, cmake(-NQ_S1, NQ_C1)) : cmul(a, cmake(-NQ_S1, -NQ_C1));
    case 6: return INV ? cscale(cmake(-a.x - a.y, a.x - a.y), NQ_R2) : cscale(cmake(a.y - a.x, -a.x - a.y), NQ_R2);
    case 7: return INV ? cmul(a, cmake(-NQ_C1, NQ_S1)) : cmul(a, cmake(-NQ_C1, -NQ_S1));
    case 8: return cmake(-a.x, -a.y);
    case 9: return INV ? cmul(a, cmake(-NQ_C1, -NQ_S1)) : cmul(a, cmake(-NQ_C1, NQ_S1));
    default: return a;   // 10..15 never needed (max index 3*3 = 9)
  }
}

template <int R, bool INV> struct Dft;

template <bool INV> struct Dft<2, INV> {
  __device__ __forceinline__ static void run(cd (&v)[2]) { dft2<INV>(v[0], v[1]); }
};
template <bool INV> struct Dft<4, INV> {
  __device__ __forceinline__ static void run(cd (&v)[4]) { dft4<INV>(v[0], v[1], v[2], v[3]); }
};
template <bool INV> struct Dft<8, INV> {
  // 8 = 2 (n1) x 4 (n2): n = 4*n1 + n2, k = k1 + 2*k2
  __device__ __forceinline__ static void run(cd (&v)[8]) {
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) dft2<INV>(v[n2], v[4 + n2]);          // over n1 -> k1 at v[4*k1+n2]
#pragma unroll
    for (int n2 = 1; n2 < 4; ++n2) v[4 + n2] = tw16<INV>(v[4 + n2], 2 * n2);   // w8^(n2*k1), k1 = 1
    dft4<INV>(v[0], v[1], v[2], v[3]);                                  // k1 = 0: over n2 -> k2
    dft4<INV>(v[4], v[5], v[6], v[7]);                                  // k1 = 1
    // now v[4*k1 + k2] = X[k1 + 2*k2]  -> reorder to natural
    cd o[8];
#pragma unroll
    for (int k1 = 0; k1 < 2; ++k1)
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) o[k1 + 2 * k2] = v[4 * k1 + k2];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = o[i];
  }
};
template <bool INV> struct Dft<16, INV> {
  // 16 = 4 (n1) x 4 (n2): n = 4*n1 + n2, k = k1 + 4*k2
  __device__ __forceinline__ static void run(cd (&v)[16]) {
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) dft4<INV>(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);   // -> v[4*k1+n2]
#pragma unroll
    for (int k1 = 1; k1 < 4; ++k1)
#pragma unroll
      for (int n2 = 1; n2 < 4; ++n2) v[4 * k1 + n2] = tw16<INV>(v[4 * k1 + n2], n2 * k1);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4<INV>(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
    cd o[16];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) o[k1 + 4 * k2] = v[4 * k1 + k2];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = o[i];
  }
};

// ---- stage twiddles ----------------------------------------------------------------------------
// v[u] *= w^u, u = 1..R-1.  Only w^1, w^4 (and w^8 for radix 16) come from the table; the others are
// products of at most two fetched/derived values (<= 2 roundings deep).  The fetched values are loaded
// ONCE per kernel into registers (WgFft::Tw): they depend on the thread and the stage only, not on the
// field, and a global load inside the transform would make every barrier-free prefetch wait (vmcnt is
// in order).
template <int R, bool INV>
__device__ __forceinline__ void twiddle_apply(cd (&v)[R], cd w1, cd w4, cd w8) {
  if (INV) {
    w1.y = -w1.y;
    w4.y = -w4.y;
    w8.y = -w8.y;
  }
  v[1] = cmul(v[1], w1);
  if constexpr (R >= 4) {
    const cd w2 = cmul(w1, w1);
    const cd w3 = cmul(w2, w1);
    v[2] = cmul(v[2], w2);
    v[3] = cmul(v[3], w3);
    if constexpr (R >= 8) {
      v[4] = cmul(v[4], w4);
      v[5] = cmul(v[5], cmul(w4, w1));
      v[6] = cmul(v[6], cmul(w4, w2));
      v[7] = cmul(v[7], cmul(w4, w3));
      if constexpr (R >= 16) {
        const cd w12 = cmul(w8, w4);
        v[8] = cmul(v[8], w8);
        v[9] = cmul(v[9], cmul(w8, w1));
        v[10] = cmul(v[10], cmul(w8, w2));
        v[11] = cmul(v[11], cmul(w8, w3));
        v[12] = cmul(v[12], w12);
        v[13] = cmul(v[13], cmul(w12, w1));
        v[14] = cmul(v[14], cmul(w12, w2));
        v[15] = cmul(v[15], cmul(w12, w3));
      }
    }
  }
}

// ---- radix-8 butterfly WITH its stage twiddles, FMA form ------------------------------------------------------
// X[k] = sum_u v[u] w^u w8^(uk), natural order in and out.  Decimation in time over the bits of u puts one twiddle on
// every radix-2 butterfly (w^4 on the first layer, w^2 w4^k' on the second, w w8^k on the third), and a twiddled
// radix-2 butterfly costs 6 fp64 instructions in FMA form (a' = a + t b by four FMAs, b' = 2a - a' by two) instead of
// 8: 72 + 11 (derived twiddles) = 83 instructions against 48 (twiddle multiplies incl. derived powers) + 56 (DFT) =
// 104 -- the row kernels are bound by fp64 issue + LDS store issue, adds dominate, and the FMA
// half of the pipe was idle.
__device__ __forceinline__ void bfly_tw(cd& a, cd& b, cd t) {
  const double xr = __builtin_fma(-t.y, b.y, __builtin_fma(t.x, b.x, a.x));
  const double xi = __builtin_fma(t.y, b.x, __builtin_fma(t.x, b.y, a.y));
  b = cmake(__builtin_fma(2.0, a.x, -xr), __builtin_fma(2.0, a.y, -xi));
  a = cmake(xr, xi);
}
template <bool INV>
__device__ __forceinline__ void dft8_twiddled(cd (&v)[8], cd w1, cd w4) {
  if (INV) {
    w1.y = -w1.y;
    w4.y = -w4.y;
  }
  const cd w2 = cmake(w1.x * w1.x - w1.y * w1.y, 2.0 * w1.x * w1.y);
  // layer A: (u, u + 4) with w^4
  bfly_tw(v[0], v[4], w4);
  bfly_tw(v[2], v[6], w4);
  bfly_tw(v[1], v[5], w4);
  bfly_tw(v[3], v[7], w4);
  // layer B: w^2 w4^k', k' = 0, 1 (w4 = -i forward, +i inverse)
  const cd w2r = rot90<INV>(w2);
  bfly_tw(v[0], v[2], w2);
  bfly_tw(v[4], v[6], w2r);
  bfly_tw(v[1], v[3], w2);
  bfly_tw(v[5], v[7], w2r);
  // layer C: w w8^k, k = 0..3 (w8 = (1 - i)/sqrt 2 forward, conjugate inverse)
  const cd w18 = INV ? cscale(cmake(w1.x - w1.y, w1.x + w1.y), NQ_R2) : cscale(cmake(w1.x + w1.y, w1.y - w1.x), NQ_R2);
  const cd w1r = rot90<INV>(w1);
  const cd w38 = INV ? cscale(cmake(w1r.x - w1r.y, w1r.x + w1r.y), NQ_R2) : cscale(cmake(w1r.x + w1r.y, w1r.y - w1r.x), NQ_R2);
  bfly_tw(v[0], v[1], w1);      // X0, X4
  bfly_tw(v[4], v[5], w18);     // X1, X5
  bfly_tw(v[2], v[3], w1r);     // X2, X6
  bfly_tw(v[6], v[7], w38);     // X3, X7
  const cd x1 = v[4], x2 = v[2], x3 = v[6], x4 = v[1], x5 = v[5], x6 = v[3];
  v[1] = x1; v[2] = x2; v[3] = x3; v[4] = x4; v[5] = x5; v[6] = x6;
}

// ---- radix plan -----------------------------------------------------------------------------
__host__ __device__ constexpr int nq_min(int a, int b) { return a < b ? a : b; }
// Radices are min(P, 16) except for one smaller "remainder" radix when N is not a power of it.  The remainder
// goes last (fewest twiddled points) except for rows of 8192 and more, where it goes FIRST: the per-stage twiddle
// tables of the row kernels grow with the product of the radices before a stage, and with the remainder last the
// 8192-point table alone would not fit in LDS beside the 128 KB exchange.
__host__ __device__ constexpr bool plan_rem_first(int N) { return N >= 8192; }
__host__ __device__ constexpr int plan_radix(int N, int P, int stage) {
  const int RM = nq_min(P, 16);
  if (plan_rem_first(N)) {
    int first = N;
    while (first > RM) first /= RM;
    return stage == 0 ? first : RM;
  }
  int rem = N, R = 1;
  for (int s = 0; s <= stage; ++s) {
    R = nq_min(rem, RM);
    rem /= R;
  }
  return R;
}
__host__ __device__ constexpr int plan_stages(int N, int P) {
  int n = 0;
  for (int rem = N; rem > 1; ++n) rem /= plan_radix(N, P, n);
  return n;
}
// twiddle powers a stage needs from a table: w^1 | w^4 (radix >= 8) | w^8 (radix 16).  8192-point rows at 8
// points per thread (1024 threads) have 18 KB less LDS than that takes: there w^4 is two squarings of w^1.
__host__ __device__ constexpr bool plan_derive_w4(int N, int P) { return N >= 8192 && P <= 8; }
__host__ __device__ constexpr int plan_tw_rows(int N, int P, int R) {
  return plan_derive_w4(N, P) ? 1 : (R >= 16 ? 3 : (R >= 8 ? 2 : 1));
}
__host__ __device__ constexpr int plan_ns(int N, int P, int stage) {   // product of radices before `stage`
  int ns = 1;
  for (int s = 0; s < stage; ++s) ns *= plan_radix(N, P, s);
  return ns;
}

// XOR swizzle of the element index inside aligned blocks of 16 (one 256-byte LDS row): the stride-16
// scatter of the first stage becomes conflict-free while 16 consecutive elements stay a permutation of
// the same row, so the canonical reads stay conflict-free as well (padding misaligned their lane groups:
// 24 % of LDS cycles were conflicts, profiles/r01_c_pmc.txt).
__host__ __device__ constexpr int plan_max_nb(int N, int P) {     // most butterflies per thread in a stage
  int m = 1;
  for (int s = 0; s < plan_stages(N, P); ++s) {
    const int nb = P / plan_radix(N, P, s);
    m = nb > m ? nb : m;
  }
  return m;
}
__host__ __device__ constexpr int plan_max_radix(int N, int P) {
  int m = 1;
  for (int s = 0; s < plan_stages(N, P); ++s) m = plan_radix(N, P, s) > m ? plan_radix(N, P, s) : m;
  return m;
}

template <int LOG> __device__ __forceinline__ int lds_swz(int p) { return p ^ ((p >> LOG) & ((1 << LOG) - 1)); }
// column tiles (C adjacent lines, element-major) keep the 1-in-16 padding: measured conflict-free there
__device__ __forceinline__ int lds_pad(int p) { return p + (p >> 4); }
template <int N> __host__ __device__ constexpr int lds_line_elems() { return N + (N >> 4) + 1; }

// Twiddle table: tw[m] = exp(-2 pi i m / NT), m in [0, NT); NT is a multiple of N.
template <int N, int P, int C, bool LINE_MAJOR>
struct WgFft {
  static constexpr int T = N / P;
  static constexpr int STAGES = plan_stages(N, P);
  static constexpr int LINE = LINE_MAJOR ? N : lds_line_elems<N>();
  static constexpr int LDS_ELEMS = (STAGES > 1) ? LINE * C : 0;   // cd elements
  static_assert(N % P == 0, "P must divide N");

  static constexpr int NBMAX = plan_max_nb(N, P);
  static constexpr int RMAX = plan_max_radix(N, P);
  struct Tw {                               // per twiddled stage, per butterfly of this thread
    cd w1[STAGES][NBMAX];
    cd w4[RMAX >= 8 ? STAGES : 1][RMAX >= 8 ? NBMAX : 1];
    cd w8[RMAX >= 16 ? STAGES : 1][RMAX >= 16 ? NBMAX : 1];
  };

  template <int STAGE>
  __device__ __forceinline__ static void load_tw_from(Tw& t, int j, const cd* __restrict__ tw, int tw_step) {
    if constexpr (STAGE < STAGES) {
      constexpr int R = plan_radix(N, P, STAGE);
      constexpr int NS = plan_ns(N, P, STAGE);
      constexpr int NB = P / R;
      if constexpr (NS > 1) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int base = ((j + b * T) % NS) * (N / (NS * R)) * tw_step;
          t.w1[STAGE][b] = tw[base];
          if constexpr (R >= 8) t.w4[STAGE][b] = tw[4 * base];
          if constexpr (R >= 16) t.w8[STAGE][b] = tw[8 * base];
        }
      }
      load_tw_from<STAGE + 1>(t, j, tw, tw_step);
    }
  }
  // tw_step = NT / N  (table stride for w_N)
  __device__ __forceinline__ static void load_tw(Tw& t, int j, const cd* __restrict__ tw, int tw_step) {
    load_tw_from<0>(t, j, tw, tw_step);
  }

  // Alternative twiddle source for the row kernels: per-stage tables in LDS, entry jr of stage s holds
  // w^1 | w^4 | w^8 for butterflies with jj % NS == jr (layout: [stage][plan_tw_rows][NS]); built on the host
  // (nq_lib.hip: build_stage_table), copied into LDS once per workgroup.  Costs no registers and no
  // VMEM inside the transforms.
  __host__ __device__ static constexpr int tw_off(int stage) {
    int o = 0;
    for (int s = 1; s < stage; ++s) o += plan_tw_rows(N, P, plan_radix(N, P, s)) * plan_ns(N, P, s);
    return o;
  }
  static constexpr int TW_LDS_ELEMS = tw_off(STAGES);
  struct TwLds {
    const cd* base;
  };

  template <int STAGE>
  __device__ __forceinline__ static void fetch_tw(const Tw& src, int b, int jr, cd& w1, cd& w4, cd& w8) {
    constexpr int R = plan_radix(N, P, STAGE);
    w1 = src.w1[STAGE][b];
    w4 = cmake(1, 0);
    w8 = cmake(1, 0);
    if constexpr (R >= 8) w4 = src.w4[STAGE][b];
    if constexpr (R >= 16) w8 = src.w8[STAGE][b];
  }
  template <int STAGE>
  __device__ __forceinline__ static void fetch_tw(const TwLds& src, int b, int jr, cd& w1, cd& w4, cd& w8) {
    constexpr int R = plan_radix(N, P, STAGE);
    constexpr int NS = plan_ns(N, P, STAGE);
    const cd* t = src.base + tw_off(STAGE);
    w1 = t[jr];
    w4 = cmake(1, 0);
    w8 = cmake(1, 0);
    if constexpr (plan_derive_w4(N, P)) {
      static_assert(!plan_derive_w4(N, P) || R <= 8, "derived w^4 only for radix <= 8");
      if constexpr (R >= 8) {
        const cd w2 = cmul(w1, w1);
        w4 = cmul(w2, w2);
      }
    } else {
      if constexpr (R >= 8) w4 = t[NS + jr];
      if constexpr (R >= 16) w8 = t[2 * NS + jr];
    }
  }

  __device__ __forceinline__ static int lds_index(int p, int c) {
    // XOR granule = first-stage radix (8 or 16): its stride-R scatter must spread over R bank groups
    return LINE_MAJOR ? c * N + lds_swz<(plan_radix(N, P, 0) >= 16 ? 4 : 3)>(p) : lds_pad(p) * C + c;
  }

  template <bool INV, int STAGE, typename Src>
  __device__ __forceinline__ static void stage(cd (&r)[P], int j, int c, cd* lds, const Src& twr) {
    constexpr int R = plan_radix(N, P, STAGE);
    constexpr int NS = plan_ns(N, P, STAGE);
    constexpr int NB = P / R;                     // butterflies per thread in this stage
    constexpr bool LAST = (STAGE == STAGES - 1);
    static_assert(P % R == 0, "radix must divide P");
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      cd v[R];
#pragma unroll
      for (int u = 0; u < R; ++u) v[u] = r[b + u * NB];
      const int jj = j + b * T;
      const int jr = jj % NS;
      if constexpr (NS > 1) {
        cd w1, w4, w8;
        fetch_tw<STAGE>(twr, b, jr, w1, w4, w8);
        if constexpr (R == 8) {
          dft8_twiddled<INV>(v, w1, w4);
        } else {
          twiddle_apply<R, INV>(v, w1, w4, w8);
          Dft<R, INV>::run(v);
        }
      } else {
        Dft<R, INV>::run(v);
      }
      if (LAST) {
#pragma unroll
        for (int u = 0; u < R; ++u) r[b + u * NB] = v[u];
      } else {
        const int pos = (jj / NS) * (NS * R) + jr;
#pragma unroll
        for (int u = 0; u < R; ++u) lds[lds_index(pos + u * NS, c)] = v[u];
      }
    }
    if (!LAST) {
      wg_barrier();
      if constexpr (LINE_MAJOR && T % 256 == 0) {
        // (j + t*T) ^ (((j + t*T) >> 4) & 15) = swz(j) + t*T when T is a multiple of 256
        const cd* src = lds + lds_index(j, c);
#pragma unroll
        for (int t = 0; t < P; ++t) r[t] = src[t * T];
      } else {
#pragma unroll
        for (int t = 0; t < P; ++t) r[t] = lds[lds_index(j + t * T, c)];
      }
      wg_barrier();
    }
  }

  template <bool INV, int STAGE, typename Src>
  __device__ __forceinline__ static void stages_from(cd (&r)[P], int j, int c, cd* lds, const Src& twr) {
    if constexpr (STAGE < STAGES) {
      stage<INV, STAGE, Src>(r, j, c, lds, twr);
      stages_from<INV, STAGE + 1, Src>(r, j, c, lds, twr);
    }
  }

  template <bool INV, typename Src>
  __device__ __forceinline__ static void run(cd (&r)[P], int j, int c, cd* lds, const Src& twr) {
    stages_from<INV, 0, Src>(r, j, c, lds, twr);
  }

  // the barriers of one run(), nothing else: for the waves of a workgroup that have no transform of their own while the
  // others run one (array-parallel small-grid kernels: every wave must arrive at every workgroup barrier)
  __device__ __forceinline__ static void idle() {
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) {
      wg_barrier();
      wg_barrier();
    }
  }

  // ---- two independent transforms, stage by stage, through two exchange areas ----------------------
  // The LDS stores of a stage are the expensive half of an exchange (ds_write_b128 moves 80 B/clk/CU against
  // 256 B/clk for the reads) and a lone transform leaves them fully exposed: every wave finishes its butterflies
  // at about the same time, then all of them store.  With two transforms in flight the stores of the first
  // drain under the butterflies of the second, and one pair of barriers serves both.
  template <int STAGE> struct StageTw {
    static constexpr int NB = P / plan_radix(N, P, STAGE);
    cd w1[NB], w4[NB], w8[NB];
  };
  template <int STAGE, typename Src>
  __device__ __forceinline__ static void fetch_stage_tw(StageTw<STAGE>& t, int j, const Src& twr) {
    constexpr int NS = plan_ns(N, P, STAGE);
    if constexpr (NS > 1) {
#pragma unroll
      for (int b = 0; b < StageTw<STAGE>::NB; ++b) fetch_tw<STAGE>(twr, b, (j + b * T) % NS, t.w1[b], t.w4[b], t.w8[b]);
    }
  }
  template <bool INV, int STAGE>
  __device__ __forceinline__ static void butterflies(cd (&r)[P], int j, int c, cd* lds, const StageTw<STAGE>& tw) {
    constexpr int R = plan_radix(N, P, STAGE);
    constexpr int NS = plan_ns(N, P, STAGE);
    constexpr int NB = P / R;
    constexpr bool LAST = (STAGE == STAGES - 1);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      cd v[R];
#pragma unroll
      for (int u = 0; u < R; ++u) v[u] = r[b + u * NB];
      const int jj = j + b * T;
      const int jr = jj % NS;
      if constexpr (NS > 1 && R == 8) {
        dft8_twiddled<INV>(v, tw.w1[b], tw.w4[b]);
      } else {
        if constexpr (NS > 1) twiddle_apply<R, INV>(v, tw.w1[b], tw.w4[b], tw.w8[b]);
        Dft<R, INV>::run(v);
      }
      if (LAST) {
#pragma unroll
        for (int u = 0; u < R; ++u) r[b + u * NB] = v[u];
      } else {
        const int pos = (jj / NS) * (NS * R) + jr;
#pragma unroll
        for (int u = 0; u < R; ++u) lds[lds_index(pos + u * NS, c)] = v[u];
      }
    }
  }
  __device__ __forceinline__ static void gather(cd (&r)[P], int j, int c, const cd* lds) {
    if constexpr (LINE_MAJOR && T % 256 == 0) {
      const cd* src = lds + lds_index(j, c);
#pragma unroll
      for (int t = 0; t < P; ++t) r[t] = src[t * T];
    } else {
#pragma unroll
      for (int t = 0; t < P; ++t) r[t] = lds[lds_index(j + t * T, c)];
    }
  }
  template <bool INV, int STAGE, typename Src>
  __device__ __forceinline__ static void stages2_from(cd (&a)[P], cd (&b)[P], int j, int c, cd* lds_a, cd* lds_b,
                                                      const Src& twr) {
    if constexpr (STAGE < STAGES) {
      // the stage's twiddles serve both transforms and are fetched BEFORE any store: LDS operations complete in
      // order, so a table read issued after the stores of `a` would wait for them
      StageTw<STAGE> tw;
      fetch_stage_tw<STAGE>(tw, j, twr);
      butterflies<INV, STAGE>(a, j, c, lds_a, tw);
      butterflies<INV, STAGE>(b, j, c, lds_b, tw);
      if constexpr (STAGE < STAGES - 1) {
        wg_barrier();
        gather(a, j, c, lds_a);
        gather(b, j, c, lds_b);
        wg_barrier();
      }
      stages2_from<INV, STAGE + 1, Src>(a, b, j, c, lds_a, lds_b, twr);
    }
  }
  template <bool INV, typename Src>
  __device__ __forceinline__ static void run2(cd (&a)[P], cd (&b)[P], int j, int c, cd* lds_a, cd* lds_b,
                                              const Src& twr) {
    stages2_from<INV, 0, Src>(a, b, j, c, lds_a, lds_b, twr);
  }
};


}  // namespace nq
