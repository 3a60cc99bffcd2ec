// Row-engine microbenchmark, round 2: candidate engines for the 4096-point rows of the fused row kernels, data never leaving
// registers / LDS.  Each workgroup loads one row, runs ITERS x (inverse, forward, scale 1/N) on it and stores it: the
// result must equal the input (checked), the time per transform per CU is what the row kernels pay per transform.
//   G8   generic WgFft, 512 threads x 8 points, three workgroup-wide exchanges            (round 1)
//   W8   RowFft<4096,8,1>: one workgroup-wide exchange + two wave-local ones             (nq_fft.hpp)
//   V16  256 threads x 16 points: radix 16 x 16 x 16, one workgroup-wide exchange + one exchange inside 16-lane groups
// Ablations (template FLAGS): 1 = no LDS stores, 2 = no LDS loads, 4 = no butterflies/twiddles -- wrong results, timing only.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../niwqg_amd/csrc/nq_fft.hpp"
#include "../niwqg_amd/csrc/nq_generic.hpp"
namespace nq {
// The wave-local engine "W8" tried in round 2 for the fused row kernels (it was wired into k_x_products / k_x_wavepv2 and
// passed every parity test, but ran 7 % slower than the round-1 engine there: kept here as the measured experiment).

// ================================================================================================================
// Row engines of the fused row kernels (k_x_products, k_x_wavepv2): RowFft<N, P, C>::run<INV>(r, j, c, ex, tab).
//
// Contract: an INVERSE transform takes the canonical distribution (thread j holds X[j + t*T]) and leaves the result in
// the engine's PHYSICAL distribution; a FORWARD transform takes the physical distribution and lands canonical.  The
// fused kernels only ever combine physical-space fields point by point, so the physical distribution may be any
// permutation -- as long as every field of a kernel goes through the same engine.
//
// Generic engine: WgFft with its per-stage twiddle tables in LDS; physical distribution = canonical.  Every one of
// its (stages-1) exchanges is workgroup-wide: two barriers each, and because all waves reach the store phase of an
// exchange together, the LDS store path (ds_write_b128: ~80 B/clk/CU, MI355X_MICROARCH.md) and the VALU never
// overlap -- 3.06 us per 4096-point transform, of which ~1.5 us VALU and ~1.4 us LDS.
template <int N, int P, int C>
struct RowFft {
  typedef WgFft<N, P, C, true> F;
  static constexpr bool WAVE_LOCAL = false;
  static constexpr int EX_ELEMS = F::LDS_ELEMS;        // exchange area (shared with hs_pack / unpack_pair_store)
  static constexpr int TAB_ELEMS = F::TW_LDS_ELEMS;    // tables behind it
  struct Tab {
    typename F::TwLds twr;
  };
  // stage_tab: host-built per-stage table (WgFft::tw_off layout); twN: exp(-2 pi i m / N), m < N (unused here)
  __device__ __forceinline__ static void load_tables(Tab& t, cd* tab_lds, const cd* __restrict__ stage_tab,
                                                     const cd* __restrict__ twN, int tid, int nthreads) {
    for (int i = tid; i < TAB_ELEMS; i += nthreads) tab_lds[i] = stage_tab[i];
    t.twr.base = tab_lds;
  }
  // a workgroup-wide LDS writer that follows a transform must call this first (the generic engine ends every
  // exchange with a barrier: nothing to do)
  __device__ __forceinline__ static void before_wg_write() {}
  template <bool INV>
  __device__ __forceinline__ static void run(cd (&r)[P], int j, int c, cd* ex, const Tab& t) {
    F::template run<INV>(r, j, c, ex, t.twr);
  }
};

// 4096-point rows, 512 threads x 8 points: ONE workgroup-wide exchange per transform, the rest wave-local.
//   inverse (decimation in frequency): radix-8 over the thread's own 8 points (stride 512), twiddle w_N^(n1 j), exchange
//     LDS[n1][j] -> wave n1 owns the 512-point problem n1 and finishes it alone: x[n1 + 8 n2], n2 = lane + 64 s;
//   forward (decimation in time): wave w transforms its 512 points alone, twiddle w_N^(w k2), exchange LDS[w][k2] ->
//     thread j gathers column j and a radix-8 butterfly lands X[j + 512 t]: canonical.
// The 512-point problems run radix 8 x 8 x 8 inside ONE wave through that wave's own row of the exchange area: LDS
// operations of a wave complete in order, so a wave needs no barrier around its private exchanges, the eight waves
// drift apart, and the stores of one wave drain under the butterflies of the others (the two waves of a SIMD
// included).  Two barriers per transform instead of six; an inverse followed by a forward runs from the inverse's
// second barrier to the forward's only one without any workgroup synchronisation.
// Every twiddle is ONE table read (no derived powers: fewer roundings and ~15 complex multiplies less per
// transform) from tables laid out [power][thread], so that a thread's reads are one base address plus immediate
// offsets and a wave's reads are contiguous: w_N^(n j) as [n][j] (64 KB), w_512^(u lane) as [u][lane], w_64^(u jr)
// as [u][jr], in LDS beside the 64 KB exchange area.
template <>
struct RowFft<4096, 8, 1> {
  static constexpr int N = 4096, P = 8, C = 1, T = 512, WAVES = 8, M = 512;
  typedef WgFft<N, P, C, true> F;                      // only for lds_index of hs_pack / unpack_pair_store
  static constexpr bool WAVE_LOCAL = true;
  static constexpr int EX_ELEMS = N;
  static constexpr int TAB_ELEMS = N + M + 64;
  struct Tab {
    const cd* twN;      // [n][j]    exp(-2 pi i n j / 4096), n < 8, j < 512
    const cd* t512;     // [u][lane] exp(-2 pi i u lane / 512), u < 8, lane < 64
    const cd* t64;      // [u][jr]   exp(-2 pi i u jr / 64), u < 8, jr < 8
  };
  // twN: the context's table exp(-2 pi i m / 4096), m < 4096, in global memory
  __device__ __forceinline__ static void load_tables(Tab& t, cd* tab_lds, const cd* __restrict__ stage_tab,
                                                     const cd* __restrict__ twN, int tid, int nthreads) {
    for (int i = tid; i < N; i += nthreads) tab_lds[i] = twN[(i >> 9) * (i & 511)];
    for (int i = tid; i < M; i += nthreads) tab_lds[N + i] = twN[8 * ((i >> 6) * (i & 63))];
    for (int i = tid; i < 64; i += nthreads) tab_lds[N + M + i] = twN[64 * ((i >> 3) * (i & 7))];
    t.twN = tab_lds;
    t.t512 = tab_lds + N;
    t.t64 = tab_lds + N + M;
  }
  __device__ __forceinline__ static void before_wg_write() { wg_barrier(); }

  template <bool INV> __device__ __forceinline__ static cd twmul(cd a, cd w) { return INV ? cmulc(a, w) : cmul(a, w); }
  __device__ __forceinline__ static int swz(int p) { return p ^ ((p >> 3) & 7); }
  // order the private exchange of one wave for the compiler; the hardware keeps a wave's LDS operations in order
  __device__ __forceinline__ static void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }

  // 512-point transform of one wave: lane holds x[lane + 64 s] before and X[lane + 64 s] after; `row` = the wave's own
  // 512 elements of the exchange area.  Stockham radix 8 x 8 x 8 (same index algebra as WgFft::stage with T = 64).
  template <bool INV>
  __device__ __forceinline__ static void wave512(cd (&r)[8], int lane, cd* row, const Tab& t) {
    Dft<8, INV>::run(r);                                            // stage 0: NS = 1, no twiddle
#pragma unroll
    for (int u = 0; u < 8; ++u) row[swz(lane * 8 + u)] = r[u];
    wave_fence();
#pragma unroll
    for (int s = 0; s < 8; ++s) r[s] = row[swz(lane + 64 * s)];
    wave_fence();
    {                                                               // stage 1: NS = 8, w_64^(jr u)
      const int jr = lane & 7;
#pragma unroll
      for (int u = 1; u < 8; ++u) r[u] = twmul<INV>(r[u], t.t64[8 * u + jr]);
      Dft<8, INV>::run(r);
      const int pos = (lane >> 3) * 64 + jr;
#pragma unroll
      for (int u = 0; u < 8; ++u) row[swz(pos + 8 * u)] = r[u];
    }
    wave_fence();
#pragma unroll
    for (int s = 0; s < 8; ++s) r[s] = row[swz(lane + 64 * s)];
    wave_fence();
#pragma unroll
    for (int u = 1; u < 8; ++u) r[u] = twmul<INV>(r[u], t.t512[64 * u + lane]);   // stage 2: NS = 64, w_512^(lane u)
    Dft<8, INV>::run(r);                                            // output position lane + 64 u: in place
  }

  template <bool INV>
  __device__ __forceinline__ static void run(cd (&r)[P], int j, int c, cd* ex, const Tab& t) {
    (void)c;
    // every LDS address of a transform derives from j: make them per-call values, or the compiler keeps some twenty
    // address registers alive across all the transforms of a row kernel and spills (a spill reload is a VMEM
    // operation: waiting for it also waits for every prefetched row, vmcnt being in order)
    asm volatile("" : "+v"(j));
    const int lane = j & 63, w = j >> 6;
    cd* row = ex + w * M;
    if (INV) {
      Dft<8, true>::run(r);                                         // over t: y[n1]
#pragma unroll
      for (int n1 = 1; n1 < 8; ++n1) r[n1] = cmulc(r[n1], t.twN[n1 * M + j]);
      wg_barrier();                                                 // everyone is done with the rows
#pragma unroll
      for (int n1 = 0; n1 < 8; ++n1) ex[n1 * M + j] = r[n1];
      wg_barrier();
#pragma unroll
      for (int s = 0; s < 8; ++s) r[s] = row[lane + 64 * s];
      wave_fence();
      wave512<true>(r, lane, row, t);                               // r[s] = x[w + 8 (lane + 64 s)]
    } else {
      wave512<false>(r, lane, row, t);                              // r[s] = Y_w[lane + 64 s]
      if (w != 0) {
#pragma unroll
        for (int s = 0; s < 8; ++s) r[s] = cmul(r[s], t.twN[w * M + lane + 64 * s]);
      }
      wave_fence();
#pragma unroll
      for (int s = 0; s < 8; ++s) row[lane + 64 * s] = r[s];
      wg_barrier();
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) r[tt] = ex[tt * M + j];
      Dft<8, false>::run(r);                                        // X[j + 512 t]
    }
  }
};

}  // namespace nq
using namespace nq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// ---- V16 ------------------------------------------------------------------------------------------------------
template <int FLAGS>
struct V16 {
  static constexpr int N = 4096, P = 16, T = 256, M = 256;
  static constexpr int EX_ELEMS = N, TAB_ELEMS = N + 256;
  struct Tab { const cd* twN; const cd* t256; };
  __device__ static void load_tables(Tab& t, cd* tab_lds, const cd* __restrict__ twN, int tid, int nthreads) {
    for (int i = tid; i < N; i += nthreads) tab_lds[i] = twN[(i >> 8) * (i & 255)];                 // [n1][j]
    for (int i = tid; i < 256; i += nthreads) tab_lds[N + i] = twN[16 * ((i >> 4) * (i & 15))];      // [u][l]: w_256^(u l)
    t.twN = tab_lds;
    t.t256 = tab_lds + N;
  }
  __device__ __forceinline__ static int swz(int p) { return p ^ ((p >> 4) & 15); }
  __device__ __forceinline__ static void fence() { __builtin_amdgcn_wave_barrier(); }
  template <bool INV> __device__ __forceinline__ static cd twmul(cd a, cd w) { return INV ? cmulc(a, w) : cmul(a, w); }
  template <bool INV> __device__ __forceinline__ static void bfly(cd (&r)[16]) { if (!(FLAGS & 4)) Dft<16, INV>::run(r); }

  // 256-point transform inside one 16-lane group: lane l holds z[l + 16 s]; `row` = the group's 256 elements
  template <bool INV>
  __device__ __forceinline__ static void group256(cd (&r)[16], int l, cd* row, const Tab& t) {
    bfly<INV>(r);
    if (!(FLAGS & 1)) {
#pragma unroll
      for (int u = 0; u < 16; ++u) row[swz(l * 16 + u)] = r[u];
    }
    fence();
    if (!(FLAGS & 2)) {
#pragma unroll
      for (int s = 0; s < 16; ++s) r[s] = row[swz(l + 16 * s)];
    }
    fence();
    if (!(FLAGS & 4)) {
#pragma unroll
      for (int u = 1; u < 16; ++u) r[u] = twmul<INV>(r[u], t.t256[16 * u + l]);
    }
    bfly<INV>(r);
  }
  template <bool INV>
  __device__ __forceinline__ static void run(cd (&r)[16], int j, cd* ex, const Tab& t) {
    asm volatile("" : "+v"(j));
    const int l = j & 15, n1 = j >> 4;            // group = subproblem n1 (16 groups of 16 lanes)
    cd* row = ex + n1 * M;
    if (INV) {
      bfly<true>(r);
      if (!(FLAGS & 4)) {
#pragma unroll
        for (int n = 1; n < 16; ++n) r[n] = cmulc(r[n], t.twN[n * M + j]);
      }
      wg_barrier();
      if (!(FLAGS & 1)) {
#pragma unroll
        for (int n = 0; n < 16; ++n) ex[n * M + j] = r[n];
      }
      wg_barrier();
      if (!(FLAGS & 2)) {
#pragma unroll
        for (int s = 0; s < 16; ++s) r[s] = row[l + 16 * s];
      }
      fence();
      group256<true>(r, l, row, t);
    } else {
      group256<false>(r, l, row, t);
      if (!(FLAGS & 4)) {
#pragma unroll
        for (int s = 0; s < 16; ++s) r[s] = cmul(r[s], t.twN[n1 * M + l + 16 * s]);
      }
      fence();
      if (!(FLAGS & 1)) {
#pragma unroll
        for (int s = 0; s < 16; ++s) row[l + 16 * s] = r[s];
      }
      wg_barrier();
      if (!(FLAGS & 2)) {
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) r[tt] = ex[tt * M + j];
      }
      bfly<false>(r);
    }
  }
};

template <int FLAGS>
__global__ void __launch_bounds__(256, 1)
k_v16(const cd* __restrict__ in, cd* __restrict__ out, const cd* __restrict__ twN, int iters) {
  typedef V16<FLAGS> E;
  const int j = threadIdx.x;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  typename E::Tab tab;
  E::load_tables(tab, lds + E::EX_ELEMS, twN, threadIdx.x, 256);
  wg_barrier_all();
  cd r[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) r[t] = in[(size_t)blockIdx.x * 4096 + j + t * 256];
  for (int it = 0; it < iters; ++it) {
    E::template run<true>(r, j, lds, tab);
#pragma unroll
    for (int t = 0; t < 16; ++t) r[t] = cscale(r[t], 1.0 / 4096.0);
    E::template run<false>(r, j, lds, tab);
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) out[(size_t)blockIdx.x * 4096 + j + t * 256] = r[t];
}

// ---- W8P: two transforms in flight per wave, wave-local stages interleaved (stores of one under the butterflies of the other)
template <int FLAGS>
struct W8P {
  static constexpr int N = 4096, M = 512;
  static constexpr int EX_ELEMS = 2 * N, TAB_ELEMS = 512 + 64 + 512 + 64;
  struct Tab { const cd* t1; const cd* t2; const cd* t512; const cd* t64; };
  __device__ static void load_tables(Tab& t, cd* tab_lds, const cd* __restrict__ twN, int tid, int nthreads) {
    for (int i = tid; i < 512; i += nthreads) tab_lds[i] = twN[(i >> 6) * (i & 63)];                    // [n][lane]  w_N^(n lane)
    for (int i = tid; i < 64; i += nthreads) tab_lds[512 + i] = twN[64 * ((i >> 3) * (i & 7))];          // [n][wv]    w_64^(n wv)
    for (int i = tid; i < 512; i += nthreads) tab_lds[576 + i] = twN[8 * ((i >> 6) * (i & 63))];         // [u][lane]  w_512^(u lane)
    for (int i = tid; i < 64; i += nthreads) tab_lds[1088 + i] = twN[64 * ((i >> 3) * (i & 7))];         // [u][jr]    w_64^(u jr)
    t.t1 = tab_lds; t.t2 = tab_lds + 512; t.t512 = tab_lds + 576; t.t64 = tab_lds + 1088;
  }
  template <bool INV> __device__ __forceinline__ static cd twmul(cd a, cd w) { return INV ? cmulc(a, w) : cmul(a, w); }
  __device__ __forceinline__ static int swz(int p) { return p ^ ((p >> 3) & 7); }
  __device__ __forceinline__ static void fence() { __builtin_amdgcn_wave_barrier(); }
  template <bool INV> __device__ __forceinline__ static void bfly(cd (&r)[8]) { if (!(FLAGS & 4)) Dft<8, INV>::run(r); }
  __device__ __forceinline__ static void scatter0(const cd (&r)[8], int lane, cd* row) {
    if (FLAGS & 1) return;
#pragma unroll
    for (int u = 0; u < 8; ++u) row[swz(lane * 8 + u)] = r[u];
  }
  __device__ __forceinline__ static void scatter1(const cd (&r)[8], int lane, cd* row) {
    if (FLAGS & 1) return;
    const int pos = (lane >> 3) * 64 + (lane & 7);
#pragma unroll
    for (int u = 0; u < 8; ++u) row[swz(pos + 8 * u)] = r[u];
  }
  __device__ __forceinline__ static void gather(cd (&r)[8], int lane, const cd* row) {
    if (FLAGS & 2) return;
#pragma unroll
    for (int s = 0; s < 8; ++s) r[s] = row[swz(lane + 64 * s)];
  }
  template <bool INV> __device__ __forceinline__ static void tw1(cd (&r)[8], int lane, const Tab& t) {
    if (FLAGS & 4) return;
    const int jr = lane & 7;
#pragma unroll
    for (int u = 1; u < 8; ++u) r[u] = twmul<INV>(r[u], t.t64[8 * u + jr]);
  }
  template <bool INV> __device__ __forceinline__ static void tw2(cd (&r)[8], int lane, const Tab& t) {
    if (FLAGS & 4) return;
#pragma unroll
    for (int u = 1; u < 8; ++u) r[u] = twmul<INV>(r[u], t.t512[64 * u + lane]);
  }
  // two 512-point transforms of one wave, rows rowa / rowb: each LDS round trip of one hides under the butterflies of the other
  template <bool INV>
  __device__ __forceinline__ static void wave512x2(cd (&a)[8], cd (&b)[8], int lane, cd* rowa, cd* rowb, const Tab& t) {
    bfly<INV>(a);
    scatter0(a, lane, rowa); fence(); gather(a, lane, rowa);
    bfly<INV>(b);
    scatter0(b, lane, rowb); fence(); gather(b, lane, rowb);
    tw1<INV>(a, lane, t); bfly<INV>(a);
    scatter1(a, lane, rowa); fence(); gather(a, lane, rowa);
    tw1<INV>(b, lane, t); bfly<INV>(b);
    scatter1(b, lane, rowb); fence(); gather(b, lane, rowb);
    tw2<INV>(a, lane, t); bfly<INV>(a);
    tw2<INV>(b, lane, t); bfly<INV>(b);
  }
  template <bool INV>
  __device__ __forceinline__ static void run2(cd (&a)[8], cd (&b)[8], int j, cd* ex0, cd* ex1, const Tab& t) {
    asm volatile("" : "+v"(j));
    const int lane = j & 63, w = j >> 6;
    cd *rowa = ex0 + w * M, *rowb = ex1 + w * M;
    if (INV) {
      bfly<true>(a);
      if (!(FLAGS & 4)) {
#pragma unroll
        for (int n = 1; n < 8; ++n) a[n] = cmulc(a[n], cmul(t.t1[64 * n + lane], t.t2[8 * n + w]));
      }
      wg_barrier();
      if (!(FLAGS & 1)) {
#pragma unroll
        for (int n = 0; n < 8; ++n) ex0[n * M + j] = a[n];
      }
      bfly<true>(b);
      if (!(FLAGS & 4)) {
#pragma unroll
        for (int n = 1; n < 8; ++n) b[n] = cmulc(b[n], cmul(t.t1[64 * n + lane], t.t2[8 * n + w]));
      }
      if (!(FLAGS & 1)) {
#pragma unroll
        for (int n = 0; n < 8; ++n) ex1[n * M + j] = b[n];
      }
      wg_barrier();
      if (!(FLAGS & 2)) {
#pragma unroll
        for (int s = 0; s < 8; ++s) a[s] = rowa[lane + 64 * s];
#pragma unroll
        for (int s = 0; s < 8; ++s) b[s] = rowb[lane + 64 * s];
      }
      fence();
      wave512x2<true>(a, b, lane, rowa, rowb, t);
    } else {
      wave512x2<false>(a, b, lane, rowa, rowb, t);
      if (!(FLAGS & 4)) {
#pragma unroll
        for (int s = 0; s < 8; ++s) a[s] = cmul(a[s], cmul(t.t1[64 * w + lane], t.t2[8 * w + s]));   // w_N^(w (lane + 64 s))
      }
      fence();
      if (!(FLAGS & 1)) {
#pragma unroll
        for (int s = 0; s < 8; ++s) rowa[lane + 64 * s] = a[s];
      }
      if (!(FLAGS & 4)) {
#pragma unroll
        for (int s = 0; s < 8; ++s) b[s] = cmul(b[s], cmul(t.t1[64 * w + lane], t.t2[8 * w + s]));
      }
      if (!(FLAGS & 1)) {
#pragma unroll
        for (int s = 0; s < 8; ++s) rowb[lane + 64 * s] = b[s];
      }
      wg_barrier();
      if (!(FLAGS & 2)) {
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) a[tt] = ex0[tt * M + j];
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) b[tt] = ex1[tt * M + j];
      }
      bfly<false>(a);
      bfly<false>(b);
    }
  }
};

template <int FLAGS>
__global__ void __launch_bounds__(512, 2)
k_w8p(const cd* __restrict__ in, cd* __restrict__ out, const cd* __restrict__ twN, int iters) {
  typedef W8P<FLAGS> E;
  const int j = threadIdx.x;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  typename E::Tab tab;
  E::load_tables(tab, lds + E::EX_ELEMS, twN, threadIdx.x, 512);
  wg_barrier_all();
  cd a[8], b[8];
  // two rows per workgroup: blockIdx.x and blockIdx.x + gridDim.x (256 rows in the buffer: wrap)
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    a[t] = in[(size_t)blockIdx.x * 4096 + j + t * 512];
    b[t] = in[(size_t)((blockIdx.x + 1) % gridDim.x) * 4096 + j + t * 512];
  }
  for (int it = 0; it < iters; ++it) {
    E::template run2<true>(a, b, j, lds, lds + 4096, tab);
#pragma unroll
    for (int t = 0; t < 8; ++t) { a[t] = cscale(a[t], 1.0 / 4096.0); b[t] = cscale(b[t], 1.0 / 4096.0); }
    E::template run2<false>(a, b, j, lds, lds + 4096, tab);
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) out[(size_t)blockIdx.x * 4096 + j + t * 512] = cadd(cscale(a[t], 1.0), cscale(csub(b[t], in[(size_t)((blockIdx.x + 1) % gridDim.x) * 4096 + j + t * 512]), 1.0));
}

// ---- W8: production engine --------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512, 2)
k_w8(const cd* __restrict__ in, cd* __restrict__ out, const cd* __restrict__ twN, int iters) {
  typedef RowFft<4096, 8, 1> E;
  const int j = threadIdx.x;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  E::Tab tab;
  E::load_tables(tab, lds + E::EX_ELEMS, nullptr, twN, threadIdx.x, 512);
  wg_barrier_all();
  cd r[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) r[t] = in[(size_t)blockIdx.x * 4096 + j + t * 512];
  for (int it = 0; it < iters; ++it) {
    E::run<true>(r, j, 0, lds, tab);
#pragma unroll
    for (int t = 0; t < 8; ++t) r[t] = cscale(r[t], 1.0 / 4096.0);
    E::run<false>(r, j, 0, lds, tab);
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) out[(size_t)blockIdx.x * 4096 + j + t * 512] = r[t];
}

// ---- G8: round-1 engine -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512, 2)
k_g8(const cd* __restrict__ in, cd* __restrict__ out, const cd* __restrict__ twx, int iters) {
  typedef WgFft<4096, 8, 1, true> F;
  const int j = threadIdx.x;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  cd* twl = lds + F::LDS_ELEMS;
  for (int i = threadIdx.x; i < F::TW_LDS_ELEMS; i += 512) twl[i] = twx[i];
  wg_barrier_all();
  F::TwLds src{twl};
  cd r[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) r[t] = in[(size_t)blockIdx.x * 4096 + j + t * 512];
  for (int it = 0; it < iters; ++it) {
    F::run<true>(r, j, 0, lds, src);
#pragma unroll
    for (int t = 0; t < 8; ++t) r[t] = cscale(r[t], 1.0 / 4096.0);
    F::run<false>(r, j, 0, lds, src);
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) out[(size_t)blockIdx.x * 4096 + j + t * 512] = r[t];
}

static std::vector<double> stage_table(int N, int PP, const std::vector<double>& twh) {
  std::vector<double> st;
  for (int sidx = 1; sidx < plan_stages(N, PP); ++sidx) {
    const int R = plan_radix(N, PP, sidx), NS = plan_ns(N, PP, sidx);
    for (int pw = 1; pw <= 8; pw *= (pw == 1 ? 4 : 2)) {
      if ((pw == 4 && plan_tw_rows(N, PP, R) < 2) || (pw == 8 && plan_tw_rows(N, PP, R) < 3)) continue;
      for (int jr = 0; jr < NS; ++jr) {
        const long long m = ((long long)pw * jr * (N / (NS * R))) % N;
        st.push_back(twh[2 * m]);
        st.push_back(twh[2 * m + 1]);
      }
    }
  }
  return st;
}

template <typename K>
static void time_kernel(const char* name, K k, int threads, size_t ldsb, const cd* in, cd* out, const cd* tab, const std::vector<double>& h,
                        bool check, double per_iter = 2.0) {
  const int nwg = 256, iters = 100;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k, dim3(nwg), dim3(threads), ldsb, 0, in, out, tab, 2);
  CK(hipDeviceSynchronize());
  double err = -1.0;
  if (check) {
    std::vector<double> o((size_t)nwg * 4096 * 2);
    CK(hipMemcpy(o.data(), out, o.size() * 8, hipMemcpyDeviceToHost));
    err = 0.0;
    for (size_t i = 0; i < o.size(); ++i) err = fmax(err, fabs(o[i] - h[i]));
  }
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k, dim3(nwg), dim3(threads), ldsb, 0, in, out, tab, iters);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  printf("%-28s threads=%3d lds=%6zu B: %.3f us per transform per CU   max|out-in| after 2 round trips: %.2e\n", name, threads, ldsb,
         ms * 1e3 / (per_iter * iters), err);
}

int main() {
  constexpr int N = 4096;
  const int rows = 256;
  std::vector<double> h((size_t)rows * N * 2), twh(2 * N);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)rand() / RAND_MAX - 0.5;
  for (int m = 0; m < N; ++m) { twh[2 * m] = cos(-2.0 * M_PI * m / N); twh[2 * m + 1] = sin(-2.0 * M_PI * m / N); }
  cd *in, *out, *tw, *twx;
  CK(hipMalloc(&in, h.size() * 8));
  CK(hipMalloc(&out, h.size() * 8));
  CK(hipMalloc(&tw, twh.size() * 8));
  CK(hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(tw, twh.data(), twh.size() * 8, hipMemcpyHostToDevice));
  std::vector<double> st = stage_table(N, 8, twh);
  CK(hipMalloc(&twx, st.size() * 8 + 16));
  CK(hipMemcpy(twx, st.data(), st.size() * 8, hipMemcpyHostToDevice));
  typedef WgFft<4096, 8, 1, true> F;
  typedef RowFft<4096, 8, 1> W;
  time_kernel("G8 generic (round 1)", k_g8, 512, (size_t)(F::LDS_ELEMS + F::TW_LDS_ELEMS) * 16 + 1024, in, out, twx, h, true);
  time_kernel("W8 wave-local", k_w8, 512, (size_t)(W::EX_ELEMS + W::TAB_ELEMS) * 16 + 1024, in, out, tw, h, true);
  const size_t lp = (size_t)(W8P<0>::EX_ELEMS + W8P<0>::TAB_ELEMS) * 16 + 1024;
  time_kernel("W8P pair (per transform)", k_w8p<0>, 512, lp, in, out, tw, h, true, 4.0);
  time_kernel("W8P no LDS loads", k_w8p<2>, 512, lp, in, out, tw, h, false, 4.0);
  time_kernel("W8P no LDS at all", k_w8p<3>, 512, lp, in, out, tw, h, false, 4.0);
  time_kernel("W8P no butterflies", k_w8p<4>, 512, lp, in, out, tw, h, false, 4.0);
  const size_t l16 = (size_t)(V16<0>::EX_ELEMS + V16<0>::TAB_ELEMS) * 16 + 1024;
  time_kernel("V16 radix-16", k_v16<0>, 256, l16, in, out, tw, h, true);
  time_kernel("V16 no LDS stores", k_v16<1>, 256, l16, in, out, tw, h, false);
  time_kernel("V16 no LDS loads", k_v16<2>, 256, l16, in, out, tw, h, false);
  time_kernel("V16 no LDS at all", k_v16<3>, 256, l16, in, out, tw, h, false);
  time_kernel("V16 no butterflies", k_v16<4>, 256, l16, in, out, tw, h, false);
  return 0;
}
