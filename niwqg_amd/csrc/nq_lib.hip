// niwqg_amd: context, precompute kernels and the C ABI (include/niwqg_amd.h).
// gfx950 only; build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC nq_lib.hip -o libniwqg_amd.so
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/niwqg_amd.h"
#include "nq_step.hpp"
#include "nq_anysize.hpp"

using namespace nq;

static thread_local std::string g_last_error;

#define NQ_FAIL(ctx, code, ...)                                  \
  do {                                                           \
    char buf_[512];                                              \
    snprintf(buf_, sizeof(buf_), __VA_ARGS__);                   \
    g_last_error = buf_;                                         \
    if (ctx) (ctx)->err = buf_;                                  \
    return (code);                                               \
  } while (0)

#define HIPCHK(ctx, call)                                                                          \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) NQ_FAIL(ctx, -5, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

#define NQ_SINGLE_RANK(c, what)                                                              \
  do {                                                                                       \
    if (!(c)) NQ_FAIL((nq_ctx*)nullptr, -1, what ": null context");                          \
    if ((c)->P != 1) NQ_FAIL(c, -4, what ": not available on a slab context (nranks > 1)");  \
  } while (0)

// ---------------------------------------------------------------------------------------------
struct EqState {           // ETDRK4 state of one equation
  cd* y[3] = {nullptr, nullptr, nullptr};   // rotating: y[cur] = y(t_n)
  int cur = 0;
  cd *fn0 = nullptr, *fna = nullptr;
  cd* coef[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // E, Eh, Q, f0, fab, fc (filter folded in)
};                                                                       // (rows 0..N/2 only when nq_ctx::cmirror is set)

struct nq_ctx {
  nq_params p;
  int N = 0, S1 = 0, S2 = 0, nk = 0;
  int CLy = CL;          // columns per workgroup of the y-side kernels: CL, or CLS with single-pass columns (S1 = N, S2 = 1)
  int small_qg = 0;      // QGModel on a small grid: columns per workgroup of the array-parallel spectral kernel k_c_qg (0 = off)
  int WhG = 0;           // global half-spectrum width N/2+1
  int Wh = 0, Ph = 0;    // valid local half-spectrum columns and their pitch (== WhG, N/2+8 when P == 1)
  bool own_stream = true;
  bool kernel_family = true;
  // c(l, k), the filter and the contour patches depend on l through l^2 only: rows l and N - l of every ETDRK4 coefficient plane
  // are bit-identical (unless the filter is not mirror-symmetric in l: the 2/3 mask).  cmirror = 1: the planes hold rows 0..N/2
  // and the spectral kernels index them through crow() -- half the coefficient memory, and, with the B-sub-pass workgroups of
  // mirrored residues launched back to back (pair_order), the second read of a line is served on chip.  NIWQG_AMD_COEF_MIRROR=0
  // keeps all N rows (A/B measurements).
  int cmirror = 0, crows = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t marks[16] = {};               // nq_event_record / nq_event_elapsed
  std::string err;
  long long bytes = 0;
  std::vector<void*> allocs;
  // tables
  cd* tw = nullptr;
  int num_cu = 256;
  cd *twx_half = nullptr;   // stage table of the N/2-point plan (k_x_products_eo / k_x_wavepv_eo: 8192-point rows)
  cd *twx = nullptr, *twx1 = nullptr;   // per-stage twiddle tables of the two row-kernel plans (WgFft::tw_off layout)
  double *kk = nullptr, *ll = nullptr, *filt_h = nullptr, *filt_f = nullptr;
  cd* contour = nullptr;
  // equations
  EqState q, w;
  // dual-copy q equation (dealias=True, or exact full-plane qh on request): second copy + unfolded filters
  bool dual = false;
  EqState q2;
  // the anti-Hermitian passenger on row l = N/2 of the reference's full-plane qh (k_s_q): one row per array, coefficient
  // pointers at row N/2 of the q planes; off in dual mode (the second copy carries it) and for QGModel
  bool pass = false;
  EqState qp;
  cd* coefu[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // q coefficients without the filter
  struct CoefPatch { int n = 0; int* l = nullptr; int* k = nullptr; cd* v = nullptr; } patch[3];   // nq_coeff_patch, per public eq
  double* filt_m = nullptr;                                                 // filter at (-l, -k)
  // half-spectrum aux spectra
  cd *qwh = nullptr, *ph = nullptr;
  // inside a slab step the A sub-passes read / write the rank's own block on the X side directly and the exchanges skip its
  // device copy (ArrayListR; NIWQG_AMD_SLAB_OWN_REDIRECT=0 keeps the copy): set by nq_slab_step while it issues steps
  bool redir_now = false;
  // nq_tick_snapshot: qh (both copies), phih, qwh as the last diagnostics tick saw them (allocated by the first call)
  cd *tick_qh = nullptr, *tick_q2 = nullptr, *tick_w = nullptr, *tick_qwh = nullptr;
  // slab decomposition; P == 1: one rank owns everything
  int P = 1, rank = 0;
  int Nloc = 0;          // local rows on the X side
  int row0 = 0, nrows = 0;   // row window of the next row-kernel launch (a chunk of the local rows; whole slab by default)
  int Wf = 0, kf0 = 0;   // full-width planes: local columns, first global column
  int Wl = 0, kh0 = 0;   // half-spectrum planes: columns per rank, first global column of this rank
  // exchange groups: G[0] X->Y {Muq,Mvq,Mw}, G[1] Y->X {Mphi,Mphiy}, G[2] X->Y {Ma,Mb},
  // G[3] Y->X {Mu,Mp,Mq,Mqw}; Gs = stale copy of G[1]'s X side (UnCoupled's frozen phix/phiy, quirk Q1)
  // G[4] (YBJModel on more than one rank): Gs with a y side of its own, the second G1-shaped group of do_step_ybj
  struct Group {
    cd *bx = nullptr, *by = nullptr;
    int pitch = 0;
    size_t elems = 0;
  } G[5];
  cd *Gs = nullptr, *Gs_y = nullptr;
  MArr mUq, mVq, mW, mPhi, mPhiy, mGx, mGy, mA, mB, mU, mP, mQ, mQw;
  // scratch for the generic transforms / downloads
  cd *scr_f0 = nullptr, *scr_f1 = nullptr, *scr_h0 = nullptr, *scr_h1 = nullptr;
  double* scr_r = nullptr;
  // in-step budget integrals (ref Kernel.py:319-322, :390-392)
  bool bud = false;
  int nww = 0, nwq = 0;                           // workgroups of the two kernels that emit partial sums
  double *partW = nullptr, *partQ = nullptr;      // [4 stages][workgroups][NQ_PARTW | 3]
  double *part0W = nullptr, *part0Q = nullptr;    // partials of set_phi / set_q / nq_invert
  double *diag_part = nullptr, *diag_out = nullptr;   // diagnostics tick: workgroup partials, 32 reduced sums
  double *carryW = nullptr, *carryQ = nullptr;    // spectral sums of the state at the start of the next step
  double *gradS1 = nullptr, *acc = nullptr;       // stale-aware sum wv2|phih_grad|^2 ; Ke,Pw,Kw increments
  double* bsums = nullptr;                        // [4 stages][11] reduced sums of one step
  int prof_class = -1;
  int prof_stride = 1;          // nq_profile_stride: every prof_stride-th launch of the enabled class is bracketed
  unsigned long long prof_seen = 0;
  std::vector<hipEvent_t> prof_ev;               // pairs
  std::vector<int> prof_cls;                     // kernel class of each pair
  size_t prof_used = 0;
  bool have_q = false, have_phi = false, stepped = false;
  // max |u|, max |v| of the FOURTH stage (what the reference's status line uses after a step without a tick: Kernel.py:594 with
  // the self.u, self.v of :364-368): recorded by the last step of a call when asked for (nq_request_stage4_max)
  bool want_uv4 = false, uv4_now = false, have_uv4 = false;
  double* uv4 = nullptr;
  // snapshots (ref niwqg/Saving.py:59-86): physical q, phi in buffers of their own, copied out on a second stream
  double* snap_q = nullptr;
  cd* snap_phi = nullptr;
  double* snap_hq = nullptr;       // pinned host
  cd* snap_hphi = nullptr;
  hipStream_t snap_stream = nullptr;
  hipEvent_t ev_snap = nullptr;
  bool snap_busy = false;
  // single-rank CoupledModel: the q update (memory-bound) runs on a second stream beside the wave-PV row kernel
  // (transform-engine bound), which then leaves `overlap_cus` CUs free for it (0 = off)
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int overlap_cus = 0;
  // ---- slab step inside the library: how the exchange groups cross between the ranks
  int link = 0;                        // LINK_*: 0 none, 1 peers in this process, 2 RCCL, 3 caller's callbacks, 4 nothing on the wire (nq_slab_set_null_link)
  std::vector<nq_ctx*> peers;          // LINK_PEERS: every rank's context (index = rank), the same list on all of them
  void* comm = nullptr;                // LINK_RCCL: ncclComm_t
  nq_exchange_fn xcb = nullptr;        // LINK_CALLBACK
  nq_allreduce_fn rcb = nullptr;
  void* cb_user = nullptr;
  int nchunk = 1;                      // row chunks per exchange (producer / consumer row kernels run chunk by chunk)
  int reserve_cus = 0;                 // CUs the persistent row kernels leave free (RCCL's copy kernels need somewhere to run:
                                       // a row-kernel workgroup owns the whole register file of its CU)
  hipStream_t mstream = nullptr;       // exchanges run here, beside the compute stream
  hipEvent_t ev_prod[8] = {}, ev_arr[5][8] = {}, ev_col = nullptr, ev_done = nullptr, ev_red = nullptr;
  bool arr_pending[5] = {false, false, false, false, false};   // group g is arriving chunk by chunk (ev_arr[g][*] recorded)
  long long n_exch = 0, n_calls = 0, n_steps = 0;       // counters since nq_slab_counters(reset)
  double bytes_sent = 0.0;
  std::vector<hipEvent_t> xev;         // timing pairs around every exchange chunk on mstream (when counting)
  size_t xev_used = 0;
  std::vector<hipEvent_t> rev;         // the same around every all-reduce
  size_t rev_used = 0;
  bool xtime = false;
  bool alloc_plain = false;            // dev_alloc: no skew (exchange-group buffers: whole rows, read by the row kernels)
  int spec_kind = -1;                  // what the scratch column slab of nq_slab_spectral holds: 1 half-spectrum, 0 full width
  bool ybj = false;
  bool passive = false;  // QGModel with its passive scalar: state cq, spectrum emitted through the qw slots of G3 / G0
  EqState cq;
  MArr mUc, mVc;      // niwqg.YBJModel: UnCoupled layouts, only phi is stepped (stage graph in do_step_ybj)
};

// Device arrays start at staggered offsets inside their allocations.  hipMalloc hands out large blocks at addresses that
// differ by multiples of 2 MiB (the 256 MiB planes: by exact multiples of their size), so element idx of every state,
// tendency and coefficient plane a spectral kernel touches in one go would sit in the same DRAM channel and bank; a few
// KB of skew per array spreads them (tools/rw_mix_bench.hip: +12-24 % for kernels with 4-8 streams).
static size_t alloc_stagger_bytes() {
  static long v = -1;
  if (v < 0) {
    const char* e = getenv("NIWQG_AMD_ALLOC_STAGGER");
    v = e ? atol(e) : 4352;
    if (v < 0 || v % 256) v = 0;
  }
  return (size_t)v;
}
template <typename Tp>
static int dev_alloc(nq_ctx* c, Tp** out, size_t count) {
  void* p = nullptr;
  const size_t skew = c->alloc_plain ? 0 : (c->allocs.size() % 16) * alloc_stagger_bytes();
  HIPCHK(c, hipMalloc(&p, count * sizeof(Tp) + skew));
  HIPCHK(c, hipMemsetAsync(p, 0, count * sizeof(Tp) + skew, c->stream));
  c->allocs.push_back(p);
  c->bytes += (long long)(count * sizeof(Tp));
  *out = reinterpret_cast<Tp*>(static_cast<char*>(p) + skew);
  return 0;
}
#define ALLOC(c, ptr, count)                    \
  do {                                          \
    int rc_ = dev_alloc((c), &(ptr), (count));  \
    if (rc_) return rc_;                        \
  } while (0)

// ---------------------------------------------------------------------------------------------
// ETDRK4 coefficient planes on the device (ref Kernel.py:417-454, QGModel.py:426-443).
// eq 0: q (Kernel family), 1: phi, 2: q (QGModel, with beta), 3: QGModel's passive scalar.  One thread per element.
__device__ __forceinline__ cd cexp_d(cd z) {
  double s, c;
  sincos(z.y, &s, &c);
  const double e = exp(z.x);
  return cmake(e * c, e * s);
}
__device__ __forceinline__ cd cdiv(cd a, cd b) {
  const double d = b.x * b.x + b.y * b.y;
  return cmake((a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d);
}

// the linear operator c(l, k) of equation eq (ref Kernel.py:417-418, :440-442, QGModel.py:426-428, :452-453)
__device__ __forceinline__ cd linear_operator(int eq, const nq_params& p, double kx, double ly) {
  const double wv2 = kx * kx + ly * ly, wv4 = wv2 * wv2;
  if (eq == 0) return cmake(-p.nu4 * wv4 - p.nu * wv2 - p.mu, -kx * p.U);
  if (eq == 1) return cmake(-p.nu4w * wv4 - p.nuw * wv2 - p.muw, -kx * p.U - 0.5 * p.f * (wv2 / p.kappa2));
  if (eq == 3) return cmake(-p.nu4c * wv4 - p.nuc * wv2 - p.muc, 0.0);      // QGModel's passive scalar: no mean flow, no beta
  const double wv2i = (wv2 != 0.0) ? 1.0 / wv2 : 0.0;
  return cmake(-p.nu4 * wv4 - p.nu * wv2 - p.mu, -kx * p.U + p.beta * kx * wv2i);
}

__global__ void k_etdrk4_coeffs(int eq, int N, int width, int pitch, int k0, nq_params p, const double* __restrict__ kk,
                                const double* __restrict__ ll, const double* __restrict__ filt,
                                const cd* __restrict__ contour, cd* E, cd* Eh, cd* Q, cd* f0, cd* fab, cd* fc) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int l = blockIdx.y;
  if (k >= width) return;
  const cd c = linear_operator(eq, p, kk[k0 + k], ll[l]);
  const cd ch = cscale(c, p.dt);
  cd sQ = cmake(0, 0), s0 = cmake(0, 0), sab = cmake(0, 0), sc = cmake(0, 0);
  for (int m = 0; m < 32; ++m) {
    const cd LR = cadd(ch, contour[m]);
    const cd LR2 = cmul(LR, LR), LR3 = cmul(LR2, LR);
    const cd eLR = cexp_d(LR), eh = cexp_d(cscale(LR, 0.5));
    sQ = cadd(sQ, cdiv(cmake(eh.x - 1.0, eh.y), LR));
    // (-4 - LR + e^LR (4 - 3 LR + LR^2)) / LR^3
    cd t = cmul(eLR, cmake(4.0 - 3.0 * LR.x + LR2.x, -3.0 * LR.y + LR2.y));
    s0 = cadd(s0, cdiv(cmake(-4.0 - LR.x + t.x, -LR.y + t.y), LR3));
    // (2 + LR + e^LR (-2 + LR)) / LR^3
    t = cmul(eLR, cmake(-2.0 + LR.x, LR.y));
    sab = cadd(sab, cdiv(cmake(2.0 + LR.x + t.x, LR.y + t.y), LR3));
    // (-4 - 3 LR - LR^2 + e^LR (4 - LR)) / LR^3
    t = cmul(eLR, cmake(4.0 - LR.x, -LR.y));
    sc = cadd(sc, cdiv(cmake(-4.0 - 3.0 * LR.x - LR2.x + t.x, -3.0 * LR.y - LR2.y + t.y), LR3));
  }
  const size_t idx = (size_t)l * pitch + k;
  const double fl = filt ? filt[idx] : 1.0;
  const double s = p.dt / 32.0 * fl;
  E[idx] = cscale(cexp_d(ch), fl);
  Eh[idx] = cscale(cexp_d(cscale(ch, 0.5)), fl);
  Q[idx] = cscale(sQ, s);
  f0[idx] = cscale(s0, s);
  fab[idx] = cscale(sab, s);
  fc[idx] = cscale(sc, s);
}

// Entries whose c dt lies within delta of MINUS a contour point: there one of the 32 terms of the contour mean is the
// removable singularity (e^z - 1 - z - ...) / z^3 at |z| < delta, evaluated by cancellation, and what the reference holds in Qh,
// f0, fab, fc at such an entry is its own libm's rounding error amplified by eps / |z|^3 -- a function of numpy's exp, not of
// the mathematics.  The host recomputes exactly these entries with the reference's numpy expression (niwqg_amd/_etdrk4.py) and
// hands them back through nq_coeff_patch; everywhere else the two evaluations agree to ~1e-13.
__global__ void k_coeff_flag(int eq, int N, int width, int k0, nq_params p, const double* __restrict__ kk,
                             const double* __restrict__ ll, const cd* __restrict__ contour, double delta2, int cap,
                             int* __restrict__ count, int* __restrict__ lo, int* __restrict__ ko) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y;
  if (k >= width) return;
  const cd ch = cscale(linear_operator(eq, p, kk[k0 + k], ll[l]), p.dt);
  double dmin = 1e300;
  for (int m = 0; m < 32; ++m) {
    const cd LR = cadd(ch, contour[m]);
    dmin = fmin(dmin, LR.x * LR.x + LR.y * LR.y);
  }
  if (!(dmin >= delta2)) {        // also catches NaN
    const int at = atomicAdd(count, 1);
    if (at < cap) { lo[at] = l; ko[at] = k0 + k; }
  }
}
// v: n x 4 values (Qh, f0, fab, fc of the reference, no filter); k is a global column
// mrows = N when the planes hold rows 0..N/2 only (row N - l shares the storage of row l: nq_ctx::cmirror), else 0
__global__ void k_coeff_patch(int n, const int* __restrict__ li, const int* __restrict__ ki, const cd* __restrict__ v, int width,
                              int pitch, int k0, const double* __restrict__ filt, cd* Q, cd* f0, cd* fab, cd* fc, int mrows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int k = ki[i] - k0;
  if (k < 0 || k >= width) return;
  const double fl = filt ? filt[(size_t)li[i] * pitch + k] : 1.0;
  const int lr = (mrows && li[i] > mrows / 2) ? mrows - li[i] : li[i];
  const size_t idx = (size_t)lr * pitch + k;
  Q[idx] = cscale(v[4 * i], fl);
  f0[idx] = cscale(v[4 * i + 1], fl);
  fab[idx] = cscale(v[4 * i + 2], fl);
  fc[idx] = cscale(v[4 * i + 3], fl);
}

// small helper kernels -------------------------------------------------------------------------
// out = mult(l,k) * in on a spectral plane; mode 0: copy, 1: -i*l, 2: i*k, 3: -wv2i (psi from q, no wave part)
__global__ void k_spec_mul(const cd* __restrict__ in, cd* __restrict__ out, int width, int pitch, int mode,
                           const double* __restrict__ kk, const double* __restrict__ ll, int N, int kernel_family) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y;
  if (k >= width) return;
  const size_t idx = (size_t)l * pitch + k;
  const cd v = in[idx];
  const double kx = kk[k], ly = ll[l];
  cd o = v;
  if (mode == 1) {
    const double lz = (kernel_family && l == N / 2 && width != N) ? 0.0 : ly;
    o = cmake(lz * v.y, -lz * v.x);
  } else if (mode == 2) {
    o = cmake(-kx * v.y, kx * v.x);
  } else if (mode == 3) {
    const double wv2 = kx * kx + ly * ly;
    const double wv2i = (wv2 != 0.0) ? 1.0 / wv2 : 0.0;
    o = cmake(-wv2i * v.x, -wv2i * v.y);
  }
  out[idx] = o;
}

// out = a on the self-mirrored columns, (a + b)/2 on the interior columns (dual-copy q-hat -> Hermitian part)
__global__ void k_avg_interior(const cd* __restrict__ a, const cd* __restrict__ b, cd* __restrict__ out, int width,
                               int pitch, int N) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y;
  if (k >= width) return;
  const size_t idx = (size_t)l * pitch + k;
  cd v = a[idx];
  if (k > 0 && k < N / 2) {
    const cd w = b[idx];
    v = cmake(0.5 * (v.x + w.x), 0.5 * (v.y + w.y));
  }
  out[idx] = v;
}

// a *= -i, contiguous array
__global__ void k_mul_minus_i(cd* __restrict__ a, size_t n) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) a[i] = cmul_mi(a[i]);
}

__global__ void k_axpy1(double* y, const double* x, double a) { y[0] += a * x[0]; }

// out = a - b on a half-spectrum plane (q_psi = q - qw, ref CoupledModel.py:145-152)
__global__ void k_sub_half(const cd* __restrict__ a, const cd* __restrict__ b, cd* __restrict__ out, int width, int pitch) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y;
  if (k >= width) return;
  const size_t idx = (size_t)l * pitch + k;
  out[idx] = csub(a[idx], b[idx]);
}

// Half spectra of REAL fields -> the reference's array layouts.  a(l,k) = f(l,k) for k <= N/2 and conj f(-l,-k) beyond
// (what numpy.fft.fft2 of the real field holds there); out has `wout` columns (N: Kernel family, N/2+1: QGModel).
//   mode 0: out = a1                               (ref CoupledModel.py:71: fft of a real field)
//   mode 1: out = i kk[k] a1 + i ll[l] a2          (ref Kernel.py:484 / QGModel.py:481: ik*fft(u q) + il*fft(v q))
__global__ void k_expand_half(const cd* __restrict__ f1, const cd* __restrict__ f2, cd* __restrict__ out, int N,
                              int pitch_h, int wout, int mode, const double* __restrict__ kk,
                              const double* __restrict__ ll) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y;
  if (k >= wout) return;
  cd a, b = cmake(0, 0);
  if (k <= N / 2) {
    const size_t at = (size_t)l * pitch_h + k;
    a = f1[at];
    if (mode == 1) b = f2[at];
  } else {
    const size_t at = (size_t)((N - l) % N) * pitch_h + (N - k);
    a = cconj(f1[at]);
    if (mode == 1) b = cconj(f2[at]);
  }
  if (mode == 1) {
    const double kx = kk[k], ly = ll[l];
    a = cmake(-(kx * a.y + ly * b.y), kx * a.x + ly * b.x);
  }
  out[(size_t)l * wout + k] = a;
}

// the same on a column slab whose first column is global column k0
__global__ void k_avg_interior_g(const cd* __restrict__ a, const cd* __restrict__ b, cd* __restrict__ out, int width,
                                 int pitch, int N, int k0) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y;
  if (k >= width) return;
  const size_t idx = (size_t)l * pitch + k;
  cd v = a[idx];
  if (k0 + k > 0 && k0 + k < N / 2) {
    const cd w = b[idx];
    v = cmake(0.5 * (v.x + w.x), 0.5 * (v.y + w.y));
  }
  out[idx] = v;
}

// reductions: sum over a real/complex plane of a pointwise expression; result in out[0..] via atomics
// kind 0: sum |a|^2 (complex plane, width N)       -> out[0]
// kind 1: sum weight_k * wv2 * |a|^2 (half spectrum psi -> 2*ke_qg*M^2), skipping [0,0]
// kind 2: sum wv2 * |a|^2 (full complex plane)
// kind 3: max |a| over complex plane (out[0] as max via atomicMax on bits, values >= 0)
__global__ void k_reduce(const cd* __restrict__ a, int width, int pitch, int N, int kind, const double* __restrict__ kk,
                         const double* __restrict__ ll, double* out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y;
  double v = 0.0;
  if (k < width) {
    cd z = a[(size_t)l * pitch + k];
    if (kind == 4 && (k == 0 || k == N / 2)) {     // Kernel family: ph = fft of the REAL p, i.e. the Hermitian part (in l) of the
      const cd zm = a[(size_t)((N - l) % N) * pitch + k];   // two self-mirrored columns (they differ under the 2/3 mask's dual copies)
      z = cmake(0.5 * (z.x + zm.x), 0.5 * (z.y - zm.y));
    }
    const double m2 = z.x * z.x + z.y * z.y;
    if (kind == 0) v = m2;
    else if (kind == 1 || kind == 4) {
      const double w = (k == 0 || k == N / 2) ? 1.0 : 2.0;
      v = (l == 0 && k == 0) ? 0.0 : w * (kk[k] * kk[k] + ll[l] * ll[l]) * m2;
    } else if (kind == 2) v = (kk[k] * kk[k] + ll[l] * ll[l]) * m2;
    else v = sqrt(m2);
  }
  __shared__ double sh[256];
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] = (kind == 3) ? fmax(sh[threadIdx.x], sh[threadIdx.x + s]) : sh[threadIdx.x] + sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (kind == 3) atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(sh[0]));
    else atomicAdd(out, sh[0]);
  }
}
__global__ void k_reduce_real_max(const double* __restrict__ a, size_t n, double* out) {
  double v = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) v = fmax(v, fabs(a[i]));
  __shared__ double sh[256];
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(sh[0]));
}

// diagnostics tick: spectral sums -----------------------------------------------------------------
// Deterministic: every block writes its partial sums (part[block][NQ]); k_reduce_partials adds them.
template <int NQ>
__device__ void diag_block_store(double (&v)[NQ], double* __restrict__ part) {
  __shared__ double sh[4][NQ];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    double x = v[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if (lane == 0) sh[wave][i] = x;
  }
  __syncthreads();
  if (threadIdx.x < NQ) part[(size_t)blockIdx.x * NQ + threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}
// phih (local columns [k0, k0+width) of the full plane): S_n = sum wv2^n |phih|^2, n = 0..3   (256 threads)
__global__ void k_diag_phi(const cd* __restrict__ phih, int N, int width, int pitch, int k0,
                           const double* __restrict__ kk, const double* __restrict__ ll, double* __restrict__ part) {
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  const size_t total = (size_t)N * width;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int l = (int)(i / width), k = (int)(i - (size_t)l * width);
    const cd z = phih[(size_t)l * pitch + k];
    const double kx = kk[k0 + k], ly = ll[l];
    const double wv2 = kx * kx + ly * ly, m2 = z.x * z.x + z.y * z.y;
    v[0] += m2;
    v[1] += wv2 * m2;
    v[2] += wv2 * wv2 * m2;
    v[3] += wv2 * wv2 * wv2 * m2;
  }
  diag_block_store<4>(v, part);
}
// half spectra qh, qwh (may be null), ph; nine sums (see nq_diagnostics).  On the two self-mirrored columns the sums
// that stand for means of REAL fields use the Hermitian part H(l) = (X(l) + conj X(-l))/2 (what `.real` keeps).
// qp, qm (dual-copy contexts, else null): the two copies X+ = qh(l,k), X- = conj qh(-l,-k) that `qh` is the mean of.  The two
// spec_var-type sums (chi_q, ke_qg_q: ref Kernel.py:644, CoupledModel.py:115-136 take |.|^2 over the reference's FULL plane,
// which is not Hermitian under the 2/3 mask) then see |X+|^2 + |X-|^2 on the interior columns, not 2 |mean|^2.  filt_p, filt_m
// (idem): the filter at (l,k) and at (-l,-k).  The stored qwh is the Hermitian part fs * qw of the reference's filtr * qw, fs =
// (filt_p + filt_m)/2; ke_qg_w = spec_var over the full plane (CoupledModel.py:108) needs (filt_p^2 + filt_m^2)/2 |qw|^2 instead.
__global__ void k_diag_q(const cd* __restrict__ qh, const cd* __restrict__ qwh, const cd* __restrict__ ph, int N,
                         int width, int pitch, int k0, const double* __restrict__ kk, const double* __restrict__ ll,
                         double* __restrict__ part, const cd* __restrict__ qp, const cd* __restrict__ qm_,
                         const double* __restrict__ filt_p, const double* __restrict__ filt_m, const cd* __restrict__ passenger) {
  double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const size_t total = (size_t)N * width;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int l = (int)(i / width), kl = (int)(i - (size_t)l * width), k = k0 + kl;      // kl: local column, k: global
    const size_t idx = (size_t)l * pitch + kl;
    const cd q = qh[idx], w = qwh ? qwh[idx] : cmake(0, 0), p = ph[idx];
    cd hq = q, hw = w, hp = p;
    double wt = 2.0;
    if (k == 0 || k == N / 2) {
      wt = 1.0;
      const size_t im = (size_t)((N - l) % N) * pitch + kl;
      const cd qm = qh[im], wm = qwh ? qwh[im] : cmake(0, 0), pm = ph[im];
      hq = cmake(0.5 * (q.x + qm.x), 0.5 * (q.y - qm.y));
      hw = cmake(0.5 * (w.x + wm.x), 0.5 * (w.y - wm.y));
      hp = cmake(0.5 * (p.x + pm.x), 0.5 * (p.y - pm.y));
    }
    const double kx = kk[k], ly = ll[l];
    const double wv2 = kx * kx + ly * ly, wv4 = wv2 * wv2, wv2i = (wv2 != 0.0) ? 1.0 / wv2 : 0.0;
    double q2 = q.x * q.x + q.y * q.y;
    if (qp != nullptr && wt == 2.0) {
      const cd a = qp[idx], b = qm_[idx];
      q2 = 0.5 * (a.x * a.x + a.y * a.y + b.x * b.x + b.y * b.y);
    }
    if (passenger != nullptr && l == N / 2 && wt == 2.0) {       // the reference's full-plane qh carries an anti-Hermitian part A on
      const cd a = passenger[kl];                                  // this row: |qh(l,k)|^2 + |qh(l,-k)|^2 = 2 |H|^2 + 2 |A|^2 (spec_var sees it)
      q2 += a.x * a.x + a.y * a.y;
    }
    v[0] += wt * (hq.x * hq.x + hq.y * hq.y);                     // sum |H q|^2            -> ens
    v[1] += wt * wv4 * q2;                                        // sum wv4 |q|^2          -> chi_q
    v[2] += wt * wv2i * q2;                                       // sum |q|^2 / wv2        -> ke_qg_q
    double w2 = w.x * w.x + w.y * w.y;
    if (filt_p != nullptr) {
      const double fp = filt_p[idx], fm = filt_m[idx], fs = 0.5 * (fp + fm);
      if (fs > 0.0) {                      // as ratios: fs * fs underflows in the far corner of the exponential filter (exact_qh)
        const double a = fp / fs, b = fm / fs;
        w2 *= (wt == 2.0) ? 0.5 * (a * a + b * b) : a * a;
      }
    }
    v[3] += wt * wv2i * w2;                                       // sum |qw|^2 / wv2       -> ke_qg_w
    // -> ke_qg_qw = mean(uq uw + vq vw) of PHYSICAL fields (CoupledModel.py:110-112): u = Re ifft(-il psi) has nothing from the
    // Nyquist row, v = Re ifft(ik psi) nothing from the Nyquist column (Hermitian psi: the two partners cancel), and qwh has no
    // anti-Hermitian part for q-hat's to pair with.  Without a filter those two lines carry energy (2 of 460 random draws, 8e-6).
    const double w2eff = ((l == N / 2) ? 0.0 : ly * ly) + ((k == N / 2) ? 0.0 : kx * kx);
    v[4] += wt * wv2i * wv2i * w2eff * (hq.x * hw.x + hq.y * hw.y);
    // -> ke_qg.  Dual-copy (2/3 mask) contexts: the reference's ph = fft of the REAL p is the Hermitian part, and there the two
    // self-mirrored columns of the stored psi-hat are not Hermitian in l.  (QGModel's ph = -wv2i qh is summed as it is.)
    const cd pk = (qp != nullptr) ? hp : p;
    v[5] += (l == 0 && k == 0) ? 0.0 : wt * wv2 * (pk.x * pk.x + pk.y * pk.y);
    const double pq = hp.x * hq.x + hp.y * hq.y;                  // Re(conj(psi) q)
    v[6] += wt * wv4 * pq;                                        // -> mean(q lap2 psi)
    v[7] += wt * wv2 * pq;                                        // -> -mean(psi lap q)
    v[8] += wt * pq;                                              // -> mean(psi q)
  }
  diag_block_store<9>(v, part);
}

// passive scalar of QGModel: S_n = sum w wv2^n |c-hat|^2, n = 0..3, over the half spectrum (w = 1 on the self-mirrored columns,
// 2 elsewhere; n = 0 without the [0,0] entry, like spec_var): C2, gradC2, mean(lap c ^2), -mean(lap^2 c lap c)
__global__ void k_diag_c(const cd* __restrict__ ch, int N, int width, int pitch, int k0, const double* __restrict__ kk,
                         const double* __restrict__ ll, double* __restrict__ part) {
  // n = 0, 1 are spec_var sums of c-hat as it is (C2, gradC2); n = 2, 3 stand for physical-space means of lap c, which only
  // sees the Hermitian part (in l) of the two self-mirrored columns -- what irfft2 keeps
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  const size_t total = (size_t)N * width;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int l = (int)(i / width), kl = (int)(i - (size_t)l * width), k = k0 + kl;
    const cd z = ch[(size_t)l * pitch + kl];
    cd h = z;
    double wt = 2.0;
    if (k == 0 || k == N / 2) {
      wt = 1.0;
      const cd zm = ch[(size_t)((N - l) % N) * pitch + kl];
      h = cmake(0.5 * (z.x + zm.x), 0.5 * (z.y - zm.y));
    }
    const double kx = kk[k], ly = ll[l], wv2 = kx * kx + ly * ly;
    const double m2 = wt * (z.x * z.x + z.y * z.y), m2h = wt * (h.x * h.x + h.y * h.y);
    v[0] += (l == 0 && k == 0) ? 0.0 : m2;
    v[1] += wv2 * m2;
    v[2] += wv2 * wv2 * m2h;
    v[3] += wv2 * wv2 * wv2 * m2h;
  }
  diag_block_store<4>(v, part);
}

// budget bookkeeping ----------------------------------------------------------------------------
struct BudgetAcc {
  int model, nww, nwq, nqs;     // nqs: doubles per workgroup in partQ (6 with QGModel's passive scalar)
  double nu4c, muc;
  const double *partW, *partQ;
  double *carryW, *carryQ, *gradS1, *acc;
  double dt, f, hslash, kappa2, nu, nu4, mu, nuw, nu4w, muw, M;
};

__device__ double block_total(const double* __restrict__ part, int n, int stride, double* sh) {
  double x = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) x += part[(size_t)i * stride];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
  __syncthreads();
  double tot = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += sh[w];
  return tot;
}

// sums workgroup partials of one emitter into dst[0..nq)
__global__ void k_reduce_partials(const double* __restrict__ part, int nwg, int stride, int nq, double* dst) {
  __shared__ double sh[16];
  for (int q = 0; q < nq; ++q) {
    const double t = block_total(part + q, nwg, stride, sh);
    if (threadIdx.x == 0) dst[q] = t;
  }
}

// Stage sums: one workgroup per (stage, quantity); quantity 0-2: partQ, 3-8: partW (S0..S3, SG, SX), 9-10 unused.
// sums[stage][11]
__global__ void k_budget_sums(BudgetAcc b, double* __restrict__ sums) {
  __shared__ double sh[16];
  const int s = blockIdx.x / 11, q = blockIdx.x % 11;
  double t = 0.0;
  if (q < 3) t = block_total(b.partQ + (size_t)s * b.nwq * b.nqs + q, b.nwq, b.nqs, sh);
  else if (b.model == NQ_MODEL_QG) {
    if (b.nqs == 6 && q < 6) t = block_total(b.partQ + (size_t)s * b.nwq * 6 + q, b.nwq, 6, sh);   // |c|^2 sums
  } else {
    if (q < 3 + NQ_PARTW) t = block_total(b.partW + (size_t)s * b.nww * NQ_PARTW + (q - 3), b.nww, NQ_PARTW, sh);
  }
  if (threadIdx.x == 0) sums[s * 11 + q] = t;
}

// One ETDRK4 step's worth of budget rates -> Ke, Pw, Kw increments (ref Kernel.py:319-322, :390-392;
// QGModel.py:355-407).  Slot s of the spectral sums = state at the start of stage s.
__device__ void budget_accumulate_body(const BudgetAcc& b, const double* __restrict__ sums) {
  double sw[5][4], sj[4][2], sq[5][3];
  const bool qg = b.model == NQ_MODEL_QG;
  for (int s = 0; s < 4; ++s) {
    for (int q = 0; q < 3; ++q) sq[qg ? s : s + 1][q] = sums[s * 11 + q];
    for (int q = 0; q < 4; ++q) sw[s + 1][q] = sums[s * 11 + 3 + q];
    for (int q = 0; q < 2; ++q) sj[s][q] = sums[s * 11 + 7 + q];
  }
  const double M2 = b.M * b.M;
  if (!qg) {
    for (int q = 0; q < 3; ++q) sq[0][q] = b.carryQ[q];
    for (int q = 0; q < 4; ++q) sw[0][q] = b.carryW[q];
  }
  double K = 0.0, Pw = 0.0, A = 0.0;
  const double wgt[4] = {1.0, 2.0, 2.0, 1.0};
  for (int s = 0; s < 4; ++s) {
    const double ep_psi = (b.nu4 * sq[s][0] + b.nu * sq[s][1] + b.mu * sq[s][2]) / M2;
    double k = ep_psi, p = 0.0, a = 0.0;
    if (!qg) {
      const double lap2 = sw[s][2] / M2, glap2 = sw[s][3] / M2, phi2 = sw[s][0] / M2;
      const double grad2 = ((b.model == NQ_MODEL_COUPLED) ? sw[s][1] : b.gradS1[0]) / M2;
      // sj[s] = {sum Re(conj(lap_h) W), sum Im(conj(diss_h) W)} with W the whole phi tendency (k_s_phi)
      const double g12 = -0.5 * b.hslash * (sj[s][0] / M2) / b.f;      // gamma1 + gamma2
      const double x12 = -(sj[s][1] / M2) / b.f;                        // xi1 + xi2
      const double chi = (-0.5 * b.nu4w * glap2 - 0.5 * b.nuw * lap2 - 0.5 * b.muw * grad2) / b.kappa2;
      a = -b.nu4w * lap2 - b.nuw * grad2 - b.muw * phi2;
      k = -g12 + x12 + ep_psi;
      p = g12 + chi;
    }
    K += wgt[s] * k;
    Pw += wgt[s] * p;
    A += wgt[s] * a;
  }
  if (qg && b.nqs == 6) {
    // cvar += dt (c1 + 2 c2 + 2 c3 + c4) / 6 with c_s = ep_c after the stage-s update (ref QGModel.py:350-394, :595-598;
    // the reference multiplies gradC2 by nu, not nuc)
    for (int s = 0; s < 4; ++s) {
      const double* sc = sums + s * 11 + 3;
      Pw += wgt[s] * (-2.0 * b.nu4c * sc[2] - 2.0 * b.nu * sc[1] - 2.0 * b.muc * sc[0]) / M2;
    }
  }
  b.acc[0] += b.dt * K / 6.0;
  b.acc[1] += b.dt * Pw / 6.0;
  b.acc[2] += b.dt * A / 6.0;
  if (!qg) {
    for (int q = 0; q < 3; ++q) b.carryQ[q] = sq[4][q];
    for (int q = 0; q < 4; ++q) b.carryW[q] = sw[4][q];
  }
}
__global__ void k_budget_accumulate(BudgetAcc b, const double* __restrict__ sums) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  budget_accumulate_body(b, sums);
}
// Both in ONE launch of one workgroup, for grids whose kernels emit few partial sums (<= 256 workgroups; at 1024 partials one
// workgroup reading them all is slower than the 44 of k_budget_sums): on the small grids a step is a chain of dependent
// 5-15 us kernels and each of the two budget launches costs 4.6 us of it.
__global__ void __launch_bounds__(1024) k_budget_small(BudgetAcc b, double* __restrict__ sums) {
  __shared__ double ss[44];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  for (int i = wave; i < 44; i += nwaves) {          // every wave sums its own quantities: no workgroup barrier inside
    const int s = i / 11, q = i % 11;
    const double* part = nullptr;
    int n = 0, stride = 0;
    if (q < 3) {
      part = b.partQ + (size_t)s * b.nwq * b.nqs + q; n = b.nwq; stride = b.nqs;
    } else if (b.model == NQ_MODEL_QG) {
      if (b.nqs == 6 && q < 6) { part = b.partQ + (size_t)s * b.nwq * 6 + q; n = b.nwq; stride = 6; }
    } else if (q < 3 + NQ_PARTW) {
      part = b.partW + (size_t)s * b.nww * NQ_PARTW + (q - 3); n = b.nww; stride = NQ_PARTW;
    }
    double x = 0.0;
    for (int k = lane; k < n; k += 64) x += part[(size_t)k * stride];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if (lane == 0) {
      ss[i] = x;
      sums[i] = x;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) budget_accumulate_body(b, ss);
}

// ---------------------------------------------------------------------------------------------
// size dispatch
#define M_SMALL(M) M(64, 8, 8) M(128, 8, 16) M(256, 16, 16) M(512, 16, 32) M(1024, 32, 32) M(2048, 32, 64)
// sizes whose fused row kernels run one transform per row (8192-point rows: the even/odd kernels, k_x_products_eo / k_x_wavepv_eo)
#define NQ_FOR_ROW_SIZES(M) M_SMALL(M) M(4096, 64, 64)
#define NQ_FOR_SIZES(M) M(64, 8, 8) M(128, 8, 16) M(256, 16, 16) M(512, 16, 32) M(1024, 32, 32) M(2048, 32, 64) M(4096, 64, 64) M(8192, 64, 128)

static bool plan_for(int N, int* S1, int* S2) {
#define CASE_(n, a, b) if (N == n) { *S1 = a; *S2 = b; return true; }
  NQ_FOR_SIZES(CASE_)
#undef CASE_
  return false;
}

// per-kernel event profiling ---------------------------------------------------------------------
enum { PK_PRODUCTS = 0, PK_WAVEPV = 1, PK_SQ = 2, PK_SPHI = 3, PK_INVERT = 4, PK_A = 5 };
struct ProfScope {
  nq_ctx* c;
  bool on;
  ProfScope(nq_ctx* c_, int cls) : c(c_), on(c_->prof_class == cls || c_->prof_class == -2) {   // -2: every class
    if (!on) return;
    if (c->prof_stride > 1 && (c->prof_seen++ % c->prof_stride) != 0) { on = false; return; }    // nq_profile_stride: a sample
    if (c->prof_used + 2 > c->prof_ev.size()) {
      for (int i = 0; i < 2; ++i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) { on = false; return; }
        c->prof_ev.push_back(e);
      }
    }
    if (c->prof_cls.size() < c->prof_ev.size() / 2) c->prof_cls.resize(c->prof_ev.size() / 2);
    c->prof_cls[c->prof_used / 2] = cls;
    (void)hipEventRecord(c->prof_ev[c->prof_used], c->stream);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(c->prof_ev[c->prof_used + 1], c->stream);
    c->prof_used += 2;
  }
};

// generic launches -----------------------------------------------------------------------------
template <int N>
static void launch_x_c2c_n(nq_ctx* c, bool inv, const cd* in, cd* out, int pin, int pout, int nrows, double scale, int mul_ik) {
  typedef XPlan<N> X;
  dim3 grid((nrows + X::C - 1) / X::C), block(X::THREADS);
  if (inv) hipLaunchKernelGGL((k_x_c2c<N, true>), grid, block, X::LDS_BYTES, c->stream, in, out, pin, pout, nrows, scale, c->tw, c->kk, mul_ik);
  else hipLaunchKernelGGL((k_x_c2c<N, false>), grid, block, X::LDS_BYTES, c->stream, in, out, pin, pout, nrows, scale, c->tw, c->kk, mul_ik);
}
static void launch_x_c2c(nq_ctx* c, bool inv, const cd* in, cd* out, int pin, int pout, double scale, int mul_ik = 0) {
  switch (c->N) {
#define CASE_(n, a, b) case n: launch_x_c2c_n<n>(c, inv, in, out, pin, pout, c->N, scale, mul_ik); break;
    NQ_FOR_SIZES(CASE_)
#undef CASE_
  }
}
static void launch_x_r2c(nq_ctx* c, const double* in, cd* out) {
  switch (c->N) {
#define CASE_(n, a, b) case n: { typedef XPlan<n> X; hipLaunchKernelGGL((k_x_r2c<n>), dim3((c->N + X::C - 1) / X::C), dim3(X::THREADS), X::LDS_BYTES, c->stream, in, out, c->N, c->Ph, c->N, c->tw); } break;
    NQ_FOR_SIZES(CASE_)
#undef CASE_
  }
}
static void launch_x_c2r(nq_ctx* c, const cd* in, double* out, double scale) {
  switch (c->N) {
#define CASE_(n, a, b) case n: { typedef XPlan<n> X; hipLaunchKernelGGL((k_x_c2r<n>), dim3((c->N + X::C - 1) / X::C), dim3(X::THREADS), X::LDS_BYTES, c->stream, in, out, c->Ph, c->N, c->N, scale, c->tw); } break;
    NQ_FOR_SIZES(CASE_)
#undef CASE_
  }
}

template <int S, typename AL>
static void launch_A_s(nq_ctx* c, bool inv, const AL& al, int n, int maxw) {
  typedef YPlan<S> Y;
  dim3 grid((maxw + CL - 1) / CL, c->S1, n), block(Y::THREADS);
  if (inv) hipLaunchKernelGGL((k_y_A<S, true, AL>), grid, block, Y::LDS_BYTES, c->stream, al, c->S1, c->tw, 1);
  else hipLaunchKernelGGL((k_y_A<S, false, AL>), grid, block, Y::LDS_BYTES, c->stream, al, c->S1, c->tw, 1);
}
template <typename AL>
static void launch_A_list(nq_ctx* c, bool inv, const AL& al, int n, int maxw) {
  if (c->S2 == 1) return;                  // single-pass columns: the B sub-pass is the whole y transform
  ProfScope ps(c, PK_A);
  switch (c->S2) {
    case 8: launch_A_s<8>(c, inv, al, n, maxw); break;
    case 16: launch_A_s<16>(c, inv, al, n, maxw); break;
    case 32: launch_A_s<32>(c, inv, al, n, maxw); break;
    case 64: launch_A_s<64>(c, inv, al, n, maxw); break;
    case 128: launch_A_s<128>(c, inv, al, n, maxw); break;
  }
}
// A sub-pass on plain arrays (generic path, P == 1): half-spectrum or full-plane geometry
static void launch_A(nq_ctx* c, bool inv, std::initializer_list<cd*> arrs, bool half) {
  ArrayList al;
  int n = 0, maxw = 0;
  for (cd* a : arrs) {
    al.ptr[n] = a;
    al.width[n] = half ? c->Wh : c->N;
    al.pitch[n] = half ? c->Ph : c->N;
    maxw = al.width[n] > maxw ? al.width[n] : maxw;
    ++n;
  }
  for (int i = n; i < 6; ++i) { al.ptr[i] = nullptr; al.width[i] = 0; al.pitch[i] = 0; }
  launch_A_list(c, inv, al, n, maxw);
}
// local valid width of a mixed-space array on the Y side
static int y_width(const nq_ctx* c, const MArr& m) {
  if (m.W == c->Wf) return c->Wf;
  const int left = c->WhG - c->kh0;
  return left < 0 ? 0 : (left < c->Wl ? left : c->Wl);
}
// A sub-pass on the Y side of exchange-group arrays
static void launch_A_m(nq_ctx* c, bool inv, std::initializer_list<const MArr*> arrs) {
  ArrayListR al;
  int n = 0, maxw = 0;
  for (const MArr* m : arrs) {
    al.ptr[n] = m->ys;
    al.alt[n] = m->xs;
    al.width[n] = y_width(c, *m);
    al.pitch[n] = m->pitch;
    maxw = al.width[n] > maxw ? al.width[n] : maxw;
    ++n;
  }
  for (int i = n; i < 6; ++i) { al.ptr[i] = al.alt[i] = nullptr; al.width[i] = 0; al.pitch[i] = 0; }
  al.own0 = c->rank * c->Nloc;
  al.own1 = al.own0 + c->Nloc;
  if (maxw <= 0) return;
  if (c->redir_now) launch_A_list(c, inv, al, n, maxw);                               // (see ArrayListR)
  else launch_A_list(c, inv, static_cast<const ArrayList&>(al), n, maxw);
}
template <int S, int CLX = CL>
static void launch_B_s(nq_ctx* c, bool inv, const cd* in, int pin, cd* out, int pout, int width, double scale) {
  typedef YPlanT<S, CLX> Y;
  dim3 grid((width + CLX - 1) / CLX, c->S2), block(Y::THREADS);
  if (inv) hipLaunchKernelGGL((k_y_B<S, true, CLX>), grid, block, Y::LDS_BYTES, c->stream, in, out, width, pin, pout, c->S2, scale, c->tw, 1);
  else hipLaunchKernelGGL((k_y_B<S, false, CLX>), grid, block, Y::LDS_BYTES, c->stream, in, out, width, pin, pout, c->S2, scale, c->tw, 1);
}
// dispatch on the y plan: S1 = one radix of the two-pass transform (tiles of CL columns) or, with single-pass columns, N
// itself (tiles of CLS columns for N >= 128)
#define NQ_S1_SWITCH(c, CALL)                 \
  if ((c)->CLy == CL) {                       \
    switch ((c)->S1) {                        \
      case 8: CALL(8, CL); break;             \
      case 16: CALL(16, CL); break;           \
      case 32: CALL(32, CL); break;           \
      case 64: CALL(64, CL); break;           \
    }                                         \
  } else {                                    \
    switch ((c)->S1) {                        \
      case 128: CALL(128, CLS); break;        \
      case 256: CALL(256, CLS); break;        \
      case 512: CALL(512, CLS); break;        \
    }                                         \
  }
static void launch_B_p(nq_ctx* c, bool inv, const cd* in, int pin, cd* out, int pout, int width, double scale) {
#define CALL_(s, clx) launch_B_s<s, clx>(c, inv, in, pin, out, pout, width, scale)
  NQ_S1_SWITCH(c, CALL_)
#undef CALL_
}
static void launch_B(nq_ctx* c, bool inv, const cd* in, cd* out, bool half, double scale) {
  const int width = half ? c->Wh : c->N, pitch = half ? c->Ph : c->N;
  launch_B_p(c, inv, in, pitch, out, pitch, width, scale);
}

// whole 2-D transforms on device arrays (generic path, P == 1) -------------------------------------
// complex physical (N,N) -> spectral (N,N), unnormalised; `tmp` is a full-plane scratch
static void fwd2d_full(nq_ctx* c, const cd* phys, cd* spec, cd* tmp) {
  launch_x_c2c(c, false, phys, tmp, c->N, c->N, 1.0);
  launch_A(c, false, {tmp}, false);
  launch_B(c, false, tmp, spec, false, 1.0);
}
static void inv2d_full(nq_ctx* c, const cd* spec, cd* phys, cd* tmp) {
  launch_B(c, true, spec, tmp, false, 1.0 / ((double)c->N * c->N));
  launch_A(c, true, {tmp}, false);
  launch_x_c2c(c, true, tmp, phys, c->N, c->N, 1.0);
}
// real physical -> half spectrum
static void fwd2d_half(nq_ctx* c, const double* phys, cd* spec, cd* tmp_h) {
  launch_x_r2c(c, phys, tmp_h);
  launch_A(c, false, {tmp_h}, true);
  launch_B(c, false, tmp_h, spec, true, 1.0);
}
static void inv2d_half(nq_ctx* c, const cd* spec, double* phys, cd* tmp_h) {
  launch_B(c, true, spec, tmp_h, true, 1.0 / ((double)c->N * c->N));
  launch_A(c, true, {tmp_h}, true);
  launch_x_c2r(c, tmp_h, phys, 1.0);
}

// column-slab geometry of the spectral kernels
static YGeom geom_half(const nq_ctx* c) {
  YGeom g;
  g.k0 = c->kh0;
  const int left = c->WhG - c->kh0;
  g.width = left < 0 ? 0 : (left < c->Wl ? left : c->Wl);
  g.pitch_s = c->Ph;
  g.S2 = c->S2;
  g.kernel_family = c->kernel_family ? 1 : 0;
  g.cmirror = c->cmirror;
  return g;
}
static YGeom geom_full(const nq_ctx* c) {
  YGeom g;
  g.k0 = c->kf0;
  g.width = c->Wf;
  g.pitch_s = c->Wf;
  g.S2 = c->S2;
  g.kernel_family = 1;
  g.cmirror = c->cmirror;
  return g;
}

// fused-stage launches ----------------------------------------------------------------------------
// X-side view of a mixed-space array that starts at local row c->row0: a row kernel launched on such views with
// c->nrows rows processes one CHUNK of the slab (the slab driver overlaps the exchange of chunk i with the kernel of
// chunk i+1); everything a row kernel addresses is relative to the row, so the kernels themselves do not change
static MArr rw(const nq_ctx* c, const MArr& m) {
  MArr r = m;
  if (r.xs) r.xs += (size_t)c->row0 * m.pitch;
  return r;
}
template <bool SLAB>
static void launch_wavepv_t(nq_ctx* c) {
  const MArr mPhi = rw(c, c->mPhi), mPhiy = rw(c, c->mPhiy), mA = rw(c, c->mA), mB = rw(c, c->mB);
  switch (c->N) {
#define CASE_(n, a, b) case n: { typedef XPlan1<n> X; hipLaunchKernelGGL((k_x_wavepv<n, SLAB>), dim3(c->nrows / X::C), dim3(X::THREADS), X::LDS_BYTES, c->stream, mPhi, mPhiy, mA, mB, c->twx1, c->kk); } break;
    case 8192:
      {                                               // even / odd samples as two 4096-point problems (no spills)
        typedef XPlan<4096> X;
        const int ncu = c->num_cu - c->reserve_cus, nb = c->nrows, grid = nb < ncu ? nb : ncu;
        const size_t ldsb = X::LDS_BYTES + (size_t)4096 * sizeof(cd) ;          // + the 64 KB of thread-private park slots
        hipLaunchKernelGGL((k_x_wavepv_eo<8192, SLAB>), dim3(grid), dim3(X::THREADS), ldsb, c->stream, mPhi, mPhiy, mA, mB, c->twx_half, c->tw, c->kk, nb);
      }
      break;
    case 4096: {                                      // long rows: two transforms in flight, no spills
      typedef XPlan<4096> X;
      const size_t ldsb = X::LDS_BYTES + X::F::LDS_ELEMS * sizeof(cd);
      int grid = c->num_cu - (c->stream2 ? c->overlap_cus : 0) - c->reserve_cus;   // one persistent workgroup per CU
      const int nb = c->nrows / X::C;
      if (grid > nb) grid = nb;
      hipLaunchKernelGGL((k_x_wavepv2<4096, SLAB>), dim3(grid), dim3(X::THREADS), ldsb, c->stream, mPhi, mPhiy, mA, mB, c->twx, c->kk, nb);
    } break;
    M_SMALL(CASE_)
#undef CASE_
  }
}
// one rank: the row kernels address their rows without the block arithmetic of the slab layout (XRowT<false>)
static void launch_wavepv(nq_ctx* c) {
  ProfScope ps(c, PK_WAVEPV);
  if (c->P > 1) launch_wavepv_t<true>(c);
  else launch_wavepv_t<false>(c);
}
template <int MODE, bool SLAB>
static void launch_products_t(nq_ctx* c, double cj, double cr, bool fresh_grad) {
  const int vz = c->kernel_family ? 1 : 0;
  const MArr mU = rw(c, c->mU), mP = rw(c, c->mP), mQ = rw(c, c->mQ), mQw = rw(c, c->mQw), mPhi = rw(c, c->mPhi);
  const MArr mUq = rw(c, c->mUq), mVq = rw(c, c->mVq), mW = rw(c, c->mW), mPhiy = rw(c, c->mPhiy);
  const MArr mGx = rw(c, c->mGx), mGy = rw(c, c->mGy), mUc = rw(c, c->mUc), mVc = rw(c, c->mVc);
  if (c->N == 8192 && c->twx_half) {
    // rows too long for the register budget as one transform: even / odd samples as two 4096-point problems
    typedef XPlan<4096> X;
    const MArr& gx8 = (MODE == MODE_QGC) ? mUc : ((MODE == MODE_UNCOUPLED && !fresh_grad) ? mGx : mPhi);
    const MArr& gy8 = (MODE == MODE_QGC) ? mVc : ((MODE == MODE_UNCOUPLED && !fresh_grad) ? mGy : mPhiy);
    const int ncu = c->num_cu - c->reserve_cus, nb = c->nrows, grid = nb < ncu ? nb : ncu;
    const size_t ldsb = X::LDS_BYTES + (size_t)4096 * sizeof(cd) ;            // + the 64 KB of thread-private park slots
    hipLaunchKernelGGL((k_x_products_eo<8192, MODE, SLAB>), dim3(grid), dim3(X::THREADS), ldsb, c->stream, mU, mP, mQ, mQw, mPhi, gx8, gy8, mUq, mVq, mW, c->twx_half, c->tw, c->kk, vz, cj, cr, nb);
    return;
  }
  const MArr& gx = (MODE == MODE_QGC) ? mUc : ((MODE == MODE_UNCOUPLED && !fresh_grad) ? mGx : mPhi);
  const MArr& gy = (MODE == MODE_QGC) ? mVc : ((MODE == MODE_UNCOUPLED && !fresh_grad) ? mGy : mPhiy);
  switch (c->N) {
#define CASE_(n, a, b) case n: { typedef XPlan<n> X; const int nb = c->nrows / X::C; \
    /* one workgroup fits per CU (LDS) and does not spill: persistent; 8192-point rows spill and do better with dynamic dispatch */ \
    const int ncu = c->num_cu - c->reserve_cus; \
    const int grid = (X::LDS_BYTES > 80 * 1024 && X::THREADS <= 512 && nb > ncu) ? ncu : nb; \
    hipLaunchKernelGGL((k_x_products<n, MODE, SLAB>), dim3(grid), dim3(X::THREADS), X::LDS_BYTES, c->stream, mU, mP, mQ, mQw, mPhi, gx, gy, mUq, mVq, mW, c->twx, c->kk, vz, cj, cr, nb); } break;
    NQ_FOR_ROW_SIZES(CASE_)
#undef CASE_
  }
}
template <int MODE>
static void launch_products_m(nq_ctx* c, double cj, double cr, bool fresh_grad) {
  if (c->P > 1) launch_products_t<MODE, true>(c, cj, cr, fresh_grad);
  else launch_products_t<MODE, false>(c, cj, cr, fresh_grad);
}
// Mw <- cj * (u phix + v phiy) + i cr * phi q_psi; a step uses the phi tendency itself: cj = -1, cr = -1/2
// fresh_grad (UnCoupled layouts only): phix, phiy from the rows of Mphi, Mphiy instead of the frozen copy
static void launch_products(nq_ctx* c, double cj = -1.0, double cr = -0.5, bool fresh_grad = false) {
  ProfScope ps(c, PK_PRODUCTS);
  if (c->p.model == NQ_MODEL_COUPLED) launch_products_m<MODE_COUPLED>(c, cj, cr, false);
  else if (c->p.model == NQ_MODEL_UNCOUPLED) launch_products_m<MODE_UNCOUPLED>(c, cj, cr, fresh_grad);
  else if (c->passive) launch_products_m<MODE_QGC>(c, cj, cr, false);
  else launch_products_m<MODE_QG>(c, cj, cr, false);
}

static EtdArrays etd_arrays(EqState& e, int stage, int* out_slot) {
  // slots: cur = y(t_n); a = (cur+1)%3 holds the stage-0 result; b = (cur+2)%3 scratch
  const int cur = e.cur, a = (cur + 1) % 3, b = (cur + 2) % 3;
  EtdArrays ea;
  ea.y_in = (stage == 2) ? e.y[a] : e.y[cur];
  const int out = (stage == 0) ? a : (stage == 3 ? cur : b);
  ea.y_out = e.y[out];
  ea.fn0 = e.fn0;
  ea.fna = e.fna;
  ea.E = e.coef[0];
  ea.Eh = e.coef[1];
  ea.Q = e.coef[2];
  ea.f0 = e.coef[3];
  ea.fab = e.coef[4];
  ea.fc = e.coef[5];
  *out_slot = out;
  return ea;
}

template <int S, int CLX>
static void launch_sq_s(nq_ctx* c, const EtdArrays& ea, int stage, const MArr& huq, const MArr& hvq, bool q_equation) {
  typedef YPlanT<S, CLX> Y;
  DualQ dq;
  EtdArrays eap = ea;
  memset(&dq, 0, sizeof(dq));
  if (c->dual) {
    int slot = 0;
    dq.minus = etd_arrays(c->q2, stage, &slot);
    dq.filt_p = c->filt_h;
    dq.filt_m = c->filt_m;
    eap.E = dq.minus.E = c->coefu[0];
    eap.Eh = dq.minus.Eh = c->coefu[1];
    eap.Q = dq.minus.Q = c->coefu[2];
    eap.f0 = dq.minus.f0 = c->coefu[3];
    eap.fab = dq.minus.fab = c->coefu[4];
    eap.fc = dq.minus.fc = c->coefu[5];
  }
  const YGeom g = geom_half(c);
  if (g.width <= 0) return;
  const dim3 grid((g.width + CLX - 1) / CLX, c->S2), block(Y::THREADS);
  EtdArrays ep;
  memset(&ep, 0, sizeof(ep));
  if (c->pass && q_equation) {             // the q equation itself, not the passive scalar's
    int slot = 0;
    ep = etd_arrays(c->qp, stage, &slot);
  }
  if (c->dual) hipLaunchKernelGGL((k_s_q<S, true, CLX>), grid, block, Y::LDS_BYTES, c->stream, huq, hvq, eap, stage, g, c->kk, c->ll, c->tw, 1, dq, ep);
  else hipLaunchKernelGGL((k_s_q<S, false, CLX>), grid, block, Y::LDS_BYTES, c->stream, huq, hvq, eap, stage, g, c->kk, c->ll, c->tw, 1, dq, ep);
}
static BudgetW budget_w(nq_ctx* c, double* part, const cd* y_start) {
  BudgetW bw;
  bw.part = c->bud ? part : nullptr;
  bw.y_start = y_start;
  bw.nu4w = c->p.nu4w;
  bw.nuw = c->p.nuw;
  bw.muw = c->p.muw;
  return bw;
}
template <int S, int CLX>
static void launch_sphi_s(nq_ctx* c, const EtdArrays& ea, int stage, const cd* y_start, const MArr& ophi, const MArr& ophiy) {
  typedef YPlanT<S, CLX> Y;
  BudgetW bw = budget_w(c, c->partW + (size_t)stage * c->nww * NQ_PARTW, y_start);
  const cd* jpass = c->ybj ? nullptr : c->mUq.ys + c->mUq.W;     // YBJModel.jacobian_psi_phi keeps [0,0] (YBJModel.py:123-133)
  const cd* jown = (jpass && c->redir_now) ? c->mUq.xs + c->mUq.W : jpass;     // the own block was not copied across (ArrayListR)
  const int own0 = c->redir_now ? c->rank * c->Nloc : 0, own1 = c->redir_now ? own0 + c->Nloc : 0;
  hipLaunchKernelGGL((k_s_phi<S, CLX>), dim3(c->Wf / CLX, c->S2), dim3(Y::THREADS), Y::LDS_BYTES, c->stream, c->mW, jpass, jown, own0, own1, c->mUq.pitch, ea, stage, geom_full(c), ophi, ophiy, 1.0 / ((double)c->N * c->N), c->kk, c->ll, c->tw, 1, bw);
}
template <int S, int CLX>
static void launch_emit_phi_s(nq_ctx* c, const cd* phih) {
  typedef YPlanT<S, CLX> Y;
  BudgetW bw = budget_w(c, c->part0W, phih);
  hipLaunchKernelGGL((k_s_emit_phi<S, CLX>), dim3(c->Wf / CLX, c->S2), dim3(Y::THREADS), Y::LDS_BYTES, c->stream, phih, geom_full(c), c->mPhi, c->mPhiy, 1.0 / ((double)c->N * c->N), c->kk, c->ll, c->tw, 1, bw);
}
template <int S, int MODE, int CLX>
static void launch_invert_sm(nq_ctx* c, const cd* qh, bool store_aux, double* part, const cd* q_bud, const cd* c_hat = nullptr) {
  typedef YPlanT<S, CLX> Y;
  // the second copy lives in the same rotating slot as qh
  const cd* qh_minus = nullptr;
  if (c->dual)
    for (int i = 0; i < 3; ++i)
      if (c->q.y[i] == qh) qh_minus = c->q2.y[i];
  const YGeom g = geom_half(c);
  if (g.width <= 0) return;
  hipLaunchKernelGGL((k_s_invert<S, MODE, CLX>), dim3((g.width + CLX - 1) / CLX, c->S2), dim3(Y::THREADS), Y::LDS_BYTES, c->stream, c->mA, c->mB, qh, c->filt_h, c->mU, c->mP, c->mQ, c->mQw, store_aux ? c->qwh : nullptr, store_aux ? c->ph : nullptr, g, 1.0 / ((double)c->N * c->N), c->p.f, c->kk, c->ll, c->tw, 1, c->bud ? part : nullptr, q_bud, qh_minus, c->dual ? c->filt_m : nullptr, c_hat);
}

static void launch_sq(nq_ctx* c, const EtdArrays& ea, int stage, const MArr* huq = nullptr, const MArr* hvq = nullptr) {
  ProfScope ps(c, PK_SQ);
  const MArr& a1 = huq ? *huq : c->mUq;
  const MArr& a2 = hvq ? *hvq : c->mVq;
#define CALL_(s, clx) launch_sq_s<s, clx>(c, ea, stage, a1, a2, huq == nullptr)
  NQ_S1_SWITCH(c, CALL_)
#undef CALL_
}
static void launch_sphi(nq_ctx* c, const EtdArrays& ea, int stage, const cd* y_start, const MArr* ophi = nullptr,
                        const MArr* ophiy = nullptr) {
  ProfScope ps(c, PK_SPHI);
  const MArr& o1 = ophi ? *ophi : c->mPhi;
  const MArr& o2 = ophiy ? *ophiy : c->mPhiy;
#define CALL_(s, clx) launch_sphi_s<s, clx>(c, ea, stage, y_start, o1, o2)
  NQ_S1_SWITCH(c, CALL_)
#undef CALL_
}
static void launch_emit_phi(nq_ctx* c, const cd* phih) {
#define CALL_(s, clx) launch_emit_phi_s<s, clx>(c, phih)
  NQ_S1_SWITCH(c, CALL_)
#undef CALL_
}
static void launch_invert(nq_ctx* c, const cd* qh, bool store_aux, double* part, const cd* q_bud, const cd* c_hat = nullptr) {
  ProfScope ps(c, PK_INVERT);
  if (c->passive) {
#define CALL_(s, clx) launch_invert_sm<s, MODE_QGC, clx>(c, qh, store_aux, part, q_bud, c_hat)
    NQ_S1_SWITCH(c, CALL_)
#undef CALL_
  } else if (c->p.model == NQ_MODEL_COUPLED) {
#define CALL_(s, clx) launch_invert_sm<s, MODE_COUPLED, clx>(c, qh, store_aux, part, q_bud)
    NQ_S1_SWITCH(c, CALL_)
#undef CALL_
  } else {
#define CALL_(s, clx) launch_invert_sm<s, MODE_UNCOUPLED, clx>(c, qh, store_aux, part, q_bud)
    NQ_S1_SWITCH(c, CALL_)
#undef CALL_
  }
}

template <int N_, int CW_>
static void launch_cqg_t(nq_ctx* c, const EtdArrays& ea, int stage, bool store_aux, double* part, const cd* q_bud) {
  typedef SmallColPlan<N_, CW_> Y;
  hipLaunchKernelGGL((k_c_qg<N_, CW_>), dim3((c->Wh + CW_ - 1) / CW_), dim3(Y::THREADS), Y::LDS_BYTES, c->stream, c->mUq, c->mVq, ea, stage,
                     geom_half(c), c->mU, c->mP, c->mQ, store_aux ? c->ph : nullptr, 1.0 / ((double)c->N * c->N), c->kk, c->ll, c->tw,
                     c->bud ? part : nullptr, q_bud);
}
// QGModel, small grid: N_q, the stage update, the inversion and its three inverse y transforms in one launch
static void launch_cqg(nq_ctx* c, const EtdArrays& ea, int stage, bool store_aux, double* part, const cd* q_bud) {
  ProfScope ps(c, PK_SQ);
  switch (c->N) {
    case 128: launch_cqg_t<128, 4>(c, ea, stage, store_aux, part, q_bud); break;
    case 256: launch_cqg_t<256, 2>(c, ea, stage, store_aux, part, q_bud); break;
    case 512: launch_cqg_t<512, 2>(c, ea, stage, store_aux, part, q_bud); break;
  }
}

static BudgetAcc budget_acc(nq_ctx* c) {
  BudgetAcc b;
  b.model = c->p.model;
  b.nww = c->nww; b.nwq = c->nwq; b.nqs = c->passive ? 6 : 3;
  b.nu4c = c->p.nu4c; b.muc = c->p.muc;
  b.partW = c->partW; b.partQ = c->partQ;
  b.carryW = c->carryW; b.carryQ = c->carryQ; b.gradS1 = c->gradS1; b.acc = c->acc;
  b.dt = c->p.dt; b.f = c->p.f; b.kappa2 = c->p.kappa2; b.hslash = c->p.f / c->p.kappa2;
  b.nu = c->p.nu; b.nu4 = c->p.nu4; b.mu = c->p.mu; b.nuw = c->p.nuw; b.nu4w = c->p.nu4w; b.muw = c->p.muw;
  b.M = (double)c->N * c->N;
  return b;
}

// a new qh from physical space (set_q) is Hermitian: the passenger row starts from zero
static int reset_passenger(nq_ctx* c) {
  if (c->pass) HIPCHK(c, hipMemsetAsync(c->qp.y[c->qp.cur], 0, sizeof(cd) * (size_t)c->Ph, c->stream));
  return 0;
}

// ---- phases of one ETDRK4 stage --------------------------------------------------------------------
// Between two phases the named exchange group has to cross from one side to the other: with P == 1 both
// sides are the same buffer and nq_step simply runs the phases back to back; with P > 1 the caller does
//      nq_phase(PRODUCTS) -> all_to_all(G0) -> nq_phase(UPDATE) -> all_to_all(G1) [-> all_to_all(G3)]
//      -> nq_phase(WAVEPV) -> all_to_all(G2) -> nq_phase(INVERT) -> all_to_all(G3)         (Coupled)
// (UnCoupled / QG have no WAVEPV/INVERT: UPDATE also emits G3.)
static const cd* stage_qh_out(nq_ctx* c, int stage) {
  const int cur = c->q.cur;
  return c->q.y[(stage == 0) ? (cur + 1) % 3 : (stage == 3 ? cur : (cur + 2) % 3)];
}
static void phase_invert_y(nq_ctx* c, const cd* qh, bool store_aux, double* part, const cd* q_bud, const cd* c_hat = nullptr) {
  if (c->p.model == NQ_MODEL_COUPLED) launch_A_m(c, false, {&c->mA, &c->mB});
  launch_invert(c, qh, store_aux, part, q_bud, c_hat);
  if (c->p.model == NQ_MODEL_COUPLED || c->passive) launch_A_m(c, true, {&c->mU, &c->mP, &c->mQ, &c->mQw});
  else launch_A_m(c, true, {&c->mU, &c->mP, &c->mQ});
}
static void phase_products(nq_ctx* c, int stage) { (void)stage; launch_products(c); }
static void phase_update(nq_ctx* c, int s) {
  const bool waves = c->p.model != NQ_MODEL_QG;
  int qslot = 0, wslot = 0;
  if (waves) launch_A_m(c, false, {&c->mW});
  if (c->passive) launch_A_m(c, false, {&c->mUq, &c->mVq, &c->mUc, &c->mVc});
  else launch_A_m(c, false, {&c->mUq, &c->mVq});
  EtdArrays eq = etd_arrays(c->q, s, &qslot);
  if (c->small_qg) {          // QGModel.py:355,:401: ep_psi of stages 0..2 with the start-of-step q
    launch_cqg(c, eq, s, s == 3, c->partQ + (size_t)s * c->nwq * 3, s < 3 ? c->q.y[c->q.cur] : nullptr);
    return;
  }
  launch_sq(c, eq, s);
  int cslot = 0;
  if (c->passive) {           // same update with the scalar's operator and its own products (ref QGModel.py:345-392)
    EtdArrays ec = etd_arrays(c->cq, s, &cslot);
    launch_sq(c, ec, s, &c->mUc, &c->mVc);
  }
  if (waves) {
    // phih at the start of this stage: y(t_n), stage-0 result, stage-1 result, stage-2 result
    const int cur = c->w.cur;
    const cd* y_start = (s == 0) ? c->w.y[cur] : (s == 1 ? c->w.y[(cur + 1) % 3] : c->w.y[(cur + 2) % 3]);
    EtdArrays ew = etd_arrays(c->w, s, &wslot);
    launch_sphi(c, ew, s, y_start);
    launch_A_m(c, true, {&c->mPhi, &c->mPhiy});
  }
  if (c->p.model != NQ_MODEL_COUPLED) {
    // no wave feedback on psi: the inversion needs no row pass, it runs here on the spectral side
    // QGModel evaluates ep_psi after each stage's inversion with the start-of-step q (QGModel.py:355,:401)
    const cd* q_bud = (!c->kernel_family && s < 3) ? c->q.y[c->q.cur] : nullptr;
    const int nqs = c->passive ? 6 : 3;
    phase_invert_y(c, c->q.y[qslot], s == 3, c->partQ + (size_t)s * c->nwq * nqs, q_bud, c->passive ? c->cq.y[cslot] : nullptr);
  }
}
static void phase_wavepv(nq_ctx* c) { launch_wavepv(c); }
static void phase_invert(nq_ctx* c, int s) {      // Coupled only
  phase_invert_y(c, stage_qh_out(c, s), s == 3, c->partQ + (size_t)s * c->nwq * 3, nullptr);
}
static void phase_budget_sums(nq_ctx* c) {
  if (c->bud) hipLaunchKernelGGL(k_budget_sums, dim3(44), dim3(1024), 0, c->stream, budget_acc(c), c->bsums);
}
static void phase_budget_finish(nq_ctx* c) {
  if (c->bud) hipLaunchKernelGGL(k_budget_accumulate, dim3(1), dim3(64), 0, c->stream, budget_acc(c), (const double*)c->bsums);
}

// inversion of the CURRENT state outside a step (set_q, nq_invert); P == 1 only.  Its spectral sums become
// the next step's slot 0.
static void do_invert_now(nq_ctx* c) {
  if (c->p.model == NQ_MODEL_COUPLED) launch_wavepv(c);
  phase_invert_y(c, c->q.y[c->q.cur], true, c->part0Q, nullptr, c->passive ? c->cq.y[c->cq.cur] : nullptr);
  if (c->bud && c->kernel_family)
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, c->stream, c->part0Q, c->nwq, 3, 3, c->carryQ);
}

// max |u|, max |v| over this rank's rows of the fields the LAST inversion emitted (Mu, Mp on the x side): during a step, right
// after stage index 2, these are the u, v of the reference's fourth jacobian_psi_q call (Kernel.py:364-368, :481-482)
static int stage4_uv_max(nq_ctx* c) {
  if (!c->uv4) ALLOC(c, c->uv4, (size_t)2);
  if (!c->scr_f0) ALLOC(c, c->scr_f0, (size_t)c->Nloc * c->N);
  HIPCHK(c, hipMemsetAsync(c->uv4, 0, sizeof(double) * 2, c->stream));
  double* scr = reinterpret_cast<double*>(c->scr_f0);
  const size_t n = (size_t)c->Nloc * c->N;
  for (int which = 0; which < 2; ++which) {
    const MArr& src = which == 0 ? c->mU : c->mP;
    switch (c->N) {
#define CASE_(nn, a, b) case nn: { typedef XPlan<nn> X; \
      hipLaunchKernelGGL((k_x_get_real<nn, true>), dim3((c->Nloc + X::C - 1) / X::C), dim3(X::THREADS), X::LDS_BYTES, c->stream, src, scr, c->Nloc, c->tw, c->kk, which, (which == 1 && c->kernel_family) ? 1 : 0); } break;
      NQ_FOR_SIZES(CASE_)
#undef CASE_
    }
    hipLaunchKernelGGL(k_reduce_real_max, dim3(1024), dim3(256), 0, c->stream, (const double*)scr, n, c->uv4 + which);
  }
  c->have_uv4 = true;
  return 0;
}

// niwqg.YBJModel._step_etdrk4 (YBJModel.py:52-87): psi, u, v, q are steady; phix, phiy are refreshed from the current
// phih before every stage, but the refraction factor phi only after the step.  Two G1-shaped buffers do it without a
// copy: A = G[1] holds phi, phiy of the start-of-step state (products read phi from it in all four stages, and the
// gradients in stage 0); B = Gs receives the stage results 0..2 and feeds the gradients of stages 1..3; the final
// state goes back to A, so B is left with the gradients of the stage-2 result -- exactly the stale phix, phiy the
// reference leaves behind (they matter to the next diagnostics tick only).
static void do_step_ybj(nq_ctx* c) {
  for (int s = 0; s < 4; ++s) {
    launch_products(c, -1.0, -0.5, s == 0);
    launch_A_m(c, false, {&c->mW});
    int wslot = 0;
    const int cur = c->w.cur;
    const cd* y_start = (s == 0) ? c->w.y[cur] : (s == 1 ? c->w.y[(cur + 1) % 3] : c->w.y[(cur + 2) % 3]);
    EtdArrays ew = etd_arrays(c->w, s, &wslot);
    if (s < 3) {
      launch_sphi(c, ew, s, y_start, &c->mGx, &c->mGy);
      launch_A_m(c, true, {&c->mGx, &c->mGy});
    } else {
      launch_sphi(c, ew, s, y_start);
      launch_A_m(c, true, {&c->mPhi, &c->mPhiy});
    }
  }
}

// CoupledModel, one rank: same kernels as do_step, but the q update waits until the wave-PV row kernel starts and
// runs beside it on a second stream.  k_x_wavepv2 is bound by its row transforms and uses a persistent grid of
// (CUs - overlap_cus) workgroups; the spectral q kernels need 34 KB of LDS, which only the CUs it left free can give.
static void do_step_overlap(nq_ctx* c) {
  hipStream_t main_stream = c->stream;
  for (int s = 0; s < 4; ++s) {
    phase_products(c, s);
    int qslot = 0, wslot = 0;
    launch_A_m(c, false, {&c->mW});
    const int cur = c->w.cur;
    const cd* y_start = (s == 0) ? c->w.y[cur] : (s == 1 ? c->w.y[(cur + 1) % 3] : c->w.y[(cur + 2) % 3]);
    EtdArrays ew = etd_arrays(c->w, s, &wslot);
    launch_sphi(c, ew, s, y_start);
    launch_A_m(c, true, {&c->mPhi, &c->mPhiy});
    (void)hipEventRecord(c->ev_fork, main_stream);
    c->stream = c->stream2;                                   // every launch helper uses c->stream
    (void)hipStreamWaitEvent(c->stream2, c->ev_fork, 0);
    launch_A_m(c, false, {&c->mUq, &c->mVq});
    EtdArrays eq = etd_arrays(c->q, s, &qslot);
    launch_sq(c, eq, s);
    (void)hipEventRecord(c->ev_join, c->stream2);
    c->stream = main_stream;
    phase_wavepv(c);
    (void)hipStreamWaitEvent(main_stream, c->ev_join, 0);
    phase_invert(c, s);
    if (s == 2 && c->uv4_now) (void)stage4_uv_max(c);
  }
  phase_budget_sums(c);
  phase_budget_finish(c);
}

static void do_step(nq_ctx* c) {      // P == 1
  if (c->ybj) return do_step_ybj(c);
  if (c->stream2 && c->p.model == NQ_MODEL_COUPLED && c->N == 4096 && !c->dual) return do_step_overlap(c);
  for (int s = 0; s < 4; ++s) {
    phase_products(c, s);
    phase_update(c, s);
    if (c->p.model == NQ_MODEL_COUPLED) {
      phase_wavepv(c);
      phase_invert(c, s);
    }
    if (s == 2 && c->uv4_now) (void)stage4_uv_max(c);
  }
  if (c->bud && c->nww <= 256 && c->nwq <= 256) {            // few partials (grids <= 512): sums and accumulation in one launch
    hipLaunchKernelGGL(k_budget_small, dim3(1), dim3(1024), 0, c->stream, budget_acc(c), c->bsums);
    return;
  }
  phase_budget_sums(c);
  phase_budget_finish(c);
}


// ==================================================================================================================
// The slab step inside the library (include/niwqg_amd.h: nq_slab_step)
// ==================================================================================================================
enum { LINK_NONE = 0, LINK_PEERS = 1, LINK_RCCL = 2, LINK_CALLBACK = 3, LINK_NULL = 4 };

// Iterating the contexts of a group makes each context's device current before its loop body runs: with peers on DIFFERENT
// devices of one process (nq_slab_attach_peers, round 4) every launch, event record and copy has to be issued with the owner's
// device current.  On one device it is a thread-local no-op.
struct EachCtx {
  std::vector<nq_ctx*>& g;
  struct It {
    nq_ctx** p;
    nq_ctx* operator*() const {
      (void)hipSetDevice((*p)->device);
      return *p;
    }
    It& operator++() { ++p; return *this; }
    bool operator!=(const It& o) const { return p != o.p; }
  };
  It begin() const { return It{g.data()}; }
  It end() const { return It{g.data() + g.size()}; }
};
static inline EachCtx each(std::vector<nq_ctx*>& g) { return EachCtx{g}; }


// RCCL, taken from the process at run time (torch ships its own librccl and has usually loaded it already)
struct NcclId { char internal[128]; };
struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
static RcclApi g_rccl;
static const int kNcclDouble = 8, kNcclSum = 0;
static bool rccl_load(std::string* err) {
  if (g_rccl.handle) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  // NIWQG_AMD_RCCL_LIB: an explicit library instead (tests/mock_rccl: rank THREADS on one GPU)
  const char* forced = getenv("NIWQG_AMD_RCCL_LIB");
  if (forced && *forced && !(h = dlopen(forced, RTLD_NOW | RTLD_LOCAL))) {
    const char* e = dlerror();                  // ONE call: dlerror() clears its state, a second call returns NULL
    *err = std::string("NIWQG_AMD_RCCL_LIB: ") + (e ? e : "?");
    return false;
  }
  for (const char* n : names)
    if (!h && (h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;          // the copy the process already uses, if any
  for (const char* n : names) {
    if (h) break;
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  }
  if (!h) {
    const char* e = dlerror();
    *err = std::string("librccl not found: ") + (e ? e : "?");
    return false;
  }
#define SYM_(field, name)                                                  \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); \
  if (!g_rccl.field) {                                                     \
    *err = std::string("librccl lacks ") + name;                           \
    return false;                                                          \
  }
  SYM_(GetUniqueId, "ncclGetUniqueId") SYM_(CommInitRank, "ncclCommInitRank") SYM_(CommDestroy, "ncclCommDestroy")
  SYM_(GroupStart, "ncclGroupStart") SYM_(GroupEnd, "ncclGroupEnd") SYM_(Send, "ncclSend") SYM_(Recv, "ncclRecv")
  SYM_(AllReduce, "ncclAllReduce") SYM_(GetErrorString, "ncclGetErrorString")
#undef SYM_
  g_rccl.handle = h;
  return true;
}
#define NCCLCHK(ctx, call)                                                                                       \
  do {                                                                                                           \
    int r_ = (call);                                                                                             \
    if (r_ != 0) NQ_FAIL(ctx, -6, "%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
  } while (0)

static int slab_link_setup(nq_ctx* c) {        // exchange stream and events, once
  if (c->mstream) return 0;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamCreateWithFlags(&c->mstream, hipStreamNonBlocking));
  for (int i = 0; i < 8; ++i) HIPCHK(c, hipEventCreateWithFlags(&c->ev_prod[i], hipEventDisableTiming));
  for (int g = 0; g < 5; ++g)
    for (int i = 0; i < 8; ++i) HIPCHK(c, hipEventCreateWithFlags(&c->ev_arr[g][i], hipEventDisableTiming));
  HIPCHK(c, hipEventCreateWithFlags(&c->ev_col, hipEventDisableTiming));
  HIPCHK(c, hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
  HIPCHK(c, hipEventCreateWithFlags(&c->ev_red, hipEventDisableTiming));
  return 0;
}
// the contexts one call drives: every rank of the process in peers mode, otherwise this rank alone
static int slab_group(nq_ctx* c, std::vector<nq_ctx*>* g) {
  if (c->link == LINK_NONE) {
    if (c->P != 1) NQ_FAIL(c, -4, "slab context without a link: call nq_comm_init, nq_slab_attach_peers or nq_slab_set_callbacks first");
    c->peers.assign(1, c);                       // one rank: its own blocks cross by device copies
    c->link = LINK_PEERS;
  }
  if (c->link == LINK_PEERS) {
    if (c->rank != 0) NQ_FAIL(c, -4, "peers mode: drive the group through its rank-0 context");
    *g = c->peers;
  } else {
    g->assign(1, c);
  }
  for (nq_ctx* x : *g) {
    int rc = slab_link_setup(x);
    if (rc) return rc;
  }
  return 0;
}
static int effective_chunks(const nq_ctx* c) {
  if (c->link == LINK_CALLBACK) return 1;
  int rows_per_wg = 1;
  switch (c->N) {
#define CASE_(n, a, b) case n: rows_per_wg = XPlan<n>::C > XPlan1<n>::C ? XPlan<n>::C : XPlan1<n>::C; break;
    NQ_FOR_SIZES(CASE_)
#undef CASE_
  }
  int n = c->nchunk < 1 ? 1 : (c->nchunk > 8 ? 8 : c->nchunk);
  while (n > 1 && (c->Nloc % (n * rows_per_wg) != 0)) n >>= 1;
  return n;
}
struct XTimer {                                  // optional HIP-event pair around one exchange chunk (or one all-reduce) on the exchange stream
  nq_ctx* c;
  bool on;
  std::vector<hipEvent_t>& ev;
  size_t& used;
  hipStream_t xs;
  explicit XTimer(nq_ctx* c_, bool reduce = false, hipStream_t xs_ = nullptr)
      : c(c_), on(c_->xtime), ev(reduce ? c_->rev : c_->xev), used(reduce ? c_->rev_used : c_->xev_used), xs(xs_ ? xs_ : c_->mstream) {
    if (!on) return;
    if (used + 2 > ev.size()) {
      // the pool only shrinks in nq_slab_counters(reset): a caller that leaves timing on for a long run() gets the first
      // 16384 exchanges / all-reduces timed and the rest untimed instead of an event pool that grows without bound
      if (ev.size() >= 2 * 16384) { on = false; return; }
      for (int i = 0; i < 2; ++i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) { on = false; return; }
        ev.push_back(e);
      }
    }
    (void)hipEventRecord(ev[used], xs);
  }
  ~XTimer() {
    if (!on) return;
    (void)hipEventRecord(ev[used + 1], xs);
    used += 2;
  }
};

// Chunk i of nch of exchange group g, for every rank of grp.  to_y: x side -> y side (the producers' row kernels recorded
// ev_prod[i]); else y side -> x side (the column kernels recorded ev_col).  Both buffers of a group are cut into P blocks of
// Nloc rows; chunk i is the same row range inside every block.
// An exchange that nothing can run under -- one chunk, and not group 1, whose transfer the q update hides -- gains nothing from the
// second stream and pays two event hand-offs between the streams (~25-35 us per exchange on one GPU, 0.3-0.4 ms of a 2.2 ms rank
// step at 4096^2 / P = 8: profiles/r04_rank_of_P_measurements.txt): it is issued on the compute stream itself.
// NIWQG_AMD_SLAB_INLINE=0 keeps every exchange on the exchange stream.
static bool inline_exchange(const nq_ctx* c, int g, int nch) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("NIWQG_AMD_SLAB_INLINE");
    on = (e && atoi(e) == 0) ? 0 : 1;
  }
  return on && nch == 1 && g != 1 && c->link != LINK_CALLBACK;
}
static int issue_chunk(std::vector<nq_ctx*>& grp, int g, bool to_y, int i, int nch) {
  nq_ctx* c0 = grp[0];
  if (c0->G[g].elems == 0) return 0;
  const size_t blk = (size_t)c0->Nloc * c0->G[g].pitch, cblk = blk / nch, off = (size_t)i * cblk;
  const bool inl = inline_exchange(c0, g, nch);
  for (nq_ctx* c : each(grp)) {
    cd* send = to_y ? c->G[g].bx : c->G[g].by;
    cd* recv = to_y ? c->G[g].by : c->G[g].bx;
    hipEvent_t mine = to_y ? c->ev_prod[i] : c->ev_col;
    if (c->link == LINK_CALLBACK) {
      if (i == 0) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (!c->xcb) NQ_FAIL(c, -4, "no exchange callback");
        const int rc = c->xcb(c->cb_user, g, to_y ? 1 : 0);
        if (rc) NQ_FAIL(c, -6, "exchange callback failed (%d)", rc);
        c->n_exch += 1;
        c->bytes_sent += (double)blk * 16.0 * (c->P - 1);
      }
      continue;
    }
    const hipStream_t xs = inl ? c->stream : c->mstream;       // the stream this exchange runs on
    if (!inl) HIPCHK(c, hipStreamWaitEvent(xs, mine, 0));
    {
      XTimer xt(c, false, xs);
      if (c->link == LINK_PEERS) {
        for (nq_ctx* s : grp)
          if (!(inl && s == c)) HIPCHK(c, hipStreamWaitEvent(xs, to_y ? s->ev_prod[i] : s->ev_col, 0));
        for (nq_ctx* s : grp) {
          if (s == c && c->redir_now) continue;  // the own block: the A sub-passes use it where it is (ArrayListR)
          const cd* src = (to_y ? s->G[g].bx : s->G[g].by) + (size_t)c->rank * blk + off;
          HIPCHK(c, hipMemcpyAsync(recv + (size_t)s->rank * blk + off, src, cblk * sizeof(cd), hipMemcpyDeviceToDevice, xs));
        }
      } else if (c->link == LINK_NULL) {         // one rank of P measured alone: only its own block crosses (device copy)
        if (!c->redir_now)
          HIPCHK(c, hipMemcpyAsync(recv + (size_t)c->rank * blk + off, send + (size_t)c->rank * blk + off, cblk * sizeof(cd), hipMemcpyDeviceToDevice, xs));
      } else {                                   // RCCL: one grouped send/recv pair per peer, the own block by a device copy
        NCCLCHK(c, g_rccl.GroupStart());
        for (int p = 0; p < c->P; ++p) {
          if (p == c->rank) continue;
          NCCLCHK(c, g_rccl.Send(send + (size_t)p * blk + off, 2 * cblk, kNcclDouble, p, c->comm, xs));
          NCCLCHK(c, g_rccl.Recv(recv + (size_t)p * blk + off, 2 * cblk, kNcclDouble, p, c->comm, xs));
        }
        NCCLCHK(c, g_rccl.GroupEnd());
        if (!c->redir_now)
          HIPCHK(c, hipMemcpyAsync(recv + (size_t)c->rank * blk + off, send + (size_t)c->rank * blk + off, cblk * sizeof(cd), hipMemcpyDeviceToDevice, xs));
      }
    }
    c->n_exch += 1;
    if (c->link != LINK_NULL) c->bytes_sent += (double)cblk * 16.0 * (c->P - 1);
    if (to_y) {
      if (i == nch - 1) HIPCHK(c, hipEventRecord(c->ev_done, xs));
    } else {
      HIPCHK(c, hipEventRecord(c->ev_arr[g][i], xs));
      c->arr_pending[g] = true;
    }
  }
  return 0;
}
static int wait_arrival(nq_ctx* c, int g, int i, int nch_now) {
  // the group was sent with the step's chunking; a caller that uses fewer chunks waits for all the chunks it covers
  if (!c->arr_pending[g] || c->link == LINK_CALLBACK) return 0;
  const int sent = effective_chunks(c);
  const int per = sent / nch_now > 0 ? sent / nch_now : 1;
  for (int k = i * per; k < (i + 1) * per && k < sent; ++k) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_arr[g][k], 0));
  return 0;
}
static void set_window(nq_ctx* c, int i, int nch) {
  c->nrows = c->Nloc / nch;
  c->row0 = i * c->nrows;
}
#define SLABTRY(call)          \
  do {                         \
    int rc__ = (call);         \
    if (rc__) return rc__;     \
  } while (0)

// whole group g at once, complete on the compute streams when this returns to the caller's next launch
static int exchange_now(std::vector<nq_ctx*>& grp, int g, bool to_y) {
  if (grp.size() == 1 && grp[0]->P == 1 && grp[0]->G[g].bx == grp[0]->G[g].by) return 0;      // one rank, one buffer
  for (nq_ctx* c : each(grp)) HIPCHK(c, hipEventRecord(to_y ? c->ev_prod[0] : c->ev_col, c->stream));
  const int sent = effective_chunks(grp[0]);
  if (to_y) {
    SLABTRY(issue_chunk(grp, g, true, 0, 1));
    for (nq_ctx* c : each(grp))
      if (c->link != LINK_CALLBACK) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_done, 0));
  } else {
    // arrival events are per chunk of the step's chunking: send it that way so that later waits find every event recorded
    for (int i = 0; i < sent; ++i) SLABTRY(issue_chunk(grp, g, false, i, sent));
    for (nq_ctx* c : each(grp))
      for (int i = 0; i < sent; ++i) SLABTRY(wait_arrival(c, g, i, sent));
  }
  return 0;
}

struct PeerBufs { double* p[8]; };
__global__ void k_peer_allreduce(PeerBufs b, int P, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int r = 0; r < P; ++r) s += b.p[r][i];
  for (int r = 0; r < P; ++r) b.p[r][i] = s;
}
// sum over ranks of n doubles at offset `lo` of each rank's reduction block (bsums), or of its 32 diagnostic sums (which = 4)
static double* red_ptr(nq_ctx* c, int which, int* n) {
  switch (which) {
    case 0: *n = 44; return c->bsums;
    case 1: *n = 4; return c->carryW;
    case 2: *n = 3; return c->carryQ;
    case 3: *n = 1; return c->gradS1;
    case 4: *n = 16; return c->diag_out;           // diagnostics tick, spectral half
    case 5: *n = 16; return c->diag_out ? c->diag_out + 16 : nullptr;   // physical half
    default: *n = 0; return nullptr;
  }
}
static int slab_allreduce(std::vector<nq_ctx*>& grp, int which) {
  nq_ctx* c0 = grp[0];
  int n = 0;
  if (!red_ptr(c0, which, &n)) NQ_FAIL(c0, -1, "slab_allreduce: which = %d", which);
  if (c0->link == LINK_CALLBACK) {
    for (nq_ctx* c : each(grp)) {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (!c->rcb) NQ_FAIL(c, -4, "no all-reduce callback");
      const int rc = c->rcb(c->cb_user, which);
      if (rc) NQ_FAIL(c, -6, "all-reduce callback failed (%d)", rc);
    }
    return 0;
  }
  if (c0->link == LINK_NULL) return 0;           // one rank of P measured alone: its partial sums stay partial
  if (c0->link == LINK_PEERS) {
    if (grp.size() == 1) return 0;
    PeerBufs pb;
    for (size_t r = 0; r < grp.size(); ++r) {
      pb.p[r] = red_ptr(grp[r], which, &n);
      HIPCHK(grp[r], hipSetDevice(grp[r]->device));
      HIPCHK(grp[r], hipEventRecord(grp[r]->ev_red, grp[r]->stream));
    }
    HIPCHK(c0, hipSetDevice(c0->device));       // the reduction runs on rank 0's device and reads / writes the peers' sums in place
    for (size_t r = 0; r < grp.size(); ++r) HIPCHK(c0, hipStreamWaitEvent(c0->mstream, grp[r]->ev_red, 0));
    {
      XTimer xt(c0, true);
      hipLaunchKernelGGL(k_peer_allreduce, dim3(1), dim3(64), 0, c0->mstream, pb, (int)grp.size(), n);
    }
    HIPCHK(c0, hipEventRecord(c0->ev_done, c0->mstream));
    for (nq_ctx* c : each(grp)) HIPCHK(c, hipStreamWaitEvent(c->stream, c0->ev_done, 0));
    return 0;
  }
  for (nq_ctx* c : each(grp)) {                        // RCCL
    double* p = red_ptr(c, which, &n);
    HIPCHK(c, hipEventRecord(c->ev_red, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->mstream, c->ev_red, 0));
    {
      XTimer xt(c, true);
      NCCLCHK(c, g_rccl.AllReduce(p, p, (size_t)n, kNcclDouble, kNcclSum, c->comm, c->mstream));
    }
    HIPCHK(c, hipEventRecord(c->ev_done, c->mstream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_done, 0));
  }
  return 0;
}

// UPDATE split in two for the coupled model: the phi half produces exchange group 1, the q half runs under its transfer
static void phase_update_phi(nq_ctx* c, int s) {
  int wslot = 0;
  launch_A_m(c, false, {&c->mW});
  const int cur = c->w.cur;
  const cd* y_start = (s == 0) ? c->w.y[cur] : (s == 1 ? c->w.y[(cur + 1) % 3] : c->w.y[(cur + 2) % 3]);
  EtdArrays ew = etd_arrays(c->w, s, &wslot);
  launch_sphi(c, ew, s, y_start);
  launch_A_m(c, true, {&c->mPhi, &c->mPhiy});
}
static void phase_update_q(nq_ctx* c, int s) {
  int qslot = 0;
  launch_A_m(c, false, {&c->mUq, &c->mVq});
  EtdArrays eq = etd_arrays(c->q, s, &qslot);
  launch_sq(c, eq, s);
}

// YBJModel's stage graph (do_step_ybj) on slabs: only the wave products cross x -> y (group 0); the stage results 0..2
// come back in group 4 (gradients for the next stage), the new state in group 1
static int slab_step_ybj(std::vector<nq_ctx*>& grp) {
  nq_ctx* c0 = grp[0];
  const int nch = effective_chunks(c0);
  for (int s = 0; s < 4; ++s) {
    for (int i = 0; i < nch; ++i) {
      for (nq_ctx* c : each(grp)) {
        SLABTRY(wait_arrival(c, 3, i, nch));
        SLABTRY(wait_arrival(c, 1, i, nch));
        SLABTRY(wait_arrival(c, 4, i, nch));
        set_window(c, i, nch);
        launch_products(c, -1.0, -0.5, s == 0);
        HIPCHK(c, hipEventRecord(c->ev_prod[i], c->stream));
      }
      SLABTRY(issue_chunk(grp, 0, true, i, nch));
    }
    for (nq_ctx* c : each(grp)) {
      set_window(c, 0, 1);
      if (c->link != LINK_CALLBACK) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_done, 0));
      launch_A_m(c, false, {&c->mW});
      int wslot = 0;
      const int cur = c->w.cur;
      const cd* y_start = (s == 0) ? c->w.y[cur] : (s == 1 ? c->w.y[(cur + 1) % 3] : c->w.y[(cur + 2) % 3]);
      EtdArrays ew = etd_arrays(c->w, s, &wslot);
      if (s < 3) {
        launch_sphi(c, ew, s, y_start, &c->mGx, &c->mGy);
        launch_A_m(c, true, {&c->mGx, &c->mGy});
      } else {
        launch_sphi(c, ew, s, y_start);
        launch_A_m(c, true, {&c->mPhi, &c->mPhiy});
      }
      HIPCHK(c, hipEventRecord(c->ev_col, c->stream));
    }
    for (int i = 0; i < nch; ++i) SLABTRY(issue_chunk(grp, s < 3 ? 4 : 1, false, i, nch));
  }
  for (nq_ctx* c : each(grp)) c->n_steps += 1;
  return 0;
}

static int slab_step_once(std::vector<nq_ctx*>& grp) {
  nq_ctx* c0 = grp[0];
  if (c0->ybj) return slab_step_ybj(grp);
  const bool coupled = c0->p.model == NQ_MODEL_COUPLED, waves = c0->kernel_family;
  const int nch = effective_chunks(c0);
  for (int s = 0; s < 4; ++s) {
    if (s == 3 && c0->uv4_now)
      for (nq_ctx* c : each(grp)) {                   // the fourth stage's u, v have just arrived on the x side (group 3)
        for (int i = 0; i < nch; ++i) SLABTRY(wait_arrival(c, 3, i, nch));
        set_window(c, 0, 1);
        SLABTRY(stage4_uv_max(c));
      }
    // rows: nonlinear products, chunk by chunk; chunk i leaves as soon as it is done
    for (int i = 0; i < nch; ++i) {
      for (nq_ctx* c : each(grp)) {
        SLABTRY(wait_arrival(c, 3, i, nch));
        if (waves) SLABTRY(wait_arrival(c, 1, i, nch));
        set_window(c, i, nch);
        phase_products(c, s);
        HIPCHK(c, hipEventRecord(c->ev_prod[i], c->stream));
      }
      SLABTRY(issue_chunk(grp, 0, true, i, nch));
    }
    for (nq_ctx* c : each(grp)) {
      set_window(c, 0, 1);
      if (c->link != LINK_CALLBACK) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_done, 0));
    }
    if (coupled) {
      for (nq_ctx* c : each(grp)) {
        phase_update_phi(c, s);
        HIPCHK(c, hipEventRecord(c->ev_col, c->stream));
      }
      for (int i = 0; i < nch; ++i) SLABTRY(issue_chunk(grp, 1, false, i, nch));
      for (nq_ctx* c : each(grp)) phase_update_q(c, s);                    // under the transfer of group 1
      for (int i = 0; i < nch; ++i) {
        for (nq_ctx* c : each(grp)) {
          SLABTRY(wait_arrival(c, 1, i, nch));
          set_window(c, i, nch);
          phase_wavepv(c);
          HIPCHK(c, hipEventRecord(c->ev_prod[i], c->stream));
        }
        SLABTRY(issue_chunk(grp, 2, true, i, nch));
      }
      for (nq_ctx* c : each(grp)) {
        set_window(c, 0, 1);
        if (c->link != LINK_CALLBACK) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_done, 0));
        phase_invert(c, s);
        HIPCHK(c, hipEventRecord(c->ev_col, c->stream));
      }
      for (int i = 0; i < nch; ++i) SLABTRY(issue_chunk(grp, 3, false, i, nch));
    } else {
      for (nq_ctx* c : each(grp)) {
        phase_update(c, s);                                            // includes the (spectral) inversion
        HIPCHK(c, hipEventRecord(c->ev_col, c->stream));
      }
      if (waves)
        for (int i = 0; i < nch; ++i) SLABTRY(issue_chunk(grp, 1, false, i, nch));
      for (int i = 0; i < nch; ++i) SLABTRY(issue_chunk(grp, 3, false, i, nch));
    }
  }
  if (c0->bud) {
    for (nq_ctx* c : each(grp)) phase_budget_sums(c);
    SLABTRY(slab_allreduce(grp, 0));
    for (nq_ctx* c : each(grp)) phase_budget_finish(c);
  }
  for (nq_ctx* c : each(grp)) c->n_steps += 1;
  return 0;
}
// every arrival of the last step has to be on the compute stream before anybody else (a read, a set_q, a sync) touches the x side
static int slab_settle(std::vector<nq_ctx*>& grp) {
  for (nq_ctx* c : each(grp)) {
    const int sent = effective_chunks(c);
    for (int g = 0; g < 5; ++g)
      if (c->arr_pending[g]) {
        for (int i = 0; i < sent; ++i) SLABTRY(wait_arrival(c, g, i, sent));
        c->arr_pending[g] = false;
      }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// diagnostics tick launches
template <int S, int CLX>
static void launch_project_s(nq_ctx* c, double* part) {
  typedef YPlanT<S, CLX> Y;
  hipLaunchKernelGGL((k_s_project<S, CLX>), dim3(c->Wf / CLX, c->S2), dim3(Y::THREADS), Y::LDS_BYTES, c->stream, c->mW,
                     (const cd*)c->w.y[c->w.cur], geom_full(c), c->kk, c->ll, c->tw, 1, c->p.nu4w, c->p.nuw, c->p.muw, part);
}
static void launch_project(nq_ctx* c, double* part) {
#define CALL_(s, clx) launch_project_s<s, clx>(c, part)
  NQ_S1_SWITCH(c, CALL_)
#undef CALL_
}
template <int MODE, bool SLAB>
static void launch_xdiag_t(nq_ctx* c, double qbar, double abar, double* part) {
  switch (c->N) {
#define CASE_(n, a, b) case n: { typedef XPlan<n> X; hipLaunchKernelGGL((k_x_diag<n, MODE, SLAB>), dim3(c->Nloc / X::C), dim3(X::THREADS), X::LDS_BYTES, c->stream, c->mQ, c->mQw, c->mPhi, c->twx, c->kk, qbar, abar, part); } break;
    NQ_FOR_SIZES(CASE_)
#undef CASE_
  }
}
template <int MODE>
static void launch_xdiag_m(nq_ctx* c, double qbar, double abar, double* part) {
  if (c->P > 1) launch_xdiag_t<MODE, true>(c, qbar, abar, part);
  else launch_xdiag_t<MODE, false>(c, qbar, abar, part);
}
static int xdiag_blocks(const nq_ctx* c) {
  switch (c->N) {
#define CASE_(n, a, b) case n: return c->Nloc / XPlan<n>::C;
    NQ_FOR_SIZES(CASE_)
#undef CASE_
  }
  return 0;
}

// 1 read + 1 write stream copy, 16 B per lane: the rate a copy kernel reaches on THIS device, the second denominator of
// bench.py's roofline fractions.  Three shapes are timed and the best is reported (tools/copy_shape_bench.hip
// swept them, 4.4-6.5 TB/s): plain grid-stride; grid-stride with U loads then U stores in flight, non-temporal both ways (what
// MI355X_MICROARCH.md's 6.29 TB/s float4 copy is); the same at twice the depth.  Buffers live only for the call.
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_stream_copy(const double2* __restrict__ src, double2* __restrict__ dst, size_t n) {
  const size_t step = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * step < n; i += U * step) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) {
        v[u].x = __builtin_nontemporal_load(&src[i + u * step].x);
        v[u].y = __builtin_nontemporal_load(&src[i + u * step].y);
      } else {
        v[u] = src[i + u * step];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) {
        __builtin_nontemporal_store(v[u].x, &dst[i + u * step].x);
        __builtin_nontemporal_store(v[u].y, &dst[i + u * step].y);
      } else {
        dst[i + u * step] = v[u];
      }
    }
  }
  for (; i < n; i += step) dst[i] = src[i];
}
extern "C" {

const char* nq_last_error(const nq_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

// ---- slab geometry (shared by nq_group_elems and the constructor) -----------------------------------
struct SlabGeom {
  int N, P, Nloc, Wf, Wl, WhG, Ph;
  bool waves, coupled, bud, passive;
  int npitch[4];                 // row pitch of the four exchange groups
  int off[4][4];                 // column offset of each array inside its group's row
};
static int round8(int x) { return (x + 7) / 8 * 8; }
static SlabGeom slab_geom(const nq_params* p, int P) {
  SlabGeom g;
  g.N = p->nx;
  g.P = P;
  g.Nloc = g.N / P;
  g.Wf = g.N / P;
  g.WhG = g.N / 2 + 1;
  g.Wl = (g.WhG + P - 1) / P;
  g.Ph = (P == 1) ? g.N / 2 + 8 : round8(g.Wl);
  g.waves = p->model != NQ_MODEL_QG;
  g.coupled = p->model == NQ_MODEL_COUPLED;
  g.bud = p->budgets != 0;
  g.passive = p->model == NQ_MODEL_QG && p->passive_scalar != 0;
  const int hs = g.Ph;                          // segment stride of a half-spectrum array inside a row
  memset(g.off, 0, sizeof(g.off));
  // G0: Muq, Mvq, [Mw | Muc, Mvc]
  g.off[0][0] = 0; g.off[0][1] = hs; g.off[0][2] = 2 * hs; g.off[0][3] = 3 * hs;
  g.npitch[0] = 2 * hs + (g.waves ? g.Wf : 0) + (g.passive ? 2 * hs : 0);
  // G1: Mphi, Mphiy
  for (int i = 0; i < 4; ++i) g.off[1][i] = i * g.Wf;
  g.npitch[1] = g.waves ? 2 * g.Wf : 0;
  // G2: Ma, Mb
  g.off[2][0] = 0; g.off[2][1] = hs;
  g.npitch[2] = g.coupled ? 2 * hs : 0;
  // G3: Mu, Mp, Mq, [Mqw]
  for (int i = 0; i < 4; ++i) g.off[3][i] = i * hs;
  g.npitch[3] = ((g.coupled || g.passive) ? 4 : 3) * hs;
  return g;
}

long long nq_group_elems(const nq_params* p, int nranks, int group) {
  if (!p || group < 0 || group > 4 || nranks < 1 || p->nx % nranks) return -1;
  const SlabGeom g = slab_geom(p, nranks);
  if (group == 4) return (p->model == NQ_MODEL_YBJ && nranks > 1) ? (long long)g.N * g.npitch[1] : 0;
  return (long long)g.N * g.npitch[group];       // one side: P blocks of Nloc rows = N rows of `pitch`
}

static MArr make_marr(const SlabGeom& g, cd* bx, cd* by, int group, int idx, bool half) {
  MArr m;
  m.xs = bx ? bx + g.off[group][idx] : nullptr;
  m.ys = by ? by + g.off[group][idx] : nullptr;
  m.pitch = g.npitch[group];
  m.W = half ? (g.P == 1 ? g.WhG : g.Wl) : g.Wf;
  m.blk = (long long)g.Nloc * g.npitch[group];
  m.shift = -1;
  m.magic = 0;
  if (g.P == 1) m.shift = 30;                    // a single block: kx / W == 0
  else if ((m.W & (m.W - 1)) == 0) {
    int sh = 0;
    while ((1 << sh) < m.W) ++sh;
    m.shift = sh;
  } else {
    m.magic = (unsigned)(((1u << 24) + m.W - 1) / m.W);
  }
  return m;
}

static int create_impl(const nq_params* p_in, const double* kk, const double* ll, const double* filtr,
                       const double* contour, int device, int P, int rank, void* const* ext, void* ext_stream,
                       nq_ctx** out) {
  if (!p_in || !kk || !ll || !filtr || !contour || !out) NQ_FAIL((nq_ctx*)nullptr, -1, "nq_create: null argument");
  // YBJModel = UnCoupled layouts and kernels with its own stage graph; its step accumulates no budgets
  nq_params pp = *p_in;
  const bool ybj = pp.model == NQ_MODEL_YBJ;
  if (ybj) {
    pp.model = NQ_MODEL_UNCOUPLED;
    pp.budgets = 0;
  }
  const nq_params* p = &pp;
  int S1, S2;
  if (!plan_for(p->nx, &S1, &S2)) NQ_FAIL((nq_ctx*)nullptr, -2, "nq_create: nx=%d unsupported (power of two in [64, 8192])", p->nx);
  if (const char* e = getenv("NIWQG_AMD_Y_SPLIT")) {       // "S1,S2": another split of the two-pass y transform (measurements)
    int a = 0, b = 0;
    if (sscanf(e, "%d,%d", &a, &b) == 2 && a * b == p->nx && (a == 8 || a == 16 || a == 32 || a == 64) &&
        (b == 8 || b == 16 || b == 32 || b == 64 || b == 128)) {
      S1 = a;
      S2 = b;
    } else {
      NQ_FAIL((nq_ctx*)nullptr, -2, "NIWQG_AMD_Y_SPLIT=%s: S1 in {8,16,32,64}, S2 in {8..128}, S1*S2 = nx", e);
    }
  }
  if (p->model < 0 || p->model > 2) NQ_FAIL((nq_ctx*)nullptr, -2, "nq_create: unknown model %d", p->model);
  if (P < 1 || rank < 0 || rank >= P || (P & (P - 1)) || p->nx / P < CL)
    NQ_FAIL((nq_ctx*)nullptr, -2, "nq_create: %d ranks unsupported for nx=%d (power of two, at least %d columns per rank)", P, p->nx, CL);
  if (P > 1) {
    // the row kernels find the block of a half-spectrum column with kx / Wl = (kx * ceil(2^24 / Wl)) >> 24 (MArr::magic, Wl is not
    // a power of two): checked here for every column this grid has, not trusted (exact while kx * Wl < 2^24)
    const int whg = p->nx / 2 + 1, wl = (whg + P - 1) / P;
    if (wl & (wl - 1)) {
      const unsigned magic = (unsigned)(((1u << 24) + wl - 1) / wl);
      for (int kx = 0; kx < whg; ++kx)
        if ((int)(((unsigned)kx * magic) >> 24) != kx / wl)
          NQ_FAIL((nq_ctx*)nullptr, -2, "nq_create: %d ranks unsupported for nx=%d (block index of column %d)", P, p->nx, kx);
    }
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) NQ_FAIL((nq_ctx*)nullptr, -3, "nq_create: no HIP device available");
  if (device < 0 || device >= ndev) NQ_FAIL((nq_ctx*)nullptr, -3, "nq_create: device %d out of range (%d devices)", device, ndev);
  nq_ctx* c = new nq_ctx();
  const SlabGeom sg = slab_geom(p, P);
  c->p = *p;
  c->ybj = ybj;
  c->passive = sg.passive;
  c->N = p->nx;
  c->S1 = S1;
  c->S2 = S2;
  c->CLy = CL;
  {
    // Small grids on one rank: SINGLE-PASS columns (a tile of whole columns per workgroup, no A sub-pass).  A step of these
    // grids is a chain of dependent 5-microsecond kernels: what counts is how many there are.  NIWQG_AMD_SINGLE_PASS=0 keeps
    // the two-pass tiles (A/B measurements; slab contexts always use them).
    const char* e = getenv("NIWQG_AMD_SINGLE_PASS");
    if (P == 1 && p->nx <= 512 && !(e && atoi(e) == 0)) {
      if (getenv("NIWQG_AMD_Y_SPLIT")) {          // the knob would be silently overridden here: refuse instead
        delete c;
        NQ_FAIL((nq_ctx*)nullptr, -2, "NIWQG_AMD_Y_SPLIT has no effect on a single-rank grid <= 512 (single-pass columns); set NIWQG_AMD_SINGLE_PASS=0 with it");
      }
      c->S1 = p->nx;
      c->S2 = 1;
      c->CLy = p->nx >= 128 ? CLS : CL;
      // QGModel without its passive scalar: k_s_q + k_s_invert of a stage as ONE array-parallel kernel (k_c_qg); one wave per
      // array and column tile (NIWQG_AMD_SMALL_QG=0: the separate kernels)
      const char* e2 = getenv("NIWQG_AMD_SMALL_QG");
      if (p->model == NQ_MODEL_QG && !p->passive_scalar && p->nx >= 128 && !(e2 && atoi(e2) == 0))
        c->small_qg = p->nx == 128 ? 4 : 2;
    }
  }
  c->P = P;
  c->rank = rank;
  c->Nloc = sg.Nloc;
  c->row0 = 0;
  c->nrows = sg.Nloc;
  c->Wf = sg.Wf;
  c->kf0 = rank * sg.Wf;
  c->Wl = (P == 1) ? sg.WhG : sg.Wl;
  c->kh0 = (P == 1) ? 0 : rank * sg.Wl;
  c->WhG = sg.WhG;
  {
    const int left = sg.WhG - c->kh0;
    c->Wh = left < 0 ? 0 : (left < c->Wl ? left : c->Wl);      // valid local half-spectrum columns
  }
  c->Ph = sg.Ph;
  c->kernel_family = p->model != NQ_MODEL_QG;
  c->nk = c->kernel_family ? c->N : c->WhG;
  c->device = device;
  const int N = c->N;
  const size_t full = (size_t)N * c->Wf, half = (size_t)N * c->Ph;     // local spectral planes
#define FAILC(rc)        \
  do {                   \
    g_last_error = c->err; \
    nq_destroy(c);       \
    return (rc);         \
  } while (0)
#define TRY(call)              \
  do {                         \
    int rc__ = (call);         \
    if (rc__) FAILC(rc__);     \
  } while (0)
  auto setup = [&]() -> int {
    HIPCHK(c, hipSetDevice(device));
    {
      int ncu = 0;
      if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) c->num_cu = ncu;
    }
    if (ext_stream) {
      c->stream = reinterpret_cast<hipStream_t>(ext_stream);
      c->own_stream = false;
    } else {
      HIPCHK(c, hipStreamCreate(&c->stream));
    }
    HIPCHK(c, hipEventCreate(&c->ev0));
    HIPCHK(c, hipEventCreate(&c->ev1));
    // twiddles, long-double accurate
    std::vector<double> twh(2 * (size_t)N);
    for (int m = 0; m < N; ++m) {
      const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)N;
      twh[2 * m] = (double)cosl(a);
      twh[2 * m + 1] = (double)sinl(a);
    }
    ALLOC(c, c->tw, (size_t)N);
    HIPCHK(c, hipMemcpyAsync(c->tw, twh.data(), sizeof(double) * 2 * N, hipMemcpyHostToDevice, c->stream));
    // stage tables for the fused row kernels: [stage >= 1][w^1 | w^4 | w^8][jr < NS]
    for (int which = 0; which < 2; ++which) {
      const int PP = which == 0 ? XPlan<8>::pts(N) : XPlan1<8>::pts(N);
      std::vector<double> st;
      for (int sidx = 1; sidx < plan_stages(N, PP); ++sidx) {
        const int R = plan_radix(N, PP, sidx), NS = plan_ns(N, PP, sidx);
        for (int pw = 1; pw <= 8; pw *= (pw == 1 ? 4 : 2)) {          // powers 1, 4, 8 as far as the radix needs
          if ((pw == 4 && plan_tw_rows(N, PP, R) < 2) || (pw == 8 && plan_tw_rows(N, PP, R) < 3)) continue;
          for (int jr = 0; jr < NS; ++jr) {
            const long long m = ((long long)pw * jr * (N / (NS * R))) % N;
            st.push_back(twh[2 * m]);
            st.push_back(twh[2 * m + 1]);
          }
        }
      }
      cd*& dst = which == 0 ? c->twx : c->twx1;
      ALLOC(c, dst, st.size() / 2 + 1);
      HIPCHK(c, hipMemcpyAsync(dst, st.data(), sizeof(double) * st.size(), hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (N == 8192) {
      // k_x_products_eo: stage table of the 4096-point plan (its own twiddles: exp(-2 pi i m / 4096) = twh[2m])
      const int Mh = N / 2, PP = XPlan<8>::pts(Mh);
      std::vector<double> st;
      for (int sidx = 1; sidx < plan_stages(Mh, PP); ++sidx) {
        const int R = plan_radix(Mh, PP, sidx), NS = plan_ns(Mh, PP, sidx);
        for (int pw = 1; pw <= 8; pw *= (pw == 1 ? 4 : 2)) {
          if ((pw == 4 && plan_tw_rows(Mh, PP, R) < 2) || (pw == 8 && plan_tw_rows(Mh, PP, R) < 3)) continue;
          for (int jr = 0; jr < NS; ++jr) {
            const long long m = ((long long)pw * jr * (Mh / (NS * R))) % Mh;
            st.push_back(twh[2 * (2 * m)]);
            st.push_back(twh[2 * (2 * m) + 1]);
          }
        }
      }
      ALLOC(c, c->twx_half, st.size() / 2 + 1);
      HIPCHK(c, hipMemcpyAsync(c->twx_half, st.data(), sizeof(double) * st.size(), hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    ALLOC(c, c->kk, (size_t)N);
    ALLOC(c, c->ll, (size_t)N);
    HIPCHK(c, hipMemcpyAsync(c->kk, kk, sizeof(double) * c->nk, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->ll, ll, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
    ALLOC(c, c->contour, (size_t)32);
    HIPCHK(c, hipMemcpyAsync(c->contour, contour, sizeof(double) * 64, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // filter planes, local columns: half-spectrum copy (pitch Ph) and, for the Kernel family, full-width copy
    ALLOC(c, c->filt_h, half);
    if (c->Wh > 0)
      HIPCHK(c, hipMemcpy2DAsync(c->filt_h, sizeof(double) * c->Ph, filtr + c->kh0, sizeof(double) * c->nk, sizeof(double) * c->Wh, N, hipMemcpyHostToDevice, c->stream));
    if (c->kernel_family) {
      ALLOC(c, c->filt_f, full);
      HIPCHK(c, hipMemcpy2DAsync(c->filt_f, sizeof(double) * c->Wf, filtr + c->kf0, sizeof(double) * c->nk, sizeof(double) * c->Wf, N, hipMemcpyHostToDevice, c->stream));
    }
    {
      // is the filter mirror-symmetric in l on this rank's columns?  (the exponential filter and "none" are; the 2/3 mask is not)
      const char* e = getenv("NIWQG_AMD_COEF_MIRROR");
      bool sym = !(e && atoi(e) == 0) && N >= 4;
      for (int l = 1; sym && l < N / 2; ++l)
        for (int k = 0; k < c->nk; ++k)
          if (filtr[(size_t)l * c->nk + k] != filtr[(size_t)(N - l) * c->nk + k]) { sym = false; break; }
      c->cmirror = sym ? 1 : 0;
      c->crows = sym ? N / 2 + 1 : N;
    }
    const size_t half_c = (size_t)c->crows * c->Ph, full_c = (size_t)c->crows * c->Wf;
    // equations
    for (int i = 0; i < 3; ++i) ALLOC(c, c->q.y[i], half);
    ALLOC(c, c->q.fn0, half);
    ALLOC(c, c->q.fna, half);
    for (int i = 0; i < 6; ++i) ALLOC(c, c->q.coef[i], half_c);
    dim3 blk(64), grdh((c->Wh + 63) / 64, c->crows), grdf((c->Wf + 63) / 64, c->crows);
    if (c->Wh > 0)
      hipLaunchKernelGGL(k_etdrk4_coeffs, grdh, blk, 0, c->stream, c->kernel_family ? 0 : 2, N, c->Wh, c->Ph, c->kh0, c->p, c->kk, c->ll, c->filt_h, c->contour, c->q.coef[0], c->q.coef[1], c->q.coef[2], c->q.coef[3], c->q.coef[4], c->q.coef[5]);
    if (sg.passive) {
      for (int i = 0; i < 3; ++i) ALLOC(c, c->cq.y[i], half);
      ALLOC(c, c->cq.fn0, half);
      ALLOC(c, c->cq.fna, half);
      for (int i = 0; i < 6; ++i) ALLOC(c, c->cq.coef[i], half_c);
      if (c->Wh > 0)
        hipLaunchKernelGGL(k_etdrk4_coeffs, grdh, blk, 0, c->stream, 3, N, c->Wh, c->Ph, c->kh0, c->p, c->kk, c->ll, c->filt_h, c->contour, c->cq.coef[0], c->cq.coef[1], c->cq.coef[2], c->cq.coef[3], c->cq.coef[4], c->cq.coef[5]);
    }
    c->dual = c->kernel_family && p->dual_q != 0;
    if (c->dual) {
      for (int i = 0; i < 3; ++i) ALLOC(c, c->q2.y[i], half);
      ALLOC(c, c->q2.fn0, half);
      ALLOC(c, c->q2.fna, half);
      for (int i = 0; i < 6; ++i) ALLOC(c, c->coefu[i], half_c);
      if (c->Wh > 0)
        hipLaunchKernelGGL(k_etdrk4_coeffs, grdh, blk, 0, c->stream, 0, N, c->Wh, c->Ph, c->kh0, c->p, c->kk, c->ll, (const double*)nullptr, c->contour, c->coefu[0], c->coefu[1], c->coefu[2], c->coefu[3], c->coefu[4], c->coefu[5]);
      std::vector<double> fm((size_t)N * c->Ph, 0.0);
      for (int l = 0; l < N; ++l)
        for (int k = 0; k < c->Wh; ++k) fm[(size_t)l * c->Ph + k] = filtr[(size_t)((N - l) % N) * c->nk + (N - (c->kh0 + k)) % N];
      ALLOC(c, c->filt_m, half);
      HIPCHK(c, hipMemcpyAsync(c->filt_m, fm.data(), sizeof(double) * half, hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    c->pass = c->kernel_family && !c->dual && !ybj;
    if (c->pass) {
      for (int i = 0; i < 3; ++i) ALLOC(c, c->qp.y[i], (size_t)c->Ph);
      ALLOC(c, c->qp.fn0, (size_t)c->Ph);
      ALLOC(c, c->qp.fna, (size_t)c->Ph);
      for (int i = 0; i < 6; ++i) c->qp.coef[i] = c->q.coef[i] + (size_t)(N / 2) * c->Ph;     // row N/2 of the q planes
    }
    ALLOC(c, c->ph, half);
    ALLOC(c, c->qwh, half);
    if (P == 1) {                                   // scratch of the generic (single-rank) transform paths
      ALLOC(c, c->scr_h0, half);
      ALLOC(c, c->scr_h1, half);
      ALLOC(c, c->scr_r, (size_t)N * N);
      ALLOC(c, c->scr_f0, (size_t)N * N);
      ALLOC(c, c->scr_f1, (size_t)N * N);
    } else {
      ALLOC(c, c->scr_h0, (size_t)64);              // reduction scratch only
    }
    if (c->kernel_family) {
      for (int i = 0; i < 3; ++i) ALLOC(c, c->w.y[i], full);
      ALLOC(c, c->w.fn0, full);
      ALLOC(c, c->w.fna, full);
      for (int i = 0; i < 6; ++i) ALLOC(c, c->w.coef[i], full_c);
      hipLaunchKernelGGL(k_etdrk4_coeffs, grdf, blk, 0, c->stream, 1, N, c->Wf, c->Wf, c->kf0, c->p, c->kk, c->ll, c->filt_f, c->contour, c->w.coef[0], c->w.coef[1], c->w.coef[2], c->w.coef[3], c->w.coef[4], c->w.coef[5]);
    }
    c->bud = p->budgets != 0;
    // exchange groups: one buffer per side (the same buffer when P == 1); external (torch) buffers when given
    for (int gi = 0; gi < 4; ++gi) {
      c->G[gi].pitch = sg.npitch[gi];
      c->G[gi].elems = (size_t)N * sg.npitch[gi];
      if (c->G[gi].elems == 0) continue;
      if (ext && ext[2 * gi] && ext[2 * gi + 1]) {
        c->G[gi].bx = reinterpret_cast<cd*>(ext[2 * gi]);
        c->G[gi].by = reinterpret_cast<cd*>(ext[2 * gi + 1]);
        HIPCHK(c, hipMemsetAsync(c->G[gi].bx, 0, c->G[gi].elems * sizeof(cd), c->stream));
        HIPCHK(c, hipMemsetAsync(c->G[gi].by, 0, c->G[gi].elems * sizeof(cd), c->stream));
      } else {
        c->alloc_plain = true;
        ALLOC(c, c->G[gi].bx, c->G[gi].elems);
        if (P == 1) c->G[gi].by = c->G[gi].bx;
        else ALLOC(c, c->G[gi].by, c->G[gi].elems);
        c->alloc_plain = false;
      }
    }
    c->mUq = make_marr(sg, c->G[0].bx, c->G[0].by, 0, 0, true);
    c->mVq = make_marr(sg, c->G[0].bx, c->G[0].by, 0, 1, true);
    c->mW = make_marr(sg, c->G[0].bx, c->G[0].by, 0, 2, false);
    c->mPhi = make_marr(sg, c->G[1].bx, c->G[1].by, 1, 0, false);
    c->mPhiy = make_marr(sg, c->G[1].bx, c->G[1].by, 1, 1, false);
    c->mA = make_marr(sg, c->G[2].bx, c->G[2].by, 2, 0, true);
    c->mB = make_marr(sg, c->G[2].bx, c->G[2].by, 2, 1, true);
    c->mU = make_marr(sg, c->G[3].bx, c->G[3].by, 3, 0, true);
    c->mP = make_marr(sg, c->G[3].bx, c->G[3].by, 3, 1, true);
    c->mQ = make_marr(sg, c->G[3].bx, c->G[3].by, 3, 2, true);
    c->mQw = make_marr(sg, c->G[3].bx, c->G[3].by, 3, 3, true);
    if (sg.passive) {
      c->mUc = make_marr(sg, c->G[0].bx, c->G[0].by, 0, 2, true);
      c->mVc = make_marr(sg, c->G[0].bx, c->G[0].by, 0, 3, true);
    }
    if (P == 1 && p->model == NQ_MODEL_COUPLED && N == 4096) {
      const char* e = getenv("NIWQG_AMD_OVERLAP_CUS");
      c->overlap_cus = e ? atoi(e) : 0;
      if (c->overlap_cus > 0 && c->overlap_cus < c->num_cu) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
      }
    }
    c->mGx = c->mPhi;
    c->mGy = c->mPhiy;
    if (c->p.model == NQ_MODEL_UNCOUPLED) {        // frozen copy of the X side of G1 (quirk Q1)
      ALLOC(c, c->Gs, c->G[1].elems);
      cd* gsy = c->Gs;
      if (ybj && P > 1) {                          // its stage results cross y -> x in a group of their own
        ALLOC(c, c->Gs_y, c->G[1].elems);
        gsy = c->Gs_y;
        c->G[4].bx = c->Gs;
        c->G[4].by = c->Gs_y;
        c->G[4].pitch = c->G[1].pitch;
        c->G[4].elems = c->G[1].elems;
      }
      c->mGx = make_marr(sg, c->Gs, gsy, 1, 0, false);
      c->mGy = make_marr(sg, c->Gs, gsy, 1, 1, false);
    }
    if (c->bud) {
      c->nww = (c->Wf / c->CLy) * c->S2;
      c->nwq = ((c->Wh + c->CLy - 1) / c->CLy) * c->S2;
      if (c->small_qg && (c->Wh + c->small_qg - 1) / c->small_qg > c->nwq) c->nwq = (c->Wh + c->small_qg - 1) / c->small_qg;
      if (c->nwq < 1) c->nwq = 1;
      ALLOC(c, c->partQ, (size_t)4 * c->nwq * (sg.passive ? 6 : 3));
      ALLOC(c, c->part0Q, (size_t)c->nwq * (sg.passive ? 6 : 3));
      ALLOC(c, c->acc, (size_t)4);
      // everything that has to be summed over ranks lives in one 64-double block (external when the caller
      // does the all-reduce): [0,44) stage sums of a step, [44,48) carried phi sums, [48,51) carried q sums,
      // [51] frozen gradient sum
      if (ext && ext[8]) {
        c->bsums = reinterpret_cast<double*>(ext[8]);
        HIPCHK(c, hipMemsetAsync(c->bsums, 0, 64 * sizeof(double), c->stream));
      } else {
        ALLOC(c, c->bsums, (size_t)64);
      }
      c->carryW = c->bsums + 44;
      c->carryQ = c->bsums + 48;
      c->gradS1 = c->bsums + 51;
      if (c->kernel_family) {
        ALLOC(c, c->partW, (size_t)4 * c->nww * NQ_PARTW);
        ALLOC(c, c->part0W, (size_t)c->nww * NQ_PARTW);
      }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    return 0;
  };
  TRY(setup());
  *out = c;
  return 0;
}

int nq_create(const nq_params* p, const double* kk, const double* ll, const double* filtr, const double* contour,
              int device, nq_ctx** out) {
  return create_impl(p, kk, ll, filtr, contour, device, 1, 0, nullptr, nullptr, out);
}

int nq_create_slab(const nq_params* p, const double* kk, const double* ll, const double* filtr, const double* contour,
                   int device, int nranks, int rank, void* const* buffers, void* stream, nq_ctx** out) {
  return create_impl(p, kk, ll, filtr, contour, device, nranks, rank, buffers, stream, out);
}

int nq_destroy(nq_ctx* c) {
  if (!c) return 0;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  for (void* p : c->allocs) hipFree(p);
  for (auto& pt : c->patch) { (void)hipFree(pt.l); (void)hipFree(pt.k); (void)hipFree(pt.v); }
  for (hipEvent_t e : c->prof_ev) hipEventDestroy(e);
  for (hipEvent_t e : c->marks)
    if (e) hipEventDestroy(e);
  if (c->ev0) hipEventDestroy(c->ev0);
  if (c->ev1) hipEventDestroy(c->ev1);
  if (c->snap_stream) {
    hipStreamSynchronize(c->snap_stream);
    hipStreamDestroy(c->snap_stream);
  }
  if (c->ev_snap) hipEventDestroy(c->ev_snap);
  if (c->snap_hq) hipHostFree(c->snap_hq);
  if (c->snap_hphi) hipHostFree(c->snap_hphi);
  if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  if (c->mstream) hipStreamSynchronize(c->mstream);
  for (hipEvent_t e : c->ev_prod)
    if (e) hipEventDestroy(e);
  for (auto& row : c->ev_arr)
    for (hipEvent_t e : row)
      if (e) hipEventDestroy(e);
  for (hipEvent_t e : {c->ev_col, c->ev_done, c->ev_red})
    if (e) hipEventDestroy(e);
  for (hipEvent_t e : c->xev) hipEventDestroy(e);
  if (c->mstream) hipStreamDestroy(c->mstream);
  if (c->stream2) hipStreamDestroy(c->stream2);
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  if (c->ev_join) hipEventDestroy(c->ev_join);
  if (c->stream && c->own_stream) hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

void* nq_stream(nq_ctx* c) { return c ? (void*)c->stream : nullptr; }
long long nq_device_bytes(const nq_ctx* c) { return c ? c->bytes : 0; }

int nq_sync(nq_ctx* c) {
  if (!c) return -1;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // ... and the exchange stream: the last y -> x group of a slab step is only waited for by the NEXT row kernel, so the
  // library's communicator can still have sends / receives queued when the compute stream is idle.  A caller that syncs
  // and then runs a collective of its own (torch.distributed) must never find two communicators active on the device.
  if (c->mstream) HIPCHK(c, hipStreamSynchronize(c->mstream));
  HIPCHK(c, hipGetLastError());
  return 0;
}

int nq_stream_copy_gbs(nq_ctx* c, long long bytes, int reps, double* gbs_out) {
  if (!c || !gbs_out || bytes < (1 << 20) || reps < 1) return -1;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)bytes / sizeof(double2);
  double2 *a = nullptr, *b = nullptr;
  HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&a), n * sizeof(double2)));
  if (hipMalloc(reinterpret_cast<void**>(&b), n * sizeof(double2)) != hipSuccess) {
    (void)hipFree(a);
    NQ_FAIL(c, -5, "nq_stream_copy_gbs: out of device memory");
  }
  (void)hipMemsetAsync(a, 1, n * sizeof(double2), c->stream);
  (void)hipMemsetAsync(b, 0, n * sizeof(double2), c->stream);
  auto launch = [&](int shape) {
    const double2* src = a;
    switch (shape) {
      case 0: hipLaunchKernelGGL((k_stream_copy<1, false>), dim3(c->num_cu * 8), dim3(256), 0, c->stream, src, b, n); break;
      case 1: hipLaunchKernelGGL((k_stream_copy<4, true>), dim3(c->num_cu * 8), dim3(256), 0, c->stream, src, b, n); break;
      case 2: hipLaunchKernelGGL((k_stream_copy<4, true>), dim3(c->num_cu * 16), dim3(256), 0, c->stream, src, b, n); break;
      default: hipLaunchKernelGGL((k_stream_copy<8, true>), dim3(c->num_cu * 8), dim3(256), 0, c->stream, src, b, n); break;
    }
  };
  launch(0);                                                                                            // untimed warm-up
  double best = 0.0;
  int rc = 0;
  for (int shape = 0; shape < 4 && rc == 0; ++shape)
    for (int r = 0; r < reps && rc == 0; ++r) {
      if (hipEventRecord(c->ev0, c->stream) != hipSuccess) { rc = -5; break; }
      launch(shape);
      float ms = 0.f;
      if (hipEventRecord(c->ev1, c->stream) != hipSuccess || hipEventSynchronize(c->ev1) != hipSuccess ||
          hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) { rc = -5; break; }
      const double g = 2.0 * (double)(n * sizeof(double2)) / (ms * 1e-3) / 1e9;
      best = g > best ? g : best;
    }
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(a);
  (void)hipFree(b);
  if (rc) NQ_FAIL(c, rc, "nq_stream_copy_gbs: event timing failed");
  *gbs_out = best;
  return 0;
}

int nq_profile_enable(nq_ctx* c, int kernel_class) {
  if (!c) return -1;
  c->prof_class = kernel_class;
  c->prof_used = 0;
  c->prof_seen = 0;
  return 0;
}
// Bracket only every stride-th launch of the enabled class(es): an event pair costs the stream about 9 microseconds, 2 % of a
// 4096^2 step when all 20 A sub-passes of a step are bracketed and 10 % of a rank's step on eight slab ranks.  A stride coprime
// with the launch pattern of a step (5 A sub-passes per stage, 20 per step: 7) samples every position equally often.
int nq_profile_stride(nq_ctx* c, int stride) {
  if (!c || stride < 1) return -1;
  c->prof_stride = stride;
  c->prof_seen = 0;
  return 0;
}
int nq_profile_read(nq_ctx* c, int* launches, float* total_ms) {
  if (!c || !launches || !total_ms) return -1;
  int rc = nq_sync(c);
  if (rc) return rc;
  float tot = 0.f;
  for (size_t i = 0; i + 1 < c->prof_used; i += 2) {
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->prof_ev[i], c->prof_ev[i + 1]));
    tot += ms;
  }
  *launches = (int)(c->prof_used / 2);
  *total_ms = tot;
  c->prof_used = 0;
  return 0;
}

int nq_profile_read_all(nq_ctx* c, int* launches6, float* total_ms6) {
  if (!c || !launches6 || !total_ms6) return -1;
  int rc = nq_sync(c);
  if (rc) return rc;
  for (int k = 0; k < 6; ++k) {
    launches6[k] = 0;
    total_ms6[k] = 0.f;
  }
  for (size_t i = 0; i + 1 < c->prof_used; i += 2) {
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->prof_ev[i], c->prof_ev[i + 1]));
    const int k = c->prof_cls[i / 2];
    if (k >= 0 && k < 6) {
      launches6[k] += 1;
      total_ms6[k] += ms;
    }
  }
  c->prof_used = 0;
  return 0;
}

int nq_timer_start(nq_ctx* c) {
  if (!c) return -1;
  HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  return 0;
}
int nq_timer_stop(nq_ctx* c, float* ms) {
  if (!c || !ms) return -1;
  HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  HIPCHK(c, hipEventSynchronize(c->ev1));
  HIPCHK(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
  return 0;
}

int nq_event_record(nq_ctx* c, int slot) {
  if (!c) return -1;
  if (slot < 0 || slot >= 16) NQ_FAIL(c, -1, "nq_event_record: slot %d (0..15)", slot);
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->marks[slot]) HIPCHK(c, hipEventCreate(&c->marks[slot]));
  HIPCHK(c, hipEventRecord(c->marks[slot], c->stream));
  return 0;
}
int nq_event_elapsed(nq_ctx* c, int slot_a, int slot_b, float* ms) {
  if (!c || !ms) return -1;
  if (slot_a < 0 || slot_a >= 16 || slot_b < 0 || slot_b >= 16 || !c->marks[slot_a] || !c->marks[slot_b])
    NQ_FAIL(c, -1, "nq_event_elapsed: slots %d, %d not recorded", slot_a, slot_b);
  HIPCHK(c, hipEventSynchronize(c->marks[slot_b]));
  HIPCHK(c, hipEventElapsedTime(ms, c->marks[slot_a], c->marks[slot_b]));
  return 0;
}

int nq_set_q(nq_ctx* c, const double* q_host) {
  if (!c || !q_host) return -1;
  NQ_SINGLE_RANK(c, "nq_set_q");
  const size_t full = (size_t)c->N * c->N;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(c->scr_r, q_host, sizeof(double) * full, hipMemcpyHostToDevice, c->stream));
  fwd2d_half(c, c->scr_r, c->q.y[c->q.cur], c->scr_h0);
  if (c->dual) HIPCHK(c, hipMemcpyAsync(c->q2.y[c->q2.cur], c->q.y[c->q.cur], sizeof(cd) * (size_t)c->N * c->Ph, hipMemcpyDeviceToDevice, c->stream));
  { int rc = reset_passenger(c); if (rc) return rc; }
  do_invert_now(c);
  c->have_q = true;
  return nq_sync(c);
}

int nq_set_c(nq_ctx* c, const double* c_host) {
  if (!c || !c_host) return -1;
  NQ_SINGLE_RANK(c, "nq_set_c");
  if (!c->passive) NQ_FAIL(c, -4, "nq_set_c: this context has no passive scalar");
  const size_t full = (size_t)c->N * c->N;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(c->scr_r, c_host, sizeof(double) * full, hipMemcpyHostToDevice, c->stream));
  fwd2d_half(c, c->scr_r, c->cq.y[c->cq.cur], c->scr_h0);
  do_invert_now(c);                 // re-emits u, psi, q and the scalar's mixed-space rows
  return nq_sync(c);
}

int nq_set_phi(nq_ctx* c, const double* phi_host) {
  if (!c || !phi_host) return -1;
  NQ_SINGLE_RANK(c, "nq_set_phi");
  if (!c->kernel_family) NQ_FAIL(c, -4, "nq_set_phi: QGModel has no wave field");
  const size_t full = (size_t)c->N * c->N;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(c->scr_f0, phi_host, sizeof(cd) * full, hipMemcpyHostToDevice, c->stream));
  fwd2d_full(c, c->scr_f0, c->w.y[c->w.cur], c->scr_f1);
  launch_emit_phi(c, c->w.y[c->w.cur]);
  launch_A_m(c, true, {&c->mPhi, &c->mPhiy});
  if (c->bud) hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, c->stream, c->part0W, c->nww, NQ_PARTW, 4, c->carryW);
  c->have_phi = true;
  int rc = nq_refresh_grad_phi(c);
  if (rc) return rc;
  return nq_sync(c);
}

int nq_invert(nq_ctx* c) {
  if (!c) return -1;
  NQ_SINGLE_RANK(c, "nq_invert");
  HIPCHK(c, hipSetDevice(c->device));
  do_invert_now(c);
  return nq_sync(c);
}

int nq_refresh_grad_phi(nq_ctx* c) {
  if (!c) return -1;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->p.model == NQ_MODEL_UNCOUPLED) {
    // freeze the X side of group 1 (phi, phiy rows as the row kernels see them)
    HIPCHK(c, hipMemcpyAsync(c->Gs, c->G[1].bx, sizeof(cd) * c->G[1].elems, hipMemcpyDeviceToDevice, c->stream));
    if (c->bud) HIPCHK(c, hipMemcpyAsync(c->gradS1, c->carryW + 1, sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  }
  return 0;
}

int nq_step(nq_ctx* c, int nsteps) {
  if (!c) return -1;
  NQ_SINGLE_RANK(c, "nq_step");
  if (nsteps < 0) NQ_FAIL(c, -1, "nq_step: nsteps < 0");
  HIPCHK(c, hipSetDevice(c->device));
  for (int i = 0; i < nsteps; ++i) {
    c->uv4_now = c->want_uv4 && i == nsteps - 1 && c->kernel_family && !c->ybj;
    do_step(c);
  }
  if (nsteps > 0) {
    c->stepped = true;
    c->have_uv4 = c->uv4_now;
    c->want_uv4 = c->uv4_now = false;
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}
// The last step of the NEXT nq_step / nq_slab_step call also records max |u|, max |v| of its fourth stage over this rank's rows
// (two extra row passes in that one step); nq_get_stage4_max reads them (-4 when the last call recorded none).
int nq_request_stage4_max(nq_ctx* c) {
  if (!c) return -1;
  c->want_uv4 = true;
  return 0;
}
int nq_get_stage4_max(nq_ctx* c, double* out2) {
  if (!c || !out2) return -1;
  if (!c->have_uv4 || !c->uv4) NQ_FAIL(c, -4, "nq_get_stage4_max: the last step call was not asked to record the fourth stage's maxima");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(out2, c->uv4, sizeof(double) * 2, hipMemcpyDeviceToHost, c->stream));
  return nq_sync(c);
}

// ---- slab decomposition API ------------------------------------------------------------------------
int nq_slab_info(const nq_ctx* c, int* info) {
  // info[0..7] = nranks, rank, local rows, full-plane local columns, first full column,
  //              half-spectrum local (valid) columns, first half-spectrum column, half-spectrum pitch
  if (!c || !info) return -1;
  info[0] = c->P; info[1] = c->rank; info[2] = c->Nloc; info[3] = c->Wf; info[4] = c->kf0;
  info[5] = c->Wh; info[6] = c->kh0; info[7] = c->Ph;
  return 0;
}

int nq_group_buffers(nq_ctx* c, int group, void** x_side, void** y_side, long long* elems) {
  if (!c || group < 0 || group > 3) return -1;
  if (x_side) *x_side = c->G[group].bx;
  if (y_side) *y_side = c->G[group].by;
  if (elems) *elems = (long long)c->G[group].elems;
  return 0;
}

// local column slab of a spectral state plane: which 0 = qh (half spectrum, (ny, local valid columns)),
// 1 = phih (full plane, (ny, local columns)); host arrays are contiguous
int nq_upload_spectral(nq_ctx* c, int which, const double* host) {
  if (!c || !host) return -1;
  HIPCHK(c, hipSetDevice(c->device));
  if (which == 0) {
    if (c->Wh > 0) HIPCHK(c, hipMemcpy2DAsync(c->q.y[c->q.cur], sizeof(cd) * c->Ph, host, sizeof(cd) * c->Wh, sizeof(cd) * c->Wh, c->N, hipMemcpyHostToDevice, c->stream));
    if (c->dual) HIPCHK(c, hipMemcpyAsync(c->q2.y[c->q2.cur], c->q.y[c->q.cur], sizeof(cd) * (size_t)c->N * c->Ph, hipMemcpyDeviceToDevice, c->stream));
    { int rc = reset_passenger(c); if (rc) return rc; }
  } else if (which == 1) {
    if (!c->kernel_family) NQ_FAIL(c, -4, "no wave field in QGModel");
    HIPCHK(c, hipMemcpyAsync(c->w.y[c->w.cur], host, sizeof(cd) * (size_t)c->N * c->Wf, hipMemcpyHostToDevice, c->stream));
  } else NQ_FAIL(c, -1, "nq_upload_spectral: which = %d", which);
  return nq_sync(c);
}
int nq_download_spectral(nq_ctx* c, int which, double* host) {
  if (!c || !host) return -1;
  HIPCHK(c, hipSetDevice(c->device));
  if (which == 0 || (which >= 2 && which <= 6) || which == 8) {   // half-spectrum planes: qh, ph, qwh, second copy of qh, ch, stage-4 qh (both copies)
    if (which == 3 && c->p.model != NQ_MODEL_COUPLED) NQ_FAIL(c, -4, "qwh exists only in the coupled model");
    if ((which == 4 || which == 8) && !c->dual) NQ_FAIL(c, -4, "no second copy of qh in this context (dual_q)");
    if (which == 5 && !c->passive) NQ_FAIL(c, -4, "no passive scalar in this context");
    const cd* src = which == 0 ? c->q.y[c->q.cur] : (which == 2 ? c->ph : (which == 3 ? c->qwh : (which == 4 ? c->q2.y[c->q2.cur] : (which == 5 ? c->cq.y[c->cq.cur] : (which == 8 ? c->q2.y[(c->q2.cur + 2) % 3] : c->q.y[(c->q.cur + 2) % 3])))));
    if (c->Wh > 0) HIPCHK(c, hipMemcpy2DAsync(host, sizeof(cd) * c->Wh, src, sizeof(cd) * c->Ph, sizeof(cd) * c->Wh, c->N, hipMemcpyDeviceToHost, c->stream));
  } else if (which == 1 || which == 7) {
    if (!c->kernel_family) NQ_FAIL(c, -4, "no wave field in QGModel");
    const cd* src = which == 1 ? c->w.y[c->w.cur] : c->w.y[(c->w.cur + 2) % 3];
    HIPCHK(c, hipMemcpyAsync(host, src, sizeof(cd) * (size_t)c->N * c->Wf, hipMemcpyDeviceToHost, c->stream));
  } else if (which == 9 || which == 11 || which == 12) {      // nq_tick_snapshot: qh, its second copy, qwh
    const cd* src = which == 9 ? c->tick_qh : (which == 11 ? c->tick_q2 : c->tick_qwh);
    if (!src) NQ_FAIL(c, -4, "nq_download_spectral: which = %d: no nq_tick_snapshot yet (or no such plane in this context)", which);
    if (c->Wh > 0) HIPCHK(c, hipMemcpy2DAsync(host, sizeof(cd) * c->Wh, src, sizeof(cd) * c->Ph, sizeof(cd) * c->Wh, c->N, hipMemcpyDeviceToHost, c->stream));
  } else if (which == 10) {                                   // ... and phih
    if (!c->tick_w) NQ_FAIL(c, -4, "nq_download_spectral: which = 10: no nq_tick_snapshot yet (or no wave field)");
    HIPCHK(c, hipMemcpyAsync(host, c->tick_w, sizeof(cd) * (size_t)c->N * c->Wf, hipMemcpyDeviceToHost, c->stream));
  } else NQ_FAIL(c, -1, "nq_download_spectral: which = %d", which);
  return nq_sync(c);
}
// What only a diagnostics tick refreshes in the reference (upsilon: Kernel.py:618; phq, phw, uq, vq, uw, vw: CoupledModel.py:
// 99-113; YBJModel's lapphi: Kernel.py:685 through the tick's _calc_energy_conversion) stays the TICK's until the next one, however
// many steps follow.  The host classes rebuild those arrays on demand; this keeps the spectra they derive from: device-to-device
// copies of this rank's qh (both copies), phih and qwh on the context's stream (planes allocated by the first call).
int nq_tick_snapshot(nq_ctx* c) {
  if (!c) return -1;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t full = (size_t)c->N * c->Wf, half = (size_t)c->N * c->Ph;
  if (!c->tick_qh) {
    ALLOC(c, c->tick_qh, half);
    if (c->dual) ALLOC(c, c->tick_q2, half);
    if (c->kernel_family) ALLOC(c, c->tick_w, full);
    if (c->qwh) ALLOC(c, c->tick_qwh, half);
  }
  HIPCHK(c, hipMemcpyAsync(c->tick_qh, c->q.y[c->q.cur], sizeof(cd) * half, hipMemcpyDeviceToDevice, c->stream));
  if (c->tick_q2) HIPCHK(c, hipMemcpyAsync(c->tick_q2, c->q2.y[c->q2.cur], sizeof(cd) * half, hipMemcpyDeviceToDevice, c->stream));
  if (c->tick_w) HIPCHK(c, hipMemcpyAsync(c->tick_w, c->w.y[c->w.cur], sizeof(cd) * full, hipMemcpyDeviceToDevice, c->stream));
  if (c->tick_qwh) HIPCHK(c, hipMemcpyAsync(c->tick_qwh, c->qwh, sizeof(cd) * half, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

// One phase of the distributed step (see "phases of one ETDRK4 stage").  Asynchronous on the stream.
//   NQ_PH_PRODUCTS(stage)  row kernel: reads G3.x, G1.x, writes G0.x             then exchange G0 (x -> y)
//   NQ_PH_UPDATE(stage)    y passes + stage updates: reads G0.y, writes G1.y (and G3.y unless Coupled)
//                                                                                  then exchange G1 (y -> x) [, G3]
//   NQ_PH_WAVEPV           Coupled row kernel: reads G1.x, writes G2.x            then exchange G2 (x -> y)
//   NQ_PH_INVERT(stage)    Coupled inversion: reads G2.y, writes G3.y             then exchange G3 (y -> x)
//   NQ_PH_EMIT_PHI         after nq_upload_spectral(phih): writes G1.y            then exchange G1 (y -> x)
//   NQ_PH_INVERT_NOW       inversion of the current qh (set_q): Coupled: run NQ_PH_WAVEPV + exchange G2 first
//   NQ_PH_BUDGET_SUMS      local sums of one step -> nq_budget_sums buffer        then all-reduce (sum) it
//   NQ_PH_BUDGET_FINISH    RK-weighted accumulation from the (reduced) sums
int nq_phase(nq_ctx* c, int phase, int stage) {
  if (!c) return -1;
  if (c->ybj) NQ_FAIL(c, -4, "nq_phase: YBJModel has a stage graph of its own (nq_step, nq_slab_step)");
  if (stage < 0 || stage > 3) NQ_FAIL(c, -1, "nq_phase: stage %d", stage);
  HIPCHK(c, hipSetDevice(c->device));
  switch (phase) {
    case NQ_PH_PRODUCTS: phase_products(c, stage); break;
    case NQ_PH_UPDATE: phase_update(c, stage); break;
    case NQ_PH_WAVEPV: phase_wavepv(c); break;
    case NQ_PH_INVERT: phase_invert(c, stage); break;
    case NQ_PH_EMIT_PHI:
      if (!c->kernel_family) NQ_FAIL(c, -4, "no wave field in QGModel");
      launch_emit_phi(c, c->w.y[c->w.cur]);
      launch_A_m(c, true, {&c->mPhi, &c->mPhiy});
      // local part of the carried sums; the caller all-reduces them (nq_reduce_buffer)
      if (c->bud) hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, c->stream, c->part0W, c->nww, NQ_PARTW, 4, c->carryW);
      break;
    case NQ_PH_INVERT_NOW:
      phase_invert_y(c, c->q.y[c->q.cur], true, c->part0Q, nullptr);
      if (c->bud && c->kernel_family)
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, c->stream, c->part0Q, c->nwq, 3, 3, c->carryQ);
      break;
    case NQ_PH_BUDGET_SUMS: phase_budget_sums(c); break;
    case NQ_PH_BUDGET_FINISH: phase_budget_finish(c); break;
    default: NQ_FAIL(c, -1, "nq_phase: unknown phase %d", phase);
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}

// The 64-double block that has to be summed over ranks at three points (see nq_phase): elements [0,44) after
// NQ_PH_BUDGET_SUMS, [44,48) after NQ_PH_EMIT_PHI, [48,51) after NQ_PH_INVERT_NOW.  It is buffers[8] of
// nq_create_slab when that was given.
int nq_reduce_buffer(nq_ctx* c, int which, void** ptr, int* count) {
  if (!c || !ptr || !count) return -1;
  if (!c->bud) NQ_FAIL(c, -4, "budgets are disabled in this context");
  switch (which) {
    case 0: *ptr = c->bsums; *count = 44; break;
    case 1: *ptr = c->carryW; *count = 4; break;
    case 2: *ptr = c->carryQ; *count = 3; break;
    case 3: *ptr = c->gradS1; *count = 1; break;
    case 4: *ptr = c->diag_out; *count = 16; break;
    case 5: *ptr = c->diag_out ? c->diag_out + 16 : nullptr; *count = 16; break;
    default: NQ_FAIL(c, -1, "nq_reduce_buffer: which = %d", which);
  }
  return 0;
}


// ---- the slab step inside the library: API ------------------------------------------------------------------------
int nq_comm_probe(void) {           // load-only: can this process resolve librccl at all?  (no communicator, no GPU work)
  std::string err;
  if (!rccl_load(&err)) NQ_FAIL((nq_ctx*)nullptr, -6, "nq_comm_probe: %s", err.c_str());
  return 0;
}
int nq_comm_unique_id(void* out128) {
  if (!out128) return -1;
  std::string err;
  if (!rccl_load(&err)) NQ_FAIL((nq_ctx*)nullptr, -6, "nq_comm_unique_id: %s", err.c_str());
  NcclId id;
  NCCLCHK((nq_ctx*)nullptr, g_rccl.GetUniqueId(&id));
  memcpy(out128, &id, sizeof(id));
  return 0;
}
int nq_comm_init(nq_ctx* c, const void* id128, int nranks, int rank) {
  if (!c || !id128) return -1;
  if (nranks != c->P || rank != c->rank) NQ_FAIL(c, -1, "nq_comm_init: rank %d of %d given to a context created as rank %d of %d", rank, nranks, c->rank, c->P);
  if (c->link != LINK_NONE) NQ_FAIL(c, -4, "nq_comm_init: the context already has a link");
  std::string err;
  if (!rccl_load(&err)) NQ_FAIL(c, -6, "nq_comm_init: %s", err.c_str());
  HIPCHK(c, hipSetDevice(c->device));
  NcclId id;
  memcpy(&id, id128, sizeof(id));
  NCCLCHK(c, g_rccl.CommInitRank(&c->comm, nranks, id, rank));
  c->link = LINK_RCCL;
  if (nranks > 1) {
    // RCCL's send/recv kernels run BESIDE the chunked row kernels, and a persistent row-kernel workgroup (512 threads x
    // ~250 VGPRs) fills its CU's register file, so they compete for whole CUs.  NIWQG_AMD_SLAB_RESERVE_CUS=R keeps R CUs
    // out of the row kernels' persistent grids.  Off by default: with 256 or 512 rows per chunk a grid of 256-R
    // workgroups needs an extra round of rows, which costs what the overlap gains -- to be settled
    // by measurement on a multi-GPU node.
    const char* e = getenv("NIWQG_AMD_SLAB_RESERVE_CUS");
    c->reserve_cus = e ? atoi(e) : 0;
    if (c->reserve_cus < 0 || c->reserve_cus >= c->num_cu) c->reserve_cus = 0;
  }
  return slab_link_setup(c);
}
int nq_slab_attach_peers(nq_ctx* const* ctxs, int nranks) {
  if (!ctxs || nranks < 1 || nranks > 8) NQ_FAIL((nq_ctx*)nullptr, -1, "nq_slab_attach_peers: 1..8 contexts");
  std::vector<nq_ctx*> all(ctxs, ctxs + nranks);
  for (int r = 0; r < nranks; ++r) {
    nq_ctx* c = all[r];
    if (!c || c->P != nranks || c->rank != r || c->link != LINK_NONE)
      NQ_FAIL(c, -1, "nq_slab_attach_peers: context %d is not rank %d of %d (or already linked)", r, r, nranks);
  }
  // peers on different devices of this process: the blocks then cross by peer copies (SDMA over xGMI, no CU involved) and the
  // all-reduce kernel reads the peers' buffers directly -- both need peer access, enabled here in both directions
  for (nq_ctx* a : all)
    for (nq_ctx* b : all) {
      if (a->device == b->device) continue;
      int can = 0;
      HIPCHK(a, hipDeviceCanAccessPeer(&can, a->device, b->device));
      if (!can) NQ_FAIL(a, -3, "nq_slab_attach_peers: device %d cannot access device %d", a->device, b->device);
      HIPCHK(a, hipSetDevice(a->device));
      const hipError_t e = hipDeviceEnablePeerAccess(b->device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) NQ_FAIL(a, -5, "hipDeviceEnablePeerAccess(%d -> %d): %s", a->device, b->device, hipGetErrorString(e));
      (void)hipGetLastError();
    }
  for (nq_ctx* c : all) {
    c->peers = all;
    c->link = LINK_PEERS;
    int rc = slab_link_setup(c);
    if (rc) return rc;
  }
  return 0;
}
// Measurement aid (bench.py --rank-of P): this context is ONE rank of its P-rank decomposition and runs alone.  Streams, events,
// chunking and every kernel launch are those of a real rank; on the wire only the rank's own block moves (the device copy the RCCL
// link makes as well), the other P-1 blocks of every group keep whatever they held, and nothing is all-reduced.  The fields are
// therefore NOT a simulation -- what is measured is a rank's compute and launch structure without the exchange.
int nq_slab_set_null_link(nq_ctx* c) {
  if (!c) return -1;
  if (c->link != LINK_NONE) NQ_FAIL(c, -4, "nq_slab_set_null_link: the context already has a link");
  const char* e = getenv("NIWQG_AMD_SLAB_RESERVE_CUS");
  c->reserve_cus = e ? atoi(e) : 0;
  if (c->reserve_cus < 0 || c->reserve_cus >= c->num_cu) c->reserve_cus = 0;
  c->link = LINK_NULL;
  return slab_link_setup(c);
}
int nq_slab_set_callbacks(nq_ctx* c, nq_exchange_fn exchange, nq_allreduce_fn allreduce, void* user) {
  if (!c || !exchange || !allreduce) return -1;
  if (c->link != LINK_NONE && c->link != LINK_CALLBACK) NQ_FAIL(c, -4, "nq_slab_set_callbacks: the context already has a link");
  c->xcb = exchange;
  c->rcb = allreduce;
  c->cb_user = user;
  c->link = LINK_CALLBACK;
  return slab_link_setup(c);
}
// caller-owned buffers for exchange group 4 (as ext_buffers of nq_create_slab for groups 0..3), before the first step
int nq_slab_set_stage_buffers(nq_ctx* c, void* bx, void* by) {
  if (!c || !bx || !by) return -1;
  if (!c->ybj || c->P == 1) NQ_FAIL(c, -4, "nq_slab_set_stage_buffers: only YBJModel on more than one rank has exchange group 4");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const SlabGeom sg = slab_geom(&c->p, c->P);
  c->Gs = c->G[4].bx = reinterpret_cast<cd*>(bx);
  c->Gs_y = c->G[4].by = reinterpret_cast<cd*>(by);
  HIPCHK(c, hipMemsetAsync(c->Gs, 0, c->G[4].elems * sizeof(cd), c->stream));
  HIPCHK(c, hipMemsetAsync(c->Gs_y, 0, c->G[4].elems * sizeof(cd), c->stream));
  c->mGx = make_marr(sg, c->Gs, c->Gs_y, 1, 0, false);
  c->mGy = make_marr(sg, c->Gs, c->Gs_y, 1, 1, false);
  return nq_sync(c);
}
int nq_slab_config(nq_ctx* c, int nchunks) {
  if (!c) return -1;
  if (nchunks != 1 && nchunks != 2 && nchunks != 4 && nchunks != 8) NQ_FAIL(c, -1, "nq_slab_config: nchunks = %d (1, 2, 4 or 8)", nchunks);
  std::vector<nq_ctx*> grp = c->link == LINK_PEERS ? c->peers : std::vector<nq_ctx*>(1, c);
  int rc = slab_settle(grp);
  if (rc) return rc;
  for (nq_ctx* x : each(grp)) x->nchunk = nchunks;
  return 0;
}
int nq_slab_step(nq_ctx* c, int nsteps) {
  if (!c) return -1;
  if (nsteps < 0) NQ_FAIL(c, -1, "nq_slab_step: nsteps < 0");
  std::vector<nq_ctx*> grp;
  SLABTRY(slab_group(c, &grp));
  for (nq_ctx* x : each(grp)) HIPCHK(x, hipSetDevice(x->device));
  c->n_calls += 1;
  // inside the step every exchange has an A sub-pass next to it on the Y side: the rank's own block is used where it lies
  // (ArrayListR) instead of being copied across -- not with caller-moved buffers (the callback moves the own block too) and not
  // on plans without an A sub-pass (single-pass columns)
  static int redirect_on = -1;
  if (redirect_on < 0) {
    const char* e = getenv("NIWQG_AMD_SLAB_OWN_REDIRECT");
    redirect_on = (e && atoi(e) == 0) ? 0 : 1;
  }
  const bool redir = redirect_on && c->link != LINK_CALLBACK && c->link != LINK_NONE && c->S2 > 1 && c->P > 1;
  for (nq_ctx* x : each(grp)) x->redir_now = redir;
  int step_rc = 0;
  for (int i = 0; i < nsteps && step_rc == 0; ++i) {
    const bool now = c->want_uv4 && i == nsteps - 1 && c->kernel_family && !c->ybj;
    for (nq_ctx* x : each(grp)) x->uv4_now = now;
    step_rc = slab_step_once(grp);
  }
  for (nq_ctx* x : each(grp)) x->redir_now = false;
  if (step_rc) return step_rc;
  if (nsteps > 0)
    for (nq_ctx* x : each(grp)) {
      x->stepped = true;
      x->have_uv4 = x->uv4_now;
      x->want_uv4 = x->uv4_now = false;
    }
  SLABTRY(slab_settle(grp));
  for (nq_ctx* x : each(grp)) HIPCHK(x, hipGetLastError());
  return 0;
}

// rows of a physical field -> the x side of the carrier array of exchange group 0 (q: the uq slot, phi: the W slot)
static int rows_scratch(nq_ctx* c, cd** out) {
  if (!c->scr_f0) ALLOC(c, c->scr_f0, (size_t)c->Nloc * c->N);
  *out = c->scr_f0;
  return 0;
}
int nq_slab_put_rows(nq_ctx* c, int which, const double* rows) {
  if (!c || !rows) return -1;
  if (which == 1 && !c->kernel_family) NQ_FAIL(c, -4, "no wave field in QGModel");
  HIPCHK(c, hipSetDevice(c->device));
  cd* scr = nullptr;
  SLABTRY(rows_scratch(c, &scr));
  const size_t n = (size_t)c->Nloc * c->N;
  if (which == 2 && !c->passive) NQ_FAIL(c, -4, "nq_slab_put_rows: this context has no passive scalar");
  if (which == 0 || which == 2) {
    HIPCHK(c, hipMemcpyAsync(scr, rows, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    switch (c->N) {
#define CASE_(nn, a, b) case nn: { typedef XPlan<nn> X; hipLaunchKernelGGL((k_x_put_real<nn, true>), dim3((c->Nloc + X::C - 1) / X::C), dim3(X::THREADS), X::LDS_BYTES, c->stream, (const double*)scr, c->mUq, c->Nloc, c->tw); } break;
      NQ_FOR_SIZES(CASE_)
#undef CASE_
    }
  } else if (which == 1) {
    HIPCHK(c, hipMemcpyAsync(scr, rows, sizeof(cd) * n, hipMemcpyHostToDevice, c->stream));
    switch (c->N) {
#define CASE_(nn, a, b) case nn: { typedef XPlan<nn> X; hipLaunchKernelGGL((k_x_put_cplx<nn, true>), dim3((c->Nloc + X::C - 1) / X::C), dim3(X::THREADS), X::LDS_BYTES, c->stream, (const cd*)scr, c->mW, c->Nloc, c->tw); } break;
      NQ_FOR_SIZES(CASE_)
#undef CASE_
    }
  } else NQ_FAIL(c, -1, "nq_slab_put_rows: which = %d", which);
  HIPCHK(c, hipGetLastError());
  return nq_sync(c);                            // the host rows may be released
}
// collective half of set_q / set_phi: exchange the carrier, finish the transform on the column slabs, then the phases of
// Kernel.set_q (Kernel.py:520-535: inversion with the current phi, quirk Q2) / Kernel.set_phi (:538-551)
int nq_slab_commit(nq_ctx* c, int which) {
  if (!c) return -1;
  std::vector<nq_ctx*> grp;
  SLABTRY(slab_group(c, &grp));
  for (nq_ctx* x : each(grp)) HIPCHK(x, hipSetDevice(x->device));
  SLABTRY(slab_settle(grp));
  nq_ctx* c0 = grp[0];
  if (which == 0) {
    SLABTRY(exchange_now(grp, 0, true));
    for (nq_ctx* x : each(grp)) {
      launch_A_m(x, false, {&x->mUq});
      if (x->Wh > 0) launch_B_p(x, false, x->mUq.ys, x->mUq.pitch, x->q.y[x->q.cur], x->Ph, x->Wh, 1.0);
      if (x->dual)      // q is real: both copies of the dual-copy equation start from the same half spectrum
        HIPCHK(x, hipMemcpyAsync(x->q2.y[x->q2.cur], x->q.y[x->q.cur], sizeof(cd) * (size_t)x->N * x->Ph, hipMemcpyDeviceToDevice, x->stream));
      SLABTRY(reset_passenger(x));
    }
    if (c0->p.model == NQ_MODEL_COUPLED) {
      for (nq_ctx* x : each(grp)) phase_wavepv(x);
      SLABTRY(exchange_now(grp, 2, true));
    }
    for (nq_ctx* x : each(grp)) {
      phase_invert_y(x, x->q.y[x->q.cur], true, x->part0Q, nullptr, x->passive ? x->cq.y[x->cq.cur] : nullptr);
      if (x->bud && x->kernel_family)
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, x->stream, x->part0Q, x->nwq, 3, 3, x->carryQ);
      x->have_q = true;
    }
    SLABTRY(exchange_now(grp, 3, false));
    if (c0->bud && c0->kernel_family) SLABTRY(slab_allreduce(grp, 2));
  } else if (which == 1) {
    if (!c0->kernel_family) NQ_FAIL(c, -4, "no wave field in QGModel");
    SLABTRY(exchange_now(grp, 0, true));
    for (nq_ctx* x : each(grp)) {
      launch_A_m(x, false, {&x->mW});
      launch_B_p(x, false, x->mW.ys, x->mW.pitch, x->w.y[x->w.cur], x->Wf, x->Wf, 1.0);
      launch_emit_phi(x, x->w.y[x->w.cur]);
      launch_A_m(x, true, {&x->mPhi, &x->mPhiy});
      if (x->bud) hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, x->stream, x->part0W, x->nww, NQ_PARTW, 4, x->carryW);
      x->have_phi = true;
    }
    SLABTRY(exchange_now(grp, 1, false));
    if (c0->bud) SLABTRY(slab_allreduce(grp, 1));
    for (nq_ctx* x : each(grp)) SLABTRY(nq_refresh_grad_phi(x));
  } else if (which == 2 || which == 3) {        // Kernel._invert on the current state (CoupledModel.py:75-97 / UnCoupledModel.py:54-64)
    if (which == 3) {                           // QGModel.set_c (QGModel.py:476-480): the scalar's spectrum first
      if (!c0->passive) NQ_FAIL(c, -4, "nq_slab_commit: this context has no passive scalar");
      SLABTRY(exchange_now(grp, 0, true));
      for (nq_ctx* x : each(grp)) {
        launch_A_m(x, false, {&x->mUq});
        if (x->Wh > 0) launch_B_p(x, false, x->mUq.ys, x->mUq.pitch, x->cq.y[x->cq.cur], x->Ph, x->Wh, 1.0);
      }
    }
    if (c0->p.model == NQ_MODEL_COUPLED) {
      for (nq_ctx* x : each(grp)) phase_wavepv(x);
      SLABTRY(exchange_now(grp, 2, true));
    }
    for (nq_ctx* x : each(grp)) {
      phase_invert_y(x, x->q.y[x->q.cur], true, x->part0Q, nullptr, x->passive ? x->cq.y[x->cq.cur] : nullptr);
      if (x->bud && x->kernel_family)
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, x->stream, x->part0Q, x->nwq, 3, 3, x->carryQ);
    }
    SLABTRY(exchange_now(grp, 3, false));
    if (c0->bud && c0->kernel_family) SLABTRY(slab_allreduce(grp, 2));
  } else NQ_FAIL(c, -1, "nq_slab_commit: which = %d", which);
  SLABTRY(slab_settle(grp));
  for (nq_ctx* x : each(grp)) {
    HIPCHK(x, hipGetLastError());
    SLABTRY(nq_sync(x));
  }
  return 0;
}
// this rank's rows of a physical field, from the mixed-space rows of the last inversion / emission on the x side
int nq_slab_get_rows(nq_ctx* c, int id, double* out) {
  if (!c || !out) return -1;
  HIPCHK(c, hipSetDevice(c->device));
  cd* scr = nullptr;
  SLABTRY(rows_scratch(c, &scr));
  const size_t n = (size_t)c->Nloc * c->N;
  const MArr* src = nullptr;
  int mode = 0, zero_nyq = 0, mul_ik = 0;
  bool real = true;
  switch (id) {
    case NQ_F_Q: src = &c->mQ; break;
    case NQ_F_P: src = &c->mP; break;
    case NQ_F_U: src = &c->mU; break;
    case NQ_F_V: src = &c->mP; mode = 1; zero_nyq = c->kernel_family ? 1 : 0; break;
    case NQ_F_QW:
      if (c->p.model != NQ_MODEL_COUPLED) NQ_FAIL(c, -4, "qw exists only in the coupled model");
      src = &c->mQw;
      break;
    case NQ_F_C:
      if (!c->passive) NQ_FAIL(c, -4, "no passive scalar in this context");
      src = &c->mQw;                            // the scalar rides in the qw slot of group 3
      break;
    case NQ_F_PHI: src = &c->mPhi; real = false; break;
    case NQ_F_PHIX: src = &c->mGx; real = false; mul_ik = 1; break;
    case NQ_F_PHIY: src = &c->mGy; real = false; break;
    default: NQ_FAIL(c, -1, "nq_slab_get_rows: field id %d", id);
  }
  if (!real && !c->kernel_family) NQ_FAIL(c, -4, "no wave field in QGModel");
  switch (c->N) {
#define CASE_(nn, a, b) case nn: { typedef XPlan<nn> X; const dim3 grid((c->Nloc + X::C - 1) / X::C), blk(X::THREADS); \
      if (real) hipLaunchKernelGGL((k_x_get_real<nn, true>), grid, blk, X::LDS_BYTES, c->stream, *src, reinterpret_cast<double*>(scr), c->Nloc, c->tw, c->kk, mode, zero_nyq); \
      else hipLaunchKernelGGL((k_x_get_cplx<nn, true>), grid, blk, X::LDS_BYTES, c->stream, *src, scr, c->Nloc, c->tw, c->kk, mul_ik); } break;
    NQ_FOR_SIZES(CASE_)
#undef CASE_
  }
  HIPCHK(c, hipMemcpyAsync(out, scr, (real ? sizeof(double) : sizeof(cd)) * n, hipMemcpyDeviceToHost, c->stream));
  return nq_sync(c);
}

// Diagnostics tick of a slab-decomposed simulation: the 32 sums of nq_diagnostics, every rank's part summed over the ranks
// (two all-reduces: the spectral half gives the two means the physical half is centred with).  Collective.
int nq_slab_diagnostics(nq_ctx* c, double* out) {
  if (!c || !out) return -1;
  std::vector<nq_ctx*> grp;
  SLABTRY(slab_group(c, &grp));
  SLABTRY(slab_settle(grp));
  nq_ctx* c0 = grp[0];
  const int N = c0->N, NB = 1024;
  const double M = (double)N * N;
  const bool waves = c0->kernel_family, coupled = c0->p.model == NQ_MODEL_COUPLED;
  for (nq_ctx* x : each(grp)) {
    HIPCHK(x, hipSetDevice(x->device));
    const int nxb = xdiag_blocks(x), nww = (x->Wf / x->CLy) * x->S2;
    if (!x->diag_part) {
      size_t need = (size_t)NB * 9;
      if ((size_t)nxb * 8 > need) need = (size_t)nxb * 8;
      if ((size_t)nww * 4 > need) need = (size_t)nww * 4;
      ALLOC(x, x->diag_part, need);
      ALLOC(x, x->diag_out, (size_t)40);
    }
    double* d = x->diag_out;
    HIPCHK(x, hipMemsetAsync(d, 0, sizeof(double) * 40, x->stream));
    const cd* qh = x->q.y[x->q.cur];
    if (x->dual) {                                  // physical space sees the mean of the two copies
      if (!x->scr_h1) ALLOC(x, x->scr_h1, (size_t)N * x->Ph);
      if (x->Wh > 0) hipLaunchKernelGGL(k_avg_interior_g, dim3((x->Wh + 63) / 64, N), dim3(64), 0, x->stream, qh, (const cd*)x->q2.y[x->q2.cur], x->scr_h1, x->Wh, x->Ph, N, x->kh0);
      qh = x->scr_h1;
    }
    if (waves) {
      const cd* phih = x->w.y[x->w.cur];
      hipLaunchKernelGGL(k_diag_phi, dim3(NB), dim3(256), 0, x->stream, phih, N, x->Wf, x->Wf, x->kf0, x->kk, x->ll, x->diag_part);
      hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, x->stream, x->diag_part, NB, 4, 4, d);
      if (x->kf0 == 0) HIPCHK(x, hipMemcpyAsync(d + 4, phih, sizeof(cd), hipMemcpyDeviceToDevice, x->stream));
    }
    if (x->Wh > 0) {
      hipLaunchKernelGGL(k_diag_q, dim3(NB), dim3(256), 0, x->stream, qh, (const cd*)(coupled ? x->qwh : nullptr), (const cd*)x->ph, N, x->Wh, x->Ph, x->kh0, x->kk, x->ll, x->diag_part,
                         (const cd*)(x->dual ? x->q.y[x->q.cur] : nullptr), (const cd*)(x->dual ? x->q2.y[x->q2.cur] : nullptr),
                         (const double*)(x->dual ? x->filt_h : nullptr), (const double*)(x->dual ? x->filt_m : nullptr),
                         (const cd*)(x->pass ? x->qp.y[x->qp.cur] : nullptr));
      hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, x->stream, x->diag_part, NB, 9, 9, d + 6);
    }
    if (x->kh0 == 0) {                              // [15] <- Re(qh - qwh)[0,0] (the owner of column 0 contributes it)
      HIPCHK(x, hipMemcpyAsync(d + 15, qh, sizeof(double), hipMemcpyDeviceToDevice, x->stream));
      if (coupled) {
        HIPCHK(x, hipMemcpyAsync(d + 32, x->qwh, sizeof(double), hipMemcpyDeviceToDevice, x->stream));
        hipLaunchKernelGGL(k_axpy1, dim3(1), dim3(1), 0, x->stream, d + 15, d + 32, -1.0);
      }
    }
  }
  SLABTRY(slab_allreduce(grp, 4));
  double h[16];
  HIPCHK(c0, hipSetDevice(c0->device));
  HIPCHK(c0, hipMemcpyAsync(h, c0->diag_out, sizeof(double) * 16, hipMemcpyDeviceToHost, c0->stream));
  SLABTRY(nq_sync(c0));
  for (int i = 0; i < 15; ++i) out[i] = h[i];
  for (int i = 15; i < 32; ++i) out[i] = 0.0;
  const double qbar = h[15] / M, abar = h[0] / (M * M);
  out[15] = qbar;
  if (!waves && c0->passive) {
    // as in nq_diagnostics: [16..19] the |c-hat|^2 sums, [20] the Gamma_c projection with the u, v of the fourth stage
    for (nq_ctx* x : each(grp)) {
      const cd* ch = x->cq.y[x->cq.cur];
      if (x->Wh > 0) {
        hipLaunchKernelGGL(k_diag_c, dim3(NB), dim3(256), 0, x->stream, ch, N, x->Wh, x->Ph, x->kh0, x->kk, x->ll, x->diag_part);
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, x->stream, x->diag_part, NB, 4, 4, x->diag_out + 16);
      }
      const cd* qh4 = x->stepped ? x->q.y[(x->q.cur + 2) % 3] : x->q.y[x->q.cur];
      phase_invert_y(x, qh4, false, x->part0Q, nullptr, ch);
    }
    SLABTRY(exchange_now(grp, 3, false));
    for (nq_ctx* x : each(grp)) launch_products(x);
    SLABTRY(exchange_now(grp, 0, true));
    for (nq_ctx* x : each(grp)) {
      launch_A_m(x, false, {&x->mUc, &x->mVc});
      const YGeom g = geom_half(x);
      if (g.width <= 0) continue;
      const int nwc = ((g.width + x->CLy - 1) / x->CLy) * x->S2;
#define CALL_(sz, clx) hipLaunchKernelGGL((k_s_project_c<sz, clx>), dim3((g.width + clx - 1) / clx, x->S2), dim3((YPlanT<sz, clx>::THREADS)), (YPlanT<sz, clx>::LDS_BYTES), x->stream, x->mUc, x->mVc, x->cq.y[x->cq.cur], g, x->kk, x->ll, x->tw, 1, x->diag_part)
      NQ_S1_SWITCH(x, CALL_)
#undef CALL_
      hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, x->stream, x->diag_part, nwc, 1, 1, x->diag_out + 20);
    }
    for (nq_ctx* x : each(grp))                           // the mixed-space rows of the CURRENT state again, for the next step
      phase_invert_y(x, x->q.y[x->q.cur], true, x->part0Q, nullptr, x->cq.y[x->cq.cur]);
    SLABTRY(exchange_now(grp, 3, false));
    SLABTRY(slab_allreduce(grp, 5));
    HIPCHK(c0, hipSetDevice(c0->device));
    HIPCHK(c0, hipMemcpyAsync(out + 16, c0->diag_out + 16, sizeof(double) * 5, hipMemcpyDeviceToHost, c0->stream));
    SLABTRY(slab_settle(grp));
    for (nq_ctx* x : each(grp)) SLABTRY(nq_sync(x));
    return 0;
  }
  if (!waves) return 0;
  for (nq_ctx* x : each(grp)) {
    if (coupled) launch_xdiag_m<MODE_COUPLED>(x, qbar, abar, x->diag_part);
    else launch_xdiag_m<MODE_UNCOUPLED>(x, qbar, abar, x->diag_part);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, x->stream, x->diag_part, xdiag_blocks(x), 8, 8, x->diag_out + 16);
  }
  for (int which = 0; which < 2; ++which) {
    for (nq_ctx* x : each(grp)) launch_products(x, which == 0 ? 1.0 : 0.0, which == 0 ? 0.0 : 1.0);
    SLABTRY(exchange_now(grp, 0, true));
    for (nq_ctx* x : each(grp)) {
      const int nww = (x->Wf / x->CLy) * x->S2;
      launch_A_m(x, false, {&x->mW});
      launch_project(x, x->diag_part);
      hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, x->stream, x->diag_part, nww, 4, 4, x->diag_out + 24 + 4 * which);
    }
  }
  SLABTRY(slab_allreduce(grp, 5));
  HIPCHK(c0, hipSetDevice(c0->device));
  HIPCHK(c0, hipMemcpyAsync(out + 16, c0->diag_out + 16, sizeof(double) * 16, hipMemcpyDeviceToHost, c0->stream));
  for (nq_ctx* x : each(grp)) SLABTRY(nq_sync(x));
  return 0;
}
// Spectra the whole-plane calls of the class API need on a slab model (fft seam, the three Jacobians): the row kernel of
// the current state (or the rows uploaded by nq_slab_put_rows), exchange, column transform into a scratch column slab.
//   what 0 / 1: F[u q] / F[v q]                 half-spectrum slab (Kernel.py:471-486, QGModel.py:469-481)
//        2:     F[u phix + v phiy]              full-width slab, [0,0] as computed (Kernel.py:457-469)
//        3:     F[i phi q_psi]                  full-width slab (the refraction source, Kernel.py:332)
//        4:     F[Re i(phix* phiy - phiy* phix)] half-spectrum slab (CoupledModel.py:59-73)
//        5 / 6: forward transform of the real / complex rows last given to nq_slab_put_rows(0 / 1)
// Groups 0 and 2 are dead between steps, so the state is left alone.
int nq_slab_spectral(nq_ctx* c, int what) {
  if (!c) return -1;
  if (what < 0 || what > 6) NQ_FAIL(c, -1, "nq_slab_spectral: what = %d", what);
  std::vector<nq_ctx*> grp;
  SLABTRY(slab_group(c, &grp));
  for (nq_ctx* x : each(grp)) HIPCHK(x, hipSetDevice(x->device));
  SLABTRY(slab_settle(grp));
  nq_ctx* c0 = grp[0];
  const bool full = what == 2 || what == 3 || what == 6;
  if (full && !c0->kernel_family) NQ_FAIL(c, -4, "nq_slab_spectral: no wave field in QGModel");
  if (what == 4 && (c0->p.model != NQ_MODEL_COUPLED || c0->ybj)) NQ_FAIL(c, -4, "jacobian_phic_phi exists only in the coupled model");
  if (what <= 3 && !c0->have_q) NQ_FAIL(c, -4, "nq_slab_spectral: set_q has not been called");
  if ((what == 2 || what == 3 || what == 4) && !c0->have_phi) NQ_FAIL(c, -4, "nq_slab_spectral: set_phi has not been called");
  for (nq_ctx* x : each(grp)) {
    if (!x->scr_f1) {
      const size_t w = (size_t)(x->Wf > x->Ph ? x->Wf : x->Ph);
      ALLOC(x, x->scr_f1, (size_t)x->N * w);
    }
    set_window(x, 0, 1);
    if (what <= 1) launch_products(x);
    else if (what == 2) launch_products(x, 1.0, 0.0);
    else if (what == 3) launch_products(x, 0.0, 1.0);
    else if (what == 4) launch_wavepv(x);
  }
  SLABTRY(exchange_now(grp, what == 4 ? 2 : 0, true));
  for (nq_ctx* x : each(grp)) {
    const MArr& m = (what == 0 || what == 5) ? x->mUq : (what == 1 ? x->mVq : (what == 4 ? x->mB : x->mW));
    launch_A_m(x, false, {&m});
    if (full) launch_B_p(x, false, m.ys, m.pitch, x->scr_f1, x->Wf, x->Wf, 1.0);
    else if (x->Wh > 0) launch_B_p(x, false, m.ys, m.pitch, x->scr_f1, x->Ph, x->Wh, 1.0);
    HIPCHK(x, hipGetLastError());
    x->spec_kind = full ? 0 : 1;
  }
  for (nq_ctx* x : each(grp)) SLABTRY(nq_sync(x));
  return 0;
}
// this rank's column slab of the last nq_slab_spectral: (nx, wh) for the half-spectrum results, (nx, wf) for the others
int nq_slab_spectral_read(nq_ctx* c, int half, double* out) {
  if (!c || !out) return -1;
  if (c->spec_kind < 0) NQ_FAIL(c, -4, "nq_slab_spectral_read: nothing computed yet");
  if (c->spec_kind != (half ? 1 : 0)) NQ_FAIL(c, -1, "nq_slab_spectral_read: the last nq_slab_spectral result is a %s slab", c->spec_kind ? "half-spectrum" : "full-width");
  HIPCHK(c, hipSetDevice(c->device));
  const int w = half ? c->Wh : c->Wf, pitch = half ? c->Ph : c->Wf;
  if (w > 0) HIPCHK(c, hipMemcpy2DAsync(out, sizeof(cd) * w, c->scr_f1, sizeof(cd) * pitch, sizeof(cd) * w, c->N, hipMemcpyDeviceToHost, c->stream));
  return nq_sync(c);
}
// max |u|, max |v|, max |phi| over THIS rank's rows (the caller takes the max over ranks: Kernel._calc_cfl, Kernel.py:660-662)
int nq_slab_local_max(nq_ctx* c, double* out3) {
  if (!c || !out3) return -1;
  HIPCHK(c, hipSetDevice(c->device));
  cd* scr = nullptr;
  SLABTRY(rows_scratch(c, &scr));
  if (!c->diag_out) {
    ALLOC(c, c->diag_out, (size_t)40);
  }
  double* d = c->diag_out + 34;
  HIPCHK(c, hipMemsetAsync(d, 0, sizeof(double) * 3, c->stream));
  const size_t n = (size_t)c->Nloc * c->N;
  for (int which = 0; which < (c->kernel_family ? 3 : 2); ++which) {
    const MArr& src = which == 0 ? c->mU : (which == 1 ? c->mP : c->mPhi);
    switch (c->N) {
#define CASE_(nn, a, b) case nn: { typedef XPlan<nn> X; const dim3 grid((c->Nloc + X::C - 1) / X::C), blk(X::THREADS); \
        if (which < 2) hipLaunchKernelGGL((k_x_get_real<nn, true>), grid, blk, X::LDS_BYTES, c->stream, src, reinterpret_cast<double*>(scr), c->Nloc, c->tw, c->kk, which, (which == 1 && c->kernel_family) ? 1 : 0); \
        else hipLaunchKernelGGL((k_x_get_cplx<nn, true>), grid, blk, X::LDS_BYTES, c->stream, src, scr, c->Nloc, c->tw, c->kk, 0); } break;
      NQ_FOR_SIZES(CASE_)
#undef CASE_
    }
    if (which < 2) hipLaunchKernelGGL(k_reduce_real_max, dim3(1024), dim3(256), 0, c->stream, reinterpret_cast<const double*>(scr), n, d + which);
    else hipLaunchKernelGGL(k_reduce, dim3((c->N + 255) / 256, c->Nloc), dim3(256), 0, c->stream, (const cd*)scr, c->N, c->N, c->N, 3, c->kk, c->ll, d + 2);
  }
  HIPCHK(c, hipMemcpyAsync(out3, d, sizeof(double) * 3, hipMemcpyDeviceToHost, c->stream));
  return nq_sync(c);
}
int nq_slab_counters(nq_ctx* c, double* out, int reset) {
  if (!c || !out) return -1;
  SLABTRY(nq_sync(c));
  if (c->mstream) HIPCHK(c, hipStreamSynchronize(c->mstream));
  float tot = 0.f;
  for (size_t i = 0; i + 1 < c->xev_used; i += 2) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->xev[i], c->xev[i + 1]) == hipSuccess) tot += ms;
  }
  out[0] = (double)c->n_calls;
  out[1] = (double)c->n_steps;
  out[2] = (double)c->n_exch;
  out[3] = c->bytes_sent;
  out[4] = (double)tot;
  out[5] = (double)effective_chunks(c);
  if (reset) {
    c->n_calls = c->n_steps = c->n_exch = 0;
    c->bytes_sent = 0.0;
    c->xev_used = 0;
    c->rev_used = 0;
    c->xtime = reset == 2;
  }
  return 0;
}
// milliseconds the exchange stream spent in all-reduces since timing was switched on (nq_slab_counters(reset = 2)), and
// how many were timed; read BEFORE the nq_slab_counters call that resets
int nq_slab_allreduce_ms(nq_ctx* c, double* out2) {
  if (!c || !out2) return -1;
  SLABTRY(nq_sync(c));
  float tot = 0.f;
  for (size_t i = 0; i + 1 < c->rev_used; i += 2) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->rev[i], c->rev[i + 1]) == hipSuccess) tot += ms;
  }
  out2[0] = (double)tot;
  out2[1] = (double)(c->rev_used / 2);
  return 0;
}


// ---- snapshots without stalling the stepper (ref niwqg/Saving.py:59-86: t, q, phi every tsave_snapshots steps) ---------
// nq_snapshot_begin: q = Re ifft(qh) (and phi = ifft(phih)) of the CURRENT state into device buffers of their own, on the
// context's stream; a second stream copies them to pinned host memory as soon as they are formed.  The caller may queue
// further steps at once.  nq_snapshot_end waits for that copy only and hands the arrays over.
int nq_snapshot_begin(nq_ctx* c, int with_phi) {
  NQ_SINGLE_RANK(c, "nq_snapshot_begin");
  if (with_phi && !c->kernel_family) NQ_FAIL(c, -4, "no wave field in QGModel");
  if (c->snap_busy) NQ_FAIL(c, -4, "nq_snapshot_begin: the previous snapshot has not been collected (nq_snapshot_end)");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t full = (size_t)c->N * c->N;
  // every resource is guarded by its own pointer: a call that failed half-way (a 4096^2 snapshot pins 128 + 256 MB of
  // host memory) leaves the rest to be made by the next call instead of running on null buffers
  if (!c->snap_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->snap_stream, hipStreamNonBlocking));
  if (!c->ev_snap) HIPCHK(c, hipEventCreateWithFlags(&c->ev_snap, hipEventDisableTiming));
  if (!c->snap_q) ALLOC(c, c->snap_q, full);
  if (!c->snap_hq) HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->snap_hq), sizeof(double) * full, hipHostMallocDefault));
  if (with_phi) {
    if (!c->snap_phi) ALLOC(c, c->snap_phi, full);
    if (!c->snap_hphi) HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->snap_hphi), sizeof(cd) * full, hipHostMallocDefault));
  }
  const cd* qh = c->q.y[c->q.cur];
  if (c->dual) {
    hipLaunchKernelGGL(k_avg_interior, dim3((c->Wh + 63) / 64, c->N), dim3(64), 0, c->stream, qh, (const cd*)c->q2.y[c->q2.cur], c->scr_f1, c->Wh, c->Ph, c->N);
    qh = c->scr_f1;
  }
  inv2d_half(c, qh, c->snap_q, c->scr_h0);
  if (with_phi) launch_x_c2c(c, true, c->mPhi.xs, c->snap_phi, c->mPhi.pitch, c->N, 1.0);
  HIPCHK(c, hipEventRecord(c->ev_snap, c->stream));
  HIPCHK(c, hipStreamWaitEvent(c->snap_stream, c->ev_snap, 0));
  HIPCHK(c, hipMemcpyAsync(c->snap_hq, c->snap_q, sizeof(double) * full, hipMemcpyDeviceToHost, c->snap_stream));
  if (with_phi) HIPCHK(c, hipMemcpyAsync(c->snap_hphi, c->snap_phi, sizeof(cd) * full, hipMemcpyDeviceToHost, c->snap_stream));
  c->snap_busy = true;
  HIPCHK(c, hipGetLastError());
  return 0;
}
int nq_snapshot_end(nq_ctx* c, double* q_out, double* phi_out) {
  NQ_SINGLE_RANK(c, "nq_snapshot_end");
  if (!c->snap_busy) NQ_FAIL(c, -4, "nq_snapshot_end: no snapshot in flight");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->snap_stream));
  const size_t full = (size_t)c->N * c->N;
  if (q_out) memcpy(q_out, c->snap_hq, sizeof(double) * full);
  if (phi_out) {
    if (!c->snap_hphi) NQ_FAIL(c, -4, "nq_snapshot_end: the snapshot was taken without phi");
    memcpy(phi_out, c->snap_hphi, sizeof(cd) * full);
  }
  c->snap_busy = false;
  return 0;
}

// host copies of the blocks that are summed over ranks (nq_reduce_buffer): what a callback link reduces
int nq_reduce_read(nq_ctx* c, int which, double* host_out) {
  if (!c || !host_out) return -1;
  int n = 0;
  double* p = red_ptr(c, which, &n);
  if (!p) NQ_FAIL(c, -1, "nq_reduce_read: block %d does not exist (yet)", which);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(host_out, p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  return nq_sync(c);
}
int nq_reduce_write(nq_ctx* c, int which, const double* host_in) {
  if (!c || !host_in) return -1;
  int n = 0;
  double* p = red_ptr(c, which, &n);
  if (!p) NQ_FAIL(c, -1, "nq_reduce_write: block %d does not exist (yet)", which);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(p, host_in, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  return nq_sync(c);
}

// ---- FFT seam ----------------------------------------------------------------------------------
int nq_fft2(nq_ctx* c, const double* in, double* out) {
  if (!c || !in || !out) return -1;
  NQ_SINGLE_RANK(c, "nq_fft2");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t full = (size_t)c->N * c->N;
  HIPCHK(c, hipMemcpyAsync(c->scr_f0, in, sizeof(cd) * full, hipMemcpyHostToDevice, c->stream));
  fwd2d_full(c, c->scr_f0, c->scr_f0, c->scr_f1);
  HIPCHK(c, hipMemcpyAsync(out, c->scr_f0, sizeof(cd) * full, hipMemcpyDeviceToHost, c->stream));
  return nq_sync(c);
}
int nq_ifft2(nq_ctx* c, const double* in, double* out) {
  if (!c || !in || !out) return -1;
  NQ_SINGLE_RANK(c, "nq_ifft2");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t full = (size_t)c->N * c->N;
  HIPCHK(c, hipMemcpyAsync(c->scr_f0, in, sizeof(cd) * full, hipMemcpyHostToDevice, c->stream));
  inv2d_full(c, c->scr_f0, c->scr_f0, c->scr_f1);
  HIPCHK(c, hipMemcpyAsync(out, c->scr_f0, sizeof(cd) * full, hipMemcpyDeviceToHost, c->stream));
  return nq_sync(c);
}
int nq_rfft2(nq_ctx* c, const double* in, double* out) {
  if (!c || !in || !out) return -1;
  NQ_SINGLE_RANK(c, "nq_rfft2");
  HIPCHK(c, hipSetDevice(c->device));
  const int N = c->N;
  HIPCHK(c, hipMemcpyAsync(c->scr_r, in, sizeof(double) * (size_t)N * N, hipMemcpyHostToDevice, c->stream));
  fwd2d_half(c, c->scr_r, c->scr_h1, c->scr_h0);
  HIPCHK(c, hipMemcpy2DAsync(out, sizeof(cd) * c->Wh, c->scr_h1, sizeof(cd) * c->Ph, sizeof(cd) * c->Wh, N, hipMemcpyDeviceToHost, c->stream));
  return nq_sync(c);
}
int nq_irfft2(nq_ctx* c, const double* in, double* out) {
  if (!c || !in || !out) return -1;
  NQ_SINGLE_RANK(c, "nq_irfft2");
  HIPCHK(c, hipSetDevice(c->device));
  const int N = c->N;
  HIPCHK(c, hipMemcpy2DAsync(c->scr_h1, sizeof(cd) * c->Ph, in, sizeof(cd) * c->Wh, sizeof(cd) * c->Wh, N, hipMemcpyHostToDevice, c->stream));
  inv2d_half(c, c->scr_h1, c->scr_r, c->scr_h0);
  HIPCHK(c, hipMemcpyAsync(out, c->scr_r, sizeof(double) * (size_t)N * N, hipMemcpyDeviceToHost, c->stream));
  return nq_sync(c);
}

// ---- downloads ---------------------------------------------------------------------------------
static int get_half_spec(nq_ctx* c, const cd* dev, double* host) {
  HIPCHK(c, hipMemcpy2DAsync(host, sizeof(cd) * c->Wh, dev, sizeof(cd) * c->Ph, sizeof(cd) * c->Wh, c->N, hipMemcpyDeviceToHost, c->stream));
  return nq_sync(c);
}
static int get_real_from_half(nq_ctx* c, const cd* spec, int mul_mode, double* host) {
  const size_t full = (size_t)c->N * c->N;
  const cd* src = spec;
  if (mul_mode) {
    hipLaunchKernelGGL(k_spec_mul, dim3((c->Wh + 63) / 64, c->N), dim3(64), 0, c->stream, spec, c->scr_h1, c->Wh, c->Ph, mul_mode, c->kk, c->ll, c->N, c->kernel_family ? 1 : 0);
    src = c->scr_h1;
  }
  inv2d_half(c, src, c->scr_r, c->scr_h0);
  HIPCHK(c, hipMemcpyAsync(host, c->scr_r, sizeof(double) * full, hipMemcpyDeviceToHost, c->stream));
  return nq_sync(c);
}

int nq_get_field(nq_ctx* c, int id, double* host) {
  if (!c || !host) return -1;
  NQ_SINGLE_RANK(c, "nq_get_field");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t full = (size_t)c->N * c->N;
  const cd* qh = c->q.y[c->q.cur];
  const bool waves = c->kernel_family;
  switch (id) {
    case NQ_F_QH: return get_half_spec(c, qh, host);
    case NQ_F_QH_MINUS:
      if (!c->dual) NQ_FAIL(c, -4, "NQ_F_QH_MINUS needs a dual_q context");
      return get_half_spec(c, c->q2.y[c->q2.cur], host);
    case NQ_F_PH: return get_half_spec(c, c->ph, host);
    case NQ_F_CH:
      if (!c->passive) NQ_FAIL(c, -4, "no passive scalar in this context");
      return get_half_spec(c, c->cq.y[c->cq.cur], host);
    case NQ_F_QH_STAGE4: return get_half_spec(c, c->q.y[(c->q.cur + 2) % 3], host);
    case NQ_F_QH_MINUS_STAGE4:
      if (!c->dual) NQ_FAIL(c, -4, "NQ_F_QH_MINUS_STAGE4 needs a dual_q context");
      return get_half_spec(c, c->q2.y[(c->q2.cur + 2) % 3], host);
    case NQ_F_PHIH_STAGE4:
      if (!waves) NQ_FAIL(c, -4, "no wave field in QGModel");
      HIPCHK(c, hipMemcpyAsync(host, c->w.y[(c->w.cur + 2) % 3], sizeof(cd) * full, hipMemcpyDeviceToHost, c->stream));
      return nq_sync(c);
    case NQ_F_QH_TICK: case NQ_F_QH_MINUS_TICK: case NQ_F_QWH_TICK: {
      const cd* src = id == NQ_F_QH_TICK ? c->tick_qh : (id == NQ_F_QH_MINUS_TICK ? c->tick_q2 : c->tick_qwh);
      if (!src) NQ_FAIL(c, -4, "field %d: no nq_tick_snapshot yet (or no such plane in this context)", id);
      return get_half_spec(c, src, host);
    }
    case NQ_F_PHIH_TICK:
      if (!c->tick_w) NQ_FAIL(c, -4, "NQ_F_PHIH_TICK: no nq_tick_snapshot yet (or no wave field)");
      HIPCHK(c, hipMemcpyAsync(host, c->tick_w, sizeof(cd) * full, hipMemcpyDeviceToHost, c->stream));
      return nq_sync(c);
    case NQ_F_C:
      if (!c->passive) NQ_FAIL(c, -4, "no passive scalar in this context");
      return get_real_from_half(c, c->cq.y[c->cq.cur], 0, host);
    case NQ_F_QWH:
      if (c->p.model != NQ_MODEL_COUPLED) NQ_FAIL(c, -4, "qwh exists only in the coupled model");
      return get_half_spec(c, c->qwh, host);
    case NQ_F_Q:
      if (c->dual) {
        hipLaunchKernelGGL(k_avg_interior, dim3((c->Wh + 63) / 64, c->N), dim3(64), 0, c->stream, qh, (const cd*)c->q2.y[c->q2.cur], c->scr_f1, c->Wh, c->Ph, c->N);
        return get_real_from_half(c, c->scr_f1, 0, host);
      }
      return get_real_from_half(c, qh, 0, host);
    case NQ_F_QPSI: {
      const cd* src = qh;
      if (c->dual) {
        hipLaunchKernelGGL(k_avg_interior, dim3((c->Wh + 63) / 64, c->N), dim3(64), 0, c->stream, qh, (const cd*)c->q2.y[c->q2.cur], c->scr_f1, c->Wh, c->Ph, c->N);
        src = c->scr_f1;
      }
      if (c->p.model == NQ_MODEL_COUPLED && !c->ybj) {
        hipLaunchKernelGGL(k_sub_half, dim3((c->Wh + 63) / 64, c->N), dim3(64), 0, c->stream, src, (const cd*)c->qwh, c->scr_f1, c->Wh, c->Ph);
        src = c->scr_f1;
      }
      return get_real_from_half(c, src, 0, host);
    }
    case NQ_F_P: return get_real_from_half(c, c->ph, 0, host);
    case NQ_F_U: return get_real_from_half(c, c->ph, 1, host);
    case NQ_F_V: {
      // v = Re ifft(ik psi): psi must be Hermitian-projected on the Nyquist column first;
      // equivalently its k = N/2 column does not contribute for the Kernel family.
      hipLaunchKernelGGL(k_spec_mul, dim3((c->Wh + 63) / 64, c->N), dim3(64), 0, c->stream, c->ph, c->scr_h1, c->Wh, c->Ph, 2, c->kk, c->ll, c->N, 1);
      if (c->kernel_family) HIPCHK(c, hipMemset2DAsync(c->scr_h1 + c->N / 2, sizeof(cd) * c->Ph, 0, sizeof(cd), c->N, c->stream));
      inv2d_half(c, c->scr_h1, c->scr_r, c->scr_h0);
      HIPCHK(c, hipMemcpyAsync(host, c->scr_r, sizeof(double) * full, hipMemcpyDeviceToHost, c->stream));
      return nq_sync(c);
    }
    case NQ_F_QW:
      if (c->p.model != NQ_MODEL_COUPLED) NQ_FAIL(c, -4, "qw exists only in the coupled model");
      return get_real_from_half(c, c->qwh, 0, host);
    case NQ_F_PHIH:
      if (!waves) NQ_FAIL(c, -4, "no wave field in QGModel");
      HIPCHK(c, hipMemcpyAsync(host, c->w.y[c->w.cur], sizeof(cd) * full, hipMemcpyDeviceToHost, c->stream));
      return nq_sync(c);
    case NQ_F_PHI:
      if (!waves) NQ_FAIL(c, -4, "no wave field in QGModel");
      launch_x_c2c(c, true, c->mPhi.xs, c->scr_f0, c->mPhi.pitch, c->N, 1.0);
      HIPCHK(c, hipMemcpyAsync(host, c->scr_f0, sizeof(cd) * full, hipMemcpyDeviceToHost, c->stream));
      return nq_sync(c);
    case NQ_F_PHIX:
    case NQ_F_PHIY: {
      if (!waves) NQ_FAIL(c, -4, "no wave field in QGModel");
      const bool unc = c->p.model == NQ_MODEL_UNCOUPLED;
      (void)unc;
      const MArr& src = (id == NQ_F_PHIX) ? c->mGx : c->mGy;       // == mPhi/mPhiy unless UnCoupled
      launch_x_c2c(c, true, src.xs, c->scr_f0, src.pitch, c->N, 1.0, id == NQ_F_PHIX ? 1 : 0);
      HIPCHK(c, hipMemcpyAsync(host, c->scr_f0, sizeof(cd) * full, hipMemcpyDeviceToHost, c->stream));
      return nq_sync(c);
    }
    default: NQ_FAIL(c, -1, "nq_get_field: unknown field id %d", id);
  }
}

// the anti-Hermitian passenger of qh on row l = N/2 (k_s_q): this context's local half-spectrum columns, zeros where there is none
int nq_get_qh_passenger(nq_ctx* c, double* out_cplx) {
  if (!c || !out_cplx) return -1;
  HIPCHK(c, hipSetDevice(c->device));
  memset(out_cplx, 0, sizeof(cd) * (size_t)(c->Wh > 0 ? c->Wh : 0));
  if (c->pass && c->Wh > 0) {
    HIPCHK(c, hipMemcpyAsync(out_cplx, c->qp.y[c->qp.cur], sizeof(cd) * (size_t)c->Wh, hipMemcpyDeviceToHost, c->stream));
    return nq_sync(c);
  }
  return 0;
}

int nq_get_scalar(nq_ctx* c, int id, double* out) {
  if (!c || !out) return -1;
  HIPCHK(c, hipSetDevice(c->device));
  const int N = c->N;
  const double M = (double)N * N;
  double* d = reinterpret_cast<double*>(c->scr_h0);    // 2 doubles of scratch for the reductions
  double h[2] = {0, 0};
  HIPCHK(c, hipMemsetAsync(d, 0, sizeof(double) * 2, c->stream));
  if (id == NQ_S_KE || id == NQ_S_PW || id == NQ_S_KW) {
    // increment accumulated by nq_step since the last read; reading resets it
    if (!c->bud) NQ_FAIL(c, -4, "budgets are disabled in this context");
    HIPCHK(c, hipMemcpyAsync(h, c->acc + id, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemsetAsync(c->acc + id, 0, sizeof(double), c->stream));
    int rc = nq_sync(c);
    *out = h[0];
    return rc;
  }
  NQ_SINGLE_RANK(c, "nq_get_scalar (ids other than the budget increments)");
  if (id == NQ_S_KE_QG) {
    hipLaunchKernelGGL(k_reduce, dim3((c->Wh + 255) / 256, N), dim3(256), 0, c->stream, c->ph, c->Wh, c->Ph, N, c->kernel_family ? 4 : 1, c->kk, c->ll, d);
    HIPCHK(c, hipMemcpyAsync(h, d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    int rc = nq_sync(c);
    *out = 0.5 * h[0] / (M * M);
    return rc;
  }
  if (!c->kernel_family && id == NQ_S_CFL) {
    const size_t full = (size_t)N * N;
    for (int which = 0; which < 2; ++which) {        // QGModel._calc_cfl: literal irfft2 (QGModel.py:621-629)
      hipLaunchKernelGGL(k_spec_mul, dim3((c->Wh + 63) / 64, N), dim3(64), 0, c->stream, c->ph, c->scr_h1, c->Wh, c->Ph, which == 0 ? 1 : 2, c->kk, c->ll, N, 0);
      inv2d_half(c, c->scr_h1, c->scr_r, c->scr_f1);
      hipLaunchKernelGGL(k_reduce_real_max, dim3(1024), dim3(256), 0, c->stream, c->scr_r, full, d);
    }
    HIPCHK(c, hipMemcpyAsync(h, d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    int rc = nq_sync(c);
    *out = h[0];
    return rc;
  }
  if (!c->kernel_family) NQ_FAIL(c, -4, "scalar %d needs the wave field", id);
  const cd* phih = c->w.y[c->w.cur];
  if (id == NQ_S_KE_NIW || id == NQ_S_PE_NIW) {
    hipLaunchKernelGGL(k_reduce, dim3((N + 255) / 256, N), dim3(256), 0, c->stream, phih, N, N, N, id == NQ_S_KE_NIW ? 0 : 2, c->kk, c->ll, d);
    HIPCHK(c, hipMemcpyAsync(h, d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    int rc = nq_sync(c);
    *out = (id == NQ_S_KE_NIW) ? 0.5 * h[0] / (M * M) : 0.25 * h[0] / (M * M) / c->p.kappa2;
    return rc;
  }
  if (id == NQ_S_CFL || id == NQ_S_MAX_PHI) {
    // max(|u|, |v|, |phi|) on the device (the caller multiplies by dt/dx): ref Kernel.py:660-662; NQ_S_MAX_PHI: max |phi| alone
    const size_t full = (size_t)N * N;
    for (int which = 0; which < (id == NQ_S_CFL ? 2 : 0); ++which) {
      hipLaunchKernelGGL(k_spec_mul, dim3((c->Wh + 63) / 64, N), dim3(64), 0, c->stream, c->ph, c->scr_h1, c->Wh, c->Ph, which == 0 ? 1 : 2, c->kk, c->ll, N, 1);
      if (which == 1 && c->kernel_family) HIPCHK(c, hipMemset2DAsync(c->scr_h1 + N / 2, sizeof(cd) * c->Ph, 0, sizeof(cd), N, c->stream));
      inv2d_half(c, c->scr_h1, c->scr_r, c->scr_f1);
      hipLaunchKernelGGL(k_reduce_real_max, dim3(1024), dim3(256), 0, c->stream, c->scr_r, full, d);
    }
    launch_x_c2c(c, true, c->mPhi.xs, c->scr_f0, c->mPhi.pitch, N, 1.0);
    hipLaunchKernelGGL(k_reduce, dim3((N + 255) / 256, N), dim3(256), 0, c->stream, c->scr_f0, N, N, N, 3, c->kk, c->ll, d);
    HIPCHK(c, hipMemcpyAsync(h, d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    int rc = nq_sync(c);
    *out = h[0];
    return rc;
  }
  NQ_FAIL(c, -1, "nq_get_scalar: id %d not available", id);
}

// overwrite the entries nq_coeff_patch recorded for public equation eq in one set of coefficient planes
static void apply_coeff_patch(nq_ctx* c, int eq, int width, int pitch, int k0, const double* filt, cd* const* coef, bool mirrored) {
  const nq_ctx::CoefPatch& pt = c->patch[eq];
  if (pt.n == 0 || width == 0) return;
  hipLaunchKernelGGL(k_coeff_patch, dim3((pt.n + 255) / 256), dim3(256), 0, c->stream, pt.n, pt.l, pt.k, pt.v, width, pitch, k0, filt,
                     coef[2], coef[3], coef[4], coef[5], mirrored ? c->N : 0);
}

int nq_get_coeff(nq_ctx* c, int eq, int which, double* out) {
  if (!c || !out || which < 0 || which > 5 || eq < 0 || eq > 2) return -1;
  NQ_SINGLE_RANK(c, "nq_get_coeff");
  if (eq == 1 && !c->kernel_family) NQ_FAIL(c, -4, "no phi equation in QGModel");
  if (eq == 2 && !c->passive) NQ_FAIL(c, -4, "no passive scalar in this context");
  HIPCHK(c, hipSetDevice(c->device));
  const int N = c->N;
  const bool half = eq != 1;
  const int width = half ? c->Wh : N, pitch = half ? c->Ph : N;
  const size_t cnt = (size_t)N * pitch;
  cd* tmp[6];
  void* blockp = nullptr;                               // one allocation: nothing to leak when it fails
  HIPCHK(c, hipMalloc(&blockp, 6 * cnt * sizeof(cd)));
  for (int i = 0; i < 6; ++i) tmp[i] = reinterpret_cast<cd*>(blockp) + (size_t)i * cnt;
  const int e = eq == 2 ? 3 : (half ? (c->kernel_family ? 0 : 2) : 1);
  hipLaunchKernelGGL(k_etdrk4_coeffs, dim3((width + 63) / 64, N), dim3(64), 0, c->stream, e, N, width, pitch, 0, c->p, c->kk, c->ll, (const double*)nullptr, c->contour, tmp[0], tmp[1], tmp[2], tmp[3], tmp[4], tmp[5]);
  apply_coeff_patch(c, eq, width, pitch, 0, nullptr, tmp, false);
  hipError_t er = hipMemcpy2DAsync(out, sizeof(cd) * width, tmp[which], sizeof(cd) * pitch, sizeof(cd) * width, N, hipMemcpyDeviceToHost, c->stream);
  int rc = nq_sync(c);
  (void)hipFree(blockp);
  if (er != hipSuccess) NQ_FAIL(c, -5, "nq_get_coeff: copy failed");
  return rc;
}

// public eq (0: q, 1: phi, 2: QGModel's passive scalar) -> k_etdrk4_coeffs' operator code, local width and first column
static int coeff_eq(nq_ctx* c, int eq, const char* what, int* code, int* width, int* k0) {
  if (!c) NQ_FAIL((nq_ctx*)nullptr, -1, "%s: null context", what);
  if (eq < 0 || eq > 2) NQ_FAIL(c, -1, "%s: eq %d (0: q, 1: phi, 2: passive scalar)", what, eq);
  if (eq == 1 && !c->kernel_family) NQ_FAIL(c, -4, "%s: no phi equation in QGModel", what);
  if (eq == 2 && !c->passive) NQ_FAIL(c, -4, "%s: no passive scalar in this context", what);
  *code = eq == 2 ? 3 : (eq == 1 ? 1 : (c->kernel_family ? 0 : 2));
  *width = eq == 1 ? c->Wf : c->Wh;
  *k0 = eq == 1 ? c->kf0 : c->kh0;
  return 0;
}

int nq_coeff_near_contour(nq_ctx* c, int eq, double delta, int cap, int* l_out, int* k_out) {
  int code, width, k0;
  if (int rc = coeff_eq(c, eq, "nq_coeff_near_contour", &code, &width, &k0)) return rc;
  if (!(delta >= 0.0) || cap < 0 || (cap > 0 && (!l_out || !k_out))) NQ_FAIL(c, -1, "nq_coeff_near_contour: bad delta / cap / outputs");
  HIPCHK(c, hipSetDevice(c->device));
  if (width == 0) return 0;
  int* d = nullptr;                                     // [count | l (cap) | k (cap)]
  HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d), sizeof(int) * (1 + 2 * (size_t)cap)));
  hipError_t er = hipMemsetAsync(d, 0, sizeof(int), c->stream);
  hipLaunchKernelGGL(k_coeff_flag, dim3((width + 63) / 64, c->N), dim3(64), 0, c->stream, code, c->N, width, k0, c->p, c->kk, c->ll,
                     c->contour, delta * delta, cap, d, d + 1, d + 1 + cap);
  int n = 0;
  if (er == hipSuccess) er = hipMemcpyAsync(&n, d, sizeof(int), hipMemcpyDeviceToHost, c->stream);
  if (er == hipSuccess) er = hipStreamSynchronize(c->stream);
  const int got = n < cap ? n : cap;
  if (er == hipSuccess && got > 0) er = hipMemcpy(l_out, d + 1, sizeof(int) * got, hipMemcpyDeviceToHost);
  if (er == hipSuccess && got > 0) er = hipMemcpy(k_out, d + 1 + cap, sizeof(int) * got, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (er != hipSuccess) NQ_FAIL(c, -5, "nq_coeff_near_contour: %s", hipGetErrorString(er));
  return n;
}

int nq_coeff_patch(nq_ctx* c, int eq, int n, const int* l, const int* k, const double* vals) {
  int code, width, k0;
  if (int rc = coeff_eq(c, eq, "nq_coeff_patch", &code, &width, &k0)) return rc;
  if (n < 0 || (n > 0 && (!l || !k || !vals))) NQ_FAIL(c, -1, "nq_coeff_patch: bad arguments");
  for (int i = 0; i < n; ++i)
    if (l[i] < 0 || l[i] >= c->N || k[i] < k0 || k[i] >= k0 + width)
      NQ_FAIL(c, -1, "nq_coeff_patch: entry %d = (l %d, k %d) outside this context's columns [%d, %d)", i, l[i], k[i], k0, k0 + width);
  HIPCHK(c, hipSetDevice(c->device));
  nq_ctx::CoefPatch& pt = c->patch[eq];
  HIPCHK(c, hipStreamSynchronize(c->stream));           // a previous patch may still be read by a queued kernel
  (void)hipFree(pt.l); (void)hipFree(pt.k); (void)hipFree(pt.v);
  pt = nq_ctx::CoefPatch();
  if (n == 0) return 0;
  HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&pt.l), sizeof(int) * n));
  HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&pt.k), sizeof(int) * n));
  HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&pt.v), sizeof(cd) * 4 * (size_t)n));
  HIPCHK(c, hipMemcpy(pt.l, l, sizeof(int) * n, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(pt.k, k, sizeof(int) * n, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(pt.v, vals, sizeof(cd) * 4 * (size_t)n, hipMemcpyHostToDevice));
  pt.n = n;
  if (eq == 0) {
    apply_coeff_patch(c, 0, c->Wh, c->Ph, c->kh0, c->filt_h, c->q.coef, c->cmirror != 0);
    if (c->dual) apply_coeff_patch(c, 0, c->Wh, c->Ph, c->kh0, nullptr, c->coefu, c->cmirror != 0);
  } else if (eq == 1) {
    apply_coeff_patch(c, 1, c->Wf, c->Wf, c->kf0, c->filt_f, c->w.coef, c->cmirror != 0);
  } else {
    apply_coeff_patch(c, 2, c->Wh, c->Ph, c->kh0, c->filt_h, c->cq.coef, c->cmirror != 0);
  }
  return nq_sync(c);
}

// ---- the three Jacobians of the public API, in the reference's own array layouts ------------------------------
// F[u q], F[v q] of the current state as two half-spectrum planes in scr_h0, scr_h1
static void products_to_scratch(nq_ctx* c) {
  launch_products(c);
  launch_A_m(c, false, {&c->mUq, &c->mVq});
  launch_B_p(c, false, c->mUq.ys, c->mUq.pitch, c->scr_h0, c->Ph, c->Wh, 1.0);
  launch_B_p(c, false, c->mVq.ys, c->mVq.pitch, c->scr_h1, c->Ph, c->Wh, 1.0);
}
// Kernel.jacobian_psi_q (ref Kernel.py:471-486): ik*fft(u q) + il*fft(v q) as the full (ny, nx) plane, [0,0] = 0;
// QGModel.jacobian_psi_q (ref QGModel.py:469-481): the same on (ny, nx/2+1), [0,0] kept.
int nq_jacobian_psi_q(nq_ctx* c, double* out_cplx) {
  NQ_SINGLE_RANK(c, "nq_jacobian_psi_q");
  if (!out_cplx) NQ_FAIL(c, -1, "nq_jacobian_psi_q: null output");
  HIPCHK(c, hipSetDevice(c->device));
  products_to_scratch(c);
  const int N = c->N, wout = c->kernel_family ? N : c->WhG;
  hipLaunchKernelGGL(k_expand_half, dim3((wout + 63) / 64, N), dim3(64), 0, c->stream, (const cd*)c->scr_h0, (const cd*)c->scr_h1, c->scr_f0, N, c->Ph, wout, 1, c->kk, c->ll);
  if (c->kernel_family) HIPCHK(c, hipMemsetAsync(c->scr_f0, 0, sizeof(cd), c->stream));
  HIPCHK(c, hipMemcpyAsync(out_cplx, c->scr_f0, sizeof(cd) * (size_t)N * wout, hipMemcpyDeviceToHost, c->stream));
  return nq_sync(c);
}
// QGModel.jacobian_psi_c (ref QGModel.py:483-495): ik*fft(u c) + il*fft(v c) on (ny, nx/2+1), with u, v of the current psi;
// formed by the row kernel exactly as inside a step (the scalar rides paired with q, its products leave as a second pair)
int nq_jacobian_psi_c(nq_ctx* c, double* out_cplx) {
  NQ_SINGLE_RANK(c, "nq_jacobian_psi_c");
  if (!out_cplx) NQ_FAIL(c, -1, "nq_jacobian_psi_c: null output");
  if (!c->passive) NQ_FAIL(c, -4, "nq_jacobian_psi_c: this context has no passive scalar");
  HIPCHK(c, hipSetDevice(c->device));
  launch_products(c);
  launch_A_m(c, false, {&c->mUc, &c->mVc});
  launch_B_p(c, false, c->mUc.ys, c->mUc.pitch, c->scr_h0, c->Ph, c->Wh, 1.0);
  launch_B_p(c, false, c->mVc.ys, c->mVc.pitch, c->scr_h1, c->Ph, c->Wh, 1.0);
  const int N = c->N, wout = c->WhG;
  hipLaunchKernelGGL(k_expand_half, dim3((wout + 63) / 64, N), dim3(64), 0, c->stream, (const cd*)c->scr_h0, (const cd*)c->scr_h1, c->scr_f0, N, c->Ph, wout, 1, c->kk, c->ll);
  HIPCHK(c, hipMemcpyAsync(out_cplx, c->scr_f0, sizeof(cd) * (size_t)N * wout, hipMemcpyDeviceToHost, c->stream));
  return nq_sync(c);
}
// the two transforms themselves, (2, ny, nx/2+1): fft(u q) then fft(v q) restricted to k = 0..nx/2 (no reference
// counterpart: exported for tests and for callers that assemble their own flux forms)
int nq_products_uq_vq(nq_ctx* c, double* out_cplx) {
  NQ_SINGLE_RANK(c, "nq_products_uq_vq");
  if (!out_cplx) NQ_FAIL(c, -1, "nq_products_uq_vq: null output");
  HIPCHK(c, hipSetDevice(c->device));
  products_to_scratch(c);
  int rc = get_half_spec(c, c->scr_h0, out_cplx);
  if (rc) return rc;
  return get_half_spec(c, c->scr_h1, out_cplx + 2 * (size_t)c->N * c->Wh);
}
// Kernel.jacobian_psi_phi (ref Kernel.py:457-469): fft(u phix + v phiy), (ny, nx), [0,0] = 0.  YBJModel's own version
// (ref YBJModel.py:123-133) keeps [0,0], and so does a YBJ context.
int nq_jacobian_psi_phi(nq_ctx* c, double* out_cplx) {
  NQ_SINGLE_RANK(c, "nq_jacobian_psi_phi");
  if (!out_cplx) NQ_FAIL(c, -1, "nq_jacobian_psi_phi: null output");
  if (!c->kernel_family) NQ_FAIL(c, -4, "no wave field in QGModel");
  HIPCHK(c, hipSetDevice(c->device));
  launch_products(c, 1.0, 0.0);                       // the Jacobian part alone
  launch_A_m(c, false, {&c->mW});
  launch_B_p(c, false, c->mW.ys, c->mW.pitch, c->scr_f0, c->N, c->N, 1.0);
  if (!c->ybj) HIPCHK(c, hipMemsetAsync(c->scr_f0, 0, sizeof(cd), c->stream));
  HIPCHK(c, hipMemcpyAsync(out_cplx, c->scr_f0, sizeof(cd) * (size_t)c->N * c->N, hipMemcpyDeviceToHost, c->stream));
  return nq_sync(c);
}
// ---- diagnostics tick on the device (ref Diagnostics.py:41-58, Kernel.py:613-706, :718-868, CoupledModel.py:99-136) --
int nq_diagnostics(nq_ctx* c, double* out) {
  NQ_SINGLE_RANK(c, "nq_diagnostics");
  if (!out) NQ_FAIL(c, -1, "nq_diagnostics: null output");
  HIPCHK(c, hipSetDevice(c->device));
  const int N = c->N, NB = 1024;
  const double M = (double)N * N;
  const bool waves = c->kernel_family;
  if (waves && !c->have_phi) NQ_FAIL(c, -4, "nq_diagnostics: set_phi has not been called");
  const int nxb = xdiag_blocks(c), nww = (c->Wf / c->CLy) * c->S2;
  if (!c->diag_part) {
    size_t need = (size_t)NB * 9;
    if ((size_t)nxb * 8 > need) need = (size_t)nxb * 8;
    if ((size_t)nww * 4 > need) need = (size_t)nww * 4;
    ALLOC(c, c->diag_part, need);
    ALLOC(c, c->diag_out, (size_t)40);
  }
  double* d = c->diag_out;
  HIPCHK(c, hipMemsetAsync(d, 0, sizeof(double) * 32, c->stream));
  const cd* qh = c->q.y[c->q.cur];
  if (c->dual) {                                      // physical space sees the mean of the two copies
    hipLaunchKernelGGL(k_avg_interior, dim3((c->Wh + 63) / 64, N), dim3(64), 0, c->stream, qh, (const cd*)c->q2.y[c->q2.cur], c->scr_f1, c->Wh, c->Ph, N);
    qh = c->scr_f1;
  }
  const cd* phih = waves ? c->w.y[c->w.cur] : nullptr;
  // spectral sums: [0,4) S0..S3, [4,6) phih[0,0], [6,15) the nine half-spectrum sums, [15] Re(qh - qwh)[0,0]
  if (waves) {
    hipLaunchKernelGGL(k_diag_phi, dim3(NB), dim3(256), 0, c->stream, phih, N, c->Wf, c->Wf, c->kf0, c->kk, c->ll, c->diag_part);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, c->stream, c->diag_part, NB, 4, 4, d);
    HIPCHK(c, hipMemcpyAsync(d + 4, phih, sizeof(cd), hipMemcpyDeviceToDevice, c->stream));
  }
  hipLaunchKernelGGL(k_diag_q, dim3(NB), dim3(256), 0, c->stream, qh, (const cd*)(c->p.model == NQ_MODEL_COUPLED ? c->qwh : nullptr), (const cd*)c->ph, N, c->Wh, c->Ph, 0, c->kk, c->ll, c->diag_part,
                     (const cd*)(c->dual ? c->q.y[c->q.cur] : nullptr), (const cd*)(c->dual ? c->q2.y[c->q2.cur] : nullptr),
                     (const double*)(c->dual ? c->filt_h : nullptr), (const double*)(c->dual ? c->filt_m : nullptr),
                     (const cd*)(c->pass ? c->qp.y[c->qp.cur] : nullptr));
  hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, c->stream, c->diag_part, NB, 9, 9, d + 6);
  double h[32];
  HIPCHK(c, hipMemcpyAsync(h, d, sizeof(double) * 32, hipMemcpyDeviceToHost, c->stream));
  cd q00 = make_double2(0.0, 0.0), w00 = make_double2(0.0, 0.0);
  HIPCHK(c, hipMemcpyAsync(&q00, qh, sizeof(cd), hipMemcpyDeviceToHost, c->stream));
  if (c->p.model == NQ_MODEL_COUPLED) HIPCHK(c, hipMemcpyAsync(&w00, c->qwh, sizeof(cd), hipMemcpyDeviceToHost, c->stream));
  {
    const int rc = nq_sync(c);
    if (rc) return rc;
  }
  for (int i = 0; i < 15; ++i) out[i] = h[i];
  for (int i = 15; i < 32; ++i) out[i] = 0.0;
  const double qbar = (q00.x - w00.x) / M, abar = h[0] / (M * M);
  out[15] = qbar;
  if (!waves && c->passive) {
    // QGModel's passive scalar (ref QGModel.py:724-737, :595-604): [16..19] the four |c-hat|^2 sums, [20] the Gamma_c projection
    const cd* ch = c->cq.y[c->cq.cur];
    hipLaunchKernelGGL(k_diag_c, dim3(NB), dim3(256), 0, c->stream, ch, N, c->Wh, c->Ph, 0, c->kk, c->ll, c->diag_part);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, c->stream, c->diag_part, NB, 4, 4, d + 16);
    // jacobian_psi_c with the u, v the reference still holds at a tick: those of the state at which the last step evaluated
    // its fourth stage (QGModel.py:375 vs :396); before any step, those of the current state
    const cd* qh4 = c->stepped ? c->q.y[(c->q.cur + 2) % 3] : c->q.y[c->q.cur];
    phase_invert_y(c, qh4, false, c->part0Q, nullptr, ch);
    launch_products(c);
    launch_A_m(c, false, {&c->mUc, &c->mVc});
    const YGeom g = geom_half(c);
    const int nwc = ((g.width + c->CLy - 1) / c->CLy) * c->S2;
#define CALL_(sz, clx) hipLaunchKernelGGL((k_s_project_c<sz, clx>), dim3((g.width + clx - 1) / clx, c->S2), dim3((YPlanT<sz, clx>::THREADS)), (YPlanT<sz, clx>::LDS_BYTES), c->stream, c->mUc, c->mVc, ch, g, c->kk, c->ll, c->tw, 1, c->diag_part)
    NQ_S1_SWITCH(c, CALL_)
#undef CALL_
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, c->stream, c->diag_part, nwc, 1, 1, d + 20);
    do_invert_now(c);                                 // the mixed-space rows of the CURRENT state again, for the next step
    HIPCHK(c, hipMemcpyAsync(out + 16, d + 16, sizeof(double) * 5, hipMemcpyDeviceToHost, c->stream));
    return nq_sync(c);
  }
  if (!waves) return 0;
  // physical-space statistics: [16,24)
  if (c->p.model == NQ_MODEL_COUPLED) launch_xdiag_m<MODE_COUPLED>(c, qbar, abar, c->diag_part);
  else launch_xdiag_m<MODE_UNCOUPLED>(c, qbar, abar, c->diag_part);
  hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, c->stream, c->diag_part, nxb, 8, 8, d + 16);
  // projections of F[u phix + v phiy] ([24,28)) and of i F[phi q_psi] ([28,32)) on lap_h and diss_h; u, v, q_psi are
  // those of the last inversion, phix / phiy as last refreshed (UnCoupled: quirk Q1), like ref Kernel.py:680-700
  for (int which = 0; which < 2; ++which) {
    launch_products(c, which == 0 ? 1.0 : 0.0, which == 0 ? 0.0 : 1.0);
    launch_A_m(c, false, {&c->mW});
    launch_project(c, c->diag_part);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, c->stream, c->diag_part, nww, 4, 4, d + 24 + 4 * which);
  }
  HIPCHK(c, hipMemcpyAsync(out + 16, d + 16, sizeof(double) * 16, hipMemcpyDeviceToHost, c->stream));
  return nq_sync(c);
}

// fft(phi * q_psi), (ny, nx): the refraction source of ref Kernel.py:332, :350, :367, :385 before its -0.5j factor,
// formed by the row kernel exactly as inside a step (mean NOT removed)
int nq_refraction(nq_ctx* c, double* out_cplx) {
  NQ_SINGLE_RANK(c, "nq_refraction");
  if (!out_cplx) NQ_FAIL(c, -1, "nq_refraction: null output");
  if (!c->kernel_family) NQ_FAIL(c, -4, "no wave field in QGModel");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t full = (size_t)c->N * c->N;
  launch_products(c, 0.0, 1.0);                       // W = i phi q_psi
  launch_A_m(c, false, {&c->mW});
  launch_B_p(c, false, c->mW.ys, c->mW.pitch, c->scr_f0, c->N, c->N, 1.0);
  hipLaunchKernelGGL(k_mul_minus_i, dim3((unsigned)((full + 255) / 256)), dim3(256), 0, c->stream, c->scr_f0, full);
  HIPCHK(c, hipMemcpyAsync(out_cplx, c->scr_f0, sizeof(cd) * full, hipMemcpyDeviceToHost, c->stream));
  return nq_sync(c);
}

// CoupledModel.jacobian_phic_phi (ref CoupledModel.py:59-73): fft(Re i(phix* phiy - phiy* phix)), (ny, nx), [0,0] = 0
int nq_jacobian_phic_phi(nq_ctx* c, double* out_cplx) {
  NQ_SINGLE_RANK(c, "nq_jacobian_phic_phi");
  if (!out_cplx) NQ_FAIL(c, -1, "nq_jacobian_phic_phi: null output");
  if (c->p.model != NQ_MODEL_COUPLED || c->ybj) NQ_FAIL(c, -4, "jacobian_phic_phi exists only in the coupled model");
  HIPCHK(c, hipSetDevice(c->device));
  const int N = c->N;
  launch_wavepv(c);
  launch_A_m(c, false, {&c->mB});
  launch_B_p(c, false, c->mB.ys, c->mB.pitch, c->scr_h0, c->Ph, c->Wh, 1.0);
  hipLaunchKernelGGL(k_expand_half, dim3((N + 63) / 64, N), dim3(64), 0, c->stream, (const cd*)c->scr_h0, (const cd*)nullptr, c->scr_f0, N, c->Ph, N, 0, c->kk, c->ll);
  HIPCHK(c, hipMemsetAsync(c->scr_f0, 0, sizeof(cd), c->stream));
  HIPCHK(c, hipMemcpyAsync(out_cplx, c->scr_f0, sizeof(cd) * (size_t)N * N, hipMemcpyDeviceToHost, c->stream));
  return nq_sync(c);
}

// number of DOUBLES nq_get_field(field_id) writes for this context (-1: unknown id)
long long nq_field_doubles(const nq_ctx* c, int id) {
  if (!c) return -1;
  const long long n = c->N, h = c->N / 2 + 1;
  switch (id) {
    case NQ_F_Q: case NQ_F_P: case NQ_F_U: case NQ_F_V: case NQ_F_QPSI: case NQ_F_QW: case NQ_F_C: return n * n;
    case NQ_F_QH: case NQ_F_PH: case NQ_F_QWH: case NQ_F_QH_MINUS: case NQ_F_CH: case NQ_F_QH_STAGE4: case NQ_F_QH_MINUS_STAGE4:
    case NQ_F_QH_TICK: case NQ_F_QH_MINUS_TICK: case NQ_F_QWH_TICK: return 2 * n * h;
    case NQ_F_PHI: case NQ_F_PHIH: case NQ_F_PHIX: case NQ_F_PHIY: case NQ_F_PHIH_STAGE4: case NQ_F_PHIH_TICK: return 2 * n * n;
    default: return -1;
  }
}

}  // extern "C"

// ==================================================================================================================
// The any-size engine (include/niwqg_amd.h: nq_any_*; csrc/nq_anysize.hpp; niwqg_amd/_anysize.py)
// ==================================================================================================================
#define ANYCHK(e, call)                                                                                      \
  do {                                                                                                       \
    hipError_t e_ = (call);                                                                                  \
    if (e_ != hipSuccess) {                                                                                  \
      char b_[384];                                                                                          \
      snprintf(b_, sizeof(b_), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);   \
      g_last_error = b_;                                                                                     \
      if (e) (e)->err = b_;                                                                                  \
      return -5;                                                                                             \
    }                                                                                                        \
  } while (0)
#define ANYFAIL(e, code, ...)                      \
  do {                                             \
    char b_[384];                                  \
    snprintf(b_, sizeof(b_), __VA_ARGS__);         \
    g_last_error = b_;                             \
    if (e) (e)->err = b_;                          \
    return (code);                                 \
  } while (0)

template <bool INV>
static int any_rows_fft(nq_any* e, const nq_any::Plan& pl, int nlines, double scale) {
  if (pl.M == 16384) {
    // four-step transform of rows of 128 x 128 points (csrc/nq_anysize.hpp: k_any_btranspose); result back in e->tmp
    const int A = 128, B = 128;
    typedef XPlan<128> X;
    const dim3 tg(A / 16, B / 16, nlines), tb(256);
    const int rows128 = nlines * 128;
    hipLaunchKernelGGL(k_any_btranspose, tg, tb, 0, e->stream, (const cd*)e->tmp, e->tmp2, B, A, (const cd*)nullptr, 0, 0);
    hipLaunchKernelGGL((k_x_c2c<128, INV>), dim3((rows128 + X::C - 1) / X::C), dim3(X::THREADS), X::LDS_BYTES, e->stream, (const cd*)e->tmp2, e->tmp2, 128, 128,
                       rows128, 1.0, (const cd*)pl.tw_small, (const double*)nullptr, 0);
    hipLaunchKernelGGL(k_any_btranspose, tg, tb, 0, e->stream, (const cd*)e->tmp2, e->tmp, A, B, (const cd*)pl.tw, 1, INV ? 1 : 0);
    hipLaunchKernelGGL((k_x_c2c<128, INV>), dim3((rows128 + X::C - 1) / X::C), dim3(X::THREADS), X::LDS_BYTES, e->stream, (const cd*)e->tmp, e->tmp, 128, 128,
                       rows128, scale, (const cd*)pl.tw_small, (const double*)nullptr, 0);
    hipLaunchKernelGGL(k_any_btranspose, tg, tb, 0, e->stream, (const cd*)e->tmp, e->tmp2, B, A, (const cd*)nullptr, 0, 0);
    std::swap(e->tmp, e->tmp2);
    std::swap(e->tmp_elems, e->tmp2_elems);
    return 0;
  }
  switch (pl.M) {
#define CASE_(n, a, b) case n: { typedef XPlan<n> X; \
      hipLaunchKernelGGL((k_x_c2c<n, INV>), dim3((nlines + X::C - 1) / X::C), dim3(X::THREADS), X::LDS_BYTES, e->stream, \
                         (const cd*)e->tmp, e->tmp, pl.M, pl.M, nlines, scale, (const cd*)pl.tw, (const double*)nullptr, 0); } break;
    NQ_FOR_SIZES(CASE_)
#undef CASE_
    default: ANYFAIL(e, -2, "any-size engine: no row plan of length %d", pl.M);
  }
  return 0;
}
// the fused Bluestein row kernel on `nlines` contiguous rows of `n` values (pitch n), in place or src -> dst
static int any_bluestein_rows(nq_any* e, const nq_any::Plan& pl, const cd* src, cd* dst, int nlines, int n, int inverse) {
  const double scale = (inverse ? 1.0 / (double)n : 1.0) / (double)pl.M;
  switch (pl.M) {
#define CASE_(m, a, b) case m: { typedef XPlan<m> X; \
      hipLaunchKernelGGL((k_any_bluestein_rows<m>), dim3((nlines + X::C - 1) / X::C), dim3(X::THREADS), X::LDS_BYTES, e->stream, src, dst, nlines, n, n, \
                         (const cd*)pl.chirp, (const cd*)pl.bhat, (const cd*)pl.tw, inverse ? 1 : 0, scale); } break;
    NQ_FOR_SIZES(CASE_)
#undef CASE_
    default: ANYFAIL(e, -2, "any-size engine: no fused row plan of length %d", pl.M);
  }
  return 0;
}
template <int R>
static int any_split_rows(nq_any* e, const nq_any::Plan& pl, const cd* src, cd* dst, int nlines, int n, int inverse) {
  const double scale = inverse ? 1.0 / (double)n : 1.0;
  switch (pl.M) {
#define CASE_(m, a, b) case m: { typedef XPlan<m> X; \
      hipLaunchKernelGGL((k_any_split_rows<m, R>), dim3((nlines + X::C - 1) / X::C), dim3(X::THREADS), X::LDS_BYTES, e->stream, src, dst, nlines, n, \
                         (const cd*)pl.chirp, (const cd*)pl.tw, inverse ? 1 : 0, scale); } break;
    M_SMALL(CASE_)
#undef CASE_
    default: ANYFAIL(e, -2, "any-size engine: no split row plan of length %d x %d", R, pl.M);
  }
  return 0;
}
// device temporaries of one call: freed when the call returns, on the error paths too
struct AnyScratch {
  std::vector<void*> held;
  ~AnyScratch() {
    for (void* q : held) (void)hipFree(q);
  }
  template <typename T>
  hipError_t get(T** out, size_t count) {
    void* q = nullptr;
    const hipError_t r = hipMalloc(&q, count * sizeof(T));
    if (r == hipSuccess) held.push_back(q);
    *out = static_cast<T*>(q);
    return r;
  }
};
static int any_buf(nq_any* e, cd** buf, size_t* have, size_t elems) {
  if (*have >= elems) return 0;
  if (*buf) {
    ANYCHK(e, hipStreamSynchronize(e->stream));
    ANYCHK(e, hipFree(*buf));
    *buf = nullptr;
    *have = 0;
  }
  ANYCHK(e, hipMalloc(reinterpret_cast<void**>(buf), elems * sizeof(cd)));
  *have = elems;
  return 0;
}
static int any_tmp(nq_any* e, size_t elems, bool four_step = false) {
  int rc = any_buf(e, &e->tmp, &e->tmp_elems, elems);
  if (rc == 0 && four_step) rc = any_buf(e, &e->tmp2, &e->tmp2_elems, elems);
  return rc;
}
// Bluestein plan for transforms of length n: chirp w[j] = exp(-i pi j^2 / n) (angle reduced exactly: j^2 mod 2n), the transform of
// conj(w) laid out circularly on M >= 2n - 1 points, and the twiddle table of the M-point row engine
static int any_plan(nq_any* e, int n, const nq_any::Plan** out) {
  for (const nq_any::Plan& p : e->plans)
    if (p.n == n) { *out = &p; return 0; }
  const bool pow2 = n >= 64 && (n & (n - 1)) == 0;
  if (n < 2 || (pow2 ? n > 16384 : n > 8192)) ANYFAIL(e, -2, "any-size engine: transform length %d (any n in [2, 8192], or 16384)", n);
  nq_any::Plan pl;
  pl.n = n;
  pl.direct = pow2;
  pl.M = 64;
  static int use_split = -1;
  if (use_split < 0) {
    const char* ev = getenv("NIWQG_AMD_ANY_SPLIT");              // 0: Bluestein for 3 m and 5 m as well (A/B measurements)
    use_split = (ev && atoi(ev) == 0) ? 0 : 1;
  }
  for (int R : {3, 5}) {
    const int m = n / R;
    if (use_split && !pow2 && n % R == 0 && m >= 64 && m <= 2048 && (m & (m - 1)) == 0) {
      pl.split = R;
      pl.M = m;
      break;
    }
  }
  if (pow2) pl.M = n;
  else if (!pl.split)
    while (pl.M < 2 * n - 1) pl.M *= 2;
  const int M = pl.M;
  const long double pi = 3.14159265358979323846264338327950288L;
  std::vector<double> tw(2 * (size_t)M);
  for (int m = 0; m < M; ++m) {
    const long double a = -2.0L * pi * (long double)m / (long double)M;
    tw[2 * m] = (double)cosl(a);
    tw[2 * m + 1] = (double)sinl(a);
  }
  ANYCHK(e, hipMalloc(reinterpret_cast<void**>(&pl.tw), sizeof(cd) * M));
  ANYCHK(e, hipMemcpy(pl.tw, tw.data(), sizeof(cd) * M, hipMemcpyHostToDevice));
  if (M == 16384) {
    std::vector<double> ts(2 * 128);
    for (int m = 0; m < 128; ++m) {
      const long double a = -2.0L * pi * (long double)m / 128.0L;
      ts[2 * m] = (double)cosl(a);
      ts[2 * m + 1] = (double)sinl(a);
    }
    ANYCHK(e, hipMalloc(reinterpret_cast<void**>(&pl.tw_small), sizeof(cd) * 128));
    ANYCHK(e, hipMemcpy(pl.tw_small, ts.data(), sizeof(cd) * 128, hipMemcpyHostToDevice));
  }
  if (pl.split) {
    // chirp slot: exp(-2 pi i q / n), q < n -- the twiddles of the radix-R combination
    std::vector<double> wn(2 * (size_t)n);
    for (int q = 0; q < n; ++q) {
      const long double a = -2.0L * pi * (long double)q / (long double)n;
      wn[2 * q] = (double)cosl(a);
      wn[2 * q + 1] = (double)sinl(a);
    }
    ANYCHK(e, hipMalloc(reinterpret_cast<void**>(&pl.chirp), sizeof(cd) * n));
    ANYCHK(e, hipMemcpy(pl.chirp, wn.data(), sizeof(cd) * n, hipMemcpyHostToDevice));
  } else if (!pl.direct) {
    std::vector<double> w(2 * (size_t)n), b(2 * (size_t)M, 0.0);
    for (int j = 0; j < n; ++j) {
      const long long r = ((long long)j * j) % (2LL * n);
      const long double a = pi * (long double)r / (long double)n;
      w[2 * j] = (double)cosl(a);
      w[2 * j + 1] = (double)(-sinl(a));
      // conj(w[j]) at +j and -j (circular)
      b[2 * j] = (double)cosl(a);
      b[2 * j + 1] = (double)sinl(a);
      if (j > 0) {
        b[2 * (size_t)(M - j)] = (double)cosl(a);
        b[2 * (size_t)(M - j) + 1] = (double)sinl(a);
      }
    }
    ANYCHK(e, hipMalloc(reinterpret_cast<void**>(&pl.chirp), sizeof(cd) * n));
    ANYCHK(e, hipMalloc(reinterpret_cast<void**>(&pl.bhat), sizeof(cd) * M));
    ANYCHK(e, hipMemcpy(pl.chirp, w.data(), sizeof(cd) * n, hipMemcpyHostToDevice));
    int rc = any_tmp(e, (size_t)M, M == 16384);
    if (rc) return rc;
    ANYCHK(e, hipMemcpy(e->tmp, b.data(), sizeof(cd) * M, hipMemcpyHostToDevice));
    rc = any_rows_fft<false>(e, pl, 1, 1.0);
    if (rc) return rc;
    ANYCHK(e, hipMemcpyAsync(pl.bhat, e->tmp, sizeof(cd) * M, hipMemcpyDeviceToDevice, e->stream));
    ANYCHK(e, hipStreamSynchronize(e->stream));
  }
  e->plans.push_back(pl);
  *out = &e->plans.back();
  return 0;
}

extern "C" {

int nq_any_create(int device, nq_any** out) {
  if (!out) return -1;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) NQ_FAIL((nq_ctx*)nullptr, -3, "nq_any_create: no HIP device available");
  if (device < 0 || device >= ndev) NQ_FAIL((nq_ctx*)nullptr, -3, "nq_any_create: device %d out of range (%d devices)", device, ndev);
  nq_any* e = new nq_any();
  e->device = device;
  e->plans.reserve(16);
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&e->part), sizeof(double) * 2 * 1024) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&e->red), sizeof(double) * 2) != hipSuccess) {
    delete e;
    NQ_FAIL((nq_ctx*)nullptr, -5, "nq_any_create: stream / scratch allocation failed");
  }
  *out = e;
  return 0;
}
int nq_any_destroy(nq_any* e) {
  if (!e) return -1;
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  for (void* p : e->allocs) (void)hipFree(p);
  for (nq_any::Plan& p : e->plans) {
    if (p.chirp) (void)hipFree(p.chirp);
    if (p.bhat) (void)hipFree(p.bhat);
    if (p.tw_small) (void)hipFree(p.tw_small);
    (void)hipFree(p.tw);
  }
  if (e->tmp) (void)hipFree(e->tmp);
  if (e->tmp2) (void)hipFree(e->tmp2);
  (void)hipFree(e->part);
  (void)hipFree(e->red);
  (void)hipStreamDestroy(e->stream);
  delete e;
  return 0;
}
const char* nq_any_last_error(const nq_any* e) { return e ? e->err.c_str() : g_last_error.c_str(); }
int nq_any_sync(nq_any* e) {
  if (!e) return -1;
  ANYCHK(e, hipStreamSynchronize(e->stream));
  ANYCHK(e, hipGetLastError());
  return 0;
}
long long nq_any_device_bytes(const nq_any* e) { return e ? e->bytes + (long long)((e->tmp_elems + e->tmp2_elems) * sizeof(cd)) : 0; }
int nq_any_alloc(nq_any* e, long long elems, void** plane) {
  if (!e || !plane || elems <= 0) return -1;
  ANYCHK(e, hipSetDevice(e->device));
  void* p = nullptr;
  ANYCHK(e, hipMalloc(&p, (size_t)elems * sizeof(cd)));
  ANYCHK(e, hipMemsetAsync(p, 0, (size_t)elems * sizeof(cd), e->stream));
  e->allocs.push_back(p);
  e->bytes += elems * (long long)sizeof(cd);
  *plane = p;
  return 0;
}
int nq_any_free(nq_any* e, void* plane, long long elems) {
  if (!e || !plane) return -1;
  for (size_t i = 0; i < e->allocs.size(); ++i)
    if (e->allocs[i] == plane) {
      ANYCHK(e, hipStreamSynchronize(e->stream));
      ANYCHK(e, hipFree(plane));
      e->allocs.erase(e->allocs.begin() + i);
      e->bytes -= elems * (long long)sizeof(cd);
      return 0;
    }
  ANYFAIL(e, -1, "nq_any_free: not a plane of this engine");
}
int nq_any_upload(nq_any* e, void* plane, const double* host_cplx, long long elems) {
  if (!e || !plane || !host_cplx || elems <= 0) return -1;
  ANYCHK(e, hipMemcpyAsync(plane, host_cplx, (size_t)elems * sizeof(cd), hipMemcpyHostToDevice, e->stream));
  return nq_any_sync(e);
}
int nq_any_download(nq_any* e, const void* plane, double* host_cplx, long long elems) {
  if (!e || !plane || !host_cplx || elems <= 0) return -1;
  ANYCHK(e, hipMemcpyAsync(host_cplx, plane, (size_t)elems * sizeof(cd), hipMemcpyDeviceToHost, e->stream));
  return nq_any_sync(e);
}
// dst <- 1-D transforms of src along `axis` (1: along the contiguous index, length cols; 0: length rows), numpy.fft conventions
// (forward unnormalised, inverse scaled by 1/n); dst may be src
int nq_any_fft(nq_any* e, void* dst, const void* src, int rows, int cols, int axis, int inverse) {
  if (!e || !dst || !src || rows < 1 || cols < 1 || (axis != 0 && axis != 1)) return -1;
  ANYCHK(e, hipSetDevice(e->device));
  const int n = axis == 1 ? cols : rows, nlines = axis == 1 ? rows : cols;
  const nq_any::Plan* pl = nullptr;
  int rc = any_plan(e, n, &pl);
  if (rc) return rc;
  cd* srcp = reinterpret_cast<cd*>(const_cast<void*>(src));
  static int fused = -1;
  if (fused < 0) {
    const char* ev = getenv("NIWQG_AMD_ANY_FUSED");             // 0: the unfused five-launch form (A/B measurements)
    fused = (ev && atoi(ev) == 0) ? 0 : 1;
  }
  auto line_kernel = [&](const cd* a, cd* b, int nl, int len) -> int {
    if (pl->split == 3) return any_split_rows<3>(e, *pl, a, b, nl, len, inverse);
    if (pl->split == 5) return any_split_rows<5>(e, *pl, a, b, nl, len, inverse);
    return any_bluestein_rows(e, *pl, a, b, nl, len, inverse);
  };
  if ((fused || pl->split) && !pl->direct && pl->M <= 8192) {
    // one kernel per line (k_any_bluestein_rows / k_any_split_rows); columns through a transpose of the plane
    if (axis == 1) {
      if (pl->split && srcp == reinterpret_cast<cd*>(dst)) {      // the split kernel writes X[k + m s] while other threads still read x[R j + r]: not in place
        rc = any_tmp(e, (size_t)rows * cols);
        if (rc) return rc;
        rc = line_kernel(srcp, e->tmp, rows, cols);
        if (rc) return rc;
        ANYCHK(e, hipMemcpyAsync(dst, e->tmp, (size_t)rows * cols * sizeof(cd), hipMemcpyDeviceToDevice, e->stream));
      } else {
        rc = line_kernel(srcp, reinterpret_cast<cd*>(dst), rows, cols);
        if (rc) return rc;
      }
    } else {
      rc = any_tmp(e, (size_t)rows * cols);
      if (rc) return rc;
      hipLaunchKernelGGL(k_any_btranspose, dim3((cols + 15) / 16, (rows + 15) / 16, 1), dim3(256), 0, e->stream, (const cd*)srcp, e->tmp, rows, cols,
                         (const cd*)nullptr, 0, 0);
      if (pl->split) {                                           // (not in place, see above: tmp -> tmp2)
        rc = any_buf(e, &e->tmp2, &e->tmp2_elems, (size_t)rows * cols);
        if (rc) return rc;
        rc = line_kernel(e->tmp, e->tmp2, cols, rows);
        if (rc) return rc;
        hipLaunchKernelGGL(k_any_btranspose, dim3((rows + 15) / 16, (cols + 15) / 16, 1), dim3(256), 0, e->stream, (const cd*)e->tmp2, reinterpret_cast<cd*>(dst), cols, rows,
                           (const cd*)nullptr, 0, 0);
        ANYCHK(e, hipGetLastError());
        return 0;
      }
      rc = line_kernel(e->tmp, e->tmp, cols, rows);
      if (rc) return rc;
      hipLaunchKernelGGL(k_any_btranspose, dim3((rows + 15) / 16, (cols + 15) / 16, 1), dim3(256), 0, e->stream, (const cd*)e->tmp, reinterpret_cast<cd*>(dst), cols, rows,
                         (const cd*)nullptr, 0, 0);
    }
    ANYCHK(e, hipGetLastError());
    return 0;
  }
  rc = any_tmp(e, (size_t)nlines * pl->M, pl->M == 16384);
  if (rc) return rc;
  const dim3 gp((pl->M + 15) / 16, (nlines + 15) / 16), gu((n + 15) / 16, (nlines + 15) / 16);
  if (pl->direct) {          // a power of two the row engine takes (four-step for 16384): no chirp, the inverse by its own kernel
    hipLaunchKernelGGL((k_any_lines<true>), gp, dim3(256), 0, e->stream, srcp, e->tmp, rows, cols, axis, pl->M, (const cd*)nullptr, 0, 1.0);
    rc = inverse ? any_rows_fft<true>(e, *pl, nlines, 1.0 / (double)n) : any_rows_fft<false>(e, *pl, nlines, 1.0);
    if (rc) return rc;
    hipLaunchKernelGGL((k_any_lines<false>), gu, dim3(256), 0, e->stream, reinterpret_cast<cd*>(dst), e->tmp, rows, cols, axis, pl->M, (const cd*)nullptr, 0, 1.0);
    ANYCHK(e, hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL((k_any_lines<true>), gp, dim3(256), 0, e->stream, srcp, e->tmp, rows, cols, axis, pl->M,
                     (const cd*)pl->chirp, inverse ? 1 : 0, 1.0);
  rc = any_rows_fft<false>(e, *pl, nlines, 1.0);
  if (rc) return rc;
  hipLaunchKernelGGL(k_any_mul_rows, dim3(2048), dim3(256), 0, e->stream, e->tmp, (const cd*)pl->bhat, nlines, pl->M);
  rc = any_rows_fft<true>(e, *pl, nlines, 1.0 / (double)pl->M);
  if (rc) return rc;
  hipLaunchKernelGGL((k_any_lines<false>), gu, dim3(256), 0, e->stream, reinterpret_cast<cd*>(dst), e->tmp, rows, cols, axis, pl->M,
                     (const cd*)pl->chirp, inverse ? 1 : 0, inverse ? 1.0 / (double)n : 1.0);
  ANYCHK(e, hipGetLastError());
  return 0;
}
// d <- op(a, b, c; scalars) element by element over `elems` complex values (NQ_EW_* in the header); operands the op does not use
// may be NULL; d may alias an operand
int nq_any_ew(nq_any* e, int op, void* d, const void* a, const void* b, const void* c, long long elems, const double* scalars6) {
  if (!e || !d || !a || elems <= 0 || op < 0 || op > EW_FILL) return -1;
  const bool need_b = (op == EW_MUL || op == EW_MULCONJ || op == EW_AXPBY || op == EW_AXPBYPCZ || op == EW_MULADD);
  const bool need_c = (op == EW_AXPBYPCZ || op == EW_MULADD);
  if ((need_b && !b) || (need_c && !c)) ANYFAIL(e, -1, "nq_any_ew: op %d needs more operands", op);
  EwScalars sc;
  for (int i = 0; i < 6; ++i) sc.s[i] = scalars6 ? scalars6[i] : (i == 0 ? 1.0 : 0.0);
  const size_t n = (size_t)elems;
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_any_ew, dim3(grid), dim3(256), 0, e->stream, op, reinterpret_cast<cd*>(d), reinterpret_cast<const cd*>(a),
                     reinterpret_cast<const cd*>(b), reinterpret_cast<const cd*>(c), n, sc);
  return 0;
}
// out2 <- reduction over `elems` complex values (NQ_RD_*): deterministic (fixed grid, partials added in order)
int nq_any_reduce(nq_any* e, int op, const void* a, const void* b, long long elems, double* out2) {
  if (!e || !a || !out2 || elems <= 0 || op < 0 || op > RD_MAXABSRE) return -1;
  if ((op == RD_DOT || op == RD_DOTC || op == RD_WSUMABS2) && !b) ANYFAIL(e, -1, "nq_any_reduce: op %d needs two operands", op);
  const size_t n = (size_t)elems;
  const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(k_any_reduce1, dim3(grid), dim3(256), 0, e->stream, op, reinterpret_cast<const cd*>(a), reinterpret_cast<const cd*>(b), n, e->part);
  hipLaunchKernelGGL(k_any_reduce2, dim3(1), dim3(256), 0, e->stream, op, (const double*)e->part, grid, e->red);
  ANYCHK(e, hipMemcpyAsync(out2, e->red, sizeof(double) * 2, hipMemcpyDeviceToHost, e->stream));
  return nq_any_sync(e);
}
// (rows, n/2+1) half spectrum -> (rows, n) Hermitian extension; project: Hermitian part (in l) of the two self-mirrored columns first
int nq_any_expand_half(nq_any* e, void* full, const void* half, int rows, int n, int project) {
  if (!e || !full || !half || rows < 1 || n < 2 || (n & 1)) return -1;
  hipLaunchKernelGGL(k_any_expand_half, dim3((n + 63) / 64, rows), dim3(64), 0, e->stream, reinterpret_cast<const cd*>(half), reinterpret_cast<cd*>(full), rows, n, project);
  return 0;
}
// the first dcols columns of a (rows, scols) plane as a (rows, dcols) plane
int nq_any_take_cols(nq_any* e, void* dst, const void* src, int rows, int scols, int dcols) {
  if (!e || !dst || !src || rows < 1 || dcols < 1 || dcols > scols) return -1;
  hipLaunchKernelGGL(k_any_take_cols, dim3((dcols + 63) / 64, rows), dim3(64), 0, e->stream, reinterpret_cast<const cd*>(src), reinterpret_cast<cd*>(dst), rows, scols, dcols);
  return 0;
}
int nq_any_set_elem(nq_any* e, void* plane, long long index, double re, double im) {
  if (!e || !plane || index < 0) return -1;
  hipLaunchKernelGGL(k_any_set_elem, dim3(1), dim3(1), 0, e->stream, reinterpret_cast<cd*>(plane), (size_t)index, re, im);
  return 0;
}
// ETDRK4 planes of the linear operator c(l, k) on the whole (n, cols) plane, no filter folded in (the any-size path multiplies by
// `filtr` as the reference does): eq as k_etdrk4_coeffs (0 q Kernel family, 1 phi, 2 QGModel's q, 3 its passive scalar);
// kk (cols values), ll (n values) host arrays; out6: E, Eh, Q, f0, fab, fc planes of this engine.
// near_*: the entries within delta of the contour (at most cap), for the host to recompute (niwqg_amd/_etdrk4.py) and hand back
int nq_any_etdrk4(nq_any* e, int eq, const nq_params* p, const double* kk, const double* ll, const double* contour32, int n, int cols,
                  void* const* out6, double delta, int cap, int* near_count, int* near_l, int* near_k) {
  if (!e || !p || !kk || !ll || !contour32 || !out6 || n < 2 || cols < 1 || eq < 0 || eq > 3) return -1;
  ANYCHK(e, hipSetDevice(e->device));
  double *dk = nullptr, *dl = nullptr;
  cd* dc = nullptr;
  int *cnt = nullptr, *lo = nullptr, *ko = nullptr;
  AnyScratch tmp;
  ANYCHK(e, tmp.get(&dk, cols));
  ANYCHK(e, tmp.get(&dl, n));
  ANYCHK(e, tmp.get(&dc, 32));
  ANYCHK(e, hipMemcpy(dk, kk, sizeof(double) * cols, hipMemcpyHostToDevice));
  ANYCHK(e, hipMemcpy(dl, ll, sizeof(double) * n, hipMemcpyHostToDevice));
  ANYCHK(e, hipMemcpy(dc, contour32, sizeof(cd) * 32, hipMemcpyHostToDevice));
  cd* o[6];
  for (int i = 0; i < 6; ++i) o[i] = reinterpret_cast<cd*>(out6[i]);
  hipLaunchKernelGGL(k_etdrk4_coeffs, dim3((cols + 63) / 64, n), dim3(64), 0, e->stream, eq, n, cols, cols, 0, *p, (const double*)dk, (const double*)dl,
                     (const double*)nullptr, (const cd*)dc, o[0], o[1], o[2], o[3], o[4], o[5]);
  int found = 0;
  if (near_count && near_l && near_k && cap > 0) {
    ANYCHK(e, tmp.get(&cnt, 1));
    ANYCHK(e, tmp.get(&lo, cap));
    ANYCHK(e, tmp.get(&ko, cap));
    ANYCHK(e, hipMemsetAsync(cnt, 0, sizeof(int), e->stream));
    hipLaunchKernelGGL(k_coeff_flag, dim3((cols + 63) / 64, n), dim3(64), 0, e->stream, eq, n, cols, 0, *p, (const double*)dk, (const double*)dl, (const cd*)dc,
                       delta * delta, cap, cnt, lo, ko);
    ANYCHK(e, hipMemcpyAsync(&found, cnt, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    ANYCHK(e, hipStreamSynchronize(e->stream));
    const int take = found < cap ? found : cap;
    if (take > 0) {
      ANYCHK(e, hipMemcpy(near_l, lo, sizeof(int) * take, hipMemcpyDeviceToHost));
      ANYCHK(e, hipMemcpy(near_k, ko, sizeof(int) * take, hipMemcpyDeviceToHost));
    }
    *near_count = found;
  }
  ANYCHK(e, hipStreamSynchronize(e->stream));
  ANYCHK(e, hipGetLastError());
  return 0;
}
// vals: count x 4 complex (Qh, f0, fab, fc) for the entries (l, k) of a (n, cols) plane set
int nq_any_etdrk4_patch(nq_any* e, void* const* out6, int cols, int count, const int* l, const int* k, const double* vals) {
  if (!e || !out6 || count < 0 || (count > 0 && (!l || !k || !vals))) return -1;
  if (count == 0) return 0;
  int *dl = nullptr, *dk = nullptr;
  cd* dv = nullptr;
  AnyScratch tmp;
  ANYCHK(e, tmp.get(&dl, count));
  ANYCHK(e, tmp.get(&dk, count));
  ANYCHK(e, tmp.get(&dv, (size_t)4 * count));
  ANYCHK(e, hipMemcpy(dl, l, sizeof(int) * count, hipMemcpyHostToDevice));
  ANYCHK(e, hipMemcpy(dk, k, sizeof(int) * count, hipMemcpyHostToDevice));
  ANYCHK(e, hipMemcpy(dv, vals, sizeof(cd) * 4 * count, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_coeff_patch, dim3((count + 255) / 256), dim3(256), 0, e->stream, count, (const int*)dl, (const int*)dk, (const cd*)dv, cols, cols, 0,
                     (const double*)nullptr, reinterpret_cast<cd*>(out6[2]), reinterpret_cast<cd*>(out6[3]), reinterpret_cast<cd*>(out6[4]), reinterpret_cast<cd*>(out6[5]), 0);
  ANYCHK(e, hipStreamSynchronize(e->stream));
  return 0;
}

}  // extern "C"
