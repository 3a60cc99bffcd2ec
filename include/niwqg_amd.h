/* niwqg_amd -- C ABI of the MI355X-native ETDRK4 pseudo-spectral stepper.
 *
 * The reference (cesar-rocha/niwqg) is pure Python and has no FFI of its own; its drop-in boundary
 * is the Python class surface.  This header is the boundary a maintainer of
 * the reference would bind with ctypes (INTEGRATION.md shows the stub); every entry point names the
 * reference method it stands in for (file:line in the reference checkout).
 *
 * Conventions
 *   - plain pointers and sizes only; host buffers are caller-owned, device buffers context-owned;
 *   - one host thread per context, one HIP stream per context;
 *   - every function returns 0 on success or a negative error code; nq_last_error() gives text;
 *   - complex arrays are interleaved (re, im) doubles, i.e. numpy complex128;
 *   - all 2-D host arrays are C-contiguous in the reference's layouts: physical (ny, nx);
 *     Kernel-family spectral (ny, nx) index [l, k]; QGModel spectral (ny, nx/2+1).
 */
#ifndef NIWQG_AMD_H
#define NIWQG_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nq_ctx nq_ctx;

enum { NQ_MODEL_COUPLED = 0, NQ_MODEL_UNCOUPLED = 1, NQ_MODEL_QG = 2,
       NQ_MODEL_YBJ = 3 /* niwqg/YBJModel.py: steady psi, nq_step advances phi only (YBJModel.py:52-87); single rank */ };

/* field ids for nq_get_field.  Exact host shapes (C-contiguous; "cplx" = interleaved re,im doubles); nq_field_doubles()
 * returns the number of doubles written.  Spectra of REAL fields (qh, ph, qwh, ch) always come as the HALF spectrum
 * (ny, nx/2+1), k = 0..nx/2, for every model: the reference's Kernel-family (ny,nx) arrays follow from
 * X(l,k) = conj X(-l,-k) for k > nx/2 (niwqg_amd/Kernel.py: hermitian_full; with dual_q the k < 0 side of qh is
 * NQ_F_QH_MINUS instead; without it row ny/2 of qh additionally gets the passenger of nq_get_qh_passenger).  phih is a
 * genuine full plane. */
enum {
  NQ_F_Q = 0,      /* real (ny,nx)        q      = Re ifft(qh)                 Kernel.py:97/CoupledModel.py:97 */
  NQ_F_QH = 1,     /* cplx (ny,nx/2+1)    qh                                                                    */
  NQ_F_P = 2,      /* real (ny,nx)        p      streamfunction                CoupledModel.py:93               */
  NQ_F_PH = 3,     /* cplx (ny,nx/2+1)    ph                                                                    */
  NQ_F_PHI = 4,    /* cplx (ny,nx)        phi                                  Kernel.py:337                    */
  NQ_F_PHIH = 5,   /* cplx (ny,nx)        phih                                                                  */
  NQ_F_U = 6,      /* real (ny,nx)        u = Re ifft(-il ph)                  Kernel.py:481                    */
  NQ_F_V = 7,      /* real (ny,nx)        v = Re ifft( ik ph)                                                   */
  NQ_F_QPSI = 8,   /* real (ny,nx)        q_psi = q - qw (q without waves)     CoupledModel.py:145-152          */
  NQ_F_QW = 9,     /* real (ny,nx)        qw                                   (coupled model only)             */
  NQ_F_QWH = 10,   /* cplx (ny,nx/2+1)    qwh                                  CoupledModel.py:86-88            */
  NQ_F_PHIX = 11,  /* cplx (ny,nx)        phix (as last refreshed: quirk Q1)   Kernel.py:610                    */
  NQ_F_PHIY = 12,  /* cplx (ny,nx)        phiy                                                                  */
  NQ_F_QH_MINUS = 13,/* cplx half spectrum conj(qh(-l,-k)), k = 0..nx/2 (dual_q contexts only)                  */
  NQ_F_C = 14,       /* real (ny,nx)       c      passive scalar of QGModel     QGModel.py:403-404               */
  NQ_F_CH = 15,      /* cplx (ny,nx/2+1)   ch                                                                     */
  NQ_F_QH_STAGE4 = 16,/* cplx half spectrum qh at which the LAST step evaluated its fourth stage: the u, v that QGModel's
                         jacobian_psi_c sees at a diagnostics tick are still those (QGModel.py:375, :483-495, :727-731) */
  NQ_F_PHIH_STAGE4 = 17,      /* cplx (ny,nx) phih of that same stage (Kernel family): with NQ_F_QH_STAGE4 it determines the
                                 self.u, self.v a step leaves behind -- the reference's last jacobian_psi_q call of a step is
                                 the fourth stage's (Kernel.py:364-368 vs :381-387), not the new state's                     */
  NQ_F_QH_MINUS_STAGE4 = 18,  /* the second copy (NQ_F_QH_MINUS) of that stage, dual_q contexts                              */
  NQ_F_QH_TICK = 19,          /* qh, phih, the second copy of qh and qwh as the last nq_tick_snapshot kept them: what the      */
  NQ_F_PHIH_TICK = 20,        /* reference's tick-only leftovers (upsilon Kernel.py:618; phq, phw, uq, vq, uw, vw              */
  NQ_F_QH_MINUS_TICK = 21,    /* CoupledModel.py:99-113; YBJModel's lapphi) are rebuilt from between ticks                     */
  NQ_F_QWH_TICK = 22
};

/* scalar ids for nq_get_scalar */
enum {
  NQ_S_KE = 0, NQ_S_PW = 1, NQ_S_KW = 2,   /* INCREMENT of the budget accumulators Ke, Pw, Kw since the last
                                              read (reading resets it)             Kernel.py:390-392;
                                              QGModel with passive scalar: NQ_S_PW is the increment of cvar
                                              (QGModel.py:394)                                        */
  NQ_S_KE_QG = 3,                          /* _calc_ke_qg()                        Kernel.py:600-602 */
  NQ_S_KE_NIW = 4,                         /* _calc_ke_niw()                       Kernel.py:604-606 */
  NQ_S_PE_NIW = 5,                         /* _calc_pe_niw() (no side effect here) Kernel.py:608-611 */
  NQ_S_CFL = 6,                            /* max(|u|,|v|,|phi|): _calc_cfl() without the dt/dx factor
                                                                                   Kernel.py:660-662 */
  NQ_S_MAX_PHI = 7                         /* max |phi| alone (the status line's CFL combines it with the
                                              fourth stage's max |u|, |v|: nq_get_stage4_max)        */
};

typedef struct nq_params {
  int model;          /* NQ_MODEL_*                                                             */
  int nx;             /* grid is nx x nx (the reference ignores ny, Kernel.py:100-101)          */
  int budgets;        /* 1: accumulate Ke,Pw,Kw inside the step like Kernel.py:319-322,:390-392 */
  int dual_q;         /* 1: keep the second q-hat copy X-(l,k) = conj(qh(-l,-k)) (needed when filtr is not
                         mirror-symmetric, i.e. dealias=True, Kernel.py:277-281; also makes qh exact on row N/2) */
  double dt;
  double U;           /* uniform zonal flow                                                     */
  double f;           /* Coriolis                                                               */
  double kappa2;      /* (m f / N)^2                                                            */
  double nu, nu4, mu;       /* q equation   Kernel.py:417-418 / QGModel.py:426-428              */
  double nuw, nu4w, muw;    /* phi equation Kernel.py:440-442                                   */
  double beta;        /* QGModel only                                                           */
  int passive_scalar; /* QGModel only: 1 = also step the passive scalar c (QGModel.py:345-404)  */
  double nu4c, nuc, muc;    /* its linear operator -nu4c wv4 - nuc wv2 - muc (QGModel.py:446-454)      */
} nq_params;

/* Kernel.__init__ / QGModel.__init__ (Kernel.py:139-152): builds grid-dependent tables on the device.
 *   kk   : nk wavenumbers (nk = nx for the Kernel family, nx/2+1 for QG)   Kernel.py:242-244 / QGModel.py:249
 *   ll   : nx wavenumbers
 *   filtr: host (nx, nk) real plane, the reference's `filtr`                  Kernel.py:267-284
 *   contour: 32 complex roots of unity r_j used by the ETDRK4 contour mean    Kernel.py:424-426
 * The ETDRK4 coefficient planes (Kernel.py:417-454) are computed on the device.                      */
int nq_create(const nq_params* p, const double* kk, const double* ll, const double* filtr,
              const double* contour, int device, nq_ctx** out);
int nq_destroy(nq_ctx* ctx);
const char* nq_last_error(const nq_ctx* ctx);     /* ctx may be NULL: last global error */

/* Kernel.set_q (Kernel.py:520-535) / QGModel.set_q (QGModel.py:507-520) */
int nq_set_q(nq_ctx* ctx, const double* q_host);
/* QGModel.set_c (QGModel.py:522-534): real (ny,nx); call after nq_set_q */
int nq_set_c(nq_ctx* ctx, const double* c_host);
/* Kernel.set_phi (Kernel.py:538-551): phi_host is complex (ny,nx) */
int nq_set_phi(nq_ctx* ctx, const double* phi_host);
/* CoupledModel._invert + _calc_rel_vorticity (CoupledModel.py:75-97,:145-152), UnCoupledModel._invert
 * (UnCoupledModel.py:54-64), QGModel._invert (QGModel.py:497-505) on the current state */
int nq_invert(nq_ctx* ctx);
/* replay of the side effect of _calc_pe_niw on phix/phiy (Kernel.py:610, quirk Q1) */
int nq_refresh_grad_phi(nq_ctx* ctx);

/* nsteps x Kernel._step_etdrk4 (Kernel.py:307-397) / QGModel._step_etdrk4 (QGModel.py:328-407).
 * Asynchronous on the context's stream.                                                              */
int nq_step(nq_ctx* ctx, int nsteps);
/* After a step WITHOUT a diagnostics tick the reference's self.u, self.v are still those of the fourth jacobian_psi_q call
 * of _step_etdrk4 (Kernel.py:364-368), and its status line's CFL (Kernel.py:594, :660-662) is taken from them.
 * nq_request_stage4_max: the last step of the NEXT nq_step / nq_slab_step call also records max |u|, max |v| of that stage
 * over this rank's rows (two extra row passes in that one step, nothing otherwise); nq_get_stage4_max reads them
 * (out2 = max|u|, max|v|; -4 when the last step call recorded none).  Kernel family only (QGModel._calc_cfl recomputes
 * u, v from psi: QGModel.py:621-629; YBJModel's u, v are steady). */
int nq_request_stage4_max(nq_ctx* ctx);
int nq_get_stage4_max(nq_ctx* ctx, double* out2);
/* Keep this rank's qh (both copies), phih and qwh as they are NOW (device-to-device, asynchronous; four planes allocated by the
 * first call): the host classes call it at every diagnostics tick (Diagnostics.py:41-58), because some of the arrays a tick
 * leaves on the reference's instance are refreshed by nothing else (see NQ_F_QH_TICK).  Read back with nq_get_field
 * (NQ_F_*_TICK) or nq_download_spectral (which 9-12). */
int nq_tick_snapshot(nq_ctx* ctx);
int nq_sync(nq_ctx* ctx);

/* copy a field to the host in the reference's layout (blocking) */
int nq_get_field(nq_ctx* ctx, int field_id, double* host_out);
/* number of doubles nq_get_field(field_id) writes for this context (-1: unknown id or NULL ctx) */
long long nq_field_doubles(const nq_ctx* ctx, int field_id);
/* The reference's full-plane qh is not exactly Hermitian: on row l = ny/2 the il term of jacobian_psi_q
 * (Kernel.py:471-486; numpy's ll[ny/2] is not odd) adds an anti-Hermitian part A_k that every stage update carries
 * along (Kernel.py:327, :347, :364, :381) and that never reaches physical space.  The device evolves it as ONE extra row
 * beside the half spectrum; out_cplx receives A_k for this context's local half-spectrum columns (nx/2+1 complex values
 * on a single context, entries k = 0 and nx/2 zero; all zeros for QGModel, YBJModel and dual_q contexts):
 *     qh_ref[ny/2, k] = qh[ny/2, k] + A_k,     qh_ref[ny/2, nx-k] = conj(qh[ny/2, k]) - conj(A_k),   0 < k < nx/2.    */
int nq_get_qh_passenger(nq_ctx* ctx, double* out_cplx);
int nq_get_scalar(nq_ctx* ctx, int scalar_id, double* out);

/* Snapshots that do not stall the stepper (niwqg/Saving.py:59-86 saves t, q, phi every tsave_snapshots steps):
 * nq_snapshot_begin forms q (real (ny,nx)) and, with_phi != 0, phi (cplx (ny,nx)) of the CURRENT state in device buffers of
 * their own and starts their copy to pinned host memory on a second stream; further nq_step calls may be queued at once.
 * nq_snapshot_end waits for that copy only and writes the arrays to q_out / phi_out (either may be NULL).  One in flight. */
int nq_snapshot_begin(nq_ctx* ctx, int with_phi);
int nq_snapshot_end(nq_ctx* ctx, double* q_out, double* phi_out);

/* the FFT seam, Kernel.fft / Kernel.ifft (Kernel.py:562-566): complex (ny,nx) -> complex (ny,nx);
 * QGModel.fft / ifft (QGModel.py:551-552): real (ny,nx) <-> complex (ny,nx/2+1).                     */
int nq_fft2(nq_ctx* ctx, const double* in_cplx, double* out_cplx);
int nq_ifft2(nq_ctx* ctx, const double* in_cplx, double* out_cplx);
int nq_rfft2(nq_ctx* ctx, const double* in_real, double* out_cplx);
int nq_irfft2(nq_ctx* ctx, const double* in_cplx, double* out_real);

/* The three Jacobians, evaluated on the current device state, in the reference's own array layouts and with its
 * [0,0] conventions (assembled on the device; nothing left for the caller to multiply or expand):
 *   nq_jacobian_psi_q    Kernel.jacobian_psi_q (Kernel.py:471-486): ik*fft(u q) + il*fft(v q), [0,0] = 0
 *                          -> cplx (ny, nx)       = 2*ny*nx doubles          (Kernel family)
 *                        QGModel.jacobian_psi_q (QGModel.py:469-481): same, [0,0] kept
 *                          -> cplx (ny, nx/2+1)   = 2*ny*(nx/2+1) doubles    (NQ_MODEL_QG)
 *   nq_jacobian_psi_phi  Kernel.jacobian_psi_phi (Kernel.py:457-469): fft(u phix + v phiy), [0,0] = 0 (kept by a
 *                        NQ_MODEL_YBJ context, YBJModel.py:123-133)         -> cplx (ny, nx) = 2*ny*nx doubles
 *   nq_jacobian_phic_phi CoupledModel.jacobian_phic_phi (CoupledModel.py:59-73), [0,0] = 0; refreshes phix, phiy
 *                                                                           -> cplx (ny, nx) = 2*ny*nx doubles
 * nq_products_uq_vq has no reference counterpart: the two transforms fft(u q), fft(v q) themselves on k = 0..nx/2,
 *                        cplx (2, ny, nx/2+1) = 4*ny*(nx/2+1) doubles (tests; callers with their own flux forms). */
int nq_jacobian_psi_q(nq_ctx* ctx, double* out_cplx);
/* QGModel with its passive scalar: ik*fft(u c) + il*fft(v c), (ny, nx/2+1) cplx (QGModel.py:483-495), u, v of the current psi */
int nq_jacobian_psi_c(nq_ctx* ctx, double* out_cplx);
int nq_jacobian_psi_phi(nq_ctx* ctx, double* out_cplx);
int nq_jacobian_phic_phi(nq_ctx* ctx, double* out_cplx);
int nq_products_uq_vq(nq_ctx* ctx, double* out_cplx2);
/* fft(phi * q_psi), cplx (ny, nx) = 2*ny*nx doubles: the refraction source of Kernel.py:332 (before its -0.5j factor;
 * mean not removed), formed by the row kernel exactly as inside a step */
int nq_refraction(nq_ctx* ctx, double* out_cplx);

/* Diagnostics tick on the device: the raw sums from which every scalar of increment_diagnostics (Diagnostics.py:41-58;
 * the 21 kernel lambdas Kernel.py:718-868 and the 3 class lambdas CoupledModel.py:115-136) follows without any plane
 * leaving the GPU.  out: 32 doubles, M = nx*nx, H = Hermitian part on the two self-mirrored columns:
 *   [0..3]  sum wv2^n |phih|^2, n = 0..3      (ke_niw, pe_niw, ep_phi, chi_phi; Kernel.py:604-611, :629-652)
 *   [4,5]   phih[0,0]                         (cke_niw, pi; Kernel.py:701-706)
 *   [6]     sum |H qh|^2          -> ens      [7] sum wv4 |qh|^2 -> chi_q        [8]  sum |qh|^2/wv2  -> ke_qg_q
 *   [9]     sum |qwh|^2/wv2       -> ke_qg_w  [10] sum Re(conj(H qh) H qwh) (l'^2 + k'^2)/wv2^2, l' = 0 on row ny/2, k' = 0 on column nx/2 -> -ke_qg_qw
 *   [11]    sum wv2 |ph|^2        -> ke_qg    [12..14] sum {wv4, wv2, 1} Re(conj(H ph) H qh) -> ep_psi (Kernel.py:635-640)
 *   [15]    mean(q_psi)
 *   [16..23] physical sums: q^2, q_psi^2, q_psi^3, (q_psi-mean)^2, ups^2, ups q_psi, q_psi Re(phi), q_psi Im(phi),
 *            ups = |phi|^2 - mean|phi|^2     (conc_niw, skew, pi; Kernel.py:613-623, :701)
 *   [24..27] sum {Re, Im}(conj(lap_h) J), {Re, Im}(conj(diss_h) J),  J = F[u phix + v phiy]        (gamma2, xi1)
 *   [28..31] the same four with i F[phi q_psi] in place of J                                        (gamma1, xi2)
 * Sums over the half spectrum carry weight 2 on the interior columns.  QGModel: entries [6..15] only; with its passive
 * scalar also [16..19] sum w wv2^n |ch|^2, n = 0..3 (n = 0 without [0,0]; C2, gradC2, ep_c, chi_c of QGModel.py:595-604,
 * :724-726) and [20] sum w Re(conj(-wv2 ch)(ik F[u c] + il F[v c])) (Gamma_c, QGModel.py:727-731; u, v of the state at
 * which the last step evaluated its fourth stage, as the reference's are at a tick).                          */
int nq_diagnostics(nq_ctx* ctx, double* out32);

/* copy of one ETDRK4 coefficient plane (0:E 1:Eh 2:Q 3:f0 4:fab 5:fc) of equation eq (0: q, (nx, nx/2+1) complex;
 * 1: phi, (nx, nx) complex; 2: QGModel's passive scalar, (nx, nx/2+1)), without the filter folded in; values as the
 * reference's expch, expch_h, Qh, f0, fab, fc (Kernel.py:417-454, QGModel.py:426-461).                           */
int nq_get_coeff(nq_ctx* ctx, int eq, int which, double* out_cplx);

/* The reference evaluates Qh, f0, fab, fc as the mean of 32 points on the unit circle around c dt (Kernel.py:424-433).  Where
 * c dt lies within delta of minus one of those points, one term is a removable singularity evaluated by cancellation and the
 * reference's value is its libm's rounding error amplified by eps / distance^3: to be identical there, those entries have to
 * come from the same numpy expression.  nq_coeff_near_contour lists them for equation eq (as nq_get_coeff; on a slab context:
 * this rank's columns): up to cap (l, k) pairs, k a GLOBAL column index; returns how many there are (call again with a larger cap
 * if that exceeds cap), negative on error.  nq_coeff_patch replaces Qh, f0, fab, fc at n entries with vals (n x 4 complex128,
 * the reference's values WITHOUT the filter; the library folds its filter in) in every plane set of that equation, and
 * nq_get_coeff returns them from then on.  The host side calls both once, right after nq_create (niwqg_amd/_etdrk4.py).     */
int nq_coeff_near_contour(nq_ctx* ctx, int eq, double delta, int cap, int* l_out, int* k_out);
int nq_coeff_patch(nq_ctx* ctx, int eq, int n, const int* l, const int* k, const double* vals_cplx);

/* ---- 1-D slab decomposition over nranks GPUs (one process per GPU) -----------------
 * Rows of the mixed-space planes are split over ranks on the "x side" (row kernels), columns on the "y side"
 * (spectral kernels).  Arrays that cross together form an exchange group g = 0..3; each group has an x-side
 * and a y-side buffer of nq_group_elems() complex128 elements, both cut into nranks equal blocks, so that ONE
 * all_to_all_single(recv = other side, send = this side) moves the group.  Two ways to drive a step: nq_slab_step
 * (below: the library runs the phases AND issues the exchanges itself over the link the context was given -- the
 * product path), or phase by phase with nq_phase, where the CALLER owns the collectives (torch.distributed / RCCL) and
 * the library only runs the phases in between, on the stream it was given (stream == NULL: a private stream the
 * library creates -- then the caller must order its collectives against nq_stream()).
 *   buffers[2g], buffers[2g+1] : x-side and y-side device buffers of group g (may be NULL for empty groups)
 *   buffers[8]                 : 64 doubles for the per-step budget sums (summed over ranks by the caller)
 * buffers == NULL: the library allocates all of them itself (nq_group_buffers / nq_reduce_buffer give the pointers).   */
long long nq_group_elems(const nq_params* p, int nranks, int group);
int nq_create_slab(const nq_params* p, const double* kk, const double* ll, const double* filtr,
                   const double* contour, int device, int nranks, int rank, void* const* buffers, void* stream,
                   nq_ctx** out);
int nq_slab_info(const nq_ctx* ctx, int* info8);
int nq_group_buffers(nq_ctx* ctx, int group, void** x_side, void** y_side, long long* elems);
/* local column slab of qh (which 0: (ny, local half-spectrum columns)) or phih (which 1: (ny, nx/nranks)); download also
 * which 2: ph, 3: qwh, 4: the second copy of qh of a dual_q context, 5: ch of QGModel's passive scalar, 6: the q-hat the
 * last step's fourth stage was evaluated at (NQ_F_QH_STAGE4) (half-spectrum slabs like qh), 7: the phih of that stage (like
 * which 1), 8: the second copy of qh of that stage (like which 4), 9-12: nq_tick_snapshot's qh, phih, second copy of qh, qwh */
int nq_upload_spectral(nq_ctx* ctx, int which, const double* host);
int nq_download_spectral(nq_ctx* ctx, int which, double* host);
enum {
  NQ_PH_PRODUCTS = 0,      /* rows: nonlinear products (Kernel.py:471-486,:457-469,:332)   then exchange group 0 (x->y) */
  NQ_PH_UPDATE = 1,        /* spectral: N_q, N_phi, ETDRK4 stage update                    then group 1 (y->x) [+3]    */
  NQ_PH_WAVEPV = 2,        /* rows: wave-PV sources (CoupledModel.py:59-88)                then group 2 (x->y)         */
  NQ_PH_INVERT = 3,        /* spectral: psi inversion (CoupledModel.py:75-97)              then group 3 (y->x)         */
  NQ_PH_EMIT_PHI = 4,      /* after uploading phih (set_phi)                               then group 1 (y->x)         */
  NQ_PH_INVERT_NOW = 5,    /* inversion of the current qh (set_q)                          then group 3 (y->x)         */
  NQ_PH_BUDGET_SUMS = 6,   /* local budget sums of the finished step                       then all-reduce buffer 0    */
  NQ_PH_BUDGET_FINISH = 7  /* Ke, Pw, Kw increments from the reduced sums (Kernel.py:390-392)                          */
};
int nq_phase(nq_ctx* ctx, int phase, int stage);
int nq_reduce_buffer(nq_ctx* ctx, int which, void** device_ptr, int* count);
/* host copies of those blocks (which 0..3 as above, 4 / 5: the spectral / physical half of the diagnostic sums, 16 doubles
 * each): read, sum over ranks by any means, write back -- what an all-reduce callback does */
int nq_reduce_read(nq_ctx* ctx, int which, double* host_out);
int nq_reduce_write(nq_ctx* ctx, int which, const double* host_in);

/* ---- the slab step INSIDE the library: one host call per nsteps, exchanges issued by the library ------------------
 * nq_slab_step runs the phase sequence of nq_phase itself and moves the exchange groups over the link the context
 * was given, on its own exchange stream, chunk by chunk: every group is cut into nchunks row chunks; for an x->y group
 * the row kernel of chunk i+1 runs while chunk i is on the wire, for a y->x group the row kernel of chunk i starts as
 * soon as chunk i has arrived; the q update runs under the transfer of the phi group; the budget sums are all-reduced
 * once per step and not at all when budgets are off.  Links:
 *   nq_comm_init          RCCL: grouped ncclSend/ncclRecv per chunk + ncclAllReduce, communicator made from a
 *                         128-byte unique id (rank 0: nq_comm_unique_id; hand it to the others by any means, e.g.
 *                         torch.distributed broadcast).  librccl is taken from the process (dlopen), not linked.
 *   nq_slab_attach_peers  all nranks contexts live in THIS process on one device (the one-GPU test double): the same
 *                         choreography of streams and events, device-to-device copies instead of the wire.  The step is
 *                         then driven through the rank-0 context and advances all of them.
 *   nq_slab_set_callbacks the caller moves the data: exchange(user, group, to_y) and allreduce(user, which) are called
 *                         with the compute stream drained and must be complete on return (gloo / host staging).
 * nq_slab_put_rows + nq_slab_commit: Kernel.set_q / set_phi (Kernel.py:520-551) from this rank's ROWS of the physical
 * field (nloc rows of nx values, real / complex): row transform on the device (put_rows, local), then exchange, column
 * transform and the phases of the single-rank calls (commit, collective) -- no rank ever holds or transforms the whole plane.
 * nq_slab_get_rows: this rank's rows (nloc, nx) of a physical field (NQ_F_Q, _P, _U, _V, _QW, _C real; _PHI, _PHIX,
 * _PHIY complex) from the mixed-space rows the last step left on the x side (q_psi = q - qw: two calls).
 * nq_comm_probe: 0 when this process can resolve librccl (load only, no communicator) -- every rank calls it and the
 * ranks agree on the outcome BEFORE rank 0 asks for an id, so that nobody is left alone in a collective. */
typedef int (*nq_exchange_fn)(void* user, int group, int to_y);
typedef int (*nq_allreduce_fn)(void* user, int which);      /* which: as nq_reduce_read: 0..3, and 4 / 5 = the spectral /
                                                              * physical half of the diagnostic sums, 16 doubles each */
int nq_comm_probe(void);
int nq_comm_unique_id(void* out128);
int nq_comm_init(nq_ctx* ctx, const void* id128, int nranks, int rank);
int nq_slab_attach_peers(nq_ctx* const* ctxs, int nranks);
int nq_slab_set_callbacks(nq_ctx* ctx, nq_exchange_fn exchange, nq_allreduce_fn allreduce, void* user);
/* Measurement aid (bench.py --rank-of P): the context is ONE rank of its nranks-rank decomposition and runs alone -- every
 * stream, event, row chunk and kernel launch of a real rank, but only the rank's own block crosses (device copy) and nothing
 * is all-reduced.  The fields are not a simulation; what it gives is a rank's compute time without the exchange. */
int nq_slab_set_null_link(nq_ctx* ctx);
/* YBJModel on more than one rank has a fifth exchange group (4, y -> x, shaped like group 1: the stage results whose
 * gradients the next stage reads, YBJModel.py:52-87).  The library owns its buffers unless the caller hands over two
 * device buffers of nq_group_elems(p, nranks, 4) complex elements here (callback link: the caller moves them). */
int nq_slab_set_stage_buffers(nq_ctx* ctx, void* x_side, void* y_side);
int nq_slab_config(nq_ctx* ctx, int nchunks);               /* 1, 2, 4 or 8; reduced if the local rows do not divide */
int nq_slab_step(nq_ctx* ctx, int nsteps);
int nq_slab_put_rows(nq_ctx* ctx, int which /* 0: q, 1: phi, 2: c */, const double* rows);   /* local: rows -> x side */
int nq_slab_commit(nq_ctx* ctx, int which);   /* collective: the rest of set_q / set_phi (rank-0 context in peers mode);
                                                  which 2: Kernel._invert on the current state, nothing uploaded;
                                                  which 3: the rest of QGModel.set_c (QGModel.py:522-534) after put_rows(2) */
int nq_slab_get_rows(nq_ctx* ctx, int field_id, double* rows_out);
/* The whole-plane calls of the class API on a slab model (Kernel.fft, jacobian_psi_q, jacobian_psi_phi,
 * CoupledModel.jacobian_phic_phi): nq_slab_spectral (collective) runs the row kernel of the current state -- or takes
 * the rows last given to nq_slab_put_rows -- through exchange and column transform into a scratch column slab;
 * nq_slab_spectral_read (local) copies this rank's slab out: (nx, wh) complex for the half-spectrum results, (nx, wf)
 * otherwise (wh, wf: nq_slab_info).  what 0 / 1: F[u q] / F[v q] (half); 2: F[u phix + v phiy] (full, [0,0] as
 * computed); 3: F[i phi q_psi] (full); 4: F[Re i(phix* phiy - phiy* phix)] (half, coupled model); 5 / 6: forward
 * transform of the real / complex rows of nq_slab_put_rows(0 / 1) (half / full).  The model state is not touched. */
int nq_slab_spectral(nq_ctx* ctx, int what);
int nq_slab_spectral_read(nq_ctx* ctx, int half, double* out_cplx);
/* nq_diagnostics of a slab-decomposed simulation: the same 32 sums, every rank's part summed over the ranks (collective);
 * nq_slab_local_max: max|u|, max|v|, max|phi| over this rank's rows (the caller takes the max over ranks for the CFL) */
int nq_slab_diagnostics(nq_ctx* ctx, double* out32);
int nq_slab_local_max(nq_ctx* ctx, double* out3);
/* counters since the last reset: out[0] host calls of nq_slab_step, [1] steps, [2] exchange chunks issued, [3] bytes this
 * rank sent to OTHER ranks, [4] milliseconds the exchange stream spent in exchanges (HIP events; 0 unless timing was
 * switched on with reset = 2), [5] nchunks in use.  reset = 2 switches the per-exchange event timing ON and it stays on (two events
 * per exchange chunk and per all-reduce, at most 16384 timed pairs each -- later ones go untimed) until a call with reset = 1,
 * which zeroes the counters and switches it off again. */
int nq_slab_counters(nq_ctx* ctx, double* out6, int reset);
/* out2[0]: milliseconds the exchange stream spent in the all-reduces of the steps since timing was switched on
 * (nq_slab_counters with reset = 2), out2[1]: how many all-reduces that was.  Read before the call that resets. */
int nq_slab_allreduce_ms(nq_ctx* ctx, double* out2);

/* timing of the hot loop with HIP events on the context's stream */
int nq_timer_start(nq_ctx* ctx);
int nq_timer_stop(nq_ctx* ctx, float* elapsed_ms);
/* 16 event slots on the context's stream: record marks between asynchronous nq_step calls, read the time between
 * two marks afterwards (blocks until slot_b has passed) */
int nq_event_record(nq_ctx* ctx, int slot);
int nq_event_elapsed(nq_ctx* ctx, int slot_a, int slot_b, float* elapsed_ms);
/* Per-kernel timing with HIP events on the context's stream.  While enabled, every launch of the
 * selected kernel class inside nq_step is bracketed by an event pair (cost ~2 us per launch).
 * class: 0 x_products, 1 x_wavepv, 2 s_q, 3 s_phi, 4 s_invert, 5 y_A (all A sub-passes)              */
/* best of `reps` timed launches of a 1-read + 1-write stream copy of `bytes` bytes (16 B per lane, grid-stride), in GB/s
 * of bytes moved (read + written): the copy rate of THIS device, bench.py's second roofline denominator */
int nq_stream_copy_gbs(nq_ctx* ctx, long long bytes, int reps, double* gbs_out);
int nq_profile_enable(nq_ctx* ctx, int kernel_class);     /* -1 disables, -2 brackets every class */
/* bracket only every stride-th launch of the enabled class(es) (>= 1; 1 = every launch): keeps the cost of measuring a kernel
 * live inside a timed region below 1 % (an event pair costs the stream ~9 us) */
int nq_profile_stride(nq_ctx* ctx, int stride);
int nq_profile_read(nq_ctx* ctx, int* launches, float* total_ms);   /* synchronises, then resets */
int nq_profile_read_all(nq_ctx* ctx, int* launches6, float* total_ms6);   /* per class, after nq_profile_enable(-2) */
/* bytes of device memory held by the context */
long long nq_device_bytes(const nq_ctx* ctx);
/* stream handle (hipStream_t) so that callers can order their own work */
void* nq_stream(nq_ctx* ctx);

/* ---- the any-size engine: grids without a fused plan -----------------------------------------------------------------------
 * The reference takes any nx (niwqg/Kernel.py:100-103; numpy.fft transforms any length, :562-566; QGModel.py:93-96, :551-552).
 * The fused step above exists for powers of two in [64, 8192].  For every other even nx in [4, 4096] the model classes run the
 * reference's own sequence of whole-plane operations (niwqg_amd/_anysize.py) on device planes through these calls: 1-D transforms
 * of any length along either axis (Bluestein's chirp-z identity on the power-of-two row engine), element-wise operations,
 * deterministic reductions.  Planes are contiguous arrays of complex128 owned by the engine; everything is asynchronous on the
 * engine's stream except the calls that return data.  Same return codes as above; nq_any_last_error for the text. */
typedef struct nq_any nq_any;
enum {  /* nq_any_ew: d = ... element by element; s0, s1, s2 complex scalars (scalars6 = re, im of each; NULL: s0 = 1, s1 = s2 = 0) */
  NQ_EW_COPY = 0,      /* a                                   */
  NQ_EW_MUL = 1,       /* s0 a b                              */
  NQ_EW_MULCONJ = 2,   /* s0 conj(a) b                        */
  NQ_EW_AXPBY = 3,     /* s0 a + s1 b                         */
  NQ_EW_AXPBYPCZ = 4,  /* s0 a + s1 b + s2 c                  */
  NQ_EW_REAL = 5,      /* Re a (imaginary part zero): numpy's .real kept as a complex plane */
  NQ_EW_ABS2 = 6,      /* |a|^2                               */
  NQ_EW_SCALE = 7,     /* s0 a                                */
  NQ_EW_CONJ = 8,      /* conj(a)                             */
  NQ_EW_ADDS = 9,      /* a + s0                              */
  NQ_EW_IMAG = 10,     /* Im a (as a real value)              */
  NQ_EW_MULADD = 11,   /* s0 a b + s1 c                       */
  NQ_EW_FILL = 12      /* s0 (a is not read)                  */
};
enum {  /* nq_any_reduce: out2 = (re, im) */
  NQ_RD_SUM = 0, NQ_RD_SUMABS2 = 1, NQ_RD_DOT = 2 /* sum a b */, NQ_RD_DOTC = 3 /* sum conj(a) b */, NQ_RD_MAXABS = 4,
  NQ_RD_WSUMABS2 = 5 /* sum Re(b) |a|^2 */, NQ_RD_MAXABSRE = 6 /* max |Re a| */
};
int nq_any_create(int device, nq_any** out);
int nq_any_destroy(nq_any* eng);
const char* nq_any_last_error(const nq_any* eng);
int nq_any_sync(nq_any* eng);
long long nq_any_device_bytes(const nq_any* eng);
int nq_any_alloc(nq_any* eng, long long elems, void** plane);                 /* zero-filled */
int nq_any_free(nq_any* eng, void* plane, long long elems);
int nq_any_upload(nq_any* eng, void* plane, const double* host_cplx, long long elems);
int nq_any_download(nq_any* eng, const void* plane, double* host_cplx, long long elems);
/* numpy.fft.fft / ifft along one axis of a (rows, cols) plane (axis 1: the contiguous index); dst may be src */
int nq_any_fft(nq_any* eng, void* dst, const void* src, int rows, int cols, int axis, int inverse);
int nq_any_ew(nq_any* eng, int op, void* d, const void* a, const void* b, const void* c, long long elems, const double* scalars6);
int nq_any_reduce(nq_any* eng, int op, const void* a, const void* b, long long elems, double* out2);
/* (rows, n/2+1) -> (rows, n): full[l, n-k] = conj(half[-l, k]); project != 0 first takes the Hermitian part (in l) of columns 0
 * and n/2, which is all numpy.fft.irfft2 sees of them (QGModel.py:552) */
int nq_any_expand_half(nq_any* eng, void* full, const void* half, int rows, int n, int project);
int nq_any_take_cols(nq_any* eng, void* dst, const void* src, int rows, int src_cols, int dst_cols);
int nq_any_set_elem(nq_any* eng, void* plane, long long index, double re, double im);
/* E = exp(c dt), Eh = exp(c dt / 2), Q, f0, fab, fc of the linear operator c(l, k) on a (n, cols) plane, WITHOUT the filter
 * (Kernel.py:417-454, QGModel.py:426-466); eq 0: q of the Kernel family, 1: phi, 2: QGModel's q (beta term), 3: its passive
 * scalar.  The entries within delta of the contour are listed (near_*; at most cap) for the host to recompute exactly as the
 * reference does (niwqg_amd/_etdrk4.py) and hand back through nq_any_etdrk4_patch (vals: count x 4 complex: Qh, f0, fab, fc). */
int nq_any_etdrk4(nq_any* eng, int eq, const nq_params* p, const double* kk, const double* ll, const double* contour32, int n, int cols,
                  void* const* out6, double delta, int cap, int* near_count, int* near_l, int* near_k);
int nq_any_etdrk4_patch(nq_any* eng, void* const* out6, int cols, int count, const int* l, const int* k, const double* vals);

#ifdef __cplusplus
}
#endif
#endif
