/* qexhip.h -- C ABI of libqexhip.so: MI355X-native staggered Dslash + CG (+ Wilson flow)
 * behind QEX's stagD / stagSolve operator API.
 *
 * This is the drop-in boundary.  Every entry point names the reference interface it
 * replaces (file:line relative to ctpeterson/qex @ 2025-02-23).  The precedent for the
 * whole boundary is QEX's own QUDA bridge, src/quda/qudaWrapperImpl.nim:165-261
 * (qudaSolveXX): host fields are handed over site-major in the V=1 layout, one C call
 * does the solve, the solution is copied back.  INTEGRATION.md shows the Nim binding.
 *
 * Conventions
 *   - all pointers are HOST pointers to fp64 data unless a name says `dev`;
 *   - site order: V=1 MILC even-odd order of the rank-local lattice
 *       lex = x0 + L0*(x1 + L1*(x2 + L2*x3)); idx = lex/2 + ((x0+x1+x2+x3)&1)*vol/2
 *     (src/layout/qlayout.nim:110-131 with innerGeom = 1);
 *   - colour vector: double[vol][3][2]; gauge field: double[vol][4][3][3][2]
 *     ([site][mu][row][col][re,im], QUDA_MILC_GAUGE_ORDER as in qudaWrapperImpl.nim:216-240);
 *   - links passed to qexhip_stag_set_links already carry boundary conditions and
 *     staggered phases (Staggered.g, src/physics/stagD.nim:19-22,72-80); they are general
 *     3x3 complex matrices (smeared links are not unitary);
 *   - parity: 0 = "even", 1 = "odd", 2 = "all" (src/layout/layoutX.nim:285-295);
 *   - return value: 0 = ok, <0 = error, message from qexhip_last_error().  Not converging
 *     within maxits is NOT an error (src/solvers/cg.nim:174): iters == maxits is returned.
 *   - threading: entry points are called from one host thread (the master thread outside
 *     any `threads:` block, src/physics/stagSolve.nim:63,78); one context per GPU/rank.
 */
#ifndef QEXHIP_H
#define QEXHIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qexhip_ctx *qexhip_handle;

#define QEXHIP_OK 0
#define QEXHIP_ERR_ARG (-1)
#define QEXHIP_ERR_HIP (-2)
#define QEXHIP_ERR_STATE (-3)
#define QEXHIP_ERR_COMM (-4)
#define QEXHIP_ERR_IO (-5)

#define QEXHIP_EVEN 0
#define QEXHIP_ODD 1
#define QEXHIP_ALL 2

/* ---------------- context ----------------
 * Replaces qudaInit/qudaSetup (src/quda/qudaWrapperImpl.nim:70-123): bind one GPU, record the
 * rank-local lattice and the rank grid (src/layout/layoutX.nim:70-125).  Only a split of the
 * outermost dimension is supported: rankGeom = {1,1,1,N} (BASELINE.json north_star).
 * Preconditions on latLocal (checked here and at set_links; QEXHIP_ERR_ARG otherwise) -- narrower than QEX's "any even
 * lattice, any rankGeom" (src/layout/layoutX.nim:46-68,81-95):
 *   * every extent even (the even/odd layout);
 *   * sharded (rankGeom[3] > 1 or qexhip_comm_force_halo): X*Y*Z/2 a multiple of 64, so that a t-slice of one parity is a whole
 *     number of 64-site wavefront tiles and the faces are contiguous tile ranges (no pack kernels); local t even; local t
 *     >= 1 for the one-hop operator, >= 3 for Naik links (ghost depth 3);
 *   * unsharded: X*Y*Z*T/2 need not be a multiple of 64 (the last tile is ragged).
 * On any failure after the context was allocated everything built so far is released; *h is written only on success. */
int qexhip_init(qexhip_handle *h, int device, const int latLocal[4],
                const int rankGeom[4], const int rankCoord[4]);
int qexhip_finalize(qexhip_handle h);
/* number of HIP devices this process can bind (qudaInit takes the device from the caller too,
 * src/quda/qudaWrapperImpl.nim:70-83): a host picks device = (rank on its node) mod this count */
int qexhip_device_count(int *n);
const char *qexhip_last_error(void);
/* wait for all work queued on the context's streams */
int qexhip_sync(qexhip_handle h);
/* library / device description, for logs */
int qexhip_device_info(qexhip_handle h, char *buf, int buflen);

/* ---------------- communicator (RCCL over xGMI, or peer-mapped memory) ----------------
 * Replaces the QMP send/recv pairs of the shifts (src/layout/qshifts.nim:51-131) and
 * threadRankSum's QMP_sum (src/comms/commsUtils.nim:195-204, commsQmp.nim:127-128).
 * Rank 0 obtains an id, the host broadcasts it (QMP_broadcast in QEX), every rank calls init (collective; an id serves ONE
 * comm_init).  Two transports sit behind the same entry points (QEXHIP_TRANSPORT / option "transport", one value for the
 * whole job):
 *   rccl   ncclSend/ncclRecv groups and ncclAllReduce between DISTINCT devices; no rendezvous; works across nodes
 *   peer   every rank maps its neighbours' receive arenas and every rank's mailbox through hipIpc; a face exchange is a push into
 *          the neighbour's HBM (from inside the sweep kernel itself where sweeps overlap: option "hop_split"), an all-reduce one
 *          single-workgroup kernel that sums the ranks' mailboxes in rank order (bit-identical on every rank).  One node only
 *          (<= 16 ranks).  It is the only transport that lets several ranks share ONE device, which RCCL refuses
 *   mbox   RCCL for the faces, the mailboxes of `peer` for the scalar rank sums of the solvers (one single-workgroup kernel of ~4 us
 *          instead of an ncclAllReduce of 15-20 us, two per CG iteration; cg.nim:206-214); one node; reported as "rccl+mbox"
 *   auto   (default) the ranks of a ONE-NODE job meet in a POSIX shared-memory segment named after the id and take `peer` if any
 *          two of them are bound to the same device, else `mbox` -- provided the control blocks map between the devices and a
 *          self-test of the mailbox all-reduce (known answers, 5 s bound) passes on every rank; if not, all ranks drop to `rccl`
 *          together.  A job that SPANS NODES takes `rccl`: by what the launcher says (QEXHIP_LOCAL_RANKS, LOCAL_WORLD_SIZE,
 *          OMPI_COMM_WORLD_LOCAL_SIZE, MPI_LOCALNRANKS, SLURM_NTASKS_PER_NODE: fewer local ranks than nranks -> no rendezvous at
 *          all) or, with no such variable, when the node-local rendezvous does not complete within QEXHIP_RENDEZVOUS_TIMEOUT
 *          (s, default 120) -- every node's ranks time out alike, so the decision stays one for the whole job; only an explicit
 *          `peer` / `mbox` turns that timeout into QEXHIP_ERR_COMM.  QEXHIP_PEER_TIMEOUT (s, default 30) bounds every device-side
 *          wait: a rank that never arrives becomes QEXHIP_ERR_COMM on the others, never a hang. */
#define QEXHIP_UNIQUE_ID_BYTES 128
int qexhip_comm_unique_id(char id[QEXHIP_UNIQUE_ID_BYTES]);
int qexhip_comm_init(qexhip_handle h, const char id[QEXHIP_UNIQUE_ID_BYTES], int nranks, int rank);
/* What RCCL reports for the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice) and the PCI bus id of
 * the bound GPU -- QEX prints the same facts from QMP at start-up (src/comms/commsQmp.nim:14-33: rank, size).
 * Without a communicator nranks = 0, rank = -1.  Any output pointer may be NULL.
 * A context with rankGeom[3] > 1 refuses every exchange / reduction until qexhip_comm_init has run (QEXHIP_ERR_STATE). */
int qexhip_comm_info(qexhip_handle h, int *nranks, int *rank, int *device, char *busid, int buslen);
/* which transport the communicator runs on: name = "none" | "rccl" | "rccl+mbox" | "peer"; stats (may be NULL) = peer-transport counters
 * {face exchanges posted, all-reduces posted, arena (re)allocations, bytes of receive arena}, zeros otherwise */
int qexhip_comm_transport(qexhip_handle h, char *name, int len, long stats[4]);
/* number of RCCL communicators the context holds: 0 before comm_init, 2 afterwards (one for the compute stream's all-reduces
 * and ghost refreshes, one -- ncclCommSplit of the first -- for the face exchanges posted on the second stream beside the
 * interior sweep, so that neither queues behind the other), 1 with QEXHIP_COMM2=0.  Peer transport: 2 (the two streams'
 * channels are independent by construction). */
int qexhip_comm_count(qexhip_handle h, int *ncomms);
/* How a one-parity sweep runs on this context's (t-sharded) field: out[0] = 1 if t-hops across the slab boundary go through
 * ghost zones, out[1] = 1 if the face exchange runs beside interior work (second stream, or inside the fused launch), out[2] = interior
 * sites of one parity, out[3] = bytes of one face message, out[4] = 1 if out[1] was MEASURED: with a communicator of more
 * than one rank, set_links times a few sweeps in every form (collective; the slowest rank decides, so all ranks agree) --
 * out[5], out[6] = microseconds per sweep it saw exchange-first / overlapped and split by sites (0: that form was not timed;
 * qexhip_stag_sweep_tuning has all three forms); otherwise out[1] follows option "overlap" or, at -1, the rule of the one-rank
 * rehearsals (overlap when the interior is >= 131072 sites and a face >= 1 MiB); out[7] = the option's value.  bench.py prints it
 * so that a scaling run explains its own launch structure. */
int qexhip_stag_sweep_info(qexhip_handle h, int out[8]);
/* ... and what set_links MEASURED for them (collective, max over ranks; qexhip_device_info prints the same):
 *   out[0] one face exchange of this operator, microseconds (measured where the communicator has more than one rank or option
 *          "overlap" = -2 asked for it; otherwise the estimate 3 us + face bytes at 45 GB/s per xGMI direction)
 *   out[1] where in its dispatch order the fused sweep puts its boundary workgroups (0.65 .. 1: out[0] over the estimated interior time)
 *   out[2..4] microseconds per sweep in the three forms: exchange first | overlapped, split by sites | fused (0: not measured / not available)
 *   out[5] the form an overlapped sweep takes: 2 fused, 0 split by sites        out[6] the fused sweep's short wait, us (-1: parks always)
 *   out[7] 1 if sweeps overlap at all
 * The decisions are timing-dependent and regroup the dot-product partials (the fused form also sums a boundary site's local hops
 * first): a job that needs its residual history bit-reproducible from run to run pins "overlap" and "hop_split". */
int qexhip_stag_sweep_tuning(qexhip_handle h, double out[8]);
/* test hook: with one rank, route the t-direction hops through the halo path
 * (pack -> RCCL self send/recv -> boundary sweep) instead of the periodic wrap. */
int qexhip_comm_force_halo(qexhip_handle h, int on);

/* ---------------- QEX's SIMD field memory <-> the host format of this header ----------------
 * A QEX Field[V,T] is double[outer][...T...][re|im][V lanes] with site i = outer*V + lane (src/field/fieldET.nim:18-22,123-128);
 * which lattice site that is follows LayoutQ (src/layout/qlayout.nim:10-66 set-up, :110-131 index, :133-185 coordinates).
 * Everything below takes the RANK-LOCAL geometry and the inner (SIMD) geometry of the Layout[V] (l.localGeom, l.innerGeom) and
 * the field's own memory, and produces / consumes the V = 1 even-odd site-major arrays every other entry point of this header
 * takes ([site][3][2], [site][4][3][3][2]) -- the per-site copy loops of the QUDA bridge (src/quda/qudaWrapperImpl.nim:198-260:
 * r.l.coord -> lo1.rankIndex) as one call.  Pure host code, no GPU needed. */
/* the inner geometry newLayoutX picks for V lanes when none is given (src/layout/layoutX.nim:19-42,98-111); {1,2,2,2} for V = 8
 * on lattices whose y, z, t extents are multiples of 4; QEXHIP_ERR_ARG where QEX itself gives up ("can't lay out inner geom") */
int qexhip_layout_default_inner(const int localGeom[4], int V, int innerGeom[4]);
/* v1_of_simd[outer*V + lane] = index of that site in the V = 1 even-odd order (Layout[1].rankIndex(Layout[V].coord(i))) */
/* (all of these refuse an ODD local extent: the even/odd split of a QEX field then depends on the rank origin, which they are not given) */
int qexhip_layout_simd_map(const int localGeom[4], const int innerGeom[4], int *v1_of_simd);
/* colour vector: simd = double[outer][3][2][V] */
int qexhip_layout_vec_simd_to_v1(const int localGeom[4], const int innerGeom[4], const double *simd, double *v1);
int qexhip_layout_vec_v1_to_simd(const int localGeom[4], const int innerGeom[4], const double *v1, double *simd);
/* gauge field: g[mu] = double[outer][3][3][2][V], one QEX field per direction (s.g[mu]; s.g[2 mu] / s.g[2 mu + 1] for the fat /
 * long links of a Naik operator, stagD.nim:552-564: the caller picks the four pointers) */
int qexhip_layout_gauge_simd_to_v1(const int localGeom[4], const int innerGeom[4], const double *const g[4], double *v1);
int qexhip_layout_gauge_v1_to_simd(const int localGeom[4], const int innerGeom[4], const double *v1, double *const g[4]);

/* ---------------- staggered operator ----------------
 * Staggered.g  (src/physics/stagD.nim:19-22; newStag :522-541, newStag3 :543-564).
 * fat: 4 links per site; lng: NULL (plain) or 4 three-hop links per site (Naik). */
int qexhip_stag_set_links(qexhip_handle h, const double *fat, const double *lng);

/* stagD2 (src/physics/stagD.nim:349-395):
 *   r[parity] = a*r + b*x + sum_mu [ U_mu(s) x(s+mu) - U_mu^+(s-mu) x(s-mu) ]  (= a r + b x + 2D x) */
int qexhip_stag_dslash(qexhip_handle h, double *r, const double *x, int parity, double a, double b);

/* stagD (src/physics/stagD.nim:406-409) on both parities = Staggered.D (sc=+1, :566-568) and
 * Staggered.Ddag (sc=-1, :569-571):  r = m*x + sc*D*x */
int qexhip_stag_D(qexhip_handle h, double *r, const double *x, double m, double sc);

/* stagD itself (src/physics/stagD.nim:406-409) on ONE subset (parity QEXHIP_EVEN / ODD / ALL): r[subset] = a*r + m*x + sc*D*x,
 * the rest of r kept.  stagDb (:425-427, no final scale) is qexhip_stag_dslash(h, r, x, parity, 0, m / (0.5 sc)). */
int qexhip_stag_stagD(qexhip_handle h, double *r, const double *x, int parity, double m, double sc, double a);

/* stagD with the accumulate coefficient: r = a*r + m*x + sc*D*x on both parities; a = 1, sc = -1 is
 * Staggered.peqDdag (src/physics/stagD.nim:572-574) */
int qexhip_stag_D_acc(qexhip_handle h, double *r, const double *x, double m, double sc, double a);

/* stagD2ee / stagD2oo (src/physics/stagD.nim:434-469): r[par] = 4 m2 x - (2D_eo)(2D_oe) x */
int qexhip_stag_op_xx(qexhip_handle h, double *r, const double *x, double m2, int par_even);

/* eoReconstruct (src/physics/stagD.nim:583-586): r.odd = (b.odd - D_oe r.even)/m, r.even kept */
int qexhip_stag_eo_reconstruct(qexhip_handle h, double *r, const double *b, double m);

/* eoReduce (src/physics/stagD.nim:575-581): r.even = (D^+ b).even = (m b - D b).even, r.odd kept */
int qexhip_stag_eo_reduce(qexhip_handle h, double *r, const double *b, double m);

/* Shifted outer product of the fermion force:
 *   f[mu](s) (+)= scale(parity of s) * x(s) (x) x(s+mu)^+     f: double[vol][4][3][3][2]
 * stagDeriv (src/physics/stagD.nim:634-664) is scale_even = +1, scale_odd = -1, accumulate = 1 (the
 * caller then rephases f); the loop of fforce (src/stagg_pv_hmc/staghmc_spv.nim:831-854) is
 * scale_even = scale_odd = scale with accumulate = 0 for the first field and 1 afterwards. */
int qexhip_stag_outer(qexhip_handle h, double *f, const double *x, double scale_even, double scale_odd,
                      int accumulate);

/* ---------------- solvers ----------------
 * solveEE / solveOO = solveXX (src/physics/stagSolve.nim:57-138), the backend seam next to
 * sbQuda (:105-118 -> qudaSolveEE/OO, src/quda/qudaWrapperImpl.nim:263-267).
 * On `par_even ? even : odd` sites solve  4(m^2 - D_eo D_oe) x = b  from x = 0 with the CG of
 * src/solvers/cg.nim:55-272; stop when |r|^2 <= r2req*|b_par|^2 or after maxits iterations.
 * x is zeroed on both parities first (stagSolve.nim:63-64).
 *   iters          <- sp.iterations (cg.nim:271)
 *   r2_over_b2     <- final recursive |r|^2/|b|^2
 *   hist[k]        <- |r|^2/|b|^2 after iteration k (k = 0 initial), the values of the
 *                     "CG iteration: k  r2/b2:" log lines (cg.nim:172,215-217); at most histcap. */
int qexhip_stag_solve_xx(qexhip_handle h, double *x, const double *b, double mass, double r2req,
                         int maxits, int par_even, int *iters, double *r2_over_b2,
                         double *hist, int histcap);

/* Staggered.solve (src/physics/stagSolve.nim:224-294): full-lattice D x = b by even-odd
 * preconditioning with the outer true-residual restart loop, x starts from 0.
 *   iters <- total CG iterations, r2_final <- |b - D x|^2/|b|^2 (sp.r2) */
int qexhip_stag_solve(qexhip_handle h, double *x, const double *b, double mass, double r2req,
                      int maxits, int *iters, double *r2_final);

/* the same with sp.usePrevSoln (src/physics/stagSolve.nim:234-243): with use_prev != 0 the solve
 * starts from the x handed in (r = b - D x) instead of x = 0 */
int qexhip_stag_solve_prev(qexhip_handle h, double *x, const double *b, double mass, double r2req,
                           int maxits, int use_prev, int *iters, double *r2_final);

/* multi-shift solveXX (src/physics/stagSolve.nim:296-345 + src/solvers/cgm.nim:84-315).
 * shifts[0] = base mass, shifts[k>0] = sigma_k added to m0^2; xs[k] full-volume vectors. */
int qexhip_stag_solve_xx_multi(qexhip_handle h, double *const *xs, const double *b,
                               const double *shifts, int nmass, double r2req, int maxits,
                               int par_even, int *iters, double *hist, int histcap);
/* Staggered.solve(xs, b, ms, sp) (src/physics/stagSolve.nim:347-446) */
int qexhip_stag_solve_multi(qexhip_handle h, double *const *xs, const double *b,
                            const double *masses, int nmass, double r2req, int maxits,
                            int *iters, double *r2_final);

/* ---------------- field algebra hooks ----------------
 * norm2 / redot with fp64 accumulation (src/field/fieldET.nim:605-625,704-724) and the
 * elementwise updates CG uses (fieldET.nim:547-598).  On host vectors; for tests. */
int qexhip_norm2(qexhip_handle h, const double *x, int parity, double *out);
int qexhip_redot(qexhip_handle h, const double *x, const double *y, int parity, double *out);
/* dotP (src/field/fieldET.nim:677-693): the complex inner product sum_s x(s)^+ y(s); out[0] = Re, out[1] = Im */
int qexhip_dot(qexhip_handle h, const double *x, const double *y, int parity, double out[2]);
/* y[parity] += a*x */
int qexhip_axpy(qexhip_handle h, double a, const double *x, double *y, int parity);
/* y[parity] = x + a*y */
int qexhip_xpay(qexhip_handle h, const double *x, double a, double *y, int parity);

/* ---------------- device-resident fields ----------------
 * QEX re-uploads per call through the QUDA seam; these keep vectors in HBM between calls
 * (the "set once per trajectory" extension of SURVEY.md 8b "Ownership"). Fields are
 * full-volume colour vectors identified by small integer ids. */
int qexhip_field_new(qexhip_handle h, int *id);
int qexhip_field_free(qexhip_handle h, int id);
int qexhip_field_upload(qexhip_handle h, int id, const double *host);
int qexhip_field_download(qexhip_handle h, int id, double *host);
int qexhip_field_zero(qexhip_handle h, int id);
/* asynchronous on the context stream; qexhip_sync() to wait */
int qexhip_dev_dslash(qexhip_handle h, int r_id, int x_id, int parity, double a, double b);
int qexhip_dev_op_xx(qexhip_handle h, int r_id, int x_id, double m2, int par_even);
/* solveXX on resident fields; blocks until finished. */
int qexhip_dev_solve_xx(qexhip_handle h, int x_id, int b_id, double mass, double r2req,
                        int maxits, int par_even, int *iters, double *r2_over_b2,
                        double *hist, int histcap);
/* Re-entry of the CG on the state the last qexhip_dev_solve_xx (or re-entry) on x_id left behind -- CgState.solve called
 * again with b2 >= 0 (src/solvers/cg.nim:21-27,85,133,155-161,256-261): no set-up, r / p / rzold kept, the stopping
 * criterion comes from the new r2req / maxits, `iters` goes on counting from where the last call stopped (maxits is the
 * cumulative limit, as sp.maxits is there).  QEXHIP_ERR_STATE if anything that uses the CG's work vectors or changes the
 * operator ran in between.  hist (may be NULL) receives the whole history from iteration 0. */
int qexhip_dev_solve_xx_continue(qexhip_handle h, int x_id, double r2req, int maxits, int *iters, double *r2_over_b2,
                                 double *hist, int histcap);
/* multi-shift solveXX (Staggered.solveXX(xs, b, ms, sp), src/physics/stagSolve.nim:296-345) on resident fields:
 * x_ids[k] receives the solution of shift k; shifts as qexhip_stag_solve_xx_multi.  Blocks until finished. */
int qexhip_dev_solve_xx_multi(qexhip_handle h, const int *x_ids, int b_id, const double *shifts, int nmass,
                              double r2req, int maxits, int par_even, int *iters, double *hist, int histcap);

/* Free the multi-shift solvers' persistent workspace (up to 3 x nmass full fields kept between solves; QEX allocates its
 * ps / ys per call with newOneOf and leaves them to the GC, src/solvers/cgm.nim:120-131, src/physics/stagSolve.nim:376-381).
 * The next multi-shift solve allocates it again.  Also frees the double-link field of the rectangle force / Symanzik flow
 * (one more field the size of the links, rebuilt at the next such call). */
int qexhip_release_workspace(qexhip_handle h);
/* norm2 / redot (src/field/fieldET.nim:605-625,704-724; rank-global sums) and Staggered.D / Ddag (r = m x + sc D x, r_id != x_id)
 * on resident fields: what a caller that keeps its vectors in HBM uses for true residuals and solution norms. */
int qexhip_dev_norm2(qexhip_handle h, int x_id, int parity, double *out);
int qexhip_dev_redot(qexhip_handle h, int x_id, int y_id, int parity, double *out);
int qexhip_dev_dot(qexhip_handle h, int x_id, int y_id, int parity, double out[2]);     /* dotP on resident fields */
int qexhip_dev_D(qexhip_handle h, int r_id, int x_id, double m, double sc);
/* r[parity] := 0 (the `phi.odd := 0` of the pseudofermion refresh, staghmc_sh.nim:748-755) */
int qexhip_dev_zero(qexhip_handle h, int id, int parity);
/* n x Staggered.solve (qexhip_stag_solve semantics per system) on resident fields, lock-step batches of four on the
 * operator's current links: faction / pbp of the HMC drivers without moving a vector (staghmc_sh.nim:260-272,339-364) */
int qexhip_dev_solve_batch(qexhip_handle h, int n, const int *x_ids, const int *b_ids, const double *mass,
                           const double *r2req, int maxits, int *iters, double *r2);

/* ---------------- gauge field, plaquette, Wilson flow ----------------
 * qexhip_gauge_set/get: the `g` of src/gauge/wflow.nim:21 (unphased links, periodic). */
int qexhip_gauge_set(qexhip_handle h, const double *g);
int qexhip_gauge_get(qexhip_handle h, double *g);
/* plaq (src/gauge/gaugeUtils.nim:213-282): six values, index mu(mu-1)/2+nu, each /(V*6*nc) */
int qexhip_plaq(qexhip_handle h, double out[6]);
/* gaugeForce with GaugeActionCoeffs(plaq: cplaq) (src/gauge/gaugeAction.nim:334-350):
 * f_mu(x) = TAH( U_mu(x) * [ (cplaq/nc) sum_staples ]^+ ), written to host f */
int qexhip_gauge_force(qexhip_handle h, double *f, double cplaq);
/* gaugeFlow(steps, eps) (src/gauge/wflow.nim:21-67): RK3 Wilson flow of the resident gauge field */
int qexhip_wflow(qexhip_handle h, int nsteps, double eps);
/* The action-selectable variants used by the fork's flow driver (src/flow/flow.nim:22-90):
 *   kind 0 ("Wilson" | "rect"): GaugeActionCoeffs(plaq: cplaq, rect: c2), gaugeForce / gaugeActionDeriv
 *           incl. the rectangle part (src/gauge/gaugeAction.nim:148-350);
 *   kind 1 ("adj"):             GaugeActionCoeffs(plaq: cplaq, adjplaq: c2), forceA / gaugeADeriv (:683-747). */
int qexhip_gauge_force_general(qexhip_handle h, double *f, double cplaq, double c2, int kind);
int qexhip_wflow_general(qexhip_handle h, int nsteps, double eps, double cplaq, double c2, int kind);
/* EQ of the flow drivers (src/flow/gauge_flow.nim:360-379, tests/base/twflow_topo.nim:4-10):
 * out = {E_s, E_t, Q} from f = g.fmunu(loop), f.densityE, f.topoQ
 * (src/gauge/gaugeUtils.nim:1162-1271); loop in {1,3,4,5} selects the clover improvement
 * (1x1 | +2x2+3x3 | +2x2+1x2+1x3 | all five loop shapes, coefficients of :1128-1146). */
int qexhip_flow_EQ(qexhip_handle h, int loop, double out[3]);
/* What a flow loop measures after every step (src/flow/gauge_flow.nim:139-156,360-379; tests/base/twflow_topo.nim:4-10:
 * `g.plaq` and `EQ` = fmunu(1) -> densityE, topoQ) in ONE pass over the resident links: the plaquette of a plane is the
 * trace of one of the four clover leaves of that plane, so plaq[6] (as qexhip_plaq) and EQ[3] (as qexhip_flow_EQ with
 * loop = 1) come out of the same kernel. */
int qexhip_flow_measure(qexhip_handle h, double plaq[6], double EQ[3]);
/* The remaining gauge-sector pieces an HMC trajectory needs, on the resident field (set with qexhip_gauge_set):
 *   action: gc.gaugeAction1(g) (crect) / gc.actionA(g) (cadjplaq) (src/gauge/gaugeAction.nim:61-142,614-681),
 *           coefficients as GaugeActionCoeffs(plaq, rect | adjplaq); at most one of crect, cadjplaq non-zero
 *   update: mdt, g[mu][s] := exp(t p[mu][s]) g[mu][s] (src/examples/staghmc_sh.nim:429-435); p host, [vol][4][3][3][2]
 *   reunit: g.projectSU (src/gauge/gaugeUtils.nim:1333-1334; `reunit` of the HMC examples, staghmc_sh.nim:247-258)
 *   wline:  g.wline(path) (gaugeUtils.nim:1079-1112): volume- and colour-averaged trace, out = {re, im}; path entries
 *           +-(mu+1); the Polyakov loops of `ploop` are path = [mu+1] * L_mu (staghmc_sh.nim:281-291).  On a t-sharded
 *           field: paths that stray at most 3 slices in t, or the straight line [+-4] * L_t (global) */
int qexhip_gauge_action(qexhip_handle h, double cplaq, double crect, double cadjplaq, double *out);
int qexhip_gauge_update(qexhip_handle h, const double *p, double t);
int qexhip_gauge_reunit(qexhip_handle h);
int qexhip_wline(qexhip_handle h, const int *path, int n, double out[2]);
/* the four Polyakov loops wline([mu+1] * L_mu), mu = 0..3, of the resident field in one call: what `meas_ploop`
 * (src/flow/gauge_flow.nim:137-156) and `ploop` (src/examples/staghmc_sh.nim:281-291) compute with four g.wline calls;
 * out[2 mu], out[2 mu + 1] = Re, Im (normalised like qexhip_wline: trace / 3, lattice average) */
int qexhip_polyakov_loops(qexhip_handle h, double out[8]);
/* `s4_gauge` of the fork's measurements (src/stagg_pv_hmc/staghmc_spv_meas.nim:25-65; the pure-gauge S4 order parameter of
 * arXiv:1111.2317, printed as "MEASplaq <dir>-dir even/odd"): the plaquette of plane (mu, nu) at x is added to peo[mu][x_mu mod 2]
 * and peo[nu][x_nu mod 2]; out[2 d + eo] = peo[d][eo] / (physVol * 0.5 * (nd - 1) * nc), of the resident field */
int qexhip_plaq_s4(qexhip_handle h, double out[8]);

/* ---------------- link construction upstream of the solver ----------------
 * g, fl, ll: double[vol][4][3][3][2].
 * qexhip_fat7: makeImpLinks (src/gauge/fat7l.nim:77-161), coef = {oneLink, threeStaple, fiveStaple,
 *   sevenStaple, lepage} (Fat7lCoefs :5-10); ll (nullable) <- naik * U U U.
 * qexhip_hisq_smear: HisqCoefs.init + smear (src/physics/hisqLinks.nim:9-43): fat7 -> projectU ->
 *   fat7 + Naik; the input already carries BC + staggered phases (tests/examples/testStagProp.nim:24-33).
 * qexhip_nhyp_smear: the forward part of HypCoefs.smear (src/gauge/hypsmear.nim:49-144,260-275). */
int qexhip_fat7(qexhip_handle h, const double *g, const double coef[5], double *fl, double *ll, double naik);
int qexhip_hisq_smear(qexhip_handle h, const double *g, double *fl, double *ll);
int qexhip_nhyp_smear(qexhip_handle h, const double *g, double *fl, double alpha1, double alpha2, double alpha3);
/* The chain rule through those link constructions (HISQ molecular dynamics):
 *   fat7_deriv: fat7lDeriv (src/gauge/fat7lderiv.nim): d = d/dU^+ of sum Re tr(dfl^+ fl(g)) + sum Re tr(dll^+ ll(g)) for
 *               (fl, ll) = makeImpLinks(g, coef, naik); dll NULL = no long links
 *   hisq_force: HisqCoefs.smearGetForce's smearedForce(dsdu, dsdsu, dsdsul) (src/gauge/hisqsmear.nim:55-90): second fat7 +
 *               Naik, projectUderiv, first fat7, in reverse; dsdsu / dsdsul = the action's derivative w.r.t. the fat / long links */
int qexhip_fat7_deriv(qexhip_handle h, const double *g, const double *dfl, const double coef[5], const double *dll, double naik, double *d);
int qexhip_hisq_force(qexhip_handle h, const double *g, const double *dsdsu, const double *dsdsul, double *f);
/* the same as the closure the reference returns (hisqsmear.nim:55-90: it retains u and the intermediate v, w):
 *   prepare: smear g, keep u, v = fat7_1(u), w = projectU(v) and the smeared su, sul on the device; fl / ll (nullable) receive
 *            su / sul; qexhip_stag_set_links_hisq(h, NULL) afterwards hands su, sul to the operator without another smearing
 *   closure_force: smearedForce(dsdu, dsdsu, dsdsul), the reverse pass only;  release: drop the state */
int qexhip_hisq_prepare(qexhip_handle h, const double *g, double *fl, double *ll);
int qexhip_hisq_closure_force(qexhip_handle h, const double *dsdsu, const double *dsdsul, double *f);
int qexhip_hisq_release(qexhip_handle h);
/* fermionForce of the HISQ HMC (src/examples/hisqhmc.nim:496-541) through the closure (prepared with the PHASED links, as
 * smearRephase does, :407-412): f1 = sum_k scale[k] p_k(x) (x) p_k(x+mu)^+, f3 the same with x+3mu, odd sites *= -1,
 * smearedForce(ff, f1, f3), f = TAH(ff u^+) */
int qexhip_hisq_fermion_force(qexhip_handle h, double *f, const double *const *psi, const double *scale, int n);

/* n (1..4) independent systems on the SAME links solved in lock-step, the links streamed once per sweep for all of
 * them (the Dslash is HBM-bound and 89 % of its bytes are links).  This is how the back-to-back solves of QEX's HMC
 * are meant to be issued: the Hasenbusch chain of faction / fforce (src/examples/staghmc_sh.nim:339-364,394-404), the
 * pbp repetitions (:260-272), the fork's fforce loop (src/stagg_pv_hmc/staghmc_spv.nim:758-830).
 *   solve_xx_batch: n x solveXX (qexhip_stag_solve_xx semantics per system: own mass, r2req, iteration count)
 *   solve_batch:    n x Staggered.solve (qexhip_stag_solve semantics per system)
 * Each system's arithmetic is that of the single-system call, so solutions and iteration counts are the same.
 * x, b: arrays of n host fields [vol][3][2].  Works t-sharded as well (faces of all systems exchanged per sweep, one
 * all-reduce for the n scalars of a reduction). */
int qexhip_stag_solve_xx_batch(qexhip_handle h, int n, double *const *x, const double *const *b, const double *mass,
                               const double *r2req, int maxits, int par_even, int *iters, double *r2_over_b2);
int qexhip_stag_solve_batch(qexhip_handle h, int n, double *const *x, const double *const *b, const double *mass,
                            const double *r2req, int maxits, int *iters, double *r2_over_b2);

/* Storage format the library chose for the operator's links at the last set_links call.  A unitary link is fixed
 * by rows 0,1 and its determinant (row2 = det * conj(row0 x row1)), and the sweep is HBM-bound, so:
 *   1: every link is SU(3) up to a sign (thin links with BC + staggered phases): rows 0,1 + a sign bit, 96 B/link
 *   2: every link is U(3) (nHYP-smeared links): rows 0,1 + det, 112 B/link
 *   0: otherwise (HISQ fat links; QEX's `random` start, 1.3 % of whose links are further than 5e-14 from unitary, the
 *      worst 2.5e-9): all 18 reals, 144 B/link
 * chosen only if ALL links reproduce their stored row 2 to 5e-14 (a few hundred ulp: exactly unitary links of a 48^3x96 lattice peak at ~2e-14); max_dev = the largest deviation found for the
 * chosen format.  Row 2 is rebuilt in registers.  QEXHIP_RECON=0|1|2 caps the format.  No counterpart in QEX (its
 * CPU Dslash always reads full links, stagD.nim:349-395); QUDA's reconstruct-12/13 is the precedent. */
int qexhip_stag_links_info(qexhip_handle h, int *nlinks, int *compressed, double *max_dev);
/* Options of a context.  Unknown names are an error (QEXHIP_ERR_ARG).
 *   "recon"        cap on the link compression (0 keep all 18 reals, 1 sign format only, 2 also the U(3) format; default 2),
 *                  effective at the next set_links
 *   "overlap"      face exchange on the second stream beside the interior sweep: 0 never, 1 always, -1 (default) measured at
 *                  set_links when the communicator has more than one rank (qexhip_stag_sweep_info), else by interior / face
 *                  size; -2: measure on one rank too (test hook)
 *                  0 / 1 also pin the launch structure, and with it the bits of a sharded residual history, from run to run
 *                  (the measurement takes the overlapped form only on a > 5 % win).  Must be the same on every rank (checked
 *                  at set_links)
 *   "transport"    before qexhip_comm_init: 0 auto, 1 rccl, 2 peer, 3 mbox (see "communicator"); the same on every rank
 *   "flow_exp"     1: closed-form exp(v) in the Wilson-flow stage (default; agrees with the reference's to ~1e-15 per element),
 *                  0: the reference's algorithm, order-4 Taylor + 20 squarings (matexp.nim:80-85,634-649)
 *   test hooks -- each selects, on any lattice, the code path that some lattices / ranks take by necessity:
 *   "multi_reduce" 1: the sharded reduction branches (all-reduce of the partial vectors) on one rank
 *   "batch_multi"  1: the same for the lock-step multi-system CG
 *   "smear_ca"     0: the nHYP levels of a t-sharded field refresh the ghost slices of every projected level field (rounds 1-4) instead of
 *                  computing them on shrinking ghost slices from one depth-3 thin-link exchange (default 1)
 *   "hop_split"    how an OVERLAPPED sweep of a t-sharded field is laid out on the peer transport (RCCL always runs 0).
 *                  2: the FUSED sweep, one kernel on one stream (shifts.nim:67-94,254-285: local terms while the faces travel, boundary
 *                  terms when they are in): its first workgroups push the faces into the neighbours' receive arenas, the interior
 *                  workgroups take every hop of their sites, the boundary workgroups take the hops inside the slab, wait SHORTLY (about
 *                  one measured exchange time) for the inbound data words and then either take the 1-2 hops per site that leave the slab
 *                  straight from the arena, or -- faces late -- park their raw accumulator and give up their slot; the last <= 32
 *                  workgroups of the grid finish the parked blocks behind the one LONG bounded wait (QEXHIP_PEER_TIMEOUT: a lost
 *                  neighbour).  No kernel holds more than those 32 slots hostage to another rank's progress -- ranks may share a chip.
 *                  0: by SITES (interior launch beside the exchange, boundary launch on the second stream behind it, device-side join).
 *                  -1 (default): whichever set_links measured faster (qexhip_stag_sweep_tuning), fused until measured.  Boundary sites
 *                  sum their local hops first under 2, parked or not: equal to 0 to rounding, and to itself to the bit.
 *   "fused_spin_us" the fused sweep's short wait: -1 (default) max(25 us, one exchange time), >= 0 microseconds, -2 park every boundary
 *                  block (test hook: the cleanup path everywhere)
 *   "chain_overlap" 0: the nHYP force chain of a t-sharded field exchanges a level's chain fields first and runs the next staple
 *                  derivative in one pass, instead of running its ghost-free slices beside the exchange (default 1; bit-identical)
 *   "force_pair"   0: k_force_lds (one tile and parity per workgroup), what shapes without paired tile positions run
 *   "obs_clover"   0: the generic path walker, what fmunu loops 3-5 run, for the clover loop as well
 *   "emu_exchange_us", "emu_allreduce_us"   transport emulation for one-GPU rehearsals: every face exchange / all-reduce is preceded,
 *                  on its stream, by a wait of that many microseconds -- the time the transfer would take between distinct
 *                  GPUs.  Results must not depend on it (a consumer that does not wait for its ghosts would show); iteration
 *                  times under it are what bench.py --halo --emulate-transport reports
 *
 * Environment (read once, at qexhip_init / qexhip_comm_init) -- the complete list:
 *   QEXHIP_RECON, QEXHIP_OVERLAP, QEXHIP_HOP_SPLIT, QEXHIP_FLOW_EXP   initial values of the options of the same (lower-case) name
 *   QEXHIP_TRANSPORT=auto|rccl|peer|mbox, QEXHIP_RENDEZVOUS_TIMEOUT, QEXHIP_PEER_TIMEOUT, QEXHIP_LOCAL_RANKS   see "communicator"
 *   QEXHIP_TEST_FAIL_INIT, QEXHIP_TEST_FAIL_MBOX   test hooks: inject a late failure into qexhip_init / into comm_init's mailbox self-test
 *                    (1: it fails; 2: and an explicit `mbox` wish falls back to rccl like `auto`), so that the clean-up and fall-back
 *                    paths run on a one-GPU box (tests/test_gpu_misc_ops.py)
 *   QEXHIP_COMM2=0   keep ONE RCCL communicator for both streams (default: the overlapped face exchange gets a communicator
 *                    of its own); the ranks agree on this by a min-all-reduce, any rank's 0 wins
 * Every other choice the kernels make (visiting orders, non-temporal accesses, LDS staging, launch shapes) is fixed to the
 * variant that won its A/B measurement on MI355X (profiles/); the losers are not in the library. */
int qexhip_set_option(qexhip_handle h, const char *name, int value);

/* Smear on the device and hand the result straight to the operator (replaces smear -> rephase ->
 * set_links without moving the smeared links over PCIe):
 *   hisq: Staggered.g <- HisqCoefs.smear(g) (fat + long); g carries BC + phases already
 *         (tests/examples/testStagProp.nim:24-40: g.setBC; g.stagPhase; hc.smear(g, fl, ll); newStag3(fl, ll))
 *   nhyp: Staggered.g <- rephase(HypCoefs.smear(g)) with the fork's per-direction boundary flags
 *         (src/stagg_pv_hmc/staghmc_spv.nim:367-401,601-604): antiperiodic[mu] != 0 flips U_mu on the last
 *         slice of direction mu (NULL: t only, as gaugeUtils.nim:124-131); phases NULL = {8,9,11,0}.
 *         g == NULL: take the links the closure of qexhip_nhyp_prepare already smeared (the alphas are
 *         ignored), which is what smearRephase does (src/examples/staghmc_sh.nim:303-312). */
int qexhip_stag_set_links_hisq(qexhip_handle h, const double *g);
int qexhip_stag_set_links_nhyp(qexhip_handle h, const double *g, double alpha1, double alpha2, double alpha3,
                               const int antiperiodic[4], const int phases[4]);

/* nHYP smeared-force chain = the closure smearGetForce returns (src/gauge/hypsmear.nim:49-247):
 *   prepare: smear g (alphas as HypCoefs) and keep the 12+12+12+12+4 intermediate link fields on the
 *            device; fl (nullable) receives the smeared links.  Replaces `coef.smearGetForce(gf, fl, info)`
 *            (src/stagg_pv_hmc/staghmc_spv.nim:989-993).  The state refers to THIS g.
 *   force:   smearedForce(f, chain): chain = dS/dV^+ w.r.t. the smeared links ([vol][4][3][3][2]),
 *            f = dS/dU^+ w.r.t. the thin links, through projectUderiv + symStapleDeriv
 *            (src/maths/matrixFunctions.nim:329-357, src/gauge/smearutil.nim:22-50); f may alias chain
 *            (the fork calls f.smeared_force(f), staghmc_spv.nim:740).
 *   release: drop the closure (QEX: GC of the closure, hypsmear.nim:249-263).
 *   prepare with g == NULL smears the links resident on the device (qexhip_gauge_set / qexhip_md_begin); the three MD
 *   forces below with f == NULL leave their result on the device for qexhip_md_kick / qexhip_md_shift_links (source 1). */
int qexhip_nhyp_prepare(qexhip_handle h, const double *g, double alpha1, double alpha2, double alpha3, double *fl);
int qexhip_nhyp_force(qexhip_handle h, double *f, const double *chain);
int qexhip_nhyp_release(qexhip_handle h);
/* The two MD forces of the fork's HMC that run through the closure, start to finish on the device:
 *   gauge:   gforce(act, g, sg, f, smear_force) (src/stagg_pv_hmc/staghmc_spv.nim:217-228): derivative of the
 *            gauge action on the SMEARED links (gaugeForceCust / forceACust, staghmc_spv_gforce.nim:17-253;
 *            coefficients as qexhip_gauge_force_general), smearedForce, projTAH(g, "adj") = TAH(g f^+).
 *   fermion: fforce + smeared_one_link_force (staghmc_spv.nim:716-865): f = sum_k scale[k] psi_k(s) (x) psi_k(s+mu)^+
 *            (psi_k full-volume vectors, host), f.rephase (setBC_cust + stagPhase), odd sites *= -1,
 *            smearedForce, projTAH(gf) = TAH(f g^+).  antiperiodic / phases as qexhip_stag_set_links_nhyp. */
int qexhip_nhyp_gauge_force(qexhip_handle h, double *f, double cplaq, double crect, double cadjplaq);
/*   fforce, solves included (src/examples/staghmc_sh.nim:387-427): psi_k = D(mass[k])^-1 phi_k for the n pseudofermion
 *            fields (Staggered.solve semantics, lock-step batches of four on the operator's CURRENT links -- set them
 *            from this closure first: qexhip_stag_set_links_nhyp(h, NULL, ...)), then the fermion force above with
 *            scale[k] = fscale(k, i, t).  Solutions never leave the GPU; iters[k] (nullable) = iterations of system k. */
int qexhip_nhyp_fforce(qexhip_handle h, double *f, int n, const double *const *phi, const double *mass, const double *scale,
                       const double *r2req, int maxits, const int antiperiodic[4], const int phases[4], int *iters);
int qexhip_nhyp_fermion_force(qexhip_handle h, double *f, const double *const *psi, const double *scale, int n,
                              const int antiperiodic[4], const int phases[4]);
/*   the same with the pseudofermion fields already resident (field ids): nothing but scalars crosses PCIe when f == NULL */
int qexhip_nhyp_fforce_dev(qexhip_handle h, double *f, int n, const int *phi_ids, const double *mass, const double *scale,
                           const double *r2req, int maxits, const int antiperiodic[4], const int phases[4], int *iters);

/* ---------------- random number fields and configuration generation (host only, no handle) ----------------
 * newRNGField (src/rng/distributionUtils.nim:306-331): one generator per site of the LOCAL lattice, seeded with
 * (seed, global lexicographic site index, x fastest); kind 0 = RngMilc6 (src/rng/milcrng.nim), 1 = MRG32k3a
 * (src/rng/mrg32k3a.nim).  glat NULL = lat, t_offset = first global t of this rank's slab.  The deviates are produced
 * with the host's libm so that they are bit-identical to QEX's.  Fields in the library's host format:
 *   uniform:          x.uniform r, ncomp reals per site           (distributionUtils.nim:23-44)
 *   gaussian_vector:  v.gaussian r, colour vector                  (:64-97)
 *   u1_vector:        v.u1 r                                       (:182-211)
 *   random_tah:       p.randomTAH r (randTah3)                     (src/gauge/gaugeUtils.nim:1356-1383)
 *   gauge_random:     g.random r = gaussian + projectSU            (:1348-1354,1424-1429)
 *   gauge_warm:       g.warm s, r = exp(s randTah3)                (:1384-1388,1431-1441) */
typedef struct qexhip_rng qexhip_rng;
int qexhip_rng_new(qexhip_rng **rng, int kind, unsigned long long seed, const int lat[4], const int glat[4], int t_offset);
int qexhip_rng_free(qexhip_rng *rng);
int qexhip_rng_uniform(qexhip_rng *rng, int ncomp, double *v);
int qexhip_rng_gaussian_vector(qexhip_rng *rng, double *v);
int qexhip_rng_u1_vector(qexhip_rng *rng, double *v);
int qexhip_rng_random_tah(qexhip_rng *rng, double *p);
int qexhip_rng_gauge_random(qexhip_rng *rng, double *g);
int qexhip_rng_gauge_warm(qexhip_rng *rng, double s, double *g);
/* generator state per site (RngMilc6: 9 uint32 = r0..r6, icState, multiplier, src/rng/milcrng.nim:12-14; MRG32k3a: 6),
 * the payload of the fork's RNG checkpoints (src/stagg_pv_hmc/staghmc_spv_rng.nim:135-182) */
int qexhip_rng_state_words(qexhip_rng *rng);
int qexhip_rng_get_state(qexhip_rng *rng, unsigned *out);
int qexhip_rng_set_state(qexhip_rng *rng, const unsigned *in);

/* The same generators writing into HBM, for the two ends of an HMC trajectory (refresh / measure of the drivers:
 * src/examples/staghmc_sh.nim:716-757,774-789, src/stagg_pv_hmc/staghmc_spv.nim:1180-1250) -- RngMilc6 fields only.  The
 * generator states go to the device (36 B per site), one lane per site advances its stream, the states come back: the field's
 * state afterwards is bit for bit the one the host call of the same name leaves (qexhip_rng_get_state), and host and device
 * calls can be mixed freely.  Deviates: formed with the device's fp64 log / cos / sqrt; after RngMilc6.gaussian's float32
 * rounding they equal the host's except where the double falls within ~1e-16 of a float32 rounding boundary (about one
 * deviate in 1e8, by one float32 ulp); u1 phases agree to 1e-16.
 *   dev_gaussian_vector: v.gaussian r into the resident colour vector `field_id`      (distributionUtils.nim:64-97)
 *   dev_u1_vector:       v.u1 r                                                       (:182-211)
 *   md_refresh_momenta:  p.randomTAH r into the resident MD momenta (gaugeUtils.nim:1356-1383); qexhip_md_begin(h, g, NULL)
 *                        keeps them, qexhip_md_momentum_norm2 gives 2 T + const. */
int qexhip_rng_dev_gaussian_vector(qexhip_handle h, qexhip_rng *rng, int field_id);
int qexhip_rng_dev_u1_vector(qexhip_handle h, qexhip_rng *rng, int field_id);
int qexhip_md_refresh_momenta(qexhip_handle h, qexhip_rng *rng);

/* ---------------- resident molecular dynamics ----------------
 * The MD loop of QEX's HMC drivers -- mdt (U <- exp(t p) U), mdv (p -= t f), mdvAllfga with its force-gradient shifts
 * (src/examples/staghmc_sh.nim:429-640, src/stagg_pv_hmc/staghmc_spv.nim:873-1061) -- with links and momenta left on the
 * device between the updates.  Forces stay in a device buffer, the "source" of a later kick or shift:
 *   source 0: qexhip_md_gauge_force (gc.forceA / gaugeForce of the resident thin links)
 *   source 1: the last force of the nHYP closure: qexhip_nhyp_gauge_force / _fermion_force / _fforce called with f = NULL
 * and qexhip_nhyp_prepare(g = NULL) smears the resident links.  begin uploads (g NULL: keep the resident links of
 * qexhip_gauge_set; p NULL: keep the resident momenta, e.g. of qexhip_md_refresh_momenta), end downloads (either pointer may be NULL).  kick: p += t * f.  shift_links: U <- exp(t f) U
 * (fgv / fgvf of the force-gradient update); save / restore bracket it (fgsave / fgload). */
int qexhip_md_begin(qexhip_handle h, const double *g, const double *p);
int qexhip_md_end(qexhip_handle h, double *g, double *p);
int qexhip_md_momentum_norm2(qexhip_handle h, double *p2);
int qexhip_md_update_links(qexhip_handle h, double t);
int qexhip_md_gauge_force(qexhip_handle h, double cplaq, double crect, double cadj);
int qexhip_md_kick(qexhip_handle h, int source, double t);
int qexhip_md_shift_links(qexhip_handle h, int source, double t);
int qexhip_md_save_links(qexhip_handle h);
int qexhip_md_restore_links(qexhip_handle h);

/* ---------------- SciDAC/LIME gauge files (host only, no handle) ----------------
 * loadGauge / saveGauge (src/gauge/gaugeUtils.nim:87-122) via Reader / Writer (src/io/readerQiolite.nim:37-239,
 * src/io/writerQiolite.nim:28-187): one record of 4 x QDP_{F,D}3_ColorMatrix per site, sites x-fastest, big-endian,
 * with the SciDAC checksum pair.  g is the library's host format (V=1 even-odd, [vol][4][3][3][2] doubles).
 *   info:  lattice, record precision ('F' | 'D') and whether the file carries checksums (getFileLattice, readerQiolite.nim:11-17)
 *   read:  fills g, returns the checksums it computed; QEXHIP_ERR_IO if they differ from the file's
 *   write: precision 'F' | 'D' (saveGauge's prec); file_md / record_md NULL = QEX's defaults (gaugeUtils.nim:108-109) */
int qexhip_io_gauge_info(const char *path, int lat[4], char *precision, int *checksums_present);
int qexhip_io_read_gauge(const char *path, const int lat[4], double *g, unsigned *suma, unsigned *sumb);
/* the slab t0 <= t < t0 + nt (both even) of a file with the GLOBAL lattice lat, into a rank-local field */
int qexhip_io_read_gauge_slab(const char *path, const int lat[4], int t0, int nt, double *g);
int qexhip_io_write_gauge(const char *path, const int lat[4], const double *g, char precision, const char *file_md,
                          const char *record_md);
/* any other field, as Writer.write / Reader.read handle it (src/io/writerQiolite.nim:96-166): site_bytes per site
 * (library's even-odd order in memory, x-fastest in the file), every word_bytes-wide word big-endian in the file;
 * datatype / precision / colors / datacount go into the record header (typesize = site_bytes / datacount).
 * The fork's RNG checkpoint (staghmc_spv_rng.nim:135-182) is write_field(state, 36, 4, "QDP_RngMilc6", 'F', 0, 1, ...). */
int qexhip_io_write_field(const char *path, const int lat[4], const void *data, int site_bytes, int word_bytes, const char *datatype,
                          char precision, int colors, int datacount, const char *file_md, const char *record_md);
int qexhip_io_read_field(const char *path, const int lat[4], void *data, int site_bytes, int word_bytes, char datatype[64]);
/* Reader.fileMetadata / Reader.recordMetadata (src/io/readerQiolite.nim:37-68,120-135; checked by tests/base/tfieldio.nim:
 * 44-62): the user strings of the file and of its first record, 0-terminated, truncated to the capacities given;
 * *file_len / *record_len = the sizes needed (incl. the 0). */
int qexhip_io_metadata(const char *path, char *file_md, int file_cap, char *record_md, int record_cap, int *file_len,
                       int *record_len);
/* crc32 of src/io/crc32.nim:35-37 (reflected 0xedb88320, start and final complement 0xffffffff): what the SciDAC checksum pair
 * is built from, suma ^= rotl(crc32(site), rank % 29), sumb ^= rotl(crc32(site), rank % 31).  The reference's own known
 * answer: crc32("The quick brown fox jumps over the lazy dog") == 0x414FA339 (crc32.nim:103-106). */
int qexhip_io_crc32(const void *data, size_t nbytes, unsigned *crc);

/* ---------------- kernel timers ----------------
 * hipEvent pairs around launches of the named kernel class on the context stream
 * (the tic/toc hooks of src/physics/stagD.nim:354-395, src/solvers/cg.nim:175-241).
 * names: "dslash" (one-parity sweep; the faces of an overlapped sweep are "dslash_bnd"), "blas", "reduce", "staple",
 * "expupdate", "plaq", "exchange" (the RCCL face exchange, on the stream it is posted on), "allreduce".
 * on = 1: every class; 2: the Dslash sweeps only (least perturbation of a timed region); 3: the anatomy of a sharded
 * iteration -- Dslash sweeps, exchange, allreduce; 0: off. */
int qexhip_timers_enable(qexhip_handle h, int on);
int qexhip_timers_reset(qexhip_handle h);
int qexhip_timers_get(qexhip_handle h, const char *name, long *count, double *total_ms);

/* ---------------- index-arithmetic test hooks (pure host functions, no GPU) ----------------
 * The site order / neighbour sense / ghost-zone positions the kernels use, evaluated on the host
 * (same inline code): layoutIndexQ with V=1 (src/layout/qlayout.nim:110-131) and the shift sense of
 * src/layout/shiftX.nim:76-81.  out[8] = {Vh, F, ntile, ghost tiles/side, tiles/parity, depth, halo, X0/2}. */
int qexhip_debug_geom(const int latLocal[4], int depth, int halo, int out[8]);
/* position, in the opposite-parity field, of site (c,parity) + hop*mu; ghost positions when halo */
int qexhip_debug_nbr_pos(const int latLocal[4], int depth, int halo, int c, int parity, int mu, int hop);
int qexhip_debug_site_coord(const int latLocal[4], int c, int parity, int x[4]);
/* the (tile, parity) visiting order of the staple kernels for gathers in the (mu, nu) plane: out[k] = 2 tile + parity or
 * -1 (empty slot); returns the number of entries, or minus the capacity needed.  Must be a permutation. */
int qexhip_debug_tile_order(const int latLocal[4], int mu, int nu, int *out, int cap);

#ifdef __cplusplus
}
#endif
#endif
