// solver.cpp -- host control flow of the operator compositions and solvers, device resident.
//
// Restates (file:line in ctpeterson/qex):
//   stagD / D / Ddag / eoReconstruct   src/physics/stagD.nim:406-409,566-586
//   stagD2ee / stagD2oo                src/physics/stagD.nim:434-469
//   solveXX (solveEE/solveOO)          src/physics/stagSolve.nim:57-138  + CG src/solvers/cg.nim:55-272
//   solveReconR / solveReconL / solve  src/physics/stagSolve.nim:141-294
//   multi-shift solveXX / solve        src/physics/stagSolve.nim:296-446 + src/solvers/cgm.nim:84-315
// All vectors stay in HBM; the CG scalars stay on the device (CgScal) and the host only reads
// the state back once per chunk of iterations.
#include "qexhip_internal.h"
#include <cmath>
#include <cstring>
#include <algorithm>


int get_work(qexhip_ctx *c, int slot, DevField **f) {
  if (slot == WK_R || slot == WK_P || slot == WK_AP) c->cg_resume.valid = 0;     // whoever takes the CG's vectors ends a resumable solve
  if (c->wk[slot] == 0) {
    DevField nf;
    CHK(field_alloc(c, nf));
    int id = -(slot + 1);
    c->fields[id] = nf;
    c->wk[slot] = id;
  }
  *f = &c->fields[c->wk[slot]];
  return 0;
}

// r[px] = 4 m2 x - (2D)(2D) x : stagDP onto the other parity, stagDM back (stagD.nim:434-456)
int op_xx(qexhip_ctx *c, DevField &r, DevField &x, double m2, int par_even, int dot, const int *done, int *ndot) {
  DevField *t;
  CHK(get_work(c, WK_T, &t));
  const int px = par_even ? 0 : 1, py = 1 - px;
  // with deferred partials (the CG loops) the caller's next call is comm_allreduce_parts, which takes the join of the second sweep's
  // boundary launch into its kernel where the sweep is split by sites and the mailboxes carry the sum
  DslashOpts o1, o2;
  o1.done = done;
  o2.cb = 4.0 * m2;
  o2.xs = &x;
  o2.neg = 1;
  o2.dot = (dot && ndot) ? 2 : dot;
  o2.nparts_out = ndot;
  o2.dot_out = &c->cg->pAp;
  o2.done = done;
  o2.defer_join = (dot && ndot) ? 1 : 0;
  o2.pair2 = (t->d != x.d && t->d != r.d) ? 1 : 0;
  if (o2.cb == 0.0 && dot) { qexhip_set_error("op_xx: dot with m2 == 0 unsupported"); return -1; }
  CHK(dslash_sweep(c, *t, x, py, o1));
  CHK(dslash_sweep(c, r, *t, px, o2));
  return 0;
}

// stagD on one parity: r = a*r + m*x + sc*D*x via stagD2(a/(.5sc), m/(.5sc)) then *(.5sc)
static int op_stagD(qexhip_ctx *c, DevField &r, DevField &x, int parity, double m, double sc, double a) {
  DslashOpts o;
  o.ca = a / (0.5 * sc);
  o.cb = m / (0.5 * sc);
  o.rin = &r;
  o.xs = &x;
  o.post = 0.5 * sc;
  return dslash_sweep(c, r, x, parity, o);
}

int op_D(qexhip_ctx *c, DevField &r, DevField &x, double m, double sc, double a) {
  CHK(op_stagD(c, r, x, 0, m, sc, a));
  CHK(op_stagD(c, r, x, 1, m, sc, a));
  return 0;
}

// r.odd = (b.odd - D_oe r.even)/m  (stagD.nim:583-586)
static int op_eo_reconstruct(qexhip_ctx *c, DevField &r, DevField &b, double m) {
  CHK(op_stagD(c, r, r, 1, 0.0, -1.0 / m, 0.0));
  CHK(blas_axpy(c, 1.0 / m, b, r, 1));
  return 0;
}
int op_eo_reconstruct_pub(qexhip_ctx *c, DevField &r, DevField &b, double m) { return op_eo_reconstruct(c, r, b, m); }
int op_stagD_pub(qexhip_ctx *c, DevField &r, DevField &x, int parity, double m, double sc, double a) { return op_stagD(c, r, x, parity, m, sc, a); }
// r.even = (D^+ b).even = (m b - D b).even  (eoReduce, stagD.nim:575-581: one stagD on the even subset with sc = -1)
int op_eo_reduce_pub(qexhip_ctx *c, DevField &r, DevField &b, double m) { return op_stagD(c, r, b, 0, m, -1.0, 0.0); }

static int ensure_hist(qexhip_ctx *c, int cap) {
  if (cap < 1) cap = 1;
  if (c->histcap >= cap) return 0;
  if (c->hist) HIPCHK(hipFree(c->hist));
  c->hist = nullptr; c->histcap = 0;
  HIPCHK(hipMalloc((void **)&c->hist, sizeof(double) * cap));
  c->histcap = cap;
  return 0;
}

static int read_cg(qexhip_ctx *c, CgScal *host) {
  HIPCHK(hipMemcpyAsync(c->pinned, c->cg, sizeof(CgScal), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  memcpy(host, c->pinned, sizeof(CgScal));
  return 0;
}

// The iterations of CgState.solve (cg.nim:174-217) from iteration k on: device state in slot k&1 (`rolled`), r / p / x as the
// previous iteration left them.  Shared by the first call and by the re-entry (cg.nim:133: `if b2<0: # first call`).
static int cg_iterate(qexhip_ctx *c, DevField &x, DevField *r, DevField *p, DevField *Ap, double m2, int par_even, int k,
                      CgScal &st, int *iters, double *r2_over_b2, double *hist, int histcap) {
  const int par = par_even ? 0 : 1;
  const int chunk = 32;
  int rolled = 1;
  bool done = st.dones[k & 1];
  double r2 = st.r2s[k & 1];
  st.itn = st.itns[k & 1];
  // (Round 4 tried forming p = r + beta p inside the first sweep -- every output site builds it for its eight neighbours, one of
  // them stores it -- to drop k_cg_xpay and a launch boundary: the sweep grew by 17 us, the iteration did not move (267.9 vs
  // 267.6 us, profiles/r04_cg_fuse_ab.log), so the separate launch stays.)
  while (!done) {
    int n = std::min(chunk, st.maxits - k);
    if (n <= 0) break;
    for (int i = 0; i < n; i++, k++) {
      CHK(cg_xpay(c, *p, *r, par, k, rolled));                        // cg.nim:186-193 (+ bookkeeping of k-1)
      rolled = 0;
      int ndot = 0;                                                   // <p,Ap> partials are summed inside cg_update
      CHK(op_xx(c, *Ap, *p, m2, par_even, 1, &c->cg->dones[k & 1], &ndot));   // cg.nim:200, qLAp :206
      CHK(cg_update(c, x, *r, *p, *Ap, par, k, ndot));                // cg.nim:208-213
    }
    CHK(cg_close(c, k));
    rolled = 1;
    CHK(read_cg(c, &st));
    CHK(comm_agree_check(c, st));
    done = st.dones[k & 1];
    r2 = st.r2s[k & 1];
    st.itn = st.itns[k & 1];
  }
  st.r2 = r2;
  if (iters) *iters = st.itn;
  if (r2_over_b2) *r2_over_b2 = (st.b2 != 0.0) ? st.r2 / st.b2 : 0.0;
  if (hist && histcap > 0) {
    int n = std::min(std::min(histcap, c->histcap), st.itn + 1);
    HIPCHK(hipMemcpyAsync(hist, c->hist, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  // what a re-entry needs: the same x, the same operator; r, p and the device scalars stay where they are
  c->cg_resume.valid = 1; c->cg_resume.x = x.d; c->cg_resume.par_even = par_even; c->cg_resume.m2 = m2; c->cg_resume.k = st.itn;
  return 0;
}

int solve_xx_dev(qexhip_ctx *c, DevField &x, DevField &b, double mass, double r2req, int maxits,
                 int par_even, int *iters, double *r2_over_b2, double *hist, int histcap) {
  const int par = par_even ? 0 : 1;
  DevField *r, *p, *Ap;
  CHK(get_work(c, WK_R, &r));
  CHK(get_work(c, WK_P, &p));
  CHK(get_work(c, WK_AP, &Ap));
  CHK(ensure_hist(c, std::max(histcap, 1)));
  const double m2 = mass * mass;
  CHK(blas_zero(c, x, 2));                       // threads: r := 0 (stagSolve.nim:63-64)
  CHK(blas_norm2(c, b, par, &c->dscal[0]));      // b2 (cg.nim:134)
  // op.apply(Ap, x) with x = 0 gives Ap = 0 exactly, so r = b - Ap = b and r2 = b2 (cg.nim:145-151)
  CHK(blas_copy(c, *r, b, par));
  CHK(blas_norm2(c, *r, par, &c->dscal[1]));
  CHK(cg_init(c, r2req, maxits));
  CgScal st;
  CHK(read_cg(c, &st));
  return cg_iterate(c, x, r, p, Ap, m2, par_even, 0, st, iters, r2_over_b2, hist, histcap);   // k_cg_init wrote slot 0
}

// Re-entry of CgState.solve with b2 >= 0 (cg.nim:21-27,85,133,155-161,256-261): nothing is set up again; r2stop and maxits are
// taken from the new SolverParams, the iteration count goes on counting, beta of the next iteration uses the kept rzold.
// Valid only directly after a solve_xx_dev / a previous re-entry on the same x and operator.
int solve_xx_continue_dev(qexhip_ctx *c, DevField &x, double r2req, int maxits, int *iters, double *r2_over_b2, double *hist, int histcap) {
  if (!c->cg_resume.valid || c->cg_resume.x != x.d) {
    qexhip_set_error("solve_xx_continue: no resumable CG state for this field (another solver, set_links or a work-vector user ran "
                     "since the last solve_xx on it)");
    return -3;
  }
  const int k = c->cg_resume.k, par_even = c->cg_resume.par_even;
  const double m2 = c->cg_resume.m2;
  DevField *r = &c->fields[c->wk[WK_R]], *p = &c->fields[c->wk[WK_P]], *Ap = &c->fields[c->wk[WK_AP]];
  CHK(cg_resume(c, k, r2req, maxits));
  CgScal st;
  CHK(read_cg(c, &st));
  return cg_iterate(c, x, r, p, Ap, m2, par_even, k, st, iters, r2_over_b2, hist, histcap);
}

// ---- full solve (stagSolve.nim:141-294) ----
static int norm2_eo(qexhip_ctx *c, DevField &f, double *e, double *o) {
  CHK(blas_norm2(c, f, 0, &c->dscal[2]));
  CHK(blas_norm2(c, f, 1, &c->dscal[3]));
  double h[2];
  CHK(read_scalars(c, &c->dscal[2], 2, h));
  *e = h[0]; *o = h[1];
  return 0;
}

static int solve_inner(qexhip_ctx *c, DevField &x, DevField &b, double m, double r2req, int maxits,
                       double b2e, double b2o, int *its) {
  const double b2 = b2e + b2o;
  const double r2stop = r2req * b2, r2stop2 = 0.5 * r2stop;
  const double r2stope = (b2o <= r2stop2) ? r2stop - b2o : r2stop2;
  const double r2stopo = (b2e <= r2stop2) ? r2stop - b2e : r2stop2;
  *its = 0;
  if (b2e <= r2stope || b2o <= r2stopo || m == 0.0) {
    // solveReconR (:141-176)
    DevField *y;
    CHK(get_work(c, WK_D, &y));
    if (b2e > r2stope) {
      CHK(solve_xx_dev(c, *y, b, m, r2stope / b2e, maxits, 1, its, nullptr, nullptr, 0));
      CHK(blas_scale(c, 4.0, *y, 0));
      CHK(op_D(c, x, *y, m, -1.0));
    } else if (b2o > r2stopo) {
      CHK(solve_xx_dev(c, *y, b, m, r2stopo / b2o, maxits, 0, its, nullptr, nullptr, 0));
      CHK(blas_scale(c, 4.0, *y, 1));
      CHK(op_D(c, x, *y, m, -1.0));
    }
  } else {
    // solveReconL (:179-208)
    DevField *d;
    CHK(get_work(c, WK_D, &d));
    CHK(op_D(c, *d, b, m, -1.0));
    CHK(blas_norm2(c, *d, 0, &c->dscal[2]));
    double d2e;
    CHK(read_scalars(c, &c->dscal[2], 1, &d2e));
    const double rr = 0.99 * r2req * (b2e + b2o) * m * m / d2e;
    CHK(solve_xx_dev(c, x, *d, m, rr, maxits, 1, its, nullptr, nullptr, 0));
    CHK(blas_scale(c, 4.0, x, 0));
    CHK(op_eo_reconstruct(c, x, b, m));
  }
  return 0;
}

int solve_full_dev(qexhip_ctx *c, DevField &x, DevField &b, double mass, double r2req, int maxits,
                   int *iters, double *r2_final, int use_prev) {
  DevField *r, *y;
  CHK(get_work(c, WK_R2, &r));
  CHK(get_work(c, WK_Y, &y));
  CHK(blas_norm2(c, b, 2, &c->dscal[2]));
  double b2;
  CHK(read_scalars(c, &c->dscal[2], 1, &b2));
  const double r2stop = r2req * b2;
  if (use_prev) {                                  // sp.usePrevSoln (stagSolve.nim:234-238)
    CHK(op_D(c, *r, x, mass, 1.0));
    CHK(blas_axpby(c, 1.0, b, -1.0, *r, *r, 2));
  } else {
    CHK(blas_zero(c, x, 2));
    CHK(blas_copy(c, *r, b, 2));
  }
  double r2e, r2o;
  CHK(norm2_eo(c, *r, &r2e, &r2o));
  double r2 = r2e + r2o;
  int its = 0;
  while (r2 > r2stop) {
    int mx = maxits - its;
    if (mx <= 0) break;
    int n = 0;
    CHK(solve_inner(c, *y, *r, mass, r2stop / r2, mx, r2e, r2o, &n));
    its += n;
    CHK(blas_axpy(c, 1.0, *y, x, 2));
    CHK(op_D(c, *r, x, mass, 1.0));
    CHK(blas_axpby(c, 1.0, b, -1.0, *r, *r, 2));   // r := b - r
    CHK(norm2_eo(c, *r, &r2e, &r2o));
    r2 = r2e + r2o;
  }
  if (iters) *iters = its;
  if (r2_final) *r2_final = (b2 != 0.0) ? r2 / b2 : 0.0;
  return 0;
}
