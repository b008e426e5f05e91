// multishift.hip -- multi-shift CG (kernel K7 of SURVEY.md 2.3) and the multi-mass solve.
//
// Restates CgmState.solve (src/solvers/cgm.nim:84-315, precon = cpNone: z = r, q = ps[0],
// LAp = Ap) driven by Staggered.solveXX(xs,b,ms,..) (src/physics/stagSolve.nim:296-345, op =
// stagD2ee|oo(mass^2 + shift)) and Staggered.solve(xs,b,ms,sp) (:347-446).  The zeta recurrences
// (cgm.nim:253-266) run in a one-thread kernel on the device; one streaming kernel then updates
// every shifted solution and search direction, reading r once.
#include "qexhip_internal.h"
#include "reduce.h"
#include <algorithm>
#include <cstring>

#define CGM_MAXM 32

struct CgmScal {
  int nmass, cont, pending, pad;
  double alpha, beta, alphaim1, betaim1;
  double sg[CGM_MAXM], zi[CGM_MAXM], zim1[CGM_MAXM], axz[CGM_MAXM], zip1[CGM_MAXM], bzz[CGM_MAXM];
  double2 *xs[CGM_MAXM], *ps[CGM_MAXM];
};

// after op.apply and redot: alpha, r -= alpha Ap, x += alpha p   (cgm.nim:226-233).  With ndot > 0 (single rank)
// every workgroup sums the <p,Ap> workgroup partials of the preceding sweep itself, in the same fixed order, so
// all agree bit for bit (same device as k_cg_update, blas.hip) and the separate reduction launch disappears.
__global__ void __launch_bounds__(256) k_cgm_base(double2 *x, double2 *r, const double2 *p, const double2 *Ap,
                                                 size_t n, CgScal *s, double *partials, const double *dotp, int ndot) {
  if (s->done) return;
  double pAp = s->pAp;
  if (ndot > 0) {
    double a = 0;
    for (int i = threadIdx.x; i < ndot; i += 256) a += dotp[i];
    pAp = block_sum_256_all(a);
  }
  const double alpha = (pAp != 0.0) ? s->r2 / pAp : 0.0;
  double acc = 0;
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    double2 pv = p[i], xv = x[i], rv = r[i], av = Ap[i];
    rv.x -= alpha * av.x; rv.y -= alpha * av.y;
    xv.x += alpha * pv.x; xv.y += alpha * pv.y;
    x[i] = xv; r[i] = rv;
    acc = fma(rv.x, rv.x, fma(rv.y, rv.y, acc));
  }
  double t = block_sum_256(acc);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
// One workgroup closes the iteration: final sum of the |r|^2 partials (sharded: the all-reduced partial vector; n == 0:
// s->tmp holds the value), alpha / beta, the loop condition, and the zeta recurrences of cgm.nim:253-266 -- thread k owns
// shift k.  `pending` tells the following k_cgm_update that this iteration is live: it has to run once more after
// the loop condition has turned false (xs[m] += alpha zr ps[m] is outside `if continuing`), so it cannot key on
// s->done; a later, dead pass through this kernel clears the flag again.
__global__ void __launch_bounds__(256) k_cgm_close(const double *partials, int n, const double *dotp, int ndot,
                                                  CgScal *s, CgmScal *m, double *hist, int histcap) {
  if (s->done) {
    if (threadIdx.x == 0) m->pending = 0;
    return;
  }
  double r2ip1 = s->tmp;
  if (n > 0) {
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partials[i];
    r2ip1 = block_sum_256_all(acc);
  }
  double pAp = s->pAp;
  if (ndot > 0) {
    double a = 0;
    for (int i = threadIdx.x; i < ndot; i += 256) a += dotp[i];
    pAp = block_sum_256_all(a);
  }
  const double r2i = s->r2;
  const double alpha = (pAp != 0.0) ? r2i / pAp : 0.0;
  const double beta = (r2i != 0.0) ? r2ip1 / r2i : 0.0;
  const int itn = s->itn + 1;
  const int cont = (itn < s->maxits) && (r2ip1 > s->r2stop);
  const double alphaim1 = m->alphaim1, betaim1 = m->betaim1, b2 = s->b2;
  const int nm = m->nmass;
  __syncthreads();   // every thread holds the old state before anyone overwrites it
  const int k = threadIdx.x;
  if (k >= 1 && k < nm) {
    double zip1d = alpha * betaim1 * (m->zim1[k] - m->zi[k]);
    zip1d += m->zim1[k] * alphaim1 * (1.0 + m->sg[k] * alpha);
    const double zip1 = (zip1d != 0.0) ? m->zi[k] * m->zim1[k] * alphaim1 / zip1d : 0.0;
    const double zr = (m->zi[k] != 0.0) ? zip1 / m->zi[k] : 0.0;
    m->axz[k] = alpha * zr;
    m->zip1[k] = zip1;
    m->bzz[k] = beta * zr * zr;
    if (cont) { m->zim1[k] = m->zi[k]; m->zi[k] = zip1; }
  }
  if (k == 0) {
    m->cont = cont; m->pending = 1;
    m->alpha = alpha; m->beta = beta;
    m->alphaim1 = alpha; m->betaim1 = beta;
    s->itn = itn;
    s->r2 = r2ip1;
    s->agree[0] = r2ip1; s->agree[1] = -r2ip1; s->agree[2] = (double)itn; s->agree[3] = -(double)itn;
    if (itn < histcap) hist[itn] = r2ip1 / b2;
    if (!cont) s->done = 1;
  }
}
// q := z + beta*q (if continuing); xs[k] += alpha*zr*ps[k]; ps[k] := zip1*r + beta*zr^2*ps[k]
__global__ void __launch_bounds__(256) k_cgm_update(const double2 *r, size_t n, const CgmScal *m) {
  if (!m->pending) return;
  const int cont = m->cont, nm = m->nmass;
  const double beta = m->beta;
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const double2 rv = r[i];
    if (cont) {
      double2 pv = m->ps[0][i];
      m->ps[0][i] = make_double2(rv.x + beta * pv.x, rv.y + beta * pv.y);
    }
    for (int k = 1; k < nm; k++) {
      double2 pv = m->ps[k][i], xv = m->xs[k][i];
      const double axz = m->axz[k];
      xv.x += axz * pv.x; xv.y += axz * pv.y;
      m->xs[k][i] = xv;
      if (cont) {
        const double z = m->zip1[k], b = m->bzz[k];
        m->ps[k][i] = make_double2(z * rv.x + b * pv.x, z * rv.y + b * pv.y);
      }
    }
  }
}
__global__ void k_cgm_init(CgScal *s, const double *dscal, double r2req, int maxits, double *hist, int histcap) {
  s->b2 = dscal[0];
  s->r2 = dscal[0];  // r := b, r2 = b2 (cgm.nim:169-175)
  s->rzo = 1.0; s->pAp = 0.0; s->tmp = 0.0;
  s->r2stop = r2req * s->b2;
  s->itn = 0; s->maxits = maxits;
  s->done = !(s->r2 > s->r2stop);
  if (histcap > 0) hist[0] = (s->b2 != 0.0) ? 1.0 : 0.0;
}
// r := b; xs[k] := 0; ps[k] := r for every shift in one pass  (cgm.nim:165-207)
__global__ void __launch_bounds__(256) k_cgm_start(double2 *r, const double2 *b, size_t n, const CgmScal *m) {
  const int nm = m->nmass;
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const double2 bv = b[i];
    r[i] = bv;
    for (int k = 0; k < nm; k++) {
      m->xs[k][i] = make_double2(0, 0);
      m->ps[k][i] = bv;
    }
  }
}

static int read_cg(qexhip_ctx *c, CgScal *host) {
  HIPCHK(hipMemcpyAsync(c->pinned, c->cg, sizeof(CgScal), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  memcpy(host, c->pinned, sizeof(CgScal));
  return 0;
}

// Persistent workspace of the multi-shift solvers: fields [POOL_PS, POOL_PS+nmass) are the search directions,
// [POOL_YS, ..) the even/odd solutions of solve_multi_dev, [POOL_XS, ..) the host-pointer entry points' solutions.
// They live in the context's field table under negative ids, so qexhip_finalize frees them and a change of the
// ghost geometry re-allocates them like every other field; nothing is allocated or freed per solve after the first.
int pool_field(qexhip_ctx *c, int idx, DevField **f) {
  const int id = -(1000 + idx);
  auto it = c->fields.find(id);
  if (it == c->fields.end()) {
    DevField nf;
    CHK(field_alloc(c, nf));
    it = c->fields.emplace(id, nf).first;
  }
  *f = &it->second;
  return 0;
}

int solve_xx_multi_dev(qexhip_ctx *c, std::vector<DevField *> &xs, DevField &b, const double *shifts,
                       int nmass, double r2req, int maxits, int par_even, int *iters, double *hist, int histcap) {
  if (nmass < 1 || nmass > CGM_MAXM) { qexhip_set_error("multishift: 1 <= nmass <= %d", CGM_MAXM); return -1; }
  c->cg_resume.valid = 0;                 // the device CG state and the history buffer are taken over
  const int par = par_even ? 0 : 1;
  const Geom &g = c->g;
  const size_t n = (size_t)g.ntile * 192;
  int nb = (int)std::min<size_t>((n + 255) / 256, 2048);
  DevField *r, *Ap;
  CHK(get_work(c, WK_R, &r));
  CHK(get_work(c, WK_AP, &Ap));
  if (c->histcap < std::max(histcap, 1)) {
    if (c->hist) HIPCHK(hipFree(c->hist));
    c->hist = nullptr; c->histcap = 0;
    HIPCHK(hipMalloc((void **)&c->hist, sizeof(double) * std::max(histcap, 1)));
    c->histcap = std::max(histcap, 1);
  }
  std::vector<DevField *> ps(nmass);
  for (int k = 0; k < nmass; k++) CHK(pool_field(c, POOL_PS + k, &ps[k]));
  if (!c->cgm_scal) HIPCHK(hipMalloc(&c->cgm_scal, sizeof(CgmScal)));
  CgmScal *g_cgm_dev = (CgmScal *)c->cgm_scal;
  CgmScal *hm = (CgmScal *)c->pinned;                 // pinned staging: sizeof(CgmScal) <= 4096
  static_assert(sizeof(CgmScal) <= 4096, "CgmScal must fit the pinned scratch page");
  memset(hm, 0, sizeof(*hm));
  hm->nmass = nmass; hm->cont = 1; hm->pending = 0;
  hm->alphaim1 = -1.0; hm->betaim1 = 0.0;
  for (int k = 0; k < nmass; k++) {
    hm->sg[k] = (k == 0) ? 0.0 : shifts[k];
    hm->zi[k] = 1.0; hm->zim1[k] = 1.0;
    hm->xs[k] = xs[k]->par(par);
    hm->ps[k] = ps[k]->par(par);
  }
  HIPCHK(hipMemcpyAsync(g_cgm_dev, hm, sizeof(*hm), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));            // the pinned page is reused by read_cg below
  const double mass = shifts[0], m2 = mass * mass;
  for (int k = 0; k < nmass; k++) CHK(blas_zero(c, *xs[k], 1 - par));   // the other parity of every solution
  k_cgm_start<<<nb, 256, 0, c->stream>>>(r->par(par), b.par(par), n, g_cgm_dev);   // q := z ; ps[m] := p (cgm.nim:196-207)
  HIPCHK(hipGetLastError());
  CHK(blas_norm2(c, b, par, &c->dscal[0]));
  k_cgm_init<<<1, 1, 0, c->stream>>>(c->cg, c->dscal, r2req, maxits, c->hist, c->histcap);
  HIPCHK(hipGetLastError());
  CgScal st;
  CHK(read_cg(c, &st));
  const bool sharded = multi_rank(c);
  double *r2p = c->partials + c->part2_off;
  while (!st.done) {
    int nn = std::min(32, std::max(1, st.maxits - st.itn));
    for (int i = 0; i < nn; i++) {
      int ndot = 0;
      CHK(op_xx(c, *Ap, *ps[0], m2, par_even, 1, &c->cg->done, &ndot));
      // sharded: the workgroup partials themselves are all-reduced (a few KB, the latency of one double), so the
      // iteration needs no one-block reduction launches; ndot == 0: big local volume, op_xx has reduced <p,Ap> already
      if (sharded && ndot > 0) CHK(comm_allreduce_parts(c, c->partials, ndot, &ndot));
      CHK(devjoin_flush(c));              // (no-op when the all-reduce has taken the second sweep's join with it)
      {
        ScopedTimer tm(c, "blas", c->stream);
        k_cgm_base<<<nb, 256, 0, c->stream>>>(xs[0]->par(par), r->par(par), ps[0]->par(par), Ap->par(par), n, c->cg, r2p,
                                              c->partials, ndot);
        HIPCHK(hipGetLastError());
      }
      int nr2 = nb;
      if (sharded) CHK(comm_allreduce_parts(c, r2p, nb, &nr2));
      {
        ScopedTimer tm(c, "reduce", c->stream);
        k_cgm_close<<<1, 256, 0, c->stream>>>(r2p, nr2, c->partials, ndot, c->cg, g_cgm_dev, c->hist, c->histcap);
        HIPCHK(hipGetLastError());
      }
      ScopedTimer tm(c, "cgm_update", c->stream);
      k_cgm_update<<<nb, 256, 0, c->stream>>>(r->par(par), n, g_cgm_dev);
      HIPCHK(hipGetLastError());
    }
    CHK(comm_agree_post(c));
    CHK(read_cg(c, &st));
    CHK(comm_agree_check(c, st));
  }
  if (iters) *iters = st.itn;
  if (hist && histcap > 0) {
    int nh = std::min(histcap, st.itn + 1);
    HIPCHK(hipMemcpyAsync(hist, c->hist, sizeof(double) * nh, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  return 0;
}

// Staggered.solve(xs, b, ms, sp)  (stagSolve.nim:347-446)
int solve_multi_dev(qexhip_ctx *c, std::vector<DevField *> &xs, DevField &b, const double *masses,
                    int nmass, double r2req, int maxits, int *iters, double *r2_final) {
  if (nmass < 1 || nmass > CGM_MAXM) { qexhip_set_error("multishift: 1 <= nmass <= %d", CGM_MAXM); return -1; }
  DevField *r, *xt;
  CHK(get_work(c, WK_R2, &r));
  CHK(get_work(c, WK_XT, &xt));
  const double mass = masses[0];
  std::vector<double> shifts(nmass);
  std::vector<DevField *> ys(nmass);
  for (int k = 0; k < nmass; k++) {
    shifts[k] = (k == 0) ? masses[0] : 4.0 * (masses[k] * masses[k] - mass * mass);
    CHK(pool_field(c, POOL_YS + k, &ys[k]));
    CHK(blas_zero(c, *xs[k], 2));
  }
  CHK(blas_zero(c, *xt, 2));
  CHK(blas_copy(c, *r, b, 2));
  CHK(blas_norm2(c, b, 2, &c->dscal[2]));
  CHK(blas_norm2(c, b, 0, &c->dscal[3]));
  CHK(blas_norm2(c, b, 1, &c->dscal[4]));
  double h[3];
  CHK(read_scalars(c, &c->dscal[2], 3, h));
  double b2 = h[0], b2e = h[1], b2o = h[2];
  double r2 = b2e + b2o;
  const double r2stop = r2req * b2;
  int its = 0;
  while (r2 > r2stop) {
    int mx = maxits - its;
    if (mx <= 0) break;
    double rq = r2stop;
    const double r2stop2 = 0.5 * rq;
    const double r2stope = (b2o <= r2stop2) ? rq - b2o : r2stop2;
    const double r2stopo = (b2e <= r2stop2) ? rq - b2e : r2stop2;
    int even = 1;
    if (b2e > r2stope) { rq = r2stope / b2e; even = 1; }
    else if (b2o > r2stopo) { rq = r2stopo / b2o; even = 0; }
    int n = 0;
    CHK(solve_xx_multi_dev(c, ys, *r, shifts.data(), nmass, rq, mx, even, &n, nullptr, 0));
    its += n;
    for (int k = 0; k < nmass; k++) CHK(blas_axpy(c, 4.0, *ys[k], *xs[k], even ? 0 : 1));
    CHK(op_D(c, *xt, *xs[0], mass, -1.0));
    CHK(op_D(c, *r, *xt, mass, 1.0));
    CHK(blas_axpby(c, 1.0, b, -1.0, *r, *r, 2));
    CHK(blas_norm2(c, *r, 0, &c->dscal[3]));
    CHK(blas_norm2(c, *r, 1, &c->dscal[4]));
    CHK(read_scalars(c, &c->dscal[3], 2, h));
    b2e = h[0]; b2o = h[1];
    r2 = b2e + b2o;
  }
  for (int k = 0; k < nmass; k++) {
    if (k != 0) CHK(op_D(c, *xt, *xs[k], masses[k], -1.0));
    CHK(blas_copy(c, *xs[k], *xt, 2));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  if (iters) *iters = its;
  if (r2_final) *r2_final = (b2 != 0.0) ? r2 / b2 : 0.0;
  return 0;
}
