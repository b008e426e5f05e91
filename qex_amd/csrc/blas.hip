// blas.hip -- streaming field algebra + reductions of the CG (kernels K4/K5 of SURVEY.md 2.3).
//
// Restates the expression-template assignments x := y, x += a*y, ... (src/field/fieldET.nim:547-598)
// and norm2P / redotP (:605-625, :704-724; always accumulated in fp64).  A one-parity vector is a
// dense array of ntile*192 double2 (padding lanes of the last tile are zero and stay zero), so
// every kernel is a grid-stride loop with 16-byte accesses.  The CG scalars live in a CgScal on
// the device: alpha/beta are formed by every lane from the same two doubles (IEEE division, so
// all lanes agree), which keeps the host out of the iteration loop.
#include "qexhip_internal.h"
#include "reduce.h"
#include "cg_device.h"
#include <cstring>
#include "peer_device.h"

static inline int grid_for(size_t n2) {
  size_t nb = (n2 + 255) / 256;
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  return (int)nb;
}
static inline size_t body2(const qexhip_ctx *c) { return (size_t)c->g.ntile * 192; }

// parity 0/1: one half body; 2: both halves (two launches keep ghost tiles untouched)
#define FOR_PAR(par, p) for (int p = ((par) == 2 ? 0 : (par)); p <= ((par) == 2 ? 1 : (par)); p++)

__global__ void __launch_bounds__(256) k_zero(double2 *y, size_t n) {
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = make_double2(0, 0);
}
__global__ void __launch_bounds__(256) k_copy(double2 *y, const double2 *x, size_t n) {
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = x[i];
}
__global__ void __launch_bounds__(256) k_axpy(double a, const double2 *x, double2 *y, size_t n) {
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    double2 xv = x[i], yv = y[i];
    yv.x += a * xv.x; yv.y += a * xv.y;
    y[i] = yv;
  }
}
__global__ void __launch_bounds__(256) k_xpay(const double2 *x, double a, double2 *y, size_t n) {
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    double2 xv = x[i], yv = y[i];
    yv.x = xv.x + a * yv.x; yv.y = xv.y + a * yv.y;
    y[i] = yv;
  }
}
__global__ void __launch_bounds__(256) k_scale(double a, double2 *y, size_t n) {
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    double2 yv = y[i];
    yv.x *= a; yv.y *= a;
    y[i] = yv;
  }
}
__global__ void __launch_bounds__(256) k_axpby(double a, const double2 *x, double b, const double2 *y, double2 *z, size_t n) {
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    double2 xv = x[i], yv = y[i];
    z[i] = make_double2(a * xv.x + b * yv.x, a * xv.y + b * yv.y);
  }
}
__global__ void __launch_bounds__(256) k_redot(const double2 *x, const double2 *y, size_t n, double *partials) {
  double acc = 0;
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    double2 xv = x[i], yv = y[i];
    acc = fma(xv.x, yv.x, fma(xv.y, yv.y, acc));   // explicit fma: the same rounding in every kernel that forms this sum
  }
  double r = block_sum_256(acc);
  if (threadIdx.x == 0) partials[blockIdx.x] = r;
}
// fixed-order final sum of the workgroup partials (deterministic)
__global__ void __launch_bounds__(256) k_reduce_final(const double *partials, int n, double *out, const int *done) {
  if (done && *done) return;
  double acc = 0;
  for (int i = threadIdx.x; i < n; i += 256) acc += partials[i];
  double r = block_sum_256(acc);
  if (threadIdx.x == 0) *out = r;
}

// Transport emulation (options "emu_exchange_us" / "emu_allreduce_us", comm.cpp): one lane waits `us` microseconds of the
// constant-rate wall clock (100 MHz on gfx950) on the stream a collective is about to be posted on -- so that a one-GPU
// rehearsal sees exchanges / all-reduces that take as long as they would between distinct GPUs.  Bounded by construction:
// the loop ends when the clock has advanced, and at the latest after 2^22 polls (~0.1 s).
__global__ void k_delay(long long ticks) {
  const long long t0 = wall_clock64();
  for (int i = 0; i < (1 << 22); i++) {
    if (wall_clock64() - t0 >= ticks) break;
    __builtin_amdgcn_s_sleep(32);
  }
}
int blas_delay(hipStream_t st, int us) {
  if (us <= 0) return 0;
  k_delay<<<1, 1, 0, st>>>((long long)us * 100);
  HIPCHK(hipGetLastError());
  return 0;
}

int blas_grid(const qexhip_ctx *c, int) { return grid_for(body2(c)); }   // workgroups (= partial sums) of the CG's BLAS kernels

int blas_zero(qexhip_ctx *c, DevField &f, int parity) {
  size_t n = body2(c);
  FOR_PAR(parity, p) k_zero<<<grid_for(n), 256, 0, c->stream>>>(f.par(p), n);
  HIPCHK(hipGetLastError());
  return 0;
}
int blas_copy(qexhip_ctx *c, DevField &dst, const DevField &src, int parity) {
  size_t n = body2(c);
  FOR_PAR(parity, p) k_copy<<<grid_for(n), 256, 0, c->stream>>>(dst.par(p), src.par(p), n);
  HIPCHK(hipGetLastError());
  return 0;
}
int blas_axpy(qexhip_ctx *c, double a, const DevField &x, DevField &y, int parity) {
  size_t n = body2(c);
  ScopedTimer tm(c, "blas", c->stream);
  FOR_PAR(parity, p) k_axpy<<<grid_for(n), 256, 0, c->stream>>>(a, x.par(p), y.par(p), n);
  HIPCHK(hipGetLastError());
  return 0;
}
int blas_xpay(qexhip_ctx *c, const DevField &x, double a, DevField &y, int parity) {
  size_t n = body2(c);
  ScopedTimer tm(c, "blas", c->stream);
  FOR_PAR(parity, p) k_xpay<<<grid_for(n), 256, 0, c->stream>>>(x.par(p), a, y.par(p), n);
  HIPCHK(hipGetLastError());
  return 0;
}
int blas_scale(qexhip_ctx *c, double a, DevField &y, int parity) {
  size_t n = body2(c);
  FOR_PAR(parity, p) k_scale<<<grid_for(n), 256, 0, c->stream>>>(a, y.par(p), n);
  HIPCHK(hipGetLastError());
  return 0;
}
int blas_axpby(qexhip_ctx *c, double a, const DevField &x, double b, const DevField &y, DevField &z, int parity) {
  size_t n = body2(c);
  FOR_PAR(parity, p) k_axpby<<<grid_for(n), 256, 0, c->stream>>>(a, x.par(p), b, y.par(p), z.par(p), n);
  HIPCHK(hipGetLastError());
  return 0;
}

int reduce_partials(qexhip_ctx *c, int n, double *dev_out) {
  {
    ScopedTimer tm(c, "reduce", c->stream);
    k_reduce_final<<<1, 256, 0, c->stream>>>(c->partials, n, dev_out, nullptr);
    HIPCHK(hipGetLastError());
  }
  if (multi_rank(c)) CHK(comm_allreduce(c, dev_out, 1));
  return 0;
}

int blas_redot(qexhip_ctx *c, const DevField &x, const DevField &y, int parity, double *dev_out) {
  size_t n = body2(c);
  int nb = grid_for(n);
  int tot = 0;
  {
    ScopedTimer tm(c, "blas", c->stream);
    FOR_PAR(parity, p) {
      k_redot<<<nb, 256, 0, c->stream>>>(x.par(p), y.par(p), n, c->partials + tot);
      tot += nb;
    }
    HIPCHK(hipGetLastError());
  }
  return reduce_partials(c, tot, dev_out);
}
// dotP (fieldET.nim:677-693): sum_s x(s)^+ y(s), complex -> dev_out[0] = Re, dev_out[1] = Im (rank-global)
__global__ void __launch_bounds__(256) k_cdot(const double2 *x, const double2 *y, size_t n, double *pre, double *pim) {
  double ar = 0, ai = 0;
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const double2 xv = x[i], yv = y[i];
    ar = fma(xv.x, yv.x, fma(xv.y, yv.y, ar));
    ai = fma(xv.x, yv.y, fma(-xv.y, yv.x, ai));
  }
  double r = block_sum_256(ar);
  if (threadIdx.x == 0) pre[blockIdx.x] = r;
  r = block_sum_256(ai);
  if (threadIdx.x == 0) pim[blockIdx.x] = r;
}
__global__ void __launch_bounds__(256) k_reduce_final2(const double *pre, const double *pim, int n, double *out) {
  const double *p = blockIdx.x ? pim : pre;
  double acc = 0;
  for (int i = threadIdx.x; i < n; i += 256) acc += p[i];
  double r = block_sum_256(acc);
  if (threadIdx.x == 0) out[blockIdx.x] = r;
}
int blas_cdot(qexhip_ctx *c, const DevField &x, const DevField &y, int parity, double *dev_out) {
  size_t n = body2(c);
  const int nb = grid_for(n);
  int tot = 0;
  double *pre = c->partials, *pim = c->partials + 2 * nb;       // up to 2 x 2 x 2048 partials: the buffer holds part2_off (>= 6144) + 2048 + 64
  if (4 * nb > c->part2_off + 2048) { qexhip_set_error("internal: partial buffer too small for cdot"); return -3; }
  {
    ScopedTimer tm(c, "blas", c->stream);
    FOR_PAR(parity, p) {
      k_cdot<<<nb, 256, 0, c->stream>>>(x.par(p), y.par(p), n, pre + tot, pim + tot);
      tot += nb;
    }
    k_reduce_final2<<<2, 256, 0, c->stream>>>(pre, pim, tot, dev_out);
    HIPCHK(hipGetLastError());
  }
  if (multi_rank(c)) CHK(comm_allreduce(c, dev_out, 2));
  return 0;
}

int blas_norm2(qexhip_ctx *c, const DevField &x, int parity, double *dev_out) {
  return blas_redot(c, x, x, parity, dev_out);
}

int read_scalars(qexhip_ctx *c, const double *dev, int n, double *host) {
  HIPCHK(hipMemcpyAsync(c->pinned, dev, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  CHK(peer_check(c));        // a bounded device-side wait of the peer transport gave up: the numbers below would be garbage
  for (int i = 0; i < n; i++) host[i] = ((double *)c->pinned)[i];
  return 0;
}

// ---------------- CG-specific fused kernels (src/solvers/cg.nim:174-214) ----------------
// One iteration is three kinds of launch: k_cg_xpay, two Dslash sweeps, k_cg_update.  The end-of-iteration bookkeeping
// (rzold := rz, r2 := |r|^2, ++itn, history, loop condition; cg.nim:174,194,214-217) is done by the k_cg_xpay that opens
// the NEXT iteration: every workgroup sums the |r|^2 workgroup partials itself (same fixed order in every workgroup, so
// all agree bit for bit) and workgroup 0 writes the state of iteration k into slot k&1 of the CgScal while the others
// still read slot (k-1)&1 -- no launch of its own, no read/write race.  The host passes k; at the end of a chunk of
// iterations k_cg_close does the same bookkeeping alone so that the host can read the state (`rolled` then tells the
// next k_cg_xpay that slot k&1 is already written).
// Sharded, the partial VECTORS are all-reduced (a few KB: the same latency as one double), which keeps the two one-block
// reduction launches out of the iteration as well.
// q := z (itn 0) | q := z + beta*q, beta = rz/rzo   (cg.nim:186-193; cpNone: z=r, q=p)
__global__ void __launch_bounds__(256) k_cg_xpay(double2 *p, const double2 *r, size_t n, CgScal *s, int k, int rolled,
                                                const double *r2parts, int nparts, double *hist, int histcap) {
  const int cur = k & 1, prv = cur ^ 1;
  double r2k;
  if (rolled) {
    if (s->dones[cur]) return;
    r2k = s->r2s[cur];
  } else {
    if (s->dones[prv]) {
      if (blockIdx.x == 0 && threadIdx.x == 0) cg_carry(s, k);
      return;
    }
    r2k = cg_sum_parts(r2parts, nparts);
    if (blockIdx.x == 0 && threadIdx.x == 0) cg_roll(s, k, r2k, hist, histcap);
    if (!(k < s->maxits && r2k > s->r2stop)) return;
  }
  const bool first = (k == 0);
  const double beta = r2k / s->r2s[prv];
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    double2 rv = r[i];
    if (first) p[i] = rv;
    else {
      double2 pv = p[i];
      p[i] = make_double2(fma(beta, pv.x, rv.x), fma(beta, pv.y, rv.y));   // explicit fma (what the compiler contracted this to anyway)
    }
  }
}
// alpha = rz/qLAp; x += alpha*p; r -= alpha*Ap; partial |r|^2   (cg.nim:208-213)
// With ndot > 0 every workgroup first sums the <p,Ap> workgroup partials of the preceding Dslash sweep itself.
__global__ void __launch_bounds__(256) k_cg_update(double2 *x, double2 *r, const double2 *p, const double2 *Ap,
                                                  size_t n, const CgScal *s, int k, double *partials,
                                                  const double *dotp, int ndot) {
  if (s->dones[k & 1]) return;
  const double pAp = (ndot > 0) ? cg_sum_parts(dotp, ndot) : s->pAp;
  const double alpha = s->r2s[k & 1] / pAp;
  double acc = 0;
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    double2 pv = p[i], xv = x[i], rv = r[i], av = Ap[i];
    xv.x += alpha * pv.x; xv.y += alpha * pv.y;
    rv.x -= alpha * av.x; rv.y -= alpha * av.y;
    x[i] = xv; r[i] = rv;
    acc = fma(rv.x, rv.x, fma(rv.y, rv.y, acc));
  }
  double t = block_sum_256(acc);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
// r2stop = r2req*b2; loop condition `itn<maxits and r2>r2stop` (cg.nim:155,174)
__global__ void k_cg_init(CgScal *s, const double *dscal, double r2req, int maxits, double *hist, int histcap) {
  s->b2 = dscal[0];
  s->r2 = dscal[1];
  s->rzo = 1.0;    // CgState.reset: rzold = 1.0 (cg.nim:21-27)
  s->pAp = 0.0;
  s->r2stop = r2req * s->b2;
  s->itn = 0;
  s->maxits = maxits;
  s->done = !(0 < maxits && s->r2 > s->r2stop);
  s->r2s[0] = s->r2; s->itns[0] = 0; s->dones[0] = s->done;
  s->r2s[1] = 1.0; s->itns[1] = 0; s->dones[1] = s->done;
  if (histcap > 0) hist[0] = (s->b2 != 0.0) ? s->r2 / s->b2 : 0.0;
}
// end of a chunk: the bookkeeping of the last iteration, so that the host can read slot k&1
__global__ void __launch_bounds__(256) k_cg_close(CgScal *s, int k, const double *r2parts, int nparts, double *hist, int histcap) {
  if (s->dones[(k & 1) ^ 1]) {
    if (threadIdx.x == 0) cg_carry(s, k);
    return;
  }
  const double r2k = cg_sum_parts(r2parts, nparts);
  if (threadIdx.x == 0) cg_roll(s, k, r2k, hist, histcap);
}

// The two rank sums of a sharded iteration (cg.nim:206-214: <p,Ap> and |r|^2 both end in threadRankSum) are launches of their own
// behind the kernels that produce the partials: one workgroup per rank, the scalar through the mailboxes (peer.hip), or RCCL's
// all-reduce of the partial vector.  Folding them into the tail of k_cg_update or the prologues of the consumers was measured twice in
// round 5 and lost / tied (profiles/r05_fold_compare.log, r05_fold2_compare.log); those forms are in the history, not in the library.
int cg_xpay(qexhip_ctx *c, DevField &p, const DevField &r, int parity, int k, int rolled) {
  size_t n = body2(c);
  ScopedTimer tm(c, "blas", c->stream);
  k_cg_xpay<<<grid_for(n), 256, 0, c->stream>>>(p.par(parity), r.par(parity), n, c->cg, k, rolled,
                                               c->partials + c->part2_off, rolled ? grid_for(n) : c->cg_r2parts, c->hist, c->histcap);
  HIPCHK(hipGetLastError());
  return 0;
}
// ndot > 0: <p,Ap> is still in workgroup partials c->partials[0..ndot) (deferred); the |r|^2 partials go to the upper
// part of the buffer and stay there for the next k_cg_xpay / k_cg_close.
int cg_update(qexhip_ctx *c, DevField &x, DevField &r, const DevField &p, const DevField &Ap, int parity, int k, int ndot) {
  size_t n = body2(c);
  int nb = grid_for(n);
  double *r2p = c->partials + c->part2_off;
  if (ndot > 0) CHK(comm_allreduce_parts(c, c->partials, ndot, &ndot));
  CHK(devjoin_flush(c));              // (no-op when the all-reduce has taken the second sweep's join with it)
  {
    ScopedTimer tm(c, "blas", c->stream);
    k_cg_update<<<nb, 256, 0, c->stream>>>(x.par(parity), r.par(parity), p.par(parity), Ap.par(parity), n, c->cg, k, r2p,
                                           c->partials, ndot);
    HIPCHK(hipGetLastError());
  }
  CHK(comm_allreduce_parts(c, r2p, nb, &c->cg_r2parts));     // how many values the next k_cg_xpay / k_cg_close has to sum
  return 0;
}
int cg_close(qexhip_ctx *c, int k) {
  k_cg_close<<<1, 256, 0, c->stream>>>(c->cg, k, c->partials + c->part2_off, c->cg_r2parts, c->hist, c->histcap);
  HIPCHK(hipGetLastError());
  return comm_agree_post(c);
}
// re-entry (cg.nim:155-161 with b2 >= 0): new r2stop / maxits, loop condition of the kept state re-evaluated in slot k&1
__global__ void k_cg_resume(CgScal *s, int k, double r2req, int maxits) {
  const int cur = k & 1;
  s->r2stop = r2req * s->b2;
  s->maxits = maxits;
  s->r2s[cur] = s->r2; s->r2s[cur ^ 1] = s->rzo; s->itns[cur] = k;      // rz, rzold, iterations as the last live iteration left them
  s->dones[cur] = !(k < maxits && s->r2s[cur] > s->r2stop);
  s->done = s->dones[cur];
  s->agree[0] = s->r2s[cur]; s->agree[1] = -s->r2s[cur]; s->agree[2] = (double)k; s->agree[3] = -(double)k;
}
int cg_resume(qexhip_ctx *c, int k, double r2req, int maxits) {
  k_cg_resume<<<1, 1, 0, c->stream>>>(c->cg, k, r2req, maxits);
  HIPCHK(hipGetLastError());
  return 0;
}
int cg_init(qexhip_ctx *c, double r2req, int maxits) {
  c->cg_resume.valid = 0;
  k_cg_init<<<1, 1, 0, c->stream>>>(c->cg, c->dscal, r2req, maxits, c->hist, c->histcap);
  HIPCHK(hipGetLastError());
  return 0;
}
