// batch.hip -- several independent even/odd CG solves on the SAME links advanced in lock-step.
//
// QEX's HMC solves one system after another on unchanged links: the Hasenbusch chain of the action and
// of the force (src/examples/staghmc_sh.nim:339-364,394-404: masses m, h0, h1 per species), the pbp
// repetitions (:260-272), the fork's fforce loop (src/stagg_pv_hmc/staghmc_spv.nim:758-830).  Each is
// `stag.solve` -> solveXX -> CG (src/physics/stagSolve.nim:57-138, src/solvers/cg.nim:55-272).  The Dslash
// is HBM-bound and 89 % of its bytes are links, so streaming the links ONCE for up to four right-hand
// sides cuts the bytes per system from 864 to (768 + 96 n)/n (352 at n = 3, compressed links).  Every
// system keeps its own CG state (alpha, beta, residual, iteration count, done flag) and its arithmetic
// is, operation for operation, that of the single-system path (same per-site accumulation order, same
// workgroup partials, same fixed-order final sums), so each solution and iteration count equals what
// qexhip_stag_solve_xx returns for that system alone.  t-sharded runs exchange the faces of all systems, then
// sweep the slab in one launch, and all-reduce the n scalars of a reduction in ONE call.
#include "qexhip_internal.h"
#include "site_index.h"
#include "reduce.h"
#include "dslash_core.h"
#include "peer_device.h"
#include <algorithm>
#include <cstring>
#include <vector>

#define QX_MAXRHS 4

struct MrhsArgs {
  Geom g;
  const double2 *W;
  const unsigned long long *S;
  const double2 *in[QX_MAXRHS];
  double2 *out[QX_MAXRHS];
  const double2 *xs[QX_MAXRHS];
  double cb[QX_MAXRHS];
  double *partials[QX_MAXRHS];
  const CgScal *st;            // st[j].done switches system j off
  int parity, nrhs, swz, ntstore;
  int c0, c1, d0, d1, nb1;     // site range [c0, c1) for workgroups < nb1, [d0, d1) for the rest (both t-faces in one launch)
  int part_off;                // first <xs, out> partial this launch writes
};

// SECOND = false: out_j = +sum_mu (U in_j(+mu) - U^+ in_j(-mu))              (stagDP, first half of stagD2ee)
// SECOND = true : out_j = cb_j xs_j - sum_mu (...), partial <xs_j, out_j>   (stagDM + 4m^2 x + <p,Ap>)
template <int NDIR, int RECON, bool SECOND, bool HALO>
__global__ void __launch_bounds__(256) k_dslash_mrhs(MrhsArgs A) {
  bool act[QX_MAXRHS];
  bool any = false;
#pragma unroll
  for (int j = 0; j < QX_MAXRHS; j++) { act[j] = j < A.nrhs && !A.st[j].done; any = any || act[j]; }
  if (!any) return;
  int bid = blockIdx.x;
  if (A.swz) bid = (bid & 7) * (A.swz >> 3) + (bid >> 3);
  int c = A.c0 + bid * 256 + threadIdx.x, clim = A.c1;
  if (bid >= A.nb1) { c = A.d0 + (bid - A.nb1) * 256 + threadIdx.x; clim = A.d1; }
  double dotv[QX_MAXRHS] = {0, 0, 0, 0};
  if (c < clim) {
    const Geom &g = A.g;
    SiteXYZT s = site_coord(g, c, A.parity);
    double2 acc[QX_MAXRHS][3], xsv[QX_MAXRHS][3];
    const double sgn = SECOND ? -1.0 : 1.0;
#pragma unroll
    for (int j = 0; j < QX_MAXRHS; j++) {
      if (!act[j]) continue;
      if (SECOND) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
          xsv[j][k] = A.xs[j][vec_off(c, k)];
          acc[j][k].x = (sgn * A.cb[j]) * xsv[j][k].x;
          acc[j][k].y = (sgn * A.cb[j]) * xsv[j][k].y;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 3; k++) acc[j][k] = make_double2(0.0, 0.0);
      }
    }
    constexpr int NLOAD = RECON == 1 ? 6 : (RECON == 2 ? 7 : 9);
    constexpr int LROW = NLOAD * 64;
    const double2 *w = A.W + (size_t)(c >> 6) * (NDIR * LROW) + (c & 63);
    const unsigned long long *sm = RECON == 1 ? A.S + (size_t)(c >> 6) * NDIR : nullptr;
#pragma unroll 1
    for (int pr = 0; pr < NDIR / 2; pr++) {
      const int mu = pr & 3;
      const int hop = pr >= 4 ? 3 : 1;
      const int pf = nbr_pos<HALO>(g, c, s, mu, hop);
      const int pb = nbr_pos<HALO>(g, c, s, mu, -hop);
      const double2 *wp = w + (size_t)pr * (2 * LROW);
      double2 U[9], Wm[9];
#pragma unroll
      for (int k = 0; k < NLOAD; k++) {
        d2v t = __builtin_nontemporal_load((const d2v *)&wp[k * 64]);
        U[k] = make_double2(t.x, t.y);
      }
#pragma unroll
      for (int k = 0; k < NLOAD; k++) {
        d2v t = __builtin_nontemporal_load((const d2v *)&wp[LROW + k * 64]);
        Wm[k] = make_double2(t.x, t.y);
      }
      if (RECON == 1) {
        const int lane = c & 63;
        recon_row2<1>(U, (sm[2 * pr] >> lane) & 1ull);
        recon_row2<1>(Wm, (sm[2 * pr + 1] >> lane) & 1ull);
      } else if (RECON == 2) {
        recon_row2<2>(U, false);
        recon_row2<2>(Wm, false);
      }
#pragma unroll
      for (int j = 0; j < QX_MAXRHS; j++) {
        if (!act[j]) continue;
        double2 vf[3], vb[3];
#pragma unroll
        for (int k = 0; k < 3; k++) vf[k] = A.in[j][vec_off(pf, k)];
#pragma unroll
        for (int k = 0; k < 3; k++) vb[k] = A.in[j][vec_off(pb, k)];
        mv3<false>(acc[j], U, vf);
        mv3<true>(acc[j], Wm, vb);
      }
    }
#pragma unroll
    for (int j = 0; j < QX_MAXRHS; j++) {
      if (!act[j]) continue;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        acc[j][k].x *= sgn; acc[j][k].y *= sgn;
        if (A.ntstore) {
          d2v t; t.x = acc[j][k].x; t.y = acc[j][k].y;
          __builtin_nontemporal_store(t, (d2v *)&A.out[j][vec_off(c, k)]);
        } else {
          A.out[j][vec_off(c, k)] = acc[j][k];
        }
      }
      if (SECOND) {
#pragma unroll
        for (int k = 0; k < 3; k++) dotv[j] = fma(xsv[j][k].x, acc[j][k].x, fma(xsv[j][k].y, acc[j][k].y, dotv[j]));
      }
    }
  }
  if (SECOND) {
#pragma unroll
    for (int j = 0; j < QX_MAXRHS; j++) {
      if (!act[j]) continue;             // uniform over the grid
      double r = block_sum_256(dotv[j]);
      if (threadIdx.x == 0) A.partials[j][A.part_off + blockIdx.x] = r;
    }
  }
}

// The lock-step sweep of a t-sharded slab on the peer transport as ONE launch on ONE stream -- k_dslash_fused (dslash.hip) for up to four
// systems: push workgroups (the faces of all systems, one piece each) | interior | boundary (hops inside the slab, SHORT wait, hops from
// the receive arena or park) | cleanup (parked blocks behind the one long wait).  A system's arithmetic is k_dslash_fused's: same hop
// order (local first), same partial slots; parked or not a block gives the same bits.
struct MrhsFusedArgs {
  MrhsArgs a;                       // c0..c1 interior, d0..d1 low face (workgroups >= nb1), e0..e1 below
  int e0, e1, nb2, nbA;
  const double2 *gh_hi[QX_MAXRHS], *gh_lo[QX_MAXRHS];
  PeerGhost pg;
  PeerPush push;
  FusedCtl fz;
};
template <int NDIR, int RECON, bool SECOND>
__global__ void __launch_bounds__(256) k_dslash_mrhs_fused(MrhsFusedArgs F) {
  const MrhsArgs &A = F.a;
  bool act[QX_MAXRHS];
  bool any = false;
#pragma unroll
  for (int j = 0; j < QX_MAXRHS; j++) { act[j] = j < A.nrhs && !A.st[j].done; any = any || act[j]; }
  const bool skip = !any;              // every system has converged: nothing is pushed, the credits still go back
  __shared__ int sh_n;
  int bid = blockIdx.x;
  if (bid < F.push.nblocks) {
    if (!skip) peer_push_block(F.push, (unsigned)bid);
    return;
  }
  bid -= F.push.nblocks;
  const int ngrid = (int)gridDim.x - F.push.nblocks - F.fz.ncl;
  const int nbnd = ngrid - A.nb1;
  const bool cleanup = bid >= ngrid;
  int npark = 0, jpark = 0;
  if (cleanup) {
    if (threadIdx.x == 0) {
      int n = 0;
      if (peer_poll_u32(F.fz.dec, (unsigned)nbnd, F.pg.err, F.pg.ticks, 0x520)) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        n = (int)__hip_atomic_load(F.fz.ndef, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n > 0) {
          if (!peer_ghost_wait(F.pg)) n = -1;
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      } else n = -1;
      sh_n = n;
    }
    __syncthreads();
    npark = sh_n;
    jpark = bid - ngrid;
  }
  bool parked = false, bnd = false;
  for (;;) {
    int lb = bid;
    if (cleanup) {
      if (jpark >= npark) break;
      lb = (int)F.fz.list[jpark];
      jpark += F.fz.ncl;
      bnd = true;
    } else {
      bnd = bid >= F.nbA && bid < F.nbA + nbnd;
      lb = bnd ? A.nb1 + (bid - F.nbA) : (bid < F.nbA ? bid : bid - nbnd);
    }
    int c = A.c0 + lb * 256 + threadIdx.x, clim = A.c1;
    if (bnd) {
      c = A.d0 + (lb - A.nb1) * 256 + threadIdx.x; clim = A.d1;
      if (lb >= F.nb2) { c = F.e0 + (lb - F.nb2) * 256 + threadIdx.x; clim = F.e1; }
    }
    double dotv[QX_MAXRHS] = {0, 0, 0, 0};
    const bool active = c < clim && !skip;
    const Geom &g = A.g;
    const SiteXYZT s = site_coord(g, c, A.parity);
    const int tu = __builtin_amdgcn_readfirstlane(s.t);
    const bool hi1 = tu + 1 >= g.X[3], lo1 = tu - 1 < 0, hi3 = tu + 3 >= g.X[3], lo3 = tu - 3 < 0;
    double2 acc[QX_MAXRHS][3], xsv[QX_MAXRHS][3];
    const double sgn = SECOND ? -1.0 : 1.0;
    constexpr int NLOAD = RECON == 1 ? 6 : (RECON == 2 ? 7 : 9);
    constexpr int LROW = NLOAD * 64;
    const double2 *w = A.W + (size_t)(c >> 6) * (NDIR * LROW) + (c & 63);
    const unsigned long long *sm = RECON == 1 ? A.S + (size_t)(c >> 6) * NDIR : nullptr;
    auto pair = [&](const int pr, const bool do_f, const bool do_b) __attribute__((always_inline)) {
      const int mu = pr & 3;
      const int hop = pr >= 4 ? 3 : 1;
      const int pf = nbr_pos<true>(g, c, s, mu, hop);
      const int pb = nbr_pos<true>(g, c, s, mu, -hop);
      const double2 *wp = w + (size_t)pr * (2 * LROW);
      double2 U[9], Wm[9];
      if (do_f) {
#pragma unroll
        for (int k = 0; k < NLOAD; k++) {
          d2v t = __builtin_nontemporal_load((const d2v *)&wp[k * 64]);
          U[k] = make_double2(t.x, t.y);
        }
      }
      if (do_b) {
#pragma unroll
        for (int k = 0; k < NLOAD; k++) {
          d2v t = __builtin_nontemporal_load((const d2v *)&wp[LROW + k * 64]);
          Wm[k] = make_double2(t.x, t.y);
        }
      }
      if (RECON == 1) {
        const int lane = c & 63;
        if (do_f) recon_row2<1>(U, (sm[2 * pr] >> lane) & 1ull);
        if (do_b) recon_row2<1>(Wm, (sm[2 * pr + 1] >> lane) & 1ull);
      } else if (RECON == 2) {
        if (do_f) recon_row2<2>(U, false);
        if (do_b) recon_row2<2>(Wm, false);
      }
      const bool xf = mu == 3 && (hop == 3 ? hi3 : hi1), xb = mu == 3 && (hop == 3 ? lo3 : lo1);     // this hop leaves the slab (wavefront-uniform)
#pragma unroll
      for (int j = 0; j < QX_MAXRHS; j++) {
        if (!act[j]) continue;
        double2 vf[3], vb[3];
        if (do_f) {
          const double2 *src = xf ? F.gh_hi[j] : A.in[j];
#pragma unroll
          for (int k = 0; k < 3; k++) vf[k] = src[vec_off(pf, k)];
          mv3<false>(acc[j], U, vf);
        }
        if (do_b) {
          const double2 *src = xb ? F.gh_lo[j] : A.in[j];
#pragma unroll
          for (int k = 0; k < 3; k++) vb[k] = src[vec_off(pb, k)];
          mv3<true>(acc[j], Wm, vb);
        }
      }
    };
    auto edge_pairs = [&](const bool crossing) __attribute__((always_inline)) {
      if (!crossing) {
        constexpr int NSP = NDIR / 2 - NDIR / 8;
#pragma unroll 1
        for (int q = 0; q < NSP; q++) pair(q + q / 3, true, true);
      }
#pragma unroll 1
      for (int pr = 3; pr < NDIR / 2; pr += 4) {
        const int hop = pr >= 4 ? 3 : 1;
        const bool xf = tu + hop >= g.X[3], xb = tu - hop < 0;
        pair(pr, crossing ? xf : !xf, crossing ? xb : !xb);
      }
    };
    if (active) {
#pragma unroll
      for (int j = 0; j < QX_MAXRHS; j++) {
        if (!act[j]) continue;
        if (SECOND) {
#pragma unroll
          for (int k = 0; k < 3; k++) xsv[j][k] = A.xs[j][vec_off(c, k)];
        }
        if (cleanup) {
#pragma unroll
          for (int k = 0; k < 3; k++) acc[j][k] = A.out[j][vec_off(c, k)];
        } else if (SECOND) {
#pragma unroll
          for (int k = 0; k < 3; k++) {
            acc[j][k].x = (sgn * A.cb[j]) * xsv[j][k].x;
            acc[j][k].y = (sgn * A.cb[j]) * xsv[j][k].y;
          }
        } else {
#pragma unroll
          for (int k = 0; k < 3; k++) acc[j][k] = make_double2(0.0, 0.0);
        }
      }
      if (!bnd) {
#pragma unroll 1
        for (int pr = 0; pr < NDIR / 2; pr++) pair(pr, true, true);
      } else if (!cleanup) {
        edge_pairs(false);
      }
    }
    if (bnd && !skip) {
      if (!cleanup) {
        if (threadIdx.x == 0) {
          const bool in = F.fz.spin_ticks >= 0 && peer_ghost_try(F.pg, F.fz.spin_ticks, F.fz.late);
          if (in) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          sh_n = in ? 1 : 0;
        }
        __syncthreads();
        parked = sh_n == 0;
      }
      if (active && !parked) edge_pairs(true);
    }
    if (active) {
#pragma unroll
      for (int j = 0; j < QX_MAXRHS; j++) {
        if (!act[j]) continue;
        if (parked) {
#pragma unroll
          for (int k = 0; k < 3; k++) A.out[j][vec_off(c, k)] = acc[j][k];
        } else {
#pragma unroll
          for (int k = 0; k < 3; k++) {
            acc[j][k].x *= sgn; acc[j][k].y *= sgn;
            d2v t; t.x = acc[j][k].x; t.y = acc[j][k].y;
            __builtin_nontemporal_store(t, (d2v *)&A.out[j][vec_off(c, k)]);
          }
          if (SECOND) {
#pragma unroll
            for (int k = 0; k < 3; k++) dotv[j] = fma(xsv[j][k].x, acc[j][k].x, fma(xsv[j][k].y, acc[j][k].y, dotv[j]));
          }
        }
      }
    }
    if (SECOND && !skip && !parked) {
      const int pidx = cleanup ? F.nbA + (lb - A.nb1) : bid;
#pragma unroll
      for (int j = 0; j < QX_MAXRHS; j++) {
        if (!act[j]) continue;             // uniform over the grid
        double r = block_sum_256(dotv[j]);
        if (threadIdx.x == 0) A.partials[j][A.part_off + pidx] = r;
      }
    }
    if (!cleanup) break;
  }
  if (!(bnd || cleanup)) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x != 0) return;
  bool credits = false;
  if (!cleanup) {
    if (parked) {
      const unsigned idx = __hip_atomic_fetch_add(F.fz.ndef, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&F.fz.list[idx], (unsigned)(A.nb1 + (bid - F.nbA)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned a = __hip_atomic_fetch_add(F.fz.dec, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a == (unsigned)nbnd - 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      credits = __hip_atomic_load(F.fz.ndef, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
    }
  } else {
    const unsigned a = __hip_atomic_fetch_add(F.fz.cl_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a == (unsigned)F.fz.ncl - 1) {
      credits = npark > 0;
      __hip_atomic_store(F.fz.ndef, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(F.fz.dec, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(F.fz.late, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(F.fz.cl_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (credits) {
    __hip_atomic_store(F.pg.credit[0], F.pg.credit_val[0], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(F.pg.credit[1], F.pg.credit_val[1], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

struct BatchBlas {
  double2 *x[QX_MAXRHS], *r[QX_MAXRHS], *p[QX_MAXRHS], *Ap[QX_MAXRHS];
  double *dotp[QX_MAXRHS], *r2p[QX_MAXRHS];
  CgScal *st;
  double *glob;      // multi-rank: [0..4) rank-summed <p,Ap>, [4..8) rank-summed |r|^2 (contiguous for one all-reduce)
  int ndot;          // > 0: deferred <p,Ap> partial sum inside k_cgb_update; 0: pAp from the state; -1: from glob
};
// the CG kernels of blas.hip in their round-1 form (xpay, update, one-block reduce + bookkeeping), system = blockIdx.y
__global__ void __launch_bounds__(256) k_cgb_xpay(BatchBlas B, size_t n) {
  const int j = blockIdx.y;
  const CgScal *s = &B.st[j];
  if (s->done) return;
  const bool first = (s->itn == 0);
  const double beta = s->r2 / s->rzo;
  double2 *p = B.p[j];
  const double2 *r = B.r[j];
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    double2 rv = r[i];
    if (first) p[i] = rv;
    else {
      double2 pv = p[i];
      p[i] = make_double2(rv.x + beta * pv.x, rv.y + beta * pv.y);
    }
  }
}
__global__ void __launch_bounds__(256) k_cgb_update(BatchBlas B, size_t n) {
  const int j = blockIdx.y;
  const CgScal *s = &B.st[j];
  if (s->done) return;
  double pAp = B.ndot < 0 ? B.glob[j] : s->pAp;
  if (B.ndot > 0) {
    double a = 0;
    for (int i = threadIdx.x; i < B.ndot; i += 256) a += B.dotp[j][i];
    pAp = block_sum_256_all(a);
  }
  const double alpha = s->r2 / pAp;
  double2 *x = B.x[j], *r = B.r[j];
  const double2 *p = B.p[j], *Ap = B.Ap[j];
  double acc = 0;
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    double2 pv = p[i], xv = x[i], rv = r[i], av = Ap[i];
    xv.x += alpha * pv.x; xv.y += alpha * pv.y;
    rv.x -= alpha * av.x; rv.y -= alpha * av.y;
    x[i] = xv; r[i] = rv;
    acc = fma(rv.x, rv.x, fma(rv.y, rv.y, acc));
  }
  double t = block_sum_256(acc);
  if (threadIdx.x == 0) B.r2p[j][blockIdx.x] = t;
}
// big local volumes: one-block final sum of the <p,Ap> partials per system (as reduce_partials does for the
// single-system path beyond 4096 workgroups); k_cgb_update then takes pAp from the state (ndot = 0)
__global__ void __launch_bounds__(256) k_cgb_reduce_dot(BatchBlas B, int n) {
  const int j = blockIdx.x;
  CgScal *s = &B.st[j];
  if (s->done) return;
  double acc = 0;
  for (int i = threadIdx.x; i < n; i += 256) acc += B.dotp[j][i];
  double r = block_sum_256(acc);
  if (threadIdx.x == 0) s->pAp = r;
}
// multi-rank pieces: local sums into the contiguous buffer (then ONE ncclAllReduce for all systems), bookkeeping from it
__global__ void __launch_bounds__(256) k_cgb_local_sum(BatchBlas B, int n, int which) {   // which 0: <p,Ap>, 1: |r|^2
  const int j = blockIdx.x;
  if (B.st[j].done) { if (threadIdx.x == 0) B.glob[4 * which + j] = 0.0; return; }
  const double *src = which ? B.r2p[j] : B.dotp[j];
  double acc = 0;
  for (int i = threadIdx.x; i < n; i += 256) acc += src[i];
  double r = block_sum_256(acc);
  if (threadIdx.x == 0) B.glob[4 * which + j] = r;
}
__global__ void k_cgb_finish_glob(BatchBlas B, int n) {
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    CgScal *s = &B.st[j];
    if (s->done) continue;
    s->rzo = s->r2;
    s->r2 = B.glob[4 + j];
    s->itn += 1;
    if (!(s->itn < s->maxits && s->r2 > s->r2stop)) s->done = 1;
  }
}
__global__ void __launch_bounds__(256) k_cgb_reduce_finish(BatchBlas B, int n) {
  const int j = blockIdx.x;
  CgScal *s = &B.st[j];
  if (s->done) return;
  double acc = 0;
  for (int i = threadIdx.x; i < n; i += 256) acc += B.r2p[j][i];
  double r = block_sum_256(acc);
  if (threadIdx.x == 0) {
    s->rzo = s->r2;
    s->r2 = r;
    s->itn += 1;
    if (!(s->itn < s->maxits && s->r2 > s->r2stop)) s->done = 1;
  }
}

struct BatchState {
  std::vector<DevField> f;     // r, p, Ap, t per system
  CgScal *st = nullptr;
  double *glob = nullptr;
  double *partials = nullptr;
  size_t npart = 0;
};
void batch_state_free(qexhip_ctx *c) {
  BatchState *b = (BatchState *)c->batch;
  if (!b) return;
  for (auto &f : b->f) (void)hipFree(f.d);
  if (b->st) (void)hipFree(b->st);
  if (b->glob) (void)hipFree(b->glob);
  if (b->partials) (void)hipFree(b->partials);
  delete b;
  c->batch = nullptr;
}

template <int NDIR, int RECON>
static void launch_mrhs(qexhip_ctx *c, MrhsArgs &A, bool second, int nb, hipStream_t st) {
  if (c->g.halo) {
    if (second) hipLaunchKernelGGL((k_dslash_mrhs<NDIR, RECON, true, true>), dim3(nb), dim3(256), 0, st, A);
    else hipLaunchKernelGGL((k_dslash_mrhs<NDIR, RECON, false, true>), dim3(nb), dim3(256), 0, st, A);
  } else {
    if (second) hipLaunchKernelGGL((k_dslash_mrhs<NDIR, RECON, true, false>), dim3(nb), dim3(256), 0, st, A);
    else hipLaunchKernelGGL((k_dslash_mrhs<NDIR, RECON, false, false>), dim3(nb), dim3(256), 0, st, A);
  }
}
// one launch over [c0, c1) (+ [d0, d1)); returns the number of workgroups = <xs, out> partials it writes from part_off on
static int launch_range(qexhip_ctx *c, MrhsArgs &A, bool second, int c0, int c1, int d0, int d1, int part_off, hipStream_t st) {
  A.c0 = c0; A.c1 = c1; A.d0 = d0; A.d1 = d1;
  A.nb1 = (c1 - c0 + 255) / 256;
  const int nb = A.nb1 + (d1 > d0 ? (d1 - d0 + 255) / 256 : 0);
  A.part_off = part_off;
  const bool whole = (c0 == 0 && c1 == c->g.Vh && d1 <= d0);
  A.swz = (whole && c->recon != 0 && nb >= 64 && (nb & 7) == 0) ? nb : 0;     // as dslash.hip: on for compressed links
  A.ntstore = 1;
  if (c->ndir == 8) {
    if (c->recon == 1) launch_mrhs<8, 1>(c, A, second, nb, st);
    else if (c->recon == 2) launch_mrhs<8, 2>(c, A, second, nb, st);
    else launch_mrhs<8, 0>(c, A, second, nb, st);
  } else {
    if (c->recon == 1) launch_mrhs<16, 1>(c, A, second, nb, st);
    else if (c->recon == 2) launch_mrhs<16, 2>(c, A, second, nb, st);
    else launch_mrhs<16, 0>(c, A, second, nb, st);
  }
  return nb;
}
// t-sharded: the faces of every system's input field travel in ONE RCCL group; either exchange-first and one launch over the
// slab, or -- where dslash_sweep overlaps (sweep_plan) -- the group and the boundary launch on the comm stream beside the
// interior launch, exactly the structure of dslash_sweep.  *ndot = number of <xs, out> partials the sweep wrote.
static int sweep_mrhs(qexhip_ctx *c, MrhsArgs &A, bool second, int *ndot, DevField *const *infield = nullptr, int inpar = 0) {
  const Geom &g = c->g;
  int lo_end = 0, hi_beg = g.Vh, overlap = 0;
  if (g.halo) sweep_plan(c, &lo_end, &hi_beg, &overlap);
  // the decision of sweep_plan is for ONE system's faces; a batch moves nrhs times as much per exchange, and between distinct
  // GPUs that transfer is what the overlap is for: with a real communicator, overlap from 1 MiB of faces per direction on
  // -- unless the single-system form was MEASURED at set_links and lost: a measurement outranks the rule (round-4 advice)
  if (g.halo && !overlap && c->opt_overlap < 0 && c->nranks > 1 && hi_beg > lo_end && c->overlap_auto[c->ndir == 16] < 0 &&
      (size_t)A.nrhs * g.depth * g.F * 48 >= ((size_t)1 << 20)) overlap = 1;
  if (g.halo && overlap && sweep_form(c, overlap) == 2) {
    // the fused lock-step sweep (peer transport): the launch pushes all systems' faces itself and reads what arrives in the arena
    MrhsFusedArgs Fz;
    memset(&Fz, 0, sizeof Fz);
    CHK(devjoin_flush(c));
    CHK(comm_halo_push_only_multi(c, A.nrhs, infield, inpar, Fz.gh_hi, Fz.gh_lo, &Fz.push));
    CHK(peer_ghost_args(c, &Fz.pg));
    const int nb_int = (hi_beg - lo_end + 255) / 256, nb_lo = (lo_end + 255) / 256, nb_hi = (g.Vh - hi_beg + 255) / 256;
    CHK(sweep_fused_ctl(c, nb_lo + nb_hi, &Fz.fz, A.nrhs));
    A.c0 = lo_end; A.c1 = hi_beg; A.d0 = 0; A.d1 = lo_end; A.nb1 = nb_int; A.part_off = 0; A.swz = 0; A.ntstore = 1;
    Fz.a = A;
    Fz.e0 = hi_beg; Fz.e1 = g.Vh; Fz.nb2 = nb_int + nb_lo;
    Fz.nbA = (int)(sweep_push_fraction(c, hi_beg - lo_end, A.nrhs) * nb_int);
    const int grid = Fz.push.nblocks + nb_int + nb_lo + nb_hi + Fz.fz.ncl;
    ScopedTimer tm(c, "dslash_batch", c->stream);
#define QX_MF(ND, RC) \
    do { \
      if (second) hipLaunchKernelGGL((k_dslash_mrhs_fused<ND, RC, true>), dim3(grid), dim3(256), 0, c->stream, Fz); \
      else hipLaunchKernelGGL((k_dslash_mrhs_fused<ND, RC, false>), dim3(grid), dim3(256), 0, c->stream, Fz); \
    } while (0)
    if (c->ndir == 8) { if (c->recon == 1) QX_MF(8, 1); else if (c->recon == 2) QX_MF(8, 2); else QX_MF(8, 0); }
    else { if (c->recon == 1) QX_MF(16, 1); else if (c->recon == 2) QX_MF(16, 2); else QX_MF(16, 0); }
#undef QX_MF
    HIPCHK(hipGetLastError());
    *ndot = nb_int + nb_lo + nb_hi;
    return 0;
  }
  if (g.halo && overlap) HIPCHK(hipEventRecord(c->ev_ready, c->stream));
  if (g.halo) CHK(comm_halo_exchange_multi(c, A.nrhs, infield, inpar, overlap));
  ScopedTimer tm(c, "dslash_batch", c->stream);
  if (!overlap) {
    *ndot = launch_range(c, A, second, 0, g.Vh, 0, 0, 0, c->stream);
  } else {
    const int nb_int = launch_range(c, A, second, lo_end, hi_beg, 0, 0, 0, c->stream);
    const int nb_bnd = launch_range(c, A, second, 0, lo_end, hi_beg, g.Vh, nb_int, c->cstream);
    CHK(devjoin_signal(c, c->cstream));               // device-side join (as dslash_sweep): ~15 us instead of the ~28 us of an event dependency
    CHK(devjoin_wait(c, c->stream, c->cstream));
    *ndot = nb_int + nb_bnd;
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// n (1..4) solveXX's in lock-step; x[j], b[j] device fields (x zeroed here, as solveXX does)
int solve_xx_batch_dev(qexhip_ctx *c, int n, DevField **x, DevField **b, const double *mass, const double *r2req,
                       int maxits, int par_even, int *iters, double *r2_over_b2) {
  const Geom &g = c->g;
  if (n < 1 || n > QX_MAXRHS) { qexhip_set_error("batch solve: 1 <= n <= %d", QX_MAXRHS); return -1; }
  if (!c->W) { qexhip_set_error("staggered links not set (qexhip_stag_set_links)"); return -3; }
  for (int j = 0; j < n; j++) if (mass[j] == 0.0) { qexhip_set_error("batch solve: mass must be non-zero"); return -1; }
  BatchState *B = (BatchState *)c->batch;
  if (!B) { B = new BatchState(); c->batch = B; }
  while ((int)B->f.size() < 4 * QX_MAXRHS) {
    DevField nf;
    CHK(field_alloc(c, nf));
    B->f.push_back(nf);
  }
  if (!B->st) HIPCHK(hipMalloc((void **)&B->st, sizeof(CgScal) * QX_MAXRHS));
  if (!B->glob) { HIPCHK(hipMalloc((void **)&B->glob, sizeof(double) * 8)); HIPCHK(hipMemsetAsync(B->glob, 0, sizeof(double) * 8, c->stream)); }
  const int nbd = (g.Vh + 255) / 256 + 4;             // room for the <p,Ap> partials of a sweep per system (a split sweep rounds up per range)
  const size_t nvec = (size_t)g.ntile * 192;
  const int nbb = (int)std::min<size_t>((nvec + 255) / 256, 2048);
  const size_t per = (size_t)nbd + nbb;
  if (B->npart < per * QX_MAXRHS) {
    if (B->partials) HIPCHK(hipFree(B->partials));
    B->partials = nullptr; B->npart = 0;
    HIPCHK(hipMalloc((void **)&B->partials, per * QX_MAXRHS * sizeof(double)));
    B->npart = per * QX_MAXRHS;
  }
  const int par = par_even ? 0 : 1;
  MrhsArgs A1, A2;
  BatchBlas L;
  DevField *pf[QX_MAXRHS], *tf[QX_MAXRHS];
  memset(&A1, 0, sizeof A1); memset(&A2, 0, sizeof A2); memset(&L, 0, sizeof L);
  for (int j = 0; j < n; j++) {
    DevField &r = B->f[4 * j], &p = B->f[4 * j + 1], &Ap = B->f[4 * j + 2], &t = B->f[4 * j + 3];
    // exactly the start of solve_xx_dev (solver.cpp): x = 0, b2, r = b, r2, cg_init -> copy the state
    CHK(blas_zero(c, *x[j], 2));
    CHK(blas_norm2(c, *b[j], par, &c->dscal[0]));
    CHK(blas_copy(c, r, *b[j], par));
    CHK(blas_norm2(c, r, par, &c->dscal[1]));
    CHK(cg_init(c, r2req[j], maxits));
    HIPCHK(hipMemcpyAsync(&B->st[j], c->cg, sizeof(CgScal), hipMemcpyDeviceToDevice, c->stream));
    pf[j] = &p; tf[j] = &t;
    A1.in[j] = p.par(par); A1.out[j] = t.par(1 - par);
    A2.in[j] = t.par(1 - par); A2.out[j] = Ap.par(par); A2.xs[j] = p.par(par);
    A2.cb[j] = 4.0 * mass[j] * mass[j];
    A2.partials[j] = B->partials + per * j;
    L.x[j] = x[j]->par(par); L.r[j] = r.par(par); L.p[j] = p.par(par); L.Ap[j] = Ap.par(par);
    L.dotp[j] = B->partials + per * j; L.r2p[j] = B->partials + per * j + nbd;
  }
  const bool multi = c->nranks > 1 || c->opt_batch_multi;
  const bool deferred = !multi && nbd <= 4096;          // same switch as dslash_sweep / k_cg_update
  L.st = B->st; L.glob = B->glob; L.ndot = multi ? -1 : (deferred ? nbd : 0);
  for (MrhsArgs *A : {&A1, &A2}) {
    A->g = g; A->st = B->st; A->nrhs = n;
    if (c->recon) {
      A->W = c->Wc + (size_t)(A == &A1 ? 1 - par : par) * g.ntile * c->ndir * (c->recon == 1 ? 384 : 448);
      A->S = c->Ws + (size_t)(A == &A1 ? 1 - par : par) * g.ntile * c->ndir;
    } else {
      A->W = c->W + (size_t)(A == &A1 ? 1 - par : par) * g.ntile * c->ndir * 576;
      A->S = nullptr;
    }
    A->parity = (A == &A1) ? 1 - par : par;
  }
  CgScal st[QX_MAXRHS];
  auto read_states = [&]() -> int {
    HIPCHK(hipMemcpyAsync(c->pinned, B->st, sizeof(CgScal) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(st, c->pinned, sizeof(CgScal) * n);
    return 0;
  };
  CHK(read_states());
  for (;;) {
    int left = 0;
    for (int j = 0; j < n; j++) if (!st[j].done) left = std::max(left, st[j].maxits - st[j].itn);
    if (left <= 0) break;
    const int chunk = std::min(32, left);
    for (int i = 0; i < chunk; i++) {
      {
        ScopedTimer tm(c, "blas", c->stream);
        k_cgb_xpay<<<dim3(nbb, n), 256, 0, c->stream>>>(L, nvec);
      }
      int nd1 = 0, ndots = 0;
      CHK(sweep_mrhs(c, A1, false, &nd1, pf, par));
      CHK(sweep_mrhs(c, A2, true, &ndots, tf, 1 - par));
      if (!multi && deferred) L.ndot = ndots;
      if (multi) {
        k_cgb_local_sum<<<n, 256, 0, c->stream>>>(L, ndots, 0);
        CHK(comm_allreduce(c, B->glob, n));
      } else if (!deferred) {
        k_cgb_reduce_dot<<<n, 256, 0, c->stream>>>(L, ndots);
      }
      {
        ScopedTimer tm(c, "blas", c->stream);
        k_cgb_update<<<dim3(nbb, n), 256, 0, c->stream>>>(L, nvec);
      }
      if (multi) {
        k_cgb_local_sum<<<n, 256, 0, c->stream>>>(L, nbb, 1);
        CHK(comm_allreduce(c, B->glob + 4, n));
        k_cgb_finish_glob<<<1, 64, 0, c->stream>>>(L, n);
      } else {
        k_cgb_reduce_finish<<<n, 256, 0, c->stream>>>(L, nbb);
      }
      HIPCHK(hipGetLastError());
    }
    CHK(read_states());
  }
  for (int j = 0; j < n; j++) {
    if (iters) iters[j] = st[j].itn;
    if (r2_over_b2) r2_over_b2[j] = (st[j].b2 != 0.0) ? st[j].r2 / st[j].b2 : 0.0;
  }
  return 0;
}

// ---- n full solves D x_j = b_j (Staggered.solve, stagSolve.nim:224-294) with the even/odd CGs batched ----
// Same decisions per system as solve_full_dev / solve_inner (solver.cpp): reconstruct-right for one-parity
// sources (the HMC case: phi.odd = 0), reconstruct-left otherwise, true-residual outer loop.
namespace {
struct Sys {
  DevField *x, *b, *r, *y, *d;    // solution, source, residual, inner solution, inner source / work
  double m, r2req, b2, r2stop, r2e, r2o;
  int its, n, kind, par;          // kind 0: ReconR, 1: ReconL
  double rr;
  DevField *src;
};
int norm2_eo(qexhip_ctx *c, DevField &f, double *e, double *o) {
  CHK(blas_norm2(c, f, 0, &c->dscal[2]));
  CHK(blas_norm2(c, f, 1, &c->dscal[3]));
  double h[2];
  CHK(read_scalars(c, &c->dscal[2], 2, h));
  *e = h[0]; *o = h[1];
  return 0;
}
}  // namespace

int solve_full_batch_dev(qexhip_ctx *c, int n, DevField **x, DevField **b, const double *mass, const double *r2req,
                         int maxits, int *iters, double *r2_final) {
  if (n < 1 || n > QX_MAXRHS) { qexhip_set_error("batch solve: 1 <= n <= %d", QX_MAXRHS); return -1; }
  BatchState *B = (BatchState *)c->batch;
  if (!B) { B = new BatchState(); c->batch = B; }
  while ((int)B->f.size() < 7 * QX_MAXRHS) {     // 4 CG fields + r, y, d per system
    DevField nf;
    CHK(field_alloc(c, nf));
    B->f.push_back(nf);
  }
  Sys S[QX_MAXRHS];
  std::vector<int> active;
  for (int j = 0; j < n; j++) {
    Sys &s = S[j];
    s.x = x[j]; s.b = b[j];
    s.r = &B->f[4 * QX_MAXRHS + 3 * j]; s.y = &B->f[4 * QX_MAXRHS + 3 * j + 1]; s.d = &B->f[4 * QX_MAXRHS + 3 * j + 2];
    s.m = mass[j]; s.r2req = r2req[j]; s.its = 0;
    CHK(blas_norm2(c, *s.b, 2, &c->dscal[2]));
    CHK(read_scalars(c, &c->dscal[2], 1, &s.b2));
    s.r2stop = s.r2req * s.b2;
    CHK(blas_zero(c, *s.x, 2));
    CHK(blas_copy(c, *s.r, *s.b, 2));
    CHK(norm2_eo(c, *s.r, &s.r2e, &s.r2o));
    if (s.r2e + s.r2o > s.r2stop) active.push_back(j);
  }
  while (!active.empty()) {
    // inner solve set-up per system (solve_inner)
    for (int j : active) {
      Sys &s = S[j];
      const double r2 = s.r2e + s.r2o, rq = s.r2stop / r2;
      const double stop = rq * r2, stop2 = 0.5 * stop;
      const double stope = (s.r2o <= stop2) ? stop - s.r2o : stop2;
      const double stopo = (s.r2e <= stop2) ? stop - s.r2e : stop2;
      s.n = 0;
      if (s.r2e <= stope || s.r2o <= stopo) {
        s.kind = 0; s.src = s.r;
        if (s.r2e > stope) { s.par = 1; s.rr = stope / s.r2e; }
        else if (s.r2o > stopo) { s.par = 0; s.rr = stopo / s.r2o; }
        else s.par = -1;                                      // nothing to do
      } else {
        s.kind = 1; s.par = 1; s.src = s.d;
        CHK(op_D(c, *s.d, *s.r, s.m, -1.0));
        CHK(blas_norm2(c, *s.d, 0, &c->dscal[2]));
        double d2e;
        CHK(read_scalars(c, &c->dscal[2], 1, &d2e));
        s.rr = 0.99 * rq * r2 * s.m * s.m / d2e;
      }
    }
    for (int par = 1; par >= 0; par--) {
      DevField *xs[QX_MAXRHS], *bs[QX_MAXRHS];
      double ms[QX_MAXRHS], rrs[QX_MAXRHS];
      int idx[QX_MAXRHS], its[QX_MAXRHS], k = 0, mx = 0;
      for (int j : active)
        if (S[j].par == par) {
          xs[k] = S[j].y; bs[k] = S[j].src; ms[k] = S[j].m; rrs[k] = S[j].rr; idx[k] = j;
          mx = std::max(mx, maxits - S[j].its);
          k++;
        }
      if (!k) continue;
      // systems of one group share maxits: the remaining budget of the one that has used least
      CHK(solve_xx_batch_dev(c, k, xs, bs, ms, rrs, mx, par, its, nullptr));
      for (int i = 0; i < k; i++) S[idx[i]].n = its[i];
    }
    std::vector<int> next;
    for (int j : active) {
      Sys &s = S[j];
      if (s.par >= 0) {
        if (s.kind == 0) {
          CHK(blas_scale(c, 4.0, *s.y, s.par ? 0 : 1));
          CHK(blas_copy(c, *s.d, *s.y, 2));
          CHK(op_D(c, *s.y, *s.d, s.m, -1.0));
        } else {
          CHK(blas_scale(c, 4.0, *s.y, 0));
          CHK(op_eo_reconstruct_pub(c, *s.y, *s.r, s.m));
        }
      } else {
        CHK(blas_zero(c, *s.y, 2));
      }
      s.its += s.n;
      CHK(blas_axpy(c, 1.0, *s.y, *s.x, 2));
      CHK(op_D(c, *s.r, *s.x, s.m, 1.0));
      CHK(blas_axpby(c, 1.0, *s.b, -1.0, *s.r, *s.r, 2));
      CHK(norm2_eo(c, *s.r, &s.r2e, &s.r2o));
      if (s.r2e + s.r2o > s.r2stop && s.its < maxits && s.par >= 0) next.push_back(j);
    }
    active.swap(next);
  }
  for (int j = 0; j < n; j++) {
    if (iters) iters[j] = S[j].its;
    if (r2_final) r2_final[j] = (S[j].b2 != 0.0) ? (S[j].r2e + S[j].r2o) / S[j].b2 : 0.0;
  }
  return 0;
}

// device fields for the systems' sources and solutions (slots of the batch state)
int batch_io_fields(qexhip_ctx *c, int n, DevField **xs, DevField **bs) {
  if (n < 1 || n > QX_MAXRHS) { qexhip_set_error("batch solve: 1 <= n <= %d", QX_MAXRHS); return -1; }
  BatchState *B = (BatchState *)c->batch;
  if (!B) { B = new BatchState(); c->batch = B; }
  while ((int)B->f.size() < 9 * QX_MAXRHS) {     // 4 CG fields + r, y, d + x, b per system
    DevField nf;
    CHK(field_alloc(c, nf));
    B->f.push_back(nf);
  }
  for (int j = 0; j < n; j++) { xs[j] = &B->f[7 * QX_MAXRHS + 2 * j]; bs[j] = &B->f[7 * QX_MAXRHS + 2 * j + 1]; }
  return 0;
}

// host-field entry points
int solve_batch_host(qexhip_ctx *c, int n, double *const *x, const double *const *b, const double *mass,
                     const double *r2req, int maxits, int xx_parity, int *iters, double *r2) {
  if (n < 1 || n > QX_MAXRHS) { qexhip_set_error("batch solve: 1 <= n <= %d", QX_MAXRHS); return -1; }
  DevField *xs[QX_MAXRHS], *bs[QX_MAXRHS];
  CHK(batch_io_fields(c, n, xs, bs));
  for (int j = 0; j < n; j++) CHK(field_upload(c, *bs[j], b[j]));
  if (xx_parity >= 0) CHK(solve_xx_batch_dev(c, n, xs, bs, mass, r2req, maxits, xx_parity, iters, r2));
  else CHK(solve_full_batch_dev(c, n, xs, bs, mass, r2req, maxits, iters, r2));
  for (int j = 0; j < n; j++) CHK(field_download(c, *xs[j], x[j]));
  return 0;
}
