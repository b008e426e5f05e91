// dslash.hip -- one-parity staggered Dslash sweep (kernels K1/K2/K3 of SURVEY.md 2.3).
//
// Restates stagD2 (src/physics/stagD.nim:349-395), stagDP (:200-237) and stagDM (:278-313):
//   out(s) = ca*rin(s) + cb*xs(s) +/- sum_mu [ U_mu(s) in(s+mu) - U_mu(s-mu)^+ in(s-mu) ]
// (+ the 3-hop terms with 16 links, initStagD3T :38-49).  One lane per output site, one
// wavefront per 64-site tile: the wavefront streams its contiguous block of links (72 KiB /
// 144 KiB) with 16-byte loads, neighbour vectors come from the opposite-parity field with
// unit-stride (x), row-stride (y), plane-stride (z) or slice-stride (t) access, all coalesced.
// HBM-bound: 1248 B and 570 flop per site (1-hop), no MFMA on purpose.
#include "qexhip_internal.h"
#include "site_index.h"
#include "reduce.h"
#include "peer_device.h"
#include <hip/hip_ext.h>
#include <cstring>
#include <algorithm>
#include <chrono>

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct DslashArgs {
  Geom g;
  const double2 *W;      // links of the output parity (RECON: rows 0,1 only, [tile][dir][6][64])
  const unsigned long long *S;   // RECON: sign masks [tile][dir], bit = lane
  const double2 *in;     // hop source (opposite parity half)
  double2 *out;          // output parity half
  const double2 *rin;    // a-term
  const double2 *xs;     // b-term / dot partner
  double ca, cb;
  double sgn;            // +1 (stagDP / stagD2) or -1 (stagDM)
  double post;           // final scale, the `r := (0.5*sc)*r` of stagD (stagD.nim:409)
  int parity, c0, c1;    // first site range [c0,c1)
  int d0, d1, nb1;       // optional second range [d0,d1) handled by workgroups >= nb1 (both t-faces in one launch)
  int e0, e1, nb2;       // PART 3: third range [e0,e1) handled by workgroups >= nb2 (interior | low face | high face)
  int nbA;               // PART 3: position of the boundary workgroups in the dispatch order (interior workgroups before and behind them)
  double *partials;
  const int *done;
  int swz;               // number of workgroups if XCD swizzle is on, else 0
  int ntstore;           // 1: non-temporal stores of the output
  const double2 *gh_hi, *gh_lo;   // GX kernels: where ghost POSITIONS are read from instead of `in` (pre-offset: gh[vec_off(pos, k)])
  // PART == 2 on the peer transport: the launch waits in its prologue for the comm stream's arrival signal and, with a zero-copy
  // receive, returns the two senders' credits from its last workgroup (peer_device.h: PeerGhost)
  PeerGhost pg;
  PeerPush push;         // PART 3 on the peer transport with a zero-copy receive: the launch's FIRST push.nblocks workgroups send the faces themselves
};

#include "dslash_core.h"

// GX (t-sharded boundary launches of the peer transport): hops that leave the slab read the neighbours' faces where the exchange
// kernel's neighbours WROTE them -- the transport's receive arena -- instead of the field's ghost tiles.  A tile lies in one
// t-slice, so which base a t-hop reads from is wavefront-uniform: four scalar selects per wavefront, nothing per lane.
// PART (t-sharded, overlapped sweeps split BY HOPS instead of by sites; shifts.nim's own order: local terms while the faces
// travel, boundary terms when they are in):
//   1  every site of the slab, every hop that stays inside it; sites with a hop that leaves it (the `depth` outermost slices
//      either side) keep their RAW accumulator in `out` -- no final scale, no dot product;
//   2  those sites only, the hops that leave the slab only (1 or 2 of 8 / 16 per site), on top of the raw accumulator; then the
//      final scale, the store and the dot partial.  ~1/4 of a site's bytes, on 1/6 of the sites of a 48^3 x 12 slab: the only
//      work left behind the exchange.  Its prologue waits for the comm stream's arrival signal;
//   3  both in ONE launch: workgroups < nb1 take the interior tiles (the one-launch loop), the workgroups behind them -- dispatched
//      last -- the boundary tiles: hops that stay inside the slab, then a bounded wait for the arrival signal (long raised by
//      then unless the exchange is the longer of the two), then the hops that leave it, accumulator in registers throughout.
// The sum of a boundary site runs local hops first, then the others: results agree with the one-launch kernel to rounding.
template <int NDIR, bool HALO, bool INIT, bool DOT, int RECON, bool GX = false, int PART = 0>
__global__ void __launch_bounds__(256) k_dslash(DslashArgs A) {
  // a finished solve turns the rest of its chunk into no-ops -- except that a launch which owes credits still returns them
  // (the exchanges on the comm stream go on, and their pushes wait for these credits)
  const bool skip = A.done && *A.done;
  if (skip && !(PART >= 2 && A.pg.ticket)) return;
  __shared__ int arrived;
  if (PART == 2 && A.pg.join) {
    if (threadIdx.x == 0) arrived = peer_poll_ge(A.pg.join, A.pg.joinval, A.pg.err, A.pg.ticks, 0x500) ? 1 : 0;
    __syncthreads();
    if (!arrived) return;
    if (!GX && threadIdx.x == 0) {        // (GX: the system-scope acquire below covers it)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (!GX) __syncthreads();
  }
  if (GX && PART != 3) {
    // the arena was written by other processes / devices: system-scope acquire on every CU that reads it
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
  int bid = blockIdx.x;
  if (PART == 3 && GX) {
    // the exchange inside the sweep: the workgroups dispatched first push this rank's faces into the neighbours' arenas -- no second
    // stream, no event, no exchange launch (peer_device.h)
    if (bid < A.push.nblocks) {
      if (!skip) peer_push_block(A.push, (unsigned)bid);
      return;
    }
    bid -= A.push.nblocks;
  }
  const int ngrid = (int)gridDim.x - ((PART == 3 && GX) ? A.push.nblocks : 0);
  // PART 3: the boundary workgroups sit at position nbA of the dispatch order, interior workgroups before AND behind them: late
  // enough for the faces to be in when they start (unless the exchange is the longer of the two), early enough for their slower
  // rolled loops not to be the tail of the launch
  const int nbnd = PART == 3 ? ngrid - A.nb1 : 0;
  const bool bnd = PART == 3 && bid >= A.nbA && bid < A.nbA + nbnd;        // workgroup-uniform
  if (PART == 3) bid = bnd ? A.nb1 + (bid - A.nbA) : (bid < A.nbA ? bid : bid - nbnd);
  if (A.swz && !bnd) {
    // XCD-aware remap: workgroups are dealt round-robin over the 8 XCDs; give every XCD a
    // contiguous run of tiles (= a contiguous t-range) so that y/z/t neighbours share its L2.
    int per = A.swz >> 3;
    bid = (bid & 7) * per + (bid >> 3);
  }
  int c = A.c0 + bid * 256 + threadIdx.x;
  int clim = A.c1;
  if (bnd || (PART != 3 && bid >= A.nb1)) {
    c = A.d0 + (bid - A.nb1) * 256 + threadIdx.x; clim = A.d1;
    if (PART == 3 && bid >= A.nb2) { c = A.e0 + (bid - A.nb2) * 256 + threadIdx.x; clim = A.e1; }
  }
  double dotv = 0;
  const bool active = c < clim && !skip;
  const Geom &g = A.g;
  const SiteXYZT s = site_coord(g, c, A.parity);     // (arithmetic only: harmless beyond clim)
  // a tile lies in one t-slice (64 | F on sharded handles): t is wavefront-uniform
  const int tu = (GX || PART != 0) ? __builtin_amdgcn_readfirstlane(s.t) : 0;
  const bool ghostdep = PART != 0 && (tu < g.depth || tu >= g.X[3] - g.depth);
  double2 acc[3];
  double2 xsv[3];
  constexpr int NLOAD = RECON == 1 ? 6 : (RECON == 2 ? 7 : 9);
  constexpr int LROW = NLOAD * 64;             // double2 per (tile, direction)
  const double2 *w = A.W + (size_t)(c >> 6) * (NDIR * LROW) + (c & 63);
  const unsigned long long *sm = RECON == 1 ? A.S + (size_t)(c >> 6) * NDIR : nullptr;
  const double2 *in_f1 = A.in, *in_b1 = A.in, *in_f3 = A.in, *in_b3 = A.in;
  if (GX) {
    in_f1 = tu + 1 >= g.X[3] ? A.gh_hi : A.in;
    in_b1 = tu - 1 < 0 ? A.gh_lo : A.in;
    in_f3 = tu + 3 >= g.X[3] ? A.gh_hi : A.in;
    in_b3 = tu - 3 < 0 ? A.gh_lo : A.in;
  }
  // One pair = the forward and the backward hop of one direction (fat links: pairs 0..3, 3-hop links: pairs 4..7);
  // do_f / do_b: which of the two this call takes (literally true on the fast path).
  auto pair = [&](const int pr, const bool do_f, const bool do_b) __attribute__((always_inline)) {
    const int mu = pr & 3;
    const int hop = pr >= 4 ? 3 : 1;
    const int pf = nbr_pos<HALO>(g, c, s, mu, hop);
    const int pb = nbr_pos<HALO>(g, c, s, mu, -hop);
    const double2 *wp = w + (size_t)pr * (2 * LROW);
    double2 U[9], W[9], vf[3], vb[3];
    // links are read exactly once per sweep: stream them past the caches (non-temporal), which
    // leaves L2 / Infinity Cache to the 8x re-read neighbour vectors.  Measured on MI355X,
    // 32^4: 120 us -> 108 us per sweep (scratch/tune_dslash.py, profiles/r01_tune_dslash.log).
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
        W[k] = make_double2(t.x, t.y);
      }
    }
    if (RECON == 1) {
      const int lane = c & 63;
      if (do_f) recon_row2<1>(U, (sm[2 * pr] >> lane) & 1ull);
      if (do_b) recon_row2<1>(W, (sm[2 * pr + 1] >> lane) & 1ull);
    } else if (RECON == 2) {
      if (do_f) recon_row2<2>(U, false);
      if (do_b) recon_row2<2>(W, false);
    }
    const double2 *srcf = (GX && mu == 3) ? (hop == 3 ? in_f3 : in_f1) : A.in;
    const double2 *srcb = (GX && mu == 3) ? (hop == 3 ? in_b3 : in_b1) : A.in;
    if (do_f) {
#pragma unroll
      for (int k = 0; k < 3; k++) vf[k] = srcf[vec_off(pf, k)];
    }
    if (do_b) {
#pragma unroll
      for (int k = 0; k < 3; k++) vb[k] = srcb[vec_off(pb, k)];
    }
    // forward hops add, backward hops subtract (compile-time sign: no per-direction multiply).
    // stagDM's overall minus sign is carried by the initial value and the final scale: negation is
    // exact, so init - sum == -((-init) + sum) bit for bit.
    if (do_f) mv3<false>(acc, U, vf);
    if (do_b) mv3<true>(acc, W, vb);
  };
  // the outermost slices of a hop-split sweep (wavefront-uniform branch): `crossing` false takes every hop but the t-hops that
  // leave the slab -- the spatial pairs in the fast loop's form, then the t pairs hop by hop --, true exactly those t-hops.
  auto edge_pairs = [&](const bool crossing) __attribute__((always_inline)) {
    if (!crossing) {
      constexpr int NSP = NDIR / 2 - NDIR / 8;       // spatial pairs: 0,1,2 (and 4,5,6)
      constexpr int UNS = NDIR == 8 ? 1 : 2;      // (8 links: rolled -- unrolled x3 the function takes 255 VGPRs, one workgroup per SIMD pair, and waiting boundary workgroups then hold most of the chip's slots)
#pragma unroll UNS
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
    if ((INIT && PART != 2) || DOT) {
#pragma unroll
      for (int k = 0; k < 3; k++) xsv[k] = A.xs[vec_off(c, k)];
    }
    if (PART == 2) {
#pragma unroll
      for (int k = 0; k < 3; k++) acc[k] = A.out[vec_off(c, k)];          // the raw accumulator PART 1 left here
    } else if (INIT) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        acc[k].x = (A.sgn * A.cb) * xsv[k].x;
        acc[k].y = (A.sgn * A.cb) * xsv[k].y;
      }
      if (A.ca != 0.0) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
          double2 r = A.rin[vec_off(c, k)];
          acc[k].x += (A.sgn * A.ca) * r.x;
          acc[k].y += (A.sgn * A.ca) * r.y;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 3; k++) acc[k] = make_double2(0.0, 0.0);
    }
    // How far the loop is unrolled decides how many link loads a wave keeps in flight.  Measured inside CG on 32^4
    // (scratch A/B builds, 2 rounds):
    //   1-hop: rolled 117 us, x2 120 us, fully unrolled 113.6 us (all 96 loads in flight, 256 VGPRs)
    //   Naik : rolled 215 us, x2 210 us, x4 222 us; FULLY unrolled hipcc hoists all 192 loads and
    //          spills to scratch (290-330 us) -- never unroll the 16-link loop completely.
    // mu/hop are wave-uniform, so the neighbour arithmetic of the rolled loop branches on scalars.
    // compressed 8-link kernel (rows 0,1 + sign, 864 B/site): rolled 80.2 us, x2 81.9, x4 83.5 (32^4, in CG)
    constexpr int UNR = (NDIR == 8) ? (RECON ? 1 : 4) : 2;
    if (PART == 0 || (PART != 2 && !ghostdep)) {
      // every hop of the site: the loop of the one-launch kernel
#pragma unroll UNR
      for (int pr = 0; pr < NDIR / 2; pr++) pair(pr, true, true);
    } else {
      edge_pairs(PART == 2);
    }
  }
  const int pidx = (PART == 3 && GX) ? (int)blockIdx.x - A.push.nblocks : (int)blockIdx.x;      // (partials: the pushing workgroups have none)
  if (PART == 3 && bnd && !skip) {
    // the faces: a bounded wait (one lane), then the acquire for what other devices / the comm stream wrote
    if (threadIdx.x == 0) {
      arrived = peer_ghost_wait(A.pg) ? 1 : 0;
      if (GX) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
      else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (active && arrived) edge_pairs(true);
  }
  if (active) {
    if (PART == 1 && ghostdep) {
      // a hop of this site leaves the slab: the raw accumulator waits in `out` for the PART 2 launch (which re-reads it soon: plain stores)
#pragma unroll
      for (int k = 0; k < 3; k++) A.out[vec_off(c, k)] = acc[k];
    } else {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        acc[k].x *= (A.sgn * A.post); acc[k].y *= (A.sgn * A.post);
        if (A.ntstore) {
          d2v t; t.x = acc[k].x; t.y = acc[k].y;
          __builtin_nontemporal_store(t, (d2v *)&A.out[vec_off(c, k)]);
        } else {
          A.out[vec_off(c, k)] = acc[k];
        }
      }
      if (DOT) {
#pragma unroll
        for (int k = 0; k < 3; k++) dotv = fma(xsv[k].x, acc[k].x, fma(xsv[k].y, acc[k].y, dotv));
      }
    }
  }
  if (DOT && !skip) {
    double r = block_sum_256(dotv);
    if (threadIdx.x == 0) A.partials[pidx] = r;
  }
  if ((PART == 2 || (PART == 3 && bnd)) && A.pg.ticket) {
    // zero-copy receive: the workgroup whose ticket comes last returns the arena halves to the two senders (every wave's loads of
    // the arena have returned before its workgroup takes the ticket)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned a = __hip_atomic_fetch_add(A.pg.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a == (unsigned)(PART == 3 ? nbnd : (int)gridDim.x) - 1) {
        __hip_atomic_store(A.pg.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(A.pg.credit[0], A.pg.credit_val[0], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(A.pg.credit[1], A.pg.credit_val[1], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// Launch with the HIP extension that attaches a start and a stop event to the kernel itself
// (hipExtLaunchKernelGGL): the pair brackets exactly the kernel's execution, like the duration
// rocprofv3 reports, instead of the record-to-record interval of two stream markers.
template <class K>
static void launch_timed(qexhip_ctx *c, const char *tname, K kernel, dim3 grid, dim3 block, DslashArgs &A, hipStream_t st) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timer_event_pair(c, tname, &e0, &e1)) hipExtLaunchKernelGGL(kernel, grid, block, 0, st, e0, e1, 0, A);
  else hipLaunchKernelGGL(kernel, grid, block, 0, st, A);
}

template <int NDIR, bool HALO>
static int launch(qexhip_ctx *c, DslashArgs &A, int c0, int c1, bool init, bool dot, int part_off,
                  int d0 = 0, int d1 = 0, const char *tname = "dslash", hipStream_t st = nullptr, bool gx = false, int part = 0,
                  int e0 = 0, int e1 = 0) {
  if (!st) st = c->stream;
  if (c1 <= c0 && d1 <= d0) return 0;
  if (c1 <= c0) { c0 = d0; c1 = d1; d0 = d1 = 0; }
  A.c0 = c0; A.c1 = c1; A.d0 = d0; A.d1 = d1; A.e0 = e0; A.e1 = e1;
  A.nb1 = (c1 - c0 + 255) / 256;
  A.nb2 = A.nb1 + (d1 > d0 ? (d1 - d0 + 255) / 256 : 0);
  int nb = A.nb2 + (e1 > e0 ? (e1 - e0 + 255) / 256 : 0);
  // XCD swizzle: measured on for compressed links, off for 18-real links (profiles/r01_tune_dslash.log); output stores are
  // non-temporal (the result is read by the NEXT kernel, after 0.6 GB of links went through the caches)
  const int nsw = part == 3 ? A.nb1 : nb;          // (the fused launch remaps its interior workgroups only)
  A.nbA = 0;
  if (part == 3) {
    // Where the boundary workgroups go in the dispatch order.  Once started they hold their slots until the faces are in, so they
    // should start about when the faces arrive: estimated transfer time (3 us + face bytes at the link rate: 45 GB/s per xGMI
    // direction unless option emu_link_gbs says otherwise) over estimated interior time (its bytes at 5.5 TB/s); never before 65 %
    // (their edge loops should not be the tail either), last of all when the exchange is the longer of the two.  How much a
    // wrong guess costs depends on the slots left: with the edge loop unrolled the 18-real 8-link function took 255 VGPRs (512
    // slots; 432 boundary workgroups on a 48^3 slab) and a fixed 65 % cost 446 instead of 378 us per iteration under 126 us of
    // transport (profiles/r05_emulated_scaling_v6.log against v5); rolled it takes 141 (1536 slots).
    const double link = (c->emu_link_gbs > 0 ? c->emu_link_gbs : 45.0) * 1e9;
    const double t_x = 3e-6 + (double)c->g.depth * c->g.F * 48.0 / link;
    const double bsite = NDIR * (c->recon == 1 ? 96.0 : (c->recon == 2 ? 112.0 : 144.0)) + 120.0;
    const double t_int = (double)(c1 - c0) * bsite / 5.5e12;
    const double f = t_int > 0 ? std::min(1.0, std::max(0.65, t_x / t_int)) : 1.0;
    A.nbA = (int)(f * A.nb1);
  }
  A.swz = (c->recon != 0 && nsw >= 64 && (nsw & 7) == 0) ? nsw : 0;
  A.ntstore = 1;
  double *psave = A.partials;
  A.partials = psave ? psave + part_off : nullptr;
  if (!(part == 3 && gx)) A.push.nblocks = 0;
  dim3 grid(nb + A.push.nblocks), block(256);
#define QX_LAUNCH(R) \
  do { \
    if (HALO && part == 3) { \
      if (gx) { \
        if (init && dot) launch_timed(c, tname, k_dslash<NDIR, HALO, true, true, R, HALO, HALO ? 3 : 0>, grid, block, A, st); \
        else if (init) launch_timed(c, tname, k_dslash<NDIR, HALO, true, false, R, HALO, HALO ? 3 : 0>, grid, block, A, st); \
        else if (dot) launch_timed(c, tname, k_dslash<NDIR, HALO, false, true, R, HALO, HALO ? 3 : 0>, grid, block, A, st); \
        else launch_timed(c, tname, k_dslash<NDIR, HALO, false, false, R, HALO, HALO ? 3 : 0>, grid, block, A, st); \
      } else { \
        if (init && dot) launch_timed(c, tname, k_dslash<NDIR, HALO, true, true, R, false, HALO ? 3 : 0>, grid, block, A, st); \
        else if (init) launch_timed(c, tname, k_dslash<NDIR, HALO, true, false, R, false, HALO ? 3 : 0>, grid, block, A, st); \
        else if (dot) launch_timed(c, tname, k_dslash<NDIR, HALO, false, true, R, false, HALO ? 3 : 0>, grid, block, A, st); \
        else launch_timed(c, tname, k_dslash<NDIR, HALO, false, false, R, false, HALO ? 3 : 0>, grid, block, A, st); \
      } \
    } else if (HALO && part == 2) { \
      if (gx && dot) launch_timed(c, tname, k_dslash<NDIR, HALO, false, true, R, HALO, HALO ? 2 : 0>, grid, block, A, st); \
      else if (gx) launch_timed(c, tname, k_dslash<NDIR, HALO, false, false, R, HALO, HALO ? 2 : 0>, grid, block, A, st); \
      else if (dot) launch_timed(c, tname, k_dslash<NDIR, HALO, false, true, R, false, HALO ? 2 : 0>, grid, block, A, st); \
      else launch_timed(c, tname, k_dslash<NDIR, HALO, false, false, R, false, HALO ? 2 : 0>, grid, block, A, st); \
    } else if (HALO && part == 1) { \
      if (init && dot) launch_timed(c, tname, k_dslash<NDIR, HALO, true, true, R, false, HALO ? 1 : 0>, grid, block, A, st); \
      else if (init) launch_timed(c, tname, k_dslash<NDIR, HALO, true, false, R, false, HALO ? 1 : 0>, grid, block, A, st); \
      else if (dot) launch_timed(c, tname, k_dslash<NDIR, HALO, false, true, R, false, HALO ? 1 : 0>, grid, block, A, st); \
      else launch_timed(c, tname, k_dslash<NDIR, HALO, false, false, R, false, HALO ? 1 : 0>, grid, block, A, st); \
    } else if (HALO && gx) { \
      if (init && dot) launch_timed(c, tname, k_dslash<NDIR, HALO, true, true, R, HALO>, grid, block, A, st); \
      else if (init) launch_timed(c, tname, k_dslash<NDIR, HALO, true, false, R, HALO>, grid, block, A, st); \
      else if (dot) launch_timed(c, tname, k_dslash<NDIR, HALO, false, true, R, HALO>, grid, block, A, st); \
      else launch_timed(c, tname, k_dslash<NDIR, HALO, false, false, R, HALO>, grid, block, A, st); \
    } else if (init && dot) launch_timed(c, tname, k_dslash<NDIR, HALO, true, true, R>, grid, block, A, st); \
    else if (init) launch_timed(c, tname, k_dslash<NDIR, HALO, true, false, R>, grid, block, A, st); \
    else if (dot) launch_timed(c, tname, k_dslash<NDIR, HALO, false, true, R>, grid, block, A, st); \
    else launch_timed(c, tname, k_dslash<NDIR, HALO, false, false, R>, grid, block, A, st); \
  } while (0)
  if (c->recon == 1) QX_LAUNCH(1);
  else if (c->recon == 2) QX_LAUNCH(2);
  else QX_LAUNCH(0);
#undef QX_LAUNCH
  A.partials = psave;
  HIPCHK(hipGetLastError());
  return 0;
}

// The fused hop-split launch waits on the device for "the faces are in".  The peer transport has its join counters; the RCCL arm (and
// the communicator-less rehearsal) gets the same from the context: a counter a one-lane kernel raises on the comm stream behind the
// exchange, an error word in pinned memory, the timeout of every other device-side wait (QEXHIP_PEER_TIMEOUT).
__global__ void k_sweep_signal(unsigned long long *ctr, unsigned long long val) {
  if (threadIdx.x == 0) __hip_atomic_store(ctr, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
static int sweep_join_args(qexhip_ctx *c, PeerGhost *G) {
  if (!c->sj_ctr) {
    HIPCHK(hipMalloc((void **)&c->sj_ctr, 128));
    HIPCHK(hipMemset(c->sj_ctr, 0, 128));
    HIPCHK(hipHostMalloc((void **)&c->sj_err, 64, hipHostMallocDefault));
    *c->sj_err = 0;
    double tmo = 30.0;
    if (const char *e = getenv("QEXHIP_PEER_TIMEOUT")) { const double v = atof(e); if (v > 0) tmo = v; }
    int khz = 0;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device);
    if (khz <= 0) khz = 100000;
    c->sj_ticks = (long long)(tmo * 1000.0 * khz);
  }
  k_sweep_signal<<<1, 64, 0, c->cstream>>>(c->sj_ctr, ++c->sj_seq);
  HIPCHK(hipGetLastError());
  memset(G, 0, sizeof *G);
  G->join = c->sj_ctr; G->joinval = c->sj_seq;
  G->err = c->sj_err; G->ticks = c->sj_ticks;
  return 0;
}

// How a sweep over a t-sharded field is laid out: boundary sites [0, lo_end) and [hi_beg, Vh), interior between them, and
// whether the face exchange goes to the second stream beside the interior launch.
// Overlap only when the interior is long enough to hide the exchange (measured on one MI355X with a one-rank communicator:
// the two cross-stream dependencies cost ~20 us per sweep; an interior of 128k sites runs ~30 us) and a face is big enough
// for its transfer to cost more than the split does: in the one-rank rehearsal (no transport time at all) the interior /
// boundary split with its two cross-stream events costs ~15 us per sweep (48^3 x 12: 417 vs 388 us per iteration; 32^3 x 16:
// 253 vs 218); a 48^3 face is 2.65 MB per direction (tens of microseconds on an xGMI link), a 32^3 face 0.79 MB.
void sweep_plan(const qexhip_ctx *c, int *lo_end_out, int *hi_beg_out, int *overlap_out) {
  const Geom &g = c->g;
  int lo_end = g.depth * g.F; if (lo_end > g.Vh) lo_end = g.Vh;
  int hi_beg = g.Vh - g.depth * g.F; if (hi_beg < lo_end) hi_beg = lo_end;
  const size_t face_bytes = (size_t)g.depth * g.F * 48;
  *lo_end_out = lo_end; *hi_beg_out = hi_beg;
  const int tuned = c->overlap_auto[c->ndir == 16];
  *overlap_out = !g.halo || hi_beg <= lo_end ? 0
                 : (c->opt_overlap >= 0 ? (c->opt_overlap != 0)
                    : (tuned >= 0 ? tuned : ((hi_beg - lo_end) >= 131072 && face_bytes >= ((size_t)1 << 20))));
}

// The static rule above was set from one-rank rehearsals, where an exchange costs one RCCL kernel and no transport.  With a real
// communicator (nranks > 1) the decision is MEASURED once per operator shape, right after the links are in place: a few
// sweeps in either mode on scratch fields, the slower rank's time decides (max-all-reduce, so every rank takes the same
// branch).  Collective over the communicator, like set_links itself (ghost links).  Option "overlap" = 0 / 1 switches the
// measurement off, -2 asks for it on one rank too (test hook).
// whether a chained pair of sweeps (dslash_sweep) is possible at all on this context: zero-copy receive on the peer transport and a
// slab deep enough for a narrowed interior
static bool chain_possible(const qexhip_ctx *c) {
  const Geom &g = c->g;
  return g.halo && c->peer && c->opt_peer_zc && c->opt_hop_split == 0 && g.Vh - 4 * g.depth * g.F > 0;
}
// ... and whether an overlapped pair runs chained: option "sweep_chain" 1 / 0, or at -1 what sweep_autotune measured (off until then)
static bool chain_on(const qexhip_ctx *c) {
  if (!chain_possible(c)) return false;
  return c->opt_sweep_chain >= 0 ? c->opt_sweep_chain != 0 : c->chain_auto[c->ndir == 16] == 1;
}
bool sweep_chain_on(const qexhip_ctx *c) { return chain_on(c); }

int sweep_autotune(qexhip_ctx *c) {
  const Geom &g = c->g;
  const int slot = c->ndir == 16;
  if (!g.halo || !c->W) return 0;
  const bool multi = c->nranks > 1 && comm_ready(c);
  if (multi) {
    // The overlap decision selects the stream (and, on RCCL, the communicator) an exchange is posted on: ranks that disagreed
    // would never match.  set_links is collective, so this is the place to find out (QEXHIP_OVERLAP / option "overlap").
    // The forms of the overlapped sweep (peer_zc, sweep_chain) ride along: they regroup the dot partials, not the messages.
    const double code = 64.0 * c->opt_overlap + 8.0 * (c->opt_sweep_chain + 1) + c->opt_peer_zc + 1024.0 * (c->opt_hop_split + 1);
    double v[2] = {code, -code};
    CHK(comm_allreduce_max(c, v, 2));
    if (v[0] != -v[1]) {
      qexhip_set_error("options overlap (QEXHIP_OVERLAP) / sweep_chain / peer_zc / hop_split differ between the ranks (1024 (hop_split + 1) + 64 overlap + 8 (sweep_chain + 1) + peer_zc = %g .. %g): "
                       "they must be the same everywhere", -v[1], v[0]);
      return -3;
    }
  }
  if (c->overlap_auto[slot] >= 0) return 0;
  if (!(c->opt_overlap == -2 || (c->opt_overlap == -1 && multi))) return 0;
  int lo_end, hi_beg, dummy;
  sweep_plan(c, &lo_end, &hi_beg, &dummy);
  if (hi_beg <= lo_end) { c->overlap_auto[slot] = 0; c->chain_auto[slot] = 0; return 0; }        // no interior to overlap with
  // three forms of a PAIR of sweeps a -> b -> c (what the normal operator runs): exchange first, overlapped, overlapped + chained
  const bool try_chain = chain_possible(c) && c->opt_sweep_chain < 0;
  DevField f[3];
  int rc = 0;
  for (int k = 0; k < 3 && !rc; k++) rc = field_alloc(c, f[k]);
  const int saved = c->opt_overlap, saved_chain = c->opt_sweep_chain, saved_timers = c->timers_on;
  c->timers_on = 0;
  double t[3] = {0, 0, 0};
  for (int mode = 0; mode < (try_chain ? 3 : 2) && !rc; mode++) {
    c->opt_overlap = mode ? 1 : 0;
    c->opt_sweep_chain = mode == 2 ? 1 : 0;
    for (int k = 0; k < 6 && !rc; k++) {
      if (k == 1) { rc = hipStreamSynchronize(c->stream) != hipSuccess ? -2 : 0; t[mode] = -now_us(); }
      DslashOpts o1, o2;
      o1.chain = 1; o2.chain = 2;
      DevField &in0 = f[(k & 1) ? 2 : 0], &out2 = f[(k & 1) ? 0 : 2];
      if (!rc) rc = dslash_sweep(c, f[1], in0, 1, o1);
      if (!rc) rc = dslash_sweep(c, out2, f[1], 0, o2);
    }
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = -2;
    t[mode] += now_us();
  }
  c->opt_overlap = saved;
  c->opt_sweep_chain = saved_chain;
  c->timers_on = saved_timers;
  for (int k = 0; k < 3; k++) if (f[k].d) (void)hipFree(f[k].d);
  // every rank reaches the collective, a failed one with a sentinel, so that all fail together instead of one hanging the others
  double v[4] = {t[0], t[1], t[2], rc ? 1.0 : 0.0};
  if (multi) {
    const int e = comm_allreduce_max(c, v, 4);
    if (e && !rc) rc = e;
  }
  if (rc) return rc;
  if (v[3] != 0.0) { qexhip_set_error("sweep_autotune: another rank failed while measuring"); return -4; }
  // Overlap only on a clear win (> 5 %), chain on top of it only on its own clear win (> 3 % of an overlapped sweep: it narrows the
  // interior that hides the exchange, a loss when the exchange is the longer of the two): the forms group the deferred dot partials
  // differently, so a decision that flips between runs on timing noise would cost run-to-run bit-reproducibility of the residual
  // history for nothing.  Options overlap / sweep_chain = 0 / 1 pin the form (and the bits) outright.
  const int chain = try_chain && v[2] < 0.97 * v[1];
  const double best = chain ? v[2] : v[1];
  c->overlap_auto[slot] = best < 0.95 * v[0] ? 1 : 0;
  c->chain_auto[slot] = chain && c->overlap_auto[slot];
  c->overlap_tune_us[slot][0] = v[0] / 10.0; c->overlap_tune_us[slot][1] = v[1] / 10.0; c->overlap_tune_us[slot][2] = try_chain ? v[2] / 10.0 : 0.0;
  return 0;
}

int dslash_sweep(qexhip_ctx *c, DevField &out, DevField &in, int parity, const DslashOpts &o) {
  const Geom &g = c->g;
  if (!c->W) { qexhip_set_error("staggered links not set (qexhip_stag_set_links)"); return -3; }
  DslashArgs A;
  A.g = g;
  if (c->recon) {
    A.W = c->Wc + (size_t)parity * g.ntile * c->ndir * (c->recon == 1 ? 384 : 448);
    A.S = c->Ws + (size_t)parity * g.ntile * c->ndir;
  } else {
    A.W = c->W + (size_t)parity * g.ntile * c->ndir * 576;
    A.S = nullptr;
  }
  A.in = in.par(1 - parity);
  A.out = out.par(parity);
  A.rin = o.rin ? o.rin->par(parity) : nullptr;
  A.xs = o.xs ? o.xs->par(parity) : nullptr;
  A.ca = o.ca; A.cb = o.cb;
  A.sgn = o.neg ? -1.0 : 1.0;
  A.post = o.post;
  A.parity = parity;
  A.partials = c->partials;
  A.done = o.done;
  A.gh_hi = A.gh_lo = nullptr;
  memset(&A.pg, 0, sizeof A.pg);
  memset(&A.push, 0, sizeof A.push);
  const bool init = (o.ca != 0.0 || o.cb != 0.0);
  if ((init || o.dot) && !A.xs) { qexhip_set_error("dslash_sweep: b-term/dot needs xs"); return -1; }
  if (o.ca != 0.0 && !A.rin) { qexhip_set_error("dslash_sweep: a-term needs rin"); return -1; }
  int nparts = 0;
  if (!g.halo) {
    if (c->ndir == 8) CHK((launch<8, false>(c, A, 0, g.Vh, init, o.dot, 0)));
    else CHK((launch<16, false>(c, A, 0, g.Vh, init, o.dot, 0)));
    nparts = (g.Vh + 255) / 256;
  } else {
    // halo: exchange faces of `in` on the comm stream, interior sweep meanwhile, then boundary
    int lo_end, hi_beg, overlap;
    sweep_plan(c, &lo_end, &hi_beg, &overlap);
    const bool zc = overlap && c->peer && c->opt_peer_zc;
    // Chained pair (o.chain 1 then 2: out2 = D (D in), three distinct fields, nothing else on the compute stream between the two):
    // no join between the sweeps.  Sweep 1's boundary launch stays unjoined on the comm stream; sweep 2's interior launch is
    // NARROWED by the stencil depth on either side -- it then reads nothing sweep 1's boundary launch wrote -- and its boundary
    // launch widened by as much, on the comm stream behind sweep 1's (in order) and behind ev_ready (sweep 1's interior).  The faces
    // sweep 2 sends are sweep 1's boundary output, the comm stream's own work: its exchange starts without waiting for anything.
    // One join per operator instead of two, and the second exchange is posted ~30 us earlier.
    // -1: the fused form where it is proven AND safe: peer transport with zero-copy receive, every rank on a GPU of its own.  Ranks
    // that SHARE a GPU keep the split by sites: with two processes' 10 000-workgroup sweeps on one chip, each holding 432 waiting
    // boundary workgroups, a 48^3 x 96 solve over 2 ranks ran into the 30 s wait bound (bench.py --gpus 2 on one device;
    // 8^4 ... 16^3 x 32 with 2 and 4 ranks are green and stay in the test suite with hop_split = 2 forced)
    const int hop_split = !overlap ? 0 : (c->opt_hop_split >= 0 ? c->opt_hop_split : ((zc && !c->ranks_share_device) ? 2 : 0));
    if (hop_split) {
      // Overlapped sweep split BY HOPS (option hop_split): the hops that stay inside the slab are taken while the faces travel, the
      // 1-2 hops per boundary site that leave it once they are in -- ~1/4 of the bytes of 2 * depth slices is all that is left behind
      // the exchange.  2 (default): ONE launch on the compute stream, interior workgroups first, the boundary workgroups behind
      // them wait on the device for the signal the comm stream raises behind the exchange (long raised by then unless the exchange
      // is the longer of the two); no second launch, no join, nothing on the comm stream but the exchange.  1: two launches (whole
      // slab; boundary sites on top of their raw accumulators), for the A/B.  Zero-copy receive on the peer transport either way:
      // the last boundary workgroup returns the credits.
      if (c->chain_pending) { c->chain_pending = 0; CHK(peer_stream_join(c, c->stream, c->cstream)); }
      CHK(peer_flush_join(c));
      // fused + zero-copy (peer transport): the exchange is INSIDE the launch -- its first workgroups push the faces, its boundary
      // workgroups poll the inbound data words -- and the comm stream is not involved at all
      const bool direct = zc && hop_split == 2;
      if (!direct) HIPCHK(hipEventRecord(c->ev_ready, c->stream));
      if (direct) CHK(comm_halo_exchange_zc(c, in, 1 - parity, &A.gh_hi, &A.gh_lo, false, false, &A.push));
      else if (zc) CHK(comm_halo_exchange_zc(c, in, 1 - parity, &A.gh_hi, &A.gh_lo));
      else CHK(comm_halo_exchange(c, in, 1 - parity, 1));
      const int nb_lo = (lo_end + 255) / 256, nb_hi = (g.Vh - hi_beg + 255) / 256;
      if (c->peer) {
        if (!direct) CHK(peer_stream_signal(c, c->cstream));
        CHK(peer_ghost_args(c, &A.pg, zc, direct));
      } else if (hop_split == 2) CHK(sweep_join_args(c, &A.pg));
      else HIPCHK(hipEventRecord(c->ev_halo, c->cstream));
      if (hop_split == 2) {
        if (c->ndir == 8) CHK((launch<8, true>(c, A, lo_end, hi_beg, init, o.dot, 0, 0, lo_end, "dslash", nullptr, zc, 3, hi_beg, g.Vh)));
        else CHK((launch<16, true>(c, A, lo_end, hi_beg, init, o.dot, 0, 0, lo_end, "dslash", nullptr, zc, 3, hi_beg, g.Vh)));
        nparts = (hi_beg - lo_end + 255) / 256 + nb_lo + nb_hi;
      } else {
        const int nb_main = (g.Vh + 255) / 256;
        PeerGhost pg = A.pg;
        memset(&A.pg, 0, sizeof A.pg);
        if (c->ndir == 8) CHK((launch<8, true>(c, A, 0, g.Vh, init, o.dot, 0, 0, 0, "dslash", nullptr, false, 1)));
        else CHK((launch<16, true>(c, A, 0, g.Vh, init, o.dot, 0, 0, 0, "dslash", nullptr, false, 1)));
        A.pg = pg;
        if (!c->peer) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_halo, 0));
        if (c->ndir == 8) CHK((launch<8, true>(c, A, 0, lo_end, false, o.dot, nb_main, hi_beg, g.Vh, "dslash_bnd", nullptr, zc, 2)));
        else CHK((launch<16, true>(c, A, 0, lo_end, false, o.dot, nb_main, hi_beg, g.Vh, "dslash_bnd", nullptr, zc, 2)));
        nparts = nb_main + nb_lo + nb_hi;
      }
      memset(&A.pg, 0, sizeof A.pg);
      memset(&A.push, 0, sizeof A.push);
    } else {
    const bool chain_ok = zc && chain_on(c);
    if (c->chain_pending && !(o.chain == 2 && chain_ok)) {       // a broken pair: join first, then an ordinary sweep
      c->chain_pending = 0;
      CHK(peer_stream_join(c, c->stream, c->cstream));
    }
    const bool chain2 = o.chain == 2 && chain_ok && c->chain_pending;
    c->chain_pending = 0;
    if (!chain2) CHK(peer_flush_join(c));
    if (chain2) { lo_end *= 2; hi_beg = g.Vh - lo_end; }
    if (overlap) HIPCHK(hipEventRecord(c->ev_ready, c->stream));
    if (zc) {
      CHK(comm_halo_exchange_zc(c, in, 1 - parity, &A.gh_hi, &A.gh_lo, !chain2));
      if (chain2) HIPCHK(hipStreamWaitEvent(c->cstream, c->ev_ready, 0));
    } else CHK(comm_halo_exchange(c, in, 1 - parity, overlap));
    if (!overlap) {
      // the exchange is already ordered before us on the compute stream: one launch over all sites
      // (small local volumes are launch-latency-bound; this drops two launches per CG iteration)
      if (c->ndir == 8) CHK((launch<8, true>(c, A, 0, g.Vh, init, o.dot, 0)));
      else CHK((launch<16, true>(c, A, 0, g.Vh, init, o.dot, 0)));
      nparts = (g.Vh + 255) / 256;
    } else {
      int nb_int = (hi_beg - lo_end + 255) / 256, nb_lo = (lo_end + 255) / 256;
      if (c->ndir == 8) CHK((launch<8, true>(c, A, lo_end, hi_beg, init, o.dot, 0)));
      else CHK((launch<16, true>(c, A, lo_end, hi_beg, init, o.dot, 0)));
      if (c->peer) {
        // Peer transport: as below -- both t-faces in one launch on the comm stream right behind the exchange, beside the
        // interior's tail -- but the join back to the compute stream is a device-side counter (a one-lane signal kernel behind
        // the boundary launch, a one-wave wait kernel on the compute stream) instead of an event: the runtime's cross-queue
        // dependency alone was ~20 us of dead time per sweep (profiles/r05_timeline_*.txt).
        // Zero-copy receive (option peer_zc): the boundary launch reads the neighbours' faces from the receive arena itself; the
        // credits go back behind it together with the join signal.  Otherwise the exchange kernel has unpacked into the ghost tiles.
        if (c->ndir == 8) CHK((launch<8, true>(c, A, 0, lo_end, init, o.dot, nb_int, hi_beg, g.Vh, "dslash_bnd", c->cstream, zc)));
        else CHK((launch<16, true>(c, A, 0, lo_end, init, o.dot, nb_int, hi_beg, g.Vh, "dslash_bnd", c->cstream, zc)));
        if (zc) CHK(peer_release_zc(c, c->cstream));
        else CHK(peer_stream_signal(c, c->cstream));
        if (o.chain == 1 && chain_ok) c->chain_pending = 1;                       // the chain-2 sweep (or whatever comes instead) joins
        else if (o.defer_join) CHK(peer_stream_join_defer(c));                    // rides in the <p,Ap> all-reduce's prologue
        else CHK(peer_stream_join(c, c->stream, c->cstream));
      } else {
      // Both t-faces in ONE launch, posted on the COMM stream right behind the exchange: it needs the ghost zones and nothing
      // of the interior launch, so it starts the moment the faces have arrived and runs beside the interior's tail instead of
      // after it (round 4: 1-6 % of an iteration on thin slabs in the one-rank rehearsal, profiles/r04_notes.md).  Everything later on the
      // compute stream waits for ev_halo, recorded behind it.
      if (c->ndir == 8) CHK((launch<8, true>(c, A, 0, lo_end, init, o.dot, nb_int, hi_beg, g.Vh, "dslash_bnd", c->cstream)));
      else CHK((launch<16, true>(c, A, 0, lo_end, init, o.dot, nb_int, hi_beg, g.Vh, "dslash_bnd", c->cstream)));
      HIPCHK(hipEventRecord(c->ev_halo, c->cstream));
      HIPCHK(hipStreamWaitEvent(c->stream, c->ev_halo, 0));
      }
      nparts = nb_int + nb_lo + (g.Vh - hi_beg + 255) / 256;
    }
    }
  }
  if (o.dot) {
    if (nparts > c->part2_off) { qexhip_set_error("internal: partial buffer too small"); return -3; }
    // deferred final sum (inside k_cg_update) only while every workgroup can afford to re-sum the
    // partials itself; big local volumes take the separate one-block reduction
    if (o.dot == 2 && o.nparts_out && nparts <= 4096) *o.nparts_out = nparts;
    else {
      if (o.nparts_out) *o.nparts_out = 0;
      CHK(peer_flush_join(c));                 // (the reduction reads the boundary launch's partials)
      CHK(reduce_partials(c, nparts, o.dot_out));
    }
  }
  return 0;
}
