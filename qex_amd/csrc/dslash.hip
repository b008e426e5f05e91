// dslash.hip -- one-parity staggered Dslash sweep (kernels K1/K2/K3 of SURVEY.md 2.3).
//
// Restates stagD2 (src/physics/stagD.nim:349-395), stagDP (:200-237) and stagDM (:278-313):
//   out(s) = ca*rin(s) + cb*xs(s) +/- sum_mu [ U_mu(s) in(s+mu) - U_mu(s-mu)^+ in(s-mu) ]
// (+ the 3-hop terms with 16 links, initStagD3T :38-49).  One lane per output site, one
// wavefront per 64-site tile: the wavefront streams its contiguous block of links (72 KiB /
// 144 KiB) with 16-byte loads, neighbour vectors come from the opposite-parity field with
// unit-stride (x), row-stride (y), plane-stride (z) or slice-stride (t) access, all coalesced.
// HBM-bound: 1248 B and 570 flop per site (1-hop), no MFMA on purpose.
//
// Two kernels with disjoint roles (round 6; rounds 4-5 had one kernel with seven template parameters):
//   k_dslash<NDIR, HALO, INIT, DOT, RECON>     every sweep that reads its t-neighbours from the FIELD: the whole lattice on one GPU, and on a
//                                              t-sharded slab the launches around an exchange into the ghost tiles (RCCL, or the peer
//                                              transport's unpacking exchange): one launch behind it, or interior | both faces
//   k_dslash_fused<NDIR, INIT, DOT, RECON>     the overlapped sweep of a t-sharded slab on the peer transport as ONE launch on ONE stream
//                                              (shifts.nim:67-94,254-285: local terms while the faces travel, boundary terms when they
//                                              are in): push | interior | boundary | cleanup workgroups, see below
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
  int d0, d1, nb1;       // second range [d0,d1) handled by workgroups >= nb1 (both t-faces in one launch)
  int e0, e1, nb2;       // fused: third range [e0,e1) handled by workgroups >= nb2 (interior | low face | high face)
  int nbA;               // fused: position of the boundary workgroups in the dispatch order (interior workgroups before and behind them)
  double *partials;
  const int *done;
  int swz;               // number of workgroups if XCD swizzle is on, else 0
  int ntstore;           // 1: non-temporal stores of the output
  // fused only
  const double2 *gh_hi, *gh_lo;   // where ghost POSITIONS are read from: the transport's receive arena (pre-offset: gh[vec_off(pos, k)])
  PeerGhost pg;          // the inbound data words, the credits owed to the two senders
  PeerPush push;         // the launch's FIRST push.nblocks workgroups send the faces
  FusedCtl fz;           // who has decided, who has parked
};

#include "dslash_core.h"

// The fused sweep, by workgroup number:
//   [0, npush)            push this rank's two faces into the neighbours' receive arenas (credits, release, data words: peer_device.h)
//   interior workgroups   every hop of their sites: the loop of the plain kernel
//   boundary workgroups   (the `depth` outermost slices either side; placed at nbA of the dispatch order, interior workgroups before AND behind
//                         them) the hops that stay inside the slab, then a SHORT wait for the inbound data words (about the transfer time):
//                         faces in -> the 1-2 hops per site that leave the slab straight from the arena, accumulator in registers throughout;
//                         faces late -> the raw accumulator is PARKED in `out`, the block appended to the parked list, the slot given up
//   cleanup workgroups    (the last fz.ncl of the grid) once every boundary workgroup has decided: nothing parked -> exit; else the LONG
//                         bounded wait for the faces (the only place a lost neighbour is noticed), then the parked blocks' remaining hops
//                         on top of their raw accumulators, final scale, store and dot partial -- in the parked block's own partial slot
// Whoever reads the arena last returns the credits.  The sum of a boundary site runs local hops first, then the others, parked or not:
// a parked block gives the same bits as an unparked one (tests/test_gpu_parity.py), and the plain kernel's to rounding.
// Why park (round 6): a boundary workgroup that spins until the faces are in holds its slot hostage to ANOTHER kernel's progress.  With
// 16 links a 48^3 face has 6 x 216 = 1296 boundary workgroups, the chip 768 slots for this kernel: on a chip shared by two ranks'
// processes the spinning ones kept the neighbour's push from ever becoming resident (profiles/r06_notes.md section 1).
template <int NDIR, bool HALO, bool INIT, bool DOT, int RECON, bool FUSED>
__device__ __forceinline__ void dslash_body(const DslashArgs &A) {
  const bool skip = A.done && *A.done;
  // a finished solve turns the rest of its chunk into no-ops -- except that a fused launch still returns the credits it owes
  if (skip && !FUSED) return;
  __shared__ int sh_n;                     // fused: faces arrived (boundary) / parked blocks to take (cleanup)
  int bid = blockIdx.x;
  int ngrid = (int)gridDim.x;              // workgroups that own sites
  bool cleanup = false;
  int npark = 0, jpark = 0;
  if (FUSED) {
    if (bid < A.push.nblocks) {
      if (!skip) peer_push_block(A.push, (unsigned)bid);
      return;
    }
    bid -= A.push.nblocks;
    ngrid -= A.push.nblocks + A.fz.ncl;
    cleanup = bid >= ngrid;
    if (cleanup) {
      if (threadIdx.x == 0) {
        int n = 0;
        if (peer_poll_u32(A.fz.dec, (unsigned)(ngrid - A.nb1), A.pg.err, A.pg.ticks, 0x520)) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // the parked accumulators, the list
          n = (int)__hip_atomic_load(A.fz.ndef, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (n > 0) {
            if (!peer_ghost_wait(A.pg)) n = -1;                       // the neighbour is gone: error word set, nothing more to do here
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");             // what other devices wrote into the arena
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
        } else n = -1;
        sh_n = n;
      }
      __syncthreads();
      npark = sh_n;
      jpark = bid - ngrid;
    }
  }
  const int nbnd = FUSED ? ngrid - A.nb1 : 0;
  bool parked = false;                     // this (boundary) workgroup gave its block to the cleanup workgroups
  bool bnd = false;
  for (;;) {                               // one pass; a cleanup workgroup takes every fz.ncl-th parked block
    int lb = bid;                          // logical workgroup: [0, nb1) first range, [nb1, nb2) second, [nb2, ..) third
    if (FUSED) {
      if (cleanup) {
        if (jpark >= npark) break;
        lb = (int)A.fz.list[jpark];
        jpark += A.fz.ncl;
        bnd = true;
      } else {
        bnd = bid >= A.nbA && bid < A.nbA + nbnd;        // workgroup-uniform
        lb = bnd ? A.nb1 + (bid - A.nbA) : (bid < A.nbA ? bid : bid - nbnd);
      }
    }
    if (A.swz && !bnd) {
      // XCD-aware remap: workgroups are dealt round-robin over the 8 XCDs; give every XCD a
      // contiguous run of tiles (= a contiguous t-range) so that y/z/t neighbours share its L2.
      int per = A.swz >> 3;
      lb = (lb & 7) * per + (lb >> 3);
    }
    int c = A.c0 + lb * 256 + threadIdx.x;
    int clim = A.c1;
    if (FUSED ? bnd : lb >= A.nb1) {
      c = A.d0 + (lb - A.nb1) * 256 + threadIdx.x; clim = A.d1;
      if (FUSED && lb >= A.nb2) { c = A.e0 + (lb - A.nb2) * 256 + threadIdx.x; clim = A.e1; }
    }
    double dotv = 0;
    const bool active = c < clim && !skip;
    const Geom &g = A.g;
    const SiteXYZT s = site_coord(g, c, A.parity);     // (arithmetic only: harmless beyond clim)
    // a tile lies in one t-slice (64 | F on sharded handles): t is wavefront-uniform
    const int tu = FUSED ? __builtin_amdgcn_readfirstlane(s.t) : 0;
    double2 acc[3];
    double2 xsv[3];
    constexpr int NLOAD = RECON == 1 ? 6 : (RECON == 2 ? 7 : 9);
    constexpr int LROW = NLOAD * 64;             // double2 per (tile, direction)
    const double2 *w = A.W + (size_t)(c >> 6) * (NDIR * LROW) + (c & 63);
    const unsigned long long *sm = RECON == 1 ? A.S + (size_t)(c >> 6) * NDIR : nullptr;
    // fused: hops that leave the slab read the neighbours' faces where the neighbours WROTE them -- the transport's receive arena --
    // instead of the field's ghost tiles: which base a t-hop reads from is wavefront-uniform, four scalar selects, nothing per lane
    const double2 *in_f1 = A.in, *in_b1 = A.in, *in_f3 = A.in, *in_b3 = A.in;
    if (FUSED) {
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
      const double2 *srcf = (FUSED && mu == 3) ? (hop == 3 ? in_f3 : in_f1) : A.in;
      const double2 *srcb = (FUSED && mu == 3) ? (hop == 3 ? in_b3 : in_b1) : A.in;
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
    // the outermost slices of the fused sweep (wavefront-uniform branch): `crossing` false takes every hop but the t-hops that
    // leave the slab -- the spatial pairs in the fast loop's form, then the t pairs hop by hop --, true exactly those t-hops.
    auto edge_pairs = [&](const bool crossing) __attribute__((always_inline)) {
      if (!crossing) {
        constexpr int NSP = NDIR / 2 - NDIR / 8;       // spatial pairs: 0,1,2 (and 4,5,6)
        constexpr int UNS = NDIR == 8 ? 1 : 2;      // (8 links: rolled -- unrolled x3 the function takes 255 VGPRs, one workgroup per SIMD pair)
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
    const bool edge = FUSED && bnd;          // (every site of a boundary block has a hop that leaves the slab)
    if (active) {
      if ((INIT && !cleanup) || DOT) {
#pragma unroll
        for (int k = 0; k < 3; k++) xsv[k] = A.xs[vec_off(c, k)];
      }
      if (FUSED && cleanup) {
#pragma unroll
        for (int k = 0; k < 3; k++) acc[k] = A.out[vec_off(c, k)];          // the raw accumulator its boundary workgroup parked here
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
      if (!edge) {
        // every hop of the site: the loop of the one-launch kernel
#pragma unroll UNR
        for (int pr = 0; pr < NDIR / 2; pr++) pair(pr, true, true);
      } else if (!cleanup) {
        edge_pairs(false);
      }
    }
    if (edge && !skip) {
      if (!cleanup) {
        // the faces: a SHORT wait (one lane), then the acquire for what other devices wrote
        if (threadIdx.x == 0) {
          const bool in = A.fz.spin_ticks >= 0 && peer_ghost_try(A.pg, A.fz.spin_ticks, A.fz.late);
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
      if (parked) {
        // the raw accumulator waits in `out` for a cleanup workgroup (which re-reads it soon: plain stores)
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
    if (DOT && !skip && !parked) {
      // (fused: the pushing workgroups have no partial; a parked block's partial goes where its boundary workgroup's would have gone)
      const int pidx = !FUSED ? (int)blockIdx.x : (cleanup ? A.nbA + (lb - A.nb1) : bid);
      double r = block_sum_256(dotv);
      if (threadIdx.x == 0) A.partials[pidx] = r;
    }
    if (!(FUSED && cleanup)) break;
  }
  if (!FUSED || !(bnd || cleanup)) return;
  // every wave's loads of the arena have returned / its parked accumulators are on their way before the workgroup counts itself
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x != 0) return;
  bool credits = false;
  if (!cleanup) {
    if (parked) {
      const unsigned idx = __hip_atomic_fetch_add(A.fz.ndef, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&A.fz.list[idx], (unsigned)(A.nb1 + (bid - A.nbA)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");     // accumulators and list entry before the count (the cleanup workgroup may sit on another XCD)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned a = __hip_atomic_fetch_add(A.fz.dec, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a == (unsigned)nbnd - 1) {
      // the last to decide: if nobody parked, every reader of the arena is through -- the halves go back to the two senders
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      credits = __hip_atomic_load(A.fz.ndef, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
    }
  } else {
    const unsigned a = __hip_atomic_fetch_add(A.fz.cl_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a == (unsigned)A.fz.ncl - 1) {
      credits = npark > 0;             // (npark < 0: a wait gave up -- the error word is set, the job is over)
      __hip_atomic_store(A.fz.ndef, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(A.fz.dec, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(A.fz.late, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(A.fz.cl_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (credits) {
    __hip_atomic_store(A.pg.credit[0], A.pg.credit_val[0], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(A.pg.credit[1], A.pg.credit_val[1], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <int NDIR, bool HALO, bool INIT, bool DOT, int RECON>
__global__ void __launch_bounds__(256) k_dslash(DslashArgs A) { dslash_body<NDIR, HALO, INIT, DOT, RECON, false>(A); }
template <int NDIR, bool INIT, bool DOT, int RECON>
__global__ void __launch_bounds__(256) k_dslash_fused(DslashArgs A) { dslash_body<NDIR, true, INIT, DOT, RECON, true>(A); }

// Launch with the HIP extension that attaches a start and a stop event to the kernel itself
// (hipExtLaunchKernelGGL): the pair brackets exactly the kernel's execution, like the duration
// rocprofv3 reports, instead of the record-to-record interval of two stream markers.
template <class K>
static void launch_timed(qexhip_ctx *c, const char *tname, K kernel, dim3 grid, dim3 block, DslashArgs &A, hipStream_t st) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timer_event_pair(c, tname, &e0, &e1)) hipExtLaunchKernelGGL(kernel, grid, block, 0, st, e0, e1, 0, A);
  else hipLaunchKernelGGL(kernel, grid, block, 0, st, A);
}

// ranges [c0,c1) (+ [d0,d1) behind it; fused: + [e0,e1)); fused: the launch pushes, parks and cleans up as described above
template <int NDIR, bool HALO>
static int launch(qexhip_ctx *c, DslashArgs &A, int c0, int c1, bool init, bool dot, int part_off,
                  int d0 = 0, int d1 = 0, const char *tname = "dslash", hipStream_t st = nullptr, bool fused = false, int e0 = 0, int e1 = 0) {
  if (!st) st = c->stream;
  if (c1 <= c0 && d1 <= d0) return 0;
  if (c1 <= c0) { c0 = d0; c1 = d1; d0 = d1 = 0; }
  A.c0 = c0; A.c1 = c1; A.d0 = d0; A.d1 = d1; A.e0 = e0; A.e1 = e1;
  A.nb1 = (c1 - c0 + 255) / 256;
  A.nb2 = A.nb1 + (d1 > d0 ? (d1 - d0 + 255) / 256 : 0);
  const int nb = A.nb2 + (e1 > e0 ? (e1 - e0 + 255) / 256 : 0);
  // XCD swizzle: measured on for compressed links, off for 18-real links (profiles/r01_tune_dslash.log); output stores are
  // non-temporal (the result is read by the NEXT kernel, after 0.6 GB of links went through the caches)
  const int nsw = fused ? A.nb1 : nb;          // (the fused launch remaps its interior workgroups only)
  A.nbA = fused ? (int)(sweep_push_fraction(c, c1 - c0) * A.nb1) : 0;
  A.swz = (c->recon != 0 && nsw >= 64 && (nsw & 7) == 0) ? nsw : 0;
  A.ntstore = 1;
  double *psave = A.partials;
  A.partials = psave ? psave + part_off : nullptr;
  dim3 grid(nb + (fused ? A.push.nblocks + A.fz.ncl : 0)), block(256);
#define QX_LAUNCH(R) \
  do { \
    if (HALO && fused) { \
      if (init && dot) launch_timed(c, tname, k_dslash_fused<NDIR, true, true, R>, grid, block, A, st); \
      else if (init) launch_timed(c, tname, k_dslash_fused<NDIR, true, false, R>, grid, block, A, st); \
      else if (dot) launch_timed(c, tname, k_dslash_fused<NDIR, false, true, R>, grid, block, A, st); \
      else launch_timed(c, tname, k_dslash_fused<NDIR, false, false, R>, grid, block, A, st); \
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

// How a sweep over a t-sharded field is laid out: boundary sites [0, lo_end) and [hi_beg, Vh), interior between them, and
// whether the face exchange overlaps the interior at all.
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

// What an overlapped sweep runs as: 2 = the fused launch (peer transport carrying the faces: the only transport a kernel of ours can push
// through), 0 = split by sites.  Option hop_split pins it; at -1 it is what set_links measured, fused until then.
int sweep_form(const qexhip_ctx *c, int overlap) {
  if (!overlap || !peer_faces(c)) return 0;
  if (c->opt_hop_split >= 0) return c->opt_hop_split ? 2 : 0;
  const int m = c->form_auto[c->ndir == 16];
  return m >= 0 ? m : 2;
}

// One face exchange of this operator in microseconds: measured at set_links where a communicator exists (sweep_autotune), else -- and
// in rehearsals that emulate a link -- the estimate 3 us + face bytes at 45 GB/s per xGMI direction (option emu_link_gbs replaces the rate).
static double exchange_estimate_us(const qexhip_ctx *c) {
  const double meas = c->xchg_us[c->ndir == 16];
  if (meas > 0) return meas;
  const double link = (c->emu_link_gbs > 0 ? c->emu_link_gbs : 45.0) * 1e3;      // bytes per us
  return 3.0 + (double)c->g.depth * c->g.F * 48.0 / link;
}
// Where the fused sweep's boundary workgroups go in the dispatch order.  They should start about when the faces arrive: exchange time
// over estimated interior time (its bytes at 5.5 TB/s); never before 65 % (their rolled edge loops should not be the tail either),
// last of all when the exchange is the longer of the two.  A wrong guess costs little since round 6 (a boundary workgroup waits
// about one exchange time, then parks), but the waiting ones do hold slots meanwhile.
// nrhs > 1 (the lock-step batch): the links once, the vectors and the faces nrhs times.
double sweep_push_fraction(const qexhip_ctx *c, int interior_sites, int nrhs) {
  const double bsite = c->ndir * (c->recon == 1 ? 96.0 : (c->recon == 2 ? 112.0 : 144.0)) + 120.0 * nrhs;
  const double t_int = (double)interior_sites * bsite / 5.5e6;       // us
  return t_int > 0 ? std::min(1.0, std::max(0.65, nrhs * exchange_estimate_us(c) / t_int)) : 1.0;
}

// FusedCtl of the next fused launch: five words on lines of their own + the parked-block list
int sweep_fused_ctl(qexhip_ctx *c, int nbnd, FusedCtl *F, int nrhs) {
  if (c->fz_cap < nbnd) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->fz_buf) HIPCHK(hipFree(c->fz_buf));
    c->fz_buf = nullptr; c->fz_cap = 0;
    const int cap = std::max(2 * nbnd, 4096);
    HIPCHK(hipMalloc((void **)&c->fz_buf, (size_t)(256 + cap) * sizeof(unsigned int)));
    HIPCHK(hipMemsetAsync(c->fz_buf, 0, (size_t)(256 + cap) * sizeof(unsigned int), c->stream));     // (ordered before the launch that uses it)
    c->fz_cap = cap;
  }
  F->dec = c->fz_buf; F->ndef = c->fz_buf + 32; F->cl_done = c->fz_buf + 64; F->late = c->fz_buf + 96;
  F->list = c->fz_buf + 256;
  // the cleanup workgroups hold their slots through the LONG wait when the faces are late: few, so that a neighbour that shares the chip
  // always finds room; enough to take a whole face's parked blocks in ~100 us when it comes to that.  Measured (48^3 x 12 / 32^3 x 4
  // slabs, profiles/r06_fused_ab.log): 8 | 16 | 32 | 64 of them cost 91 | 92.5 | 95 | 97.6 us per iteration on the thin slab's normal
  // path (they poll while the whole launch is resident) and 663 | 513 | 450 | 438 us with EVERY block parked on the 48^3 one:
  // one per eight boundary workgroups (a parked face then costs each of them the same ~8-14 blocks whatever the face), at most 32
  F->ncl = std::max(std::min(nbnd, 4), std::min(nbnd / 8, 32));       // one per eight boundary workgroups, 4 .. 32
  const double tick_per_us = (double)c->dj.ticks / (c->dj.timeout_s * 1e6);
  if (c->opt_fused_spin_us == -2) F->spin_ticks = -1;
  else {
    const double us = c->opt_fused_spin_us >= 0 ? (double)c->opt_fused_spin_us : std::max(25.0, nrhs * exchange_estimate_us(c));
    F->spin_ticks = (long long)(us * tick_per_us);
  }
  return 0;
}

// The static rule in sweep_plan was set from one-rank rehearsals, where an exchange costs one kernel and no transport.  With a real
// communicator (nranks > 1) the decisions are MEASURED once per operator shape, right after the links are in place: the face
// exchange alone (what the fused sweep's placement and its short wait go by), then a few pairs of sweeps in every form -- exchange
// first, overlapped and split by sites, fused -- on scratch fields; the slower rank's time decides (max-all-reduce, so every rank
// takes the same branch).  Collective over the communicator, like set_links itself (ghost links).  Options overlap = 0 / 1 and
// hop_split = 0 / 2 switch the respective measurement off and pin form and bits; overlap = -2 asks for it on one rank too (test hook).
int sweep_autotune(qexhip_ctx *c) {
  const Geom &g = c->g;
  const int slot = c->ndir == 16;
  if (!g.halo || !c->W) return 0;
  const bool multi = c->nranks > 1 && comm_ready(c);
  if (multi) {
    // The forms select the stream (and, on RCCL, the communicator) an exchange is posted on: ranks that disagreed would never match.
    // set_links is collective, so this is the place to find out.
    const double code = 64.0 * c->opt_overlap + 1024.0 * (c->opt_hop_split + 1) + 65536.0 * (c->opt_fused_spin_us + 2);
    double v[2] = {code, -code};
    CHK(comm_allreduce_max(c, v, 2));
    if (v[0] != -v[1]) {
      qexhip_set_error("options overlap (QEXHIP_OVERLAP) / hop_split / fused_spin_us differ between the ranks (65536 (fused_spin_us + 2) + 1024 (hop_split + 1) "
                       "+ 64 overlap = %g .. %g): they must be the same everywhere", -v[1], v[0]);
      return -3;
    }
  }
  if (c->form_auto[slot] >= 0) return 0;           // measured already for this operator shape
  const bool measure = c->opt_overlap == -2 || multi;
  if (!measure) return 0;
  int lo_end, hi_beg, dummy;
  sweep_plan(c, &lo_end, &hi_beg, &dummy);
  if (hi_beg <= lo_end) { c->overlap_auto[slot] = 0; c->form_auto[slot] = 0; return 0; }        // no interior to overlap with
  DevField f[3];
  int rc = 0;
  for (int k = 0; k < 3 && !rc; k++) rc = field_alloc(c, f[k]);
  const int saved = c->opt_overlap, saved_form = c->opt_hop_split, saved_timers = c->timers_on;
  c->timers_on = 0;
  // (1) the exchange alone, on the compute stream: 2 to warm up (arena growth is collective and synchronous), 8 timed
  double tx = 0;
  for (int k = 0; k < 10 && !rc; k++) {
    if (k == 2) { rc = hipStreamSynchronize(c->stream) != hipSuccess ? -2 : 0; tx = -now_us(); }
    if (!rc) rc = comm_halo_exchange(c, f[0], 0, 0);
  }
  if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = -2;
  tx = (tx + now_us()) / 8.0;
  if (!rc && multi) {
    double v[2] = {tx, 0};
    rc = comm_allreduce_max(c, v, 1);
    tx = v[0];
  }
  if (!rc) c->xchg_us[slot] = tx;
  // (2) pairs of sweeps a -> b -> c (what the normal operator runs) in the forms this context can take
  const bool try_fused = peer_faces(c) && (c->opt_hop_split < 0 || c->opt_hop_split == 2);
  const bool try_sites = !(peer_faces(c) && c->opt_hop_split == 2);
  double t[3] = {0, 0, 0};
  for (int mode = 0; mode < 3 && !rc; mode++) {
    if ((mode == 1 && !try_sites) || (mode == 2 && !try_fused)) continue;
    if ((mode && saved == 0) || (!mode && saved == 1)) continue;           // overlap pinned: only the forms it leaves
    c->opt_overlap = mode ? 1 : 0;
    c->opt_hop_split = mode == 2 ? 2 : 0;
    for (int k = 0; k < 6 && !rc; k++) {
      if (k == 1) { rc = hipStreamSynchronize(c->stream) != hipSuccess ? -2 : 0; t[mode] = -now_us(); }
      DslashOpts o1, o2;
      DevField &in0 = f[(k & 1) ? 2 : 0], &out2 = f[(k & 1) ? 0 : 2];
      if (!rc) rc = dslash_sweep(c, f[1], in0, 1, o1);
      if (!rc) rc = dslash_sweep(c, out2, f[1], 0, o2);
    }
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = -2;
    t[mode] += now_us();
  }
  c->opt_overlap = saved;
  c->opt_hop_split = saved_form;
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
  // The fused form unless the split by sites is clearly (> 3 %) faster; overlap at all only on a clear win (> 5 %) over exchange-first:
  // the forms group the dot partials differently (and the fused one sums a boundary site's local hops first), so a decision that flips
  // between runs on timing noise would cost run-to-run bit-reproducibility of the residual history for nothing.
  const bool have1 = v[1] > 0, have2 = v[2] > 0;
  const int form = have2 && !(have1 && v[1] < 0.97 * v[2]) ? 2 : 0;
  const double best = form == 2 ? v[2] : v[1];
  c->form_auto[slot] = form;
  if (saved < 0) c->overlap_auto[slot] = (best > 0 && v[0] > 0 && best < 0.95 * v[0]) ? 1 : 0;
  for (int k = 0; k < 3; k++) c->overlap_tune_us[slot][k] = v[k] / 10.0;
  return 0;
}

int dslash_sweep(qexhip_ctx *c, DevField &out, DevField &in, int parity, const DslashOpts &o) {
  const Geom &g = c->g;
  if (!c->W) { qexhip_set_error("staggered links not set (qexhip_stag_set_links)"); return -3; }
  DslashArgs A;
  memset(&A, 0, sizeof A);
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
  const bool init = (o.ca != 0.0 || o.cb != 0.0);
  if ((init || o.dot) && !A.xs) { qexhip_set_error("dslash_sweep: b-term/dot needs xs"); return -1; }
  if (o.ca != 0.0 && !A.rin) { qexhip_set_error("dslash_sweep: a-term needs rin"); return -1; }
  int nparts = 0;
  const double2 *faces_on_cstream = nullptr;
  if (!g.halo) {
    if (c->ndir == 8) CHK((launch<8, false>(c, A, 0, g.Vh, init, o.dot, 0)));
    else CHK((launch<16, false>(c, A, 0, g.Vh, init, o.dot, 0)));
    nparts = (g.Vh + 255) / 256;
  } else {
    int lo_end, hi_beg, overlap;
    sweep_plan(c, &lo_end, &hi_beg, &overlap);
    CHK(devjoin_flush(c));
    const int nb_int = (hi_beg - lo_end + 255) / 256, nb_lo = (lo_end + 255) / 256, nb_hi = (g.Vh - hi_beg + 255) / 256;
    if (sweep_form(c, overlap) == 2) {
      // The fused sweep: the exchange is INSIDE the launch -- its first workgroups push the faces, its boundary workgroups poll the inbound
      // data words, park when those are late, its last workgroups clean up -- and neither the comm stream nor an event is involved.
      CHK(comm_halo_push_only(c, in, 1 - parity, &A.gh_hi, &A.gh_lo, &A.push));
      CHK(peer_ghost_args(c, &A.pg));
      CHK(sweep_fused_ctl(c, nb_lo + nb_hi, &A.fz));
      if (c->ndir == 8) CHK((launch<8, true>(c, A, lo_end, hi_beg, init, o.dot, 0, 0, lo_end, "dslash", nullptr, true, hi_beg, g.Vh)));
      else CHK((launch<16, true>(c, A, lo_end, hi_beg, init, o.dot, 0, 0, lo_end, "dslash", nullptr, true, hi_beg, g.Vh)));
      nparts = nb_int + nb_lo + nb_hi;
    } else if (!overlap) {
      // the exchange is ordered before us on the compute stream: one launch over all sites
      // (small local volumes are launch-latency-bound; this drops two launches per CG iteration)
      CHK(comm_halo_exchange(c, in, 1 - parity, 0));
      if (c->ndir == 8) CHK((launch<8, true>(c, A, 0, g.Vh, init, o.dot, 0)));
      else CHK((launch<16, true>(c, A, 0, g.Vh, init, o.dot, 0)));
      nparts = (g.Vh + 255) / 256;
    } else {
      // Split by sites: the face exchange on the comm stream (behind ev_ready: the producer of `in`), the interior launch on the
      // compute stream meanwhile, both t-faces in ONE launch on the COMM stream right behind the exchange -- it needs the ghost zones
      // and nothing of the interior launch, so it starts the moment the faces have arrived and runs beside the interior's tail
      // (round 4: 1-6 % of an iteration on thin slabs).  The join back is a device-side counter on BOTH transports since round 6 (a
      // one-lane signal kernel behind the boundary launch, a one-wave wait on the compute stream -- or, with o.defer_join, in the
      // prologue of the mailbox all-reduce that comes next): the runtime's cross-queue event dependency was ~28 us of dead time per
      // sweep (profiles/r05_timeline_*.txt).
      // Second sweep of a pair (op_xx) behind a first one of this form: the faces it sends ARE the output of the first sweep's boundary
      // launch, the comm stream's own previous kernel -- the exchange is posted at once (~28 us earlier: it no longer waits for the first
      // sweep's interior launch and the join), and only the boundary launch, which also reads the slices next to the faces, waits for
      // ev_ready.  (Round 5's "chained pair" did this and more -- it also dropped the join by narrowing the second interior -- and never
      // won; this keeps every launch as it is.)
      const bool early = o.pair2 && c->bnd_out_on_cstream == in.par(1 - parity);
      HIPCHK(hipEventRecord(c->ev_ready, c->stream));
      CHK(comm_halo_exchange(c, in, 1 - parity, 1, !early));
      if (early) HIPCHK(hipStreamWaitEvent(c->cstream, c->ev_ready, 0));
      if (c->ndir == 8) CHK((launch<8, true>(c, A, lo_end, hi_beg, init, o.dot, 0)));
      else CHK((launch<16, true>(c, A, lo_end, hi_beg, init, o.dot, 0)));
      if (c->ndir == 8) CHK((launch<8, true>(c, A, 0, lo_end, init, o.dot, nb_int, hi_beg, g.Vh, "dslash_bnd", c->cstream)));
      else CHK((launch<16, true>(c, A, 0, lo_end, init, o.dot, nb_int, hi_beg, g.Vh, "dslash_bnd", c->cstream)));
      CHK(devjoin_signal(c, c->cstream));
      if (o.defer_join && c->peer) CHK(devjoin_defer(c));
      else CHK(devjoin_wait(c, c->stream, c->cstream));
      nparts = nb_int + nb_lo + nb_hi;
      faces_on_cstream = out.par(parity);
    }
  }
  c->bnd_out_on_cstream = faces_on_cstream;
  if (o.dot) {
    if (nparts > c->part2_off) { qexhip_set_error("internal: partial buffer too small"); return -3; }
    // deferred final sum (inside k_cg_update) only while every workgroup can afford to re-sum the
    // partials itself; big local volumes take the separate one-block reduction
    if (o.dot == 2 && o.nparts_out && nparts <= 4096) *o.nparts_out = nparts;
    else {
      if (o.nparts_out) *o.nparts_out = 0;
      CHK(devjoin_flush(c));                 // (the reduction reads the boundary launch's partials)
      CHK(reduce_partials(c, nparts, o.dot_out));
    }
  }
  return 0;
}
