// gauge.hip -- plaquette, staple force and Wilson-flow step (kernels K8/K9/K10 of SURVEY.md 2.3).
//
// Restates (file:line in ctpeterson/qex):
//   plaq                       src/gauge/gaugeUtils.nim:213-282
//   makeStaples                src/gauge/staples.nim:153-238
//   gaugeActionDeriv (plaq)    src/gauge/gaugeAction.nim:148-204
//   contractProjectTAH         src/gauge/gaugeUtils.nim:389-398
//   gaugeFlow (RK3)            src/gauge/wflow.nim:21-67
// Layout: natural links G[parity][tile][mu][9][64] double2 (one 16-byte load per lane and entry).
// t-sharded runs: every parity half is [body ntile | ghost_hi 3F/64 tiles | ghost_lo 3F/64 tiles]; ghost_hi holds
// the upper neighbour's slices t = Xt, Xt+1, Xt+2, ghost_lo the lower neighbour's t = -3, -2, -1 ("virtual slices":
// a t-hop never wraps, link_off() maps the virtual slice to its ghost tile).  The ghosts of U are refreshed after
// every change of U to the depth the next kernel needs (1: plaquette action, 2: rectangles, 3: 3x3 clover loops);
// reductions end in an all-reduce.  Wilson lines that wind around t are not available sharded.
#include "qexhip_internal.h"
#include "reduce.h"
#include "su3.h"
#include "gauge_index.h"
#include <utility>
#include <algorithm>
#include <cstdlib>
#include <initializer_list>

struct GaugeNat {
  double2 *U = nullptr, *F = nullptr, *P = nullptr;
  double2 *U2 = nullptr;   // second link buffer: the fused flow stage reads U and writes exp(v) U here, then they swap
  double2 *D2 = nullptr;   // double links U_a(x) U_a(x+a) of the rectangle force (k_double_links / k_force_rect)
  size_t n2 = 0;  // double2 elements per field (incl. ghost tiles when t is sharded)
  int ghost_valid = 0;   // depth to which the ghost slices of U are current
  double *pp = nullptr; int npp = 0;   // per-workgroup plaquette partials of k_plaq
  double2 *M = nullptr;      // resident MD momenta (qexhip_md_*)
  double2 *Usave = nullptr;  // links saved around a force-gradient shift
  int save_ghost_valid = 0;
};

static int gauge_ghosts(qexhip_ctx *c, int depth);
static int read_global(qexhip_ctx *c, double *dev, int n, double *host);
static int ordered_sites(qexhip_ctx *c, const int **order, int *chunk, int *nb, double **part);

__device__ __forceinline__ size_t link_off(const Geom &g, const int x[4], int mu) {
  return g.halo ? link_off_t<true>(g, x, mu) : link_off_t<false>(g, x, mu);
}
__device__ __forceinline__ void shifted(const Geom &g, const int x[4], int mu, int d, int y[4]) {
  if (g.halo) shifted_t<true>(g, x, mu, d, y); else shifted_t<false>(g, x, mu, d, y);
}

// host [idx][mu][9] <-> tiles
__global__ void __launch_bounds__(256) k_gauge_to_tiles(Geom g, const double2 *__restrict__ host, double2 *G) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.V) return;
  int p = i >= g.Vh, c = i - p * g.Vh;
  for (int mu = 0; mu < 4; mu++) {
    double2 *w = G + (((size_t)p * g.etile + (c >> 6)) * 4 + mu) * 576 + (c & 63);
    for (int k = 0; k < 9; k++) w[k * 64] = host[((size_t)i * 4 + mu) * 9 + k];
  }
}
__global__ void __launch_bounds__(256) k_gauge_from_tiles(Geom g, double2 *__restrict__ host, const double2 *G) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.V) return;
  int p = i >= g.Vh, c = i - p * g.Vh;
  for (int mu = 0; mu < 4; mu++) {
    const double2 *w = G + (((size_t)p * g.etile + (c >> 6)) * 4 + mu) * 576 + (c & 63);
    for (int k = 0; k < 9; k++) host[((size_t)i * 4 + mu) * 9 + k] = w[k * 64];
  }
}

// plaquette: per site six Re tr[(U_mu(x)U_nu(x+mu))^+ (U_nu(x)U_mu(x+nu))], ip = mu(mu-1)/2+nu
// Visiting order from tile_order_table: wavefront w of workgroup b takes table slot 4*(b>>3)+w of XCD b&7.
// The four local links stay in registers (16 link loads per site).  Walking the six planes one at a time instead
// (24 loads, 8 of them L1 hits, 132 VGPR = 3 waves/SIMD instead of 1) measured 383 us against 277 us at 32^4: the
// kernel is bound by the number of L2->L1 requests, not by occupancy.  (Round 2: a workgroup per tile with six
// wavefronts, one per plane, and the four site links through LDS -- the same 16 loads at three wavefronts per SIMD -- measured
// 285-291 us against 230: rejected as well.  Round 6: gathering rows 0,1 of SU(3) links and rebuilding row 2: 227 -> 227 us, no change --
// the kernel is latency-bound at one wavefront per SIMD, 256 VGPRs + 35 AGPRs; __launch_bounds__(256, 2): two wavefronts per SIMD with
// 136-192 B/lane of scratch, 277-286 us.  profiles/r06_notes.md section 4.)
// S4: instead of the six plane sums, the eight sums of `s4_gauge` (stagg_pv_hmc/staghmc_spv_meas.nim:25-65, the S4 order
// parameter of arXiv:1111.2317): the plaquette of plane (mu, nu) at x is added to peo[mu][x_mu mod 2] and to peo[nu][x_nu mod 2];
// partial k = 2 d + (x_d mod 2).
template <bool HALO, bool S4 = false>
__global__ void __launch_bounds__(256) k_plaq(Geom g, const double2 *__restrict__ G, double *partials, const int *order, int chunk) {
  double pl[S4 ? 8 : 6] = {0, 0, 0, 0, 0, 0};
  const int slot = 4 * (blockIdx.x >> 3) + (threadIdx.x >> 6);
  const int e = slot < chunk ? order[(blockIdx.x & 7) * chunk + slot] : -1;
  const int p = e & 1, c = (e >> 1) * 64 + (threadIdx.x & 63);
  if (e >= 0 && c < g.Vh) {
    int x[4], y[4];
    coords_of(g, c, p, x);
    M3 U[4];
#pragma unroll
    for (int mu = 0; mu < 4; mu++) U[mu] = m3_load(G + link_off_t<HALO>(g, x, mu), 64);
#pragma unroll
    for (int mu = 1; mu < 4; mu++) {
#pragma unroll
      for (int nu = 0; nu < mu; nu++) {
        shifted_t<HALO>(g, x, nu, 1, y);
        M3 unumu = m3_mul(U[nu], m3_load(G + link_off_t<HALO>(g, y, mu), 64));
        shifted_t<HALO>(g, x, mu, 1, y);
        M3 umunu = m3_mul(U[mu], m3_load(G + link_off_t<HALO>(g, y, nu), 64));
        const double ps = m3_redot(umunu, unumu);
        if (S4) {
          const bool om = x[mu] & 1, on = x[nu] & 1;     // (a t-sharded slab starts at an even global t: local parity = global)
          pl[2 * mu] += om ? 0.0 : ps; pl[2 * mu + 1] += om ? ps : 0.0;
          pl[2 * nu] += on ? 0.0 : ps; pl[2 * nu + 1] += on ? ps : 0.0;
        } else {
          pl[(mu * (mu - 1)) / 2 + nu] += ps;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < (S4 ? 8 : 6); k++) {
    double r = block_sum_256(pl[k]);
    if (threadIdx.x == 0) partials[(size_t)k * gridDim.x + blockIdx.x] = r;
  }
}

__global__ void __launch_bounds__(256) k_plaq_final(const double *partials, int nb, double norm, double *out) {
  const int k = blockIdx.x;                    // one workgroup per plane
  double acc = 0;
  for (int i = threadIdx.x; i < nb; i += 256) acc += partials[(size_t)k * nb + i];
  double r = block_sum_256(acc);
  if (threadIdx.x == 0) out[k] = r / norm;
}

// Plaquette force / Wilson-flow stage.  F_mu(x) = TAH( U_mu(x) [cp * sum_nu (fwd + bwd staples)]^+ ), one lane per (mu, site).
// A workgroup = one 64-site tile x 4 directions (wavefront w handles mu = w), so the four wavefronts that share most of their
// neighbour links run together.  Which tile a workgroup takes comes from tile_order_table: one contiguous (t,z) region per
// XCD, walked in compact blocks with both parities adjacent -- every link is used by 19 staple terms, and tiles dealt
// round-robin over the XCDs would each re-fetch it from beyond their own L2 (cdna_hip_programming.md T1).
// flow mode (Pm != nullptr): the RK3 combination v = cf*f + cpm*p (wflow.nim:39,48,57) is formed here and written over the
// momentum field, and U' = exp(v) U goes to the other link buffer in the same kernel (wflow.nim:40-43): the compute-bound
// exp overlaps the L2-bound staple gathers of other waves and v, U are not re-read.
// CLOSED: exp(v) in closed form (m3_exp_tah, su3.h) instead of the reference's Taylor + 20 squarings: option "flow_exp".
// Momentum / force / new-link traffic is streamed past the caches (non-temporal): it is touched once per stage.

// what k_force_lds and k_force_lds2 do with a link's staple sum: f = TAH(U acc^+) (gaugeUtils.nim:389-398), then either the
// force itself (F = cp f) or the RK3 stage v = cf cp f + cpm p -> p, U' = exp(v) U (wflow.nim:36-62)
template <bool CLOSED>
__device__ __forceinline__ void force_finish(const M3 &U, const M3 &acc, bool live, size_t o, double2 *F, double cp, double2 *Pm,
                                             double cf, double cpm, double2 *Uout, int nt) {
  M3 f = m3_tah(m3_mul_na(U, acc));
  if (!live) return;
  if (Pm) {
    M3 v;
    const double cfp = cf * cp;
    if (cpm != 0.0) {
      const M3 pm = nt ? m3_load_nt(Pm + o, 64) : m3_load(Pm + o, 64);
#pragma unroll
      for (int k = 0; k < 9; k++) v.e[k] = make_double2(cfp * f.e[k].x + cpm * pm.e[k].x, cfp * f.e[k].y + cpm * pm.e[k].y);
    } else {
#pragma unroll
      for (int k = 0; k < 9; k++) v.e[k] = make_double2(cfp * f.e[k].x, cfp * f.e[k].y);
    }
    if (nt) m3_store_nt(Pm + o, 64, v); else m3_store(Pm + o, 64, v);
    if (Uout) {
      const M3 un = m3_mul(CLOSED ? m3_exp_tah(v) : m3_exp(v), U);
      if (nt) m3_store_nt(Uout + o, 64, un); else m3_store(Uout + o, 64, un);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 9; k++) { f.e[k].x *= cp; f.e[k].y *= cp; }
    if (nt) m3_store_nt(F + o, 64, f); else m3_store(F + o, 64, f);
  }
}

// The links that the four directions of a tile share are passed through LDS.  The kernel pays for
// its gathers at the CU's L2->L1 rate (profiles/r02_kforce_experiments.md): of the 19 matrices a lane fetches, U_nu(x) and
// U_nu(x-nu) (nu != mu) are the same for the three wavefronts with mu != nu -- and they are the workgroup's own links
// U_mu(x), U_mu(x-mu) of wavefront nu.  Every wavefront therefore loads its own two once, puts them into LDS (8 x 9 KiB),
// ONE barrier, and reads its six shared operands back from there: 14 global matrix loads per lane instead of 19.
template <bool CLOSED, bool HALO>
__global__ void __launch_bounds__(256) k_force_lds(Geom g, const double2 *__restrict__ G, double2 *F, double cp,
                                                   double2 *Pm, double cf, double cpm, double2 *Uout, const int *order, int chunk) {
  extern __shared__ double2 smU[];                    // [2 nu + (0: U_nu(x) | 1: U_nu(x-nu))][9][64]
  const int e = order[(blockIdx.x & 7) * chunk + (blockIdx.x >> 3)];
  if (e < 0) return;                                  // the whole workgroup together
  const int mu = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p = e & 1;
  const int c0 = (e >> 1) * 64 + lane;
  const bool live = c0 < g.Vh;
  const int c = live ? c0 : g.Vh - 1;                 // padding lanes of the last tile work on a valid site and store nothing
  int x[4], xpm[4], y[4], z[4];
  coords_of(g, c, p, x);
  shifted_t<HALO>(g, x, mu, 1, xpm);
  const size_t o = link_off_t<HALO>(g, x, mu);
  {
    shifted_t<HALO>(g, x, mu, -1, y);
    const M3 a = m3_load(G + o, 64), b = m3_load(G + link_off_t<HALO>(g, y, mu), 64);
    double2 *s0 = smU + (size_t)(2 * mu) * 576 + lane;
#pragma unroll
    for (int k = 0; k < 9; k++) { s0[k * 64] = a.e[k]; s0[576 + k * 64] = b.e[k]; }
  }
  __syncthreads();
  M3 acc = m3_zero();
#pragma unroll 1
  for (int nu = 0; nu < 4; nu++) {
    if (nu == mu) continue;
    const double2 *sn = smU + (size_t)(2 * nu) * 576 + lane;
    // forward: U_nu(x) U_mu(x+nu) U_nu(x+mu)^+          (stf[mu,nu], staples.nim:181-183)
    shifted_t<HALO>(g, x, nu, 1, y);
    M3 t = m3_mul_na(m3_load(G + link_off_t<HALO>(g, y, mu), 64), m3_load(G + link_off_t<HALO>(g, xpm, nu), 64));
    m3_mac(acc, m3_load(sn, 64), t);
    // backward: U_nu(x-nu)^+ U_mu(x-nu) U_nu(x-nu+mu)   (stu[mu,nu] shifted down, staples.nim:184-186)
    shifted_t<HALO>(g, x, nu, -1, y);
    shifted_t<HALO>(g, y, mu, 1, z);
    t = m3_mul_an(m3_load(sn + 576, 64), m3_load(G + link_off_t<HALO>(g, y, mu), 64));
    m3_mac(acc, t, m3_load(G + link_off_t<HALO>(g, z, nu), 64));
  }
  const M3 U = m3_load(smU + (size_t)(2 * mu) * 576 + lane, 64);
  force_finish<CLOSED>(U, acc, live, o, F, cp, Pm, cf, cpm, Uout, 1);
}

// k_force_lds over BOTH parities of a tile position in one workgroup (8 wavefronts: parity x direction).  The two tiles
// (tile, 0) and (tile, 1) are the same 128 consecutive lattice sites (whole x rows), so a one-hop neighbour in x -- and in
// y for three rows of four -- is a site of the OTHER wavefront group whose own links are already in LDS.  After the one
// barrier a lane therefore takes U_a(s) (slot 0) or U_a(s - a) (slot 1) of any site s inside the tile position from LDS
// and goes to global memory only for the rest: of the 48 neighbour matrices of a site 21 come from LDS on a 32-wide
// lattice (every operand with a hop in x, 3/4 of those with a hop in y), 35 + 4 global matrix loads per site instead of
// 56 + 4.  The stage is bound by the CUs' L2->L1 gather rate (profiles/r03_flow_stage_experiments.md), which is what
// this cuts.  Same products in the same order as k_force_lds: bit-identical results.
template <bool HALO>
__device__ __forceinline__ int site_cidx(const Geom &g, const int x[4]) {
  int t = x[3];
  if (HALO) t = t < 0 ? t + g.X[3] + 6 : t;
  return (x[0] + g.X[0] * (x[1] + g.X[1] * (x[2] + g.X[2] * t))) >> 1;
}
// U_a(gs) from global memory, unless the site s (parity q = the other one) lies in this tile position: then from the
// workgroup's LDS copy, slot 0 = U_a(s), slot 1 = U_a(s - a)
template <bool HALO>
__device__ __forceinline__ M3 link_lds_or_global(const Geom &g, const double2 *__restrict__ G, const double2 *smq, int tile,
                                                 const int s[4], int a, int slot, const int gs[4]) {
  const int cs = site_cidx<HALO>(g, s);
  if ((cs >> 6) == tile) return m3_load(smq + (size_t)(2 * a + slot) * 576 + (cs & 63), 64);
  return m3_load(G + link_off_t<HALO>(g, gs, a), 64);
}
// the same choice as an ADDRESS (generic pointer: the flat load that follows serves either aperture): one load sequence per
// operand instead of a divergent branch around two, which is what keeps k_flow_obs_clover2 inside its register budget
template <bool HALO>
__device__ __forceinline__ const double2 *link_ptr_lds_or_global(const Geom &g, const double2 *__restrict__ G, const double2 *smq, int tile,
                                                                 const int s[4], int a, int slot, const int gs[4]) {
  const int cs = site_cidx<HALO>(g, s);
  const double2 *pl = smq + (size_t)(2 * a + slot) * 576 + (cs & 63);
  const double2 *pg = G + link_off_t<HALO>(g, gs, a);
  return (cs >> 6) == tile ? pl : pg;
}
// (Round 6 tried fetching only rows 0,1 of every globally gathered SU(3) link and rebuilding row 2 in registers: the gather stream alone
// runs 527 -> 291 us that way, this kernel 687 -> 948 us -- 104 B/lane of spills and 44 % more fp64 work on a pipe that is already at
// its power limit; profiles/r06_notes.md section 4.  Not in the library.)
template <bool CLOSED, bool HALO>
__global__ void __launch_bounds__(512) k_force_lds2(Geom g, const double2 *__restrict__ G, double2 *F, double cp,
                                                    double2 *Pm, double cf, double cpm, double2 *Uout, const int *order, int chunk) {
  extern __shared__ double2 smU[];                    // [parity][2 nu + (0: U_nu(x) | 1: U_nu(x-nu))][9][64]
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int e = order[(blockIdx.x & 7) * chunk + 2 * (blockIdx.x >> 3) + (w >> 2)];   // the table holds (tile,0),(tile,1) adjacent
  if (order[(blockIdx.x & 7) * chunk + 2 * (blockIdx.x >> 3)] < 0) return;            // padding pair: the whole workgroup together
  const int mu = w & 3;
  const int p = e & 1, tile = e >> 1;
  const int c0 = tile * 64 + lane;
  const bool live = c0 < g.Vh;
  const int c = live ? c0 : g.Vh - 1;                 // padding lanes of the last tile work on a valid site and store nothing
  int x[4], xpm[4], y[4], z[4];
  coords_of(g, c, p, x);
  shifted_t<HALO>(g, x, mu, 1, xpm);
  const size_t o = link_off_t<HALO>(g, x, mu);
  double2 *smp = smU + (size_t)p * 8 * 576;           // this parity's own links
  const double2 *smq = smU + (size_t)(1 - p) * 8 * 576;   // the other parity's: every one-hop neighbour
  {
    shifted_t<HALO>(g, x, mu, -1, y);
    const M3 a = m3_load(G + o, 64), b = m3_load(G + link_off_t<HALO>(g, y, mu), 64);
    double2 *s0 = smp + (size_t)(2 * mu) * 576 + lane;
#pragma unroll
    for (int k = 0; k < 9; k++) { s0[k * 64] = a.e[k]; s0[576 + k * 64] = b.e[k]; }
  }
  __syncthreads();
  M3 acc = m3_zero();
#pragma unroll 1
  for (int nu = 0; nu < 4; nu++) {
    if (nu == mu) continue;
    const double2 *sn = smp + (size_t)(2 * nu) * 576 + lane;
    // forward: U_nu(x) U_mu(x+nu) U_nu(x+mu)^+          (stf[mu,nu], staples.nim:181-183)
    shifted_t<HALO>(g, x, nu, 1, y);
    M3 t = m3_mul_na(link_lds_or_global<HALO>(g, G, smq, tile, y, mu, 0, y), link_lds_or_global<HALO>(g, G, smq, tile, xpm, nu, 0, xpm));
    m3_mac(acc, m3_load(sn, 64), t);
    // backward: U_nu(x-nu)^+ U_mu(x-nu) U_nu(x-nu+mu)   (stu[mu,nu] shifted down, staples.nim:184-186)
    shifted_t<HALO>(g, x, nu, -1, y);
    shifted_t<HALO>(g, y, mu, 1, z);
    t = m3_mul_an(m3_load(sn + 576, 64), link_lds_or_global<HALO>(g, G, smq, tile, y, mu, 0, y));
    m3_mac(acc, t, link_lds_or_global<HALO>(g, G, smq, tile, xpm, nu, 1, z));     // U_nu(x+mu-nu) = slot 1 of site x+mu
  }
  const M3 U = m3_load(smp + (size_t)(2 * mu) * 576 + lane, 64);
  force_finish<CLOSED>(U, acc, live, o, F, cp, Pm, cf, cpm, Uout, 1);
}

// RK3 stage, second half: U <- exp(v) U with v already in the momentum field (wflow.nim:40-43)
// body link-tiles of both parity halves: linear index -> offset (skips the ghost tiles of a sharded field)
__device__ __forceinline__ size_t body_tile_off(size_t tile, size_t ntile4, size_t etile4) {
  const size_t p = tile >= ntile4;
  return (p * etile4 + (tile - p * ntile4)) * 576;
}
__global__ void __launch_bounds__(256) k_exp_update(size_t nlinks_tiles, double2 *G, const double2 *V, double t, size_t ntile4, size_t etile4) {
  size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;  // (tile-of-links, lane)
  size_t tile = j >> 6;
  if (tile >= nlinks_tiles) return;
  size_t o = body_tile_off(tile, ntile4, etile4) + (j & 63);
  M3 v = m3_load(V + o, 64);
  if (t != 1.0) {
#pragma unroll
    for (int k = 0; k < 9; k++) { v.e[k].x *= t; v.e[k].y *= t; }
  }
  M3 e = m3_exp(v);
  M3 u = m3_load(G + o, 64);
  m3_store(G + o, 64, m3_mul(e, u));
}

// ---------------- flow observables: F_munu (clover loops), E_s, E_t, Q   ----------------
// fmunu / densityE / topoQ (src/gauge/gaugeUtils.nim:1162-1271).  The closed paths (all corner
// rotations of the 1x1, 2x2, 1x2, 2x1, 1x3, 3x1, 3x3 loops, gaugeUtils.nim:1114-1160) are generated on
// the host into a table; one lane per site walks them (links come from L2), forms the two
// traceless anti-Hermitian F of a dual pair at a time (F10&F32, F20&F31, F21&F30) and
// accumulates -Re tr(F F) and the Q density, so no F field is ever written.
struct ObsPath { signed char step[12]; int len; double coef; };
struct ObsTable { int np; ObsPath p[28]; };   // paths of plane (mu,nu) = (1,0); others by substitution

__device__ __forceinline__ M3 path_prod(const Geom &g, const double2 *__restrict__ G, const int x0[4],
                                        const ObsPath &P, int mu, int nu) {
  int x[4] = {x0[0], x0[1], x0[2], x0[3]};
  M3 m;
  bool first = true;
  for (int i = 0; i < P.len; i++) {
    const int s = P.step[i];                 // +-1: direction mu, +-2: direction nu
    const int d = (s == 1 || s == -1) ? mu : nu;
    M3 u;
    if (s > 0) {
      u = m3_load(G + link_off(g, x, d), 64);
      x[d] = (g.halo && d == 3) ? x[d] + 1 : (x[d] + 1 >= g.X[d] ? 0 : x[d] + 1);
      m = first ? u : m3_mul(m, u);
    } else {
      x[d] = (g.halo && d == 3) ? x[d] - 1 : (x[d] == 0 ? g.X[d] - 1 : x[d] - 1);
      u = m3_load(G + link_off(g, x, d), 64);
      if (first) {
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int q = 0; q < 3; q++) m.e[3 * r + q] = make_double2(u.e[3 * q + r].x, -u.e[3 * q + r].y);
      } else {
        m = m3_mul_na(m, u);
      }
    }
    first = false;
  }
  return m;
}
__device__ __forceinline__ M3 fmunu_site(const Geom &g, const double2 *__restrict__ G, const int x[4],
                                         const ObsTable *T, int mu, int nu) {
  M3 acc = m3_zero();
  for (int p = 0; p < T->np; p++) {
    M3 m = path_prod(g, G, x, T->p[p], mu, nu);
    m3_axpy(acc, T->p[p].coef, m);
  }
  return m3_tah(acc);
}
// Re tr(a b)
__device__ __forceinline__ double m3_retr_mul(const M3 &a, const M3 &b) {
  double s = 0;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int q = 0; q < 3; q++) s += a.e[3 * r + q].x * b.e[3 * q + r].x - a.e[3 * r + q].y * b.e[3 * q + r].y;
  return s;
}
__global__ void __launch_bounds__(256) k_flow_obs(Geom g, const double2 *__restrict__ G, const ObsTable *T, double *partials,
                                                  const int *order, int chunk) {
  double es = 0, et = 0, q = 0;
  const int slot = 4 * (blockIdx.x >> 3) + (threadIdx.x >> 6);     // visiting order: tile_order_table, as k_plaq
  const int e = slot < chunk ? order[(blockIdx.x & 7) * chunk + slot] : -1;
  const int p = e & 1, c = (e >> 1) * 64 + (threadIdx.x & 63);
  if (e >= 0 && c < g.Vh) {
    int x[4];
    coords_of(g, c, p, x);
    // dual pairs and the sign of their term in Q = -(1/4pi^2) (F10 F32 - F20 F31 + F21 F30)
    const int pm[3][4] = {{1, 0, 3, 2}, {2, 0, 3, 1}, {2, 1, 3, 0}};
    const double sg[3] = {1.0, -1.0, 1.0};
#pragma unroll 1
    for (int k = 0; k < 3; k++) {
      M3 fa = fmunu_site(g, G, x, T, pm[k][0], pm[k][1]);
      M3 fb = fmunu_site(g, G, x, T, pm[k][2], pm[k][3]);
      es += m3_retr_mul(fa, fa);          // (mu,nu) with mu < 3 is spatial
      et += m3_retr_mul(fb, fb);          // the partner always has mu = 3
      q += sg[k] * m3_retr_mul(fa, fb);
    }
  }
  double r;
  r = block_sum_256(es); if (threadIdx.x == 0) partials[blockIdx.x] = r;
  r = block_sum_256(et); if (threadIdx.x == 0) partials[gridDim.x + blockIdx.x] = r;
  r = block_sum_256(q);  if (threadIdx.x == 0) partials[2 * gridDim.x + blockIdx.x] = r;
}
// loop == 1 (the plain clover: four plaquette leaves per plane), the observable of every flow step.  The path walker above
// fetches 96 matrices per site for it; here a workgroup is one tile x SIX wavefronts, wavefront w = plane w.  The eight
// links every plane of a site touches, U_mu(x) and U_mu(x-mu), go through LDS once (72 KiB, as k_force_lds), each wavefront
// gathers the eight remaining links of its plane: 56 matrices per site.  The leaves are multiplied and summed in the order
// of the path table, so F is bit for bit the path walker's.  The F of a dual pair (F10 & F32, F20 & F31, F21 & F30) meet
// through the same LDS for the Q density.
template <bool HALO>
__global__ void __launch_bounds__(384, 3) k_flow_obs_clover(Geom g, const double2 *__restrict__ G, double *partials,
                                                            const int *order, int chunk) {
  extern __shared__ double2 smO[];                    // [2 mu + (0: U_mu(x) | 1: U_mu(x-mu))][9][64], later F[plane][9][64]
  __shared__ double red[4][6];
  const int e = order[(blockIdx.x & 7) * chunk + (blockIdx.x >> 3)];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double es = 0, et = 0, q = 0, pl = 0;               // pl: Re tr of the plaquette of this wavefront's plane at x
  if (e >= 0) {                                       // the whole workgroup together
    const int p = e & 1;
    const int c0 = (e >> 1) * 64 + lane;
    const bool live = c0 < g.Vh;
    const int c = live ? c0 : g.Vh - 1;               // padding lanes of the last tile work on a valid site and count nothing
    int x[4], y[4], z[4];
    coords_of(g, c, p, x);
    for (int id = w; id < 8; id += 6) {
      const int mu = id >> 1;
      shifted_dyn<HALO>(g, x, mu, (id & 1) ? -1 : 0, y);
      const M3 m = m3_load(G + link_off_t<HALO>(g, y, mu), 64);
      double2 *d = smO + (size_t)id * 576 + lane;
#pragma unroll
      for (int k = 0; k < 9; k++) d[k * 64] = m.e[k];
    }
    __syncthreads();
    const int a = (w & 1) ? 3 : (w == 0 ? 1 : 2);     // planes (1,0) (3,2) (2,0) (3,1) (2,1) (3,0)
    const int b = (w == 0 || w == 2 || w == 5) ? 0 : (w == 1 ? 2 : 1);
    const double2 *Ua = smO + (size_t)(2 * a) * 576 + lane, *Ub = smO + (size_t)(2 * b) * 576 + lane;
    M3 acc = m3_zero();
    {                                                 // {-a,-b,a,b}: U_a(x-a)^+ U_b(x-a-b)^+ U_a(x-a-b) U_b(x-b)
      M3 m = m3_adj(m3_load(Ua + 576, 64));
      shifted_dyn<HALO>(g, x, a, -1, y);
      shifted_dyn<HALO>(g, y, b, -1, z);
      m = m3_mul_na(m, m3_load(G + link_off_t<HALO>(g, z, b), 64));
      __builtin_amdgcn_sched_barrier(0);
      m = m3_mul(m, m3_load(G + link_off_t<HALO>(g, z, a), 64));
      __builtin_amdgcn_sched_barrier(0);
      m = m3_mul(m, m3_load(Ub + 576, 64));
      __builtin_amdgcn_sched_barrier(0);
      m3_axpy(acc, 0.25, m);
      __builtin_amdgcn_sched_barrier(0);
    }
    {                                                 // {-b,a,b,-a}: U_b(x-b)^+ U_a(x-b) U_b(x-b+a) U_a(x)^+
      M3 m = m3_adj(m3_load(Ub + 576, 64));
      shifted_dyn<HALO>(g, x, b, -1, y);
      shifted_dyn<HALO>(g, y, a, 1, z);
      m = m3_mul(m, m3_load(G + link_off_t<HALO>(g, y, a), 64));
      __builtin_amdgcn_sched_barrier(0);
      m = m3_mul(m, m3_load(G + link_off_t<HALO>(g, z, b), 64));
      __builtin_amdgcn_sched_barrier(0);
      m = m3_mul_na(m, m3_load(Ua, 64));
      __builtin_amdgcn_sched_barrier(0);
      m3_axpy(acc, 0.25, m);
      __builtin_amdgcn_sched_barrier(0);
    }
    {                                                 // {a,b,-a,-b}: U_a(x) U_b(x+a) U_a(x+b)^+ U_b(x)^+
      M3 m = m3_load(Ua, 64);
      shifted_dyn<HALO>(g, x, a, 1, y);
      m = m3_mul(m, m3_load(G + link_off_t<HALO>(g, y, b), 64));
      __builtin_amdgcn_sched_barrier(0);
      shifted_dyn<HALO>(g, x, b, 1, y);
      m = m3_mul_na(m, m3_load(G + link_off_t<HALO>(g, y, a), 64));
      __builtin_amdgcn_sched_barrier(0);
      m = m3_mul_na(m, m3_load(Ub, 64));
      __builtin_amdgcn_sched_barrier(0);
      // this leaf IS the plaquette U_a(x) U_b(x+a) U_a(x+b)^+ U_b(x)^+ of plaq (gaugeUtils.nim:213-282): its trace is the
      // plaquette observable of the plane, for free
      if (live) pl = m.e[0].x + m.e[4].x + m.e[8].x;
      m3_axpy(acc, 0.25, m);
      __builtin_amdgcn_sched_barrier(0);
    }
    {                                                 // {b,-a,-b,a}: U_b(x) U_a(x+b-a)^+ U_b(x-a)^+ U_a(x-a)
      M3 m = m3_load(Ub, 64);
      shifted_dyn<HALO>(g, x, a, -1, y);
      shifted_dyn<HALO>(g, y, b, 1, z);
      m = m3_mul_na(m, m3_load(G + link_off_t<HALO>(g, z, a), 64));
      __builtin_amdgcn_sched_barrier(0);
      m = m3_mul_na(m, m3_load(G + link_off_t<HALO>(g, y, b), 64));
      __builtin_amdgcn_sched_barrier(0);
      m = m3_mul(m, m3_load(Ua + 576, 64));
      __builtin_amdgcn_sched_barrier(0);
      m3_axpy(acc, 0.25, m);
      __builtin_amdgcn_sched_barrier(0);
    }
    const M3 F = m3_tah(acc);
    const double ff = m3_retr_mul(F, F);
    __syncthreads();                                  // every wavefront is done with the links
    {
      double2 *d = smO + (size_t)w * 576 + lane;
#pragma unroll
      for (int k = 0; k < 9; k++) d[k * 64] = F.e[k];
    }
    __syncthreads();
    if (live) {
      if (w & 1) et = ff;                             // the partner of a pair always has mu = 3
      else {
        es = ff;
        const M3 fb = m3_load(smO + (size_t)(w + 1) * 576 + lane, 64);
        q = (w == 2 ? -1.0 : 1.0) * m3_retr_mul(F, fb);   // Q = -(1/4pi^2) (F10 F32 - F20 F31 + F21 F30)
      }
    }
  }
  es = wave_sum(es); et = wave_sum(et); q = wave_sum(q); pl = wave_sum(pl);
  if (lane == 0) { red[0][w] = es; red[1][w] = et; red[2][w] = q; red[3][w] = pl; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const double *r = red[threadIdx.x];
    partials[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = ((r[0] + r[1]) + (r[2] + r[3])) + (r[4] + r[5]);
  } else if (threadIdx.x < 9) {
    // planes of wavefronts 0..5: (1,0) (3,2) (2,0) (3,1) (2,1) (3,0) -> plaq's index mu (mu - 1) / 2 + nu: 0 5 1 4 2 3
    const int wv = threadIdx.x - 3;
    const int ip = wv == 0 ? 0 : (wv == 1 ? 5 : (wv == 2 ? 1 : (wv == 3 ? 4 : (wv == 4 ? 2 : 3))));
    partials[(size_t)(3 + ip) * gridDim.x + blockIdx.x] = red[3][wv];
  }
}
// k_flow_obs_clover over BOTH parities of a tile position in one workgroup (12 wavefronts: parity x plane), as k_force_lds2:
// the two tiles are the same 128 consecutive lattice sites (whole x rows), so every operand of a leaf that sits one hop in
// x -- and in y for three rows of four -- from the lane's site is a link the OTHER parity group already holds in LDS:
// U_d(s) = slot 0 of site s, U_d(s - d) = slot 1.  All eight gathered operands of a plane can be named that way through a
// one-hop neighbour (x+-a or x+-b); on a 32-wide lattice 21 of the 48 per site then come from LDS: 27 + 8 global matrix loads
// per site instead of 48 + 8.  Same products in the same order as k_flow_obs_clover: bit-identical F, E, Q, plaquettes.
template <bool HALO>
__global__ void __launch_bounds__(768) k_flow_obs_clover2(Geom g, const double2 *__restrict__ G, double *partials,
                                                             const int *order, int chunk) {
  extern __shared__ double2 smO[];                    // [parity][2 mu + (0: U_mu(x) | 1: U_mu(x-mu))][9][64], later F[parity][plane][9][64]
  __shared__ double red[4][12];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int pgrp = wv / 6, w = wv - 6 * pgrp;         // parity group, plane
  const int slot0 = (blockIdx.x & 7) * chunk + 2 * (blockIdx.x >> 3);   // the table holds (tile,0),(tile,1) adjacent
  double es = 0, et = 0, q = 0, pl = 0;
  if (order[slot0] >= 0) {                            // padding pair: the whole workgroup together
    const int e = order[slot0 + pgrp];
    const int p = e & 1, tile = e >> 1;
    const int c0 = tile * 64 + lane;
    const bool live = c0 < g.Vh;
    const int c = live ? c0 : g.Vh - 1;
    int x[4], xma[4], xmb[4], xpa[4], xpb[4], z[4];
    coords_of(g, c, p, x);
    double2 *smp = smO + (size_t)p * 8 * 576;
    const double2 *smq = smO + (size_t)(1 - p) * 8 * 576;
    for (int id = w; id < 8; id += 6) {
      const int mu = id >> 1;
      shifted_dyn<HALO>(g, x, mu, (id & 1) ? -1 : 0, z);
      const M3 m = m3_load(G + link_off_t<HALO>(g, z, mu), 64);
      double2 *d = smp + (size_t)id * 576 + lane;
#pragma unroll
      for (int k = 0; k < 9; k++) d[k * 64] = m.e[k];
    }
    __syncthreads();
    const int a = (w & 1) ? 3 : (w == 0 ? 1 : 2);     // planes (1,0) (3,2) (2,0) (3,1) (2,1) (3,0)
    const int b = (w == 0 || w == 2 || w == 5) ? 0 : (w == 1 ? 2 : 1);
    const double2 *Ua = smp + (size_t)(2 * a) * 576 + lane, *Ub = smp + (size_t)(2 * b) * 576 + lane;
    shifted_dyn<HALO>(g, x, a, -1, xma);
    shifted_dyn<HALO>(g, x, b, -1, xmb);
    shifted_dyn<HALO>(g, x, a, 1, xpa);
    shifted_dyn<HALO>(g, x, b, 1, xpb);
    M3 acc = m3_zero();
    {                                                 // {-a,-b,a,b}: U_a(x-a)^+ U_b(x-a-b)^+ U_a(x-a-b) U_b(x-b)
      M3 m = m3_adj(m3_load(Ua + 576, 64));
      shifted_dyn<HALO>(g, xma, b, -1, z);
      m = m3_mul_na(m, m3_load(link_ptr_lds_or_global<HALO>(g, G, smq, tile, xma, b, 1, z), 64));     // U_b(x-a-b) = slot 1 of site x-a
      __builtin_amdgcn_sched_barrier(0);
      m = m3_mul(m, m3_load(link_ptr_lds_or_global<HALO>(g, G, smq, tile, xmb, a, 1, z), 64));        // U_a(x-a-b) = slot 1 of site x-b
      __builtin_amdgcn_sched_barrier(0);
      m = m3_mul(m, m3_load(Ub + 576, 64));
      __builtin_amdgcn_sched_barrier(0);
      m3_axpy(acc, 0.25, m);
      __builtin_amdgcn_sched_barrier(0);
    }
    {                                                 // {-b,a,b,-a}: U_b(x-b)^+ U_a(x-b) U_b(x-b+a) U_a(x)^+
      M3 m = m3_adj(m3_load(Ub + 576, 64));
      shifted_dyn<HALO>(g, xmb, a, 1, z);
      m = m3_mul(m, m3_load(link_ptr_lds_or_global<HALO>(g, G, smq, tile, xmb, a, 0, xmb), 64));
      __builtin_amdgcn_sched_barrier(0);
      m = m3_mul(m, m3_load(link_ptr_lds_or_global<HALO>(g, G, smq, tile, xpa, b, 1, z), 64));        // U_b(x+a-b) = slot 1 of site x+a
      __builtin_amdgcn_sched_barrier(0);
      m = m3_mul_na(m, m3_load(Ua, 64));
      __builtin_amdgcn_sched_barrier(0);
      m3_axpy(acc, 0.25, m);
      __builtin_amdgcn_sched_barrier(0);
    }
    {                                                 // {a,b,-a,-b}: U_a(x) U_b(x+a) U_a(x+b)^+ U_b(x)^+
      M3 m = m3_load(Ua, 64);
      m = m3_mul(m, m3_load(link_ptr_lds_or_global<HALO>(g, G, smq, tile, xpa, b, 0, xpa), 64));
      __builtin_amdgcn_sched_barrier(0);
      m = m3_mul_na(m, m3_load(link_ptr_lds_or_global<HALO>(g, G, smq, tile, xpb, a, 0, xpb), 64));
      __builtin_amdgcn_sched_barrier(0);
      m = m3_mul_na(m, m3_load(Ub, 64));
      __builtin_amdgcn_sched_barrier(0);
      if (live) pl = m.e[0].x + m.e[4].x + m.e[8].x;  // this leaf is the plaquette of plaq (gaugeUtils.nim:213-282)
      m3_axpy(acc, 0.25, m);
      __builtin_amdgcn_sched_barrier(0);
    }
    {                                                 // {b,-a,-b,a}: U_b(x) U_a(x+b-a)^+ U_b(x-a)^+ U_a(x-a)
      M3 m = m3_load(Ub, 64);
      shifted_dyn<HALO>(g, xpb, a, -1, z);
      m = m3_mul_na(m, m3_load(link_ptr_lds_or_global<HALO>(g, G, smq, tile, xpb, a, 1, z), 64));     // U_a(x+b-a) = slot 1 of site x+b
      __builtin_amdgcn_sched_barrier(0);
      m = m3_mul_na(m, m3_load(link_ptr_lds_or_global<HALO>(g, G, smq, tile, xma, b, 0, xma), 64));
      __builtin_amdgcn_sched_barrier(0);
      m = m3_mul(m, m3_load(Ua + 576, 64));
      __builtin_amdgcn_sched_barrier(0);
      m3_axpy(acc, 0.25, m);
      __builtin_amdgcn_sched_barrier(0);
    }
    const M3 F = m3_tah(acc);
    const double ff = m3_retr_mul(F, F);
    __syncthreads();                                  // every wavefront is done with the links
    {
      double2 *d = smO + (size_t)wv * 576 + lane;
#pragma unroll
      for (int k = 0; k < 9; k++) d[k * 64] = F.e[k];
    }
    __syncthreads();
    if (live) {
      if (w & 1) et = ff;                             // the partner of a pair always has mu = 3
      else {
        es = ff;
        const M3 fb = m3_load(smO + (size_t)(wv + 1) * 576 + lane, 64);
        q = (w == 2 ? -1.0 : 1.0) * m3_retr_mul(F, fb);   // Q = -(1/4pi^2) (F10 F32 - F20 F31 + F21 F30)
      }
    }
  }
  es = wave_sum(es); et = wave_sum(et); q = wave_sum(q); pl = wave_sum(pl);
  if (lane == 0) { red[0][wv] = es; red[1][wv] = et; red[2][wv] = q; red[3][wv] = pl; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const double *r = red[threadIdx.x];
    partials[(size_t)threadIdx.x * gridDim.x + blockIdx.x] =
        (((r[0] + r[1]) + (r[2] + r[3])) + (r[4] + r[5])) + (((r[6] + r[7]) + (r[8] + r[9])) + (r[10] + r[11]));
  } else if (threadIdx.x < 9) {
    // planes of wavefronts 0..5: (1,0) (3,2) (2,0) (3,1) (2,1) (3,0) -> plaq's index mu (mu - 1) / 2 + nu: 0 5 1 4 2 3
    const int k = threadIdx.x - 3;
    const int ip = k == 0 ? 0 : (k == 1 ? 5 : (k == 2 ? 1 : (k == 3 ? 4 : (k == 4 ? 2 : 3))));
    partials[(size_t)(3 + ip) * gridDim.x + blockIdx.x] = red[3][k] + red[3][k + 6];
  }
}
__global__ void __launch_bounds__(256) k_obs_final(const double *partials, int nb, double vol, double *out) {
  const int k = blockIdx.x;                    // one workgroup per observable
  double acc = 0;
  for (int i = threadIdx.x; i < nb; i += 256) acc += partials[(size_t)k * nb + i];
  double r = block_sum_256(acc);
  if (threadIdx.x == 0) out[k] = r;            // raw sums: rank-summed and normalised by the caller
}

// host: build the path table for plane (mu,nu) -> (step +-1, step +-2)
static int obs_all_corners(const int *path, int np, int out[][12]) {
  int n = 0, old = 0;
  for (int i = 0; i < np; i++)
    if (path[i] != old) {
      for (int j = 0; j < np; j++) out[n][j] = path[(j + i) % np];
      n++;
      old = path[i];
    }
  return n;
}
static int obs_build_table(int loop, ObsTable &T) {
  if (loop != 1 && loop != 3 && loop != 4 && loop != 5) { qexhip_set_error("fmunu uses loop in [1,3,4,5], but got %d", loop); return -1; }
  double k[5] = {0, 0, 0, 0, 0};
  if (loop == 1) k[0] = 1.0;
  else {
    k[4] = (loop == 3) ? 1.0 / 90.0 : (loop == 5 ? 1.0 / 180.0 : 0.0);
    k[0] = 19.0 / 9.0 - 55.0 * k[4];
    k[1] = 1.0 / 36.0 - 16.0 * k[4];
    k[2] = 64.0 * k[4] - 32.0 / 45.0;
    k[3] = 1.0 / 15.0 - 6.0 * k[4];
  }
  static const int lpc[5] = {4, 4, 8, 8, 4};
  T.np = 0;
  auto add = [&](int grp, std::initializer_list<int> l) {
    int path[12], n = 0, tmp[8][12];
    for (int v : l) path[n++] = v;
    int c = obs_all_corners(path, n, tmp);
    for (int q = 0; q < c; q++) {
      ObsPath &P = T.p[T.np++];
      for (int j = 0; j < n; j++) P.step[j] = (signed char)tmp[q][j];
      P.len = n;
      P.coef = k[grp] / lpc[grp];
    }
  };
  const int a = 1, b = 2;
  add(0, {-a, -b, a, b});
  if (loop >= 3) add(1, {-a, -a, -b, -b, a, a, b, b});
  if (loop >= 4) {
    add(2, {-a, -a, -b, a, a, b});
    add(2, {-a, -b, -b, a, b, b});
    add(3, {-a, -a, -a, -b, a, a, a, b});
    add(3, {-a, -b, -b, -b, a, b, b, b});
  }
  if (loop == 3 || loop == 5) add(4, {-a, -a, -a, -b, -b, -b, a, a, a, b, b, b});
  return 0;
}

// plaq6 != nullptr (loop 1, clover kernel only): the six plaquettes of plaq come out of the same pass (qexhip_flow_measure)
int gauge_flow_obs(qexhip_ctx *c, int loop, double out[3], double *plaq6) {
  if (!c->gn) { qexhip_set_error("gauge field not set (qexhip_gauge_set)"); return -3; }
  for (int d = 0; d < 4; d++)
    if (loop > 1 && c->g.X[d] < 4) { qexhip_set_error("improved fmunu needs extents >= 4"); return -1; }
  CHK(gauge_ghosts(c, loop > 1 ? 3 : 1));      // the improved clover reaches three sites from x (3x3, 1x3 loops)
  ObsTable T;
  CHK(obs_build_table(loop, T));
  if (!c->obs_table) HIPCHK(hipMalloc(&c->obs_table, sizeof(ObsTable)));
  ObsTable *dT = (ObsTable *)c->obs_table;
  HIPCHK(hipMemcpyAsync(dT, &T, sizeof(T), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));  // T is a stack object
  const int *order = nullptr; int chunk = 0, nb = 0;
  double *part = nullptr;
  CHK(ordered_sites(c, &order, &chunk, &nb, &part));
  const size_t shb = (size_t)8 * 576 * sizeof(double2);
  const bool clover = loop == 1 && c->opt_obs_clover;
  if (plaq6 && !clover) { qexhip_set_error("flow_measure: the fused pass is the clover kernel's (loop 1, option obs_clover)"); return -3; }
  if (clover) {
    if (!(c->lds_attr_done & 2)) {
      HIPCHK(hipFuncSetAttribute((const void *)k_flow_obs_clover<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
      HIPCHK(hipFuncSetAttribute((const void *)k_flow_obs_clover<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
      c->lds_attr_done |= 2;
    }
    ScopedTimer tm(c, "flowobs", c->stream);
    if (c->opt_force_pair && c->tile_pairs_ok) {
      // both parities of a tile position per workgroup: 144 KiB of LDS, one workgroup of 12 wavefronts per CU
      if (!(c->lds_attr_done & 4)) {
        HIPCHK(hipFuncSetAttribute((const void *)k_flow_obs_clover2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * shb)));
        HIPCHK(hipFuncSetAttribute((const void *)k_flow_obs_clover2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * shb)));
        c->lds_attr_done |= 4;
      }
      nb = 4 * chunk;                                 // one workgroup per tile position
      if (c->g.halo) k_flow_obs_clover2<true><<<nb, 768, 2 * shb, c->stream>>>(c->g, c->gn->U, part, order, chunk);
      else k_flow_obs_clover2<false><<<nb, 768, 2 * shb, c->stream>>>(c->g, c->gn->U, part, order, chunk);
    } else {
      nb = 8 * chunk;                                 // one workgroup per tile
      if (c->g.halo) k_flow_obs_clover<true><<<nb, 384, shb, c->stream>>>(c->g, c->gn->U, part, order, chunk);
      else k_flow_obs_clover<false><<<nb, 384, shb, c->stream>>>(c->g, c->gn->U, part, order, chunk);
    }
    HIPCHK(hipGetLastError());
  } else {
    ScopedTimer tm(c, "flowobs", c->stream);
    k_flow_obs<<<nb, 256, 0, c->stream>>>(c->g, c->gn->U, dT, part, order, chunk);
    HIPCHK(hipGetLastError());
  }
  const int nobs = plaq6 ? 9 : 3;
  k_obs_final<<<nobs, 256, 0, c->stream>>>(part, nb, (double)c->g.V, &c->dscal[24]);
  HIPCHK(hipGetLastError());
  double res[9];
  CHK(read_global(c, &c->dscal[24], nobs, res));
  for (int k = 0; k < 3; k++) out[k] = res[k];
  const double vol = (double)c->g.V * (double)c->nranks;
  if (plaq6) for (int k = 0; k < 6; k++) plaq6[k] = res[3 + k] / (vol * 18.0);    // pl[i]/(physVol*np*nc), gaugeUtils.nim:277
  out[0] = -out[0] / vol; out[1] = -out[1] / vol;
  out[2] = -out[2] / (4.0 * 3.14159265358979323846 * 3.14159265358979323846);
  return 0;
}

// ---------------- general gauge actions: plaq + rect, plaq + adjplaq (completes row a14) ----------------
// gaugeActionDeriv with c.rect (src/gauge/gaugeAction.nim:205-241,275-331) and gaugeADeriv / forceA
// (:683-747), as used by the action-selectable flow of src/flow/flow.nim:22-90.  Same lane
// mapping as k_force_lds.  The 18 rectangle staples of a link are walked as 5-link paths.
__device__ const signed char RECT_STEPS[6][5] = {
    {2, 2, 1, -2, -2}, {2, 1, 1, -2, -1}, {-1, 2, 1, 1, -2},          // +nu
    {-2, -2, 1, 2, 2}, {-2, 1, 1, 2, -1}, {-1, -2, 1, 1, 2}};         // -nu
template <bool CLOSED>
__global__ void __launch_bounds__(256) k_force_gen(Geom g, const double2 *__restrict__ G, double2 *F, double cp,
                                                   double c2, int kind, double2 *Pm, double cf, double cpm, int raw, double2 *Uout,
                                                   const int *order, int chunk) {
  const int e = order[(blockIdx.x & 7) * chunk + (blockIdx.x >> 3)];   // tile_order_table
  if (e < 0) return;
  const int mu = threadIdx.x >> 6;
  const int p = e & 1;
  const int c = (e >> 1) * 64 + (threadIdx.x & 63);
  if (c >= g.Vh) return;
  int x[4], xpm[4], y[4], z[4];
  coords_of(g, c, p, x);
  shifted(g, x, mu, 1, xpm);
  const size_t o = link_off(g, x, mu);
  const M3 U = m3_load(G + o, 64);
  M3 acc = m3_zero();
#pragma unroll 1
  for (int nu = 0; nu < 4; nu++) {
    if (nu == mu) continue;
#pragma unroll 1
    for (int dir = 0; dir < 2; dir++) {
      M3 s;
      if (dir == 0) {
        shifted(g, x, nu, 1, y);
        M3 t = m3_mul_na(m3_load(G + link_off(g, y, mu), 64), m3_load(G + link_off(g, xpm, nu), 64));
        s = m3_mul(m3_load(G + link_off(g, x, nu), 64), t);
      } else {
        shifted(g, x, nu, -1, y);
        shifted(g, y, mu, 1, z);
        M3 t = m3_mul_an(m3_load(G + link_off(g, y, nu), 64), m3_load(G + link_off(g, y, mu), 64));
        s = m3_mul(t, m3_load(G + link_off(g, z, nu), 64));
      }
      if (kind == 1) {
        // weight cp + ca*tr(S^+ U), ca = 2 c.adjplaq/nc^2 (gaugeAction.nim:694,705-706)
        double tr = 0, ti = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
          tr += s.e[k].x * U.e[k].x + s.e[k].y * U.e[k].y;
          ti += s.e[k].x * U.e[k].y - s.e[k].y * U.e[k].x;
        }
        const double wr = cp + c2 * tr, wi = c2 * ti;
#pragma unroll
        for (int k = 0; k < 9; k++) {
          acc.e[k].x += wr * s.e[k].x - wi * s.e[k].y;
          acc.e[k].y += wr * s.e[k].y + wi * s.e[k].x;
        }
      } else {
        m3_axpy(acc, cp, s);
      }
    }
    if (kind == 0 && c2 != 0.0) {
#pragma unroll 1
      for (int q = 0; q < 6; q++) {
        ObsPath P;
        P.len = 5;
#pragma unroll
        for (int k = 0; k < 5; k++) P.step[k] = RECT_STEPS[q][k];
        M3 m = path_prod(g, G, x, P, mu, nu);
        m3_axpy(acc, c2, m);
      }
    }
  }
  if (raw) {   // the derivative itself (gaugeActionDeriv / gaugeForceCust), no projection
    m3_store(F + o, 64, acc);
    return;
  }
  M3 f = m3_tah(m3_mul_na(U, acc));
  if (Pm) {
    M3 v;
    if (cpm != 0.0) {
      M3 pm = m3_load(Pm + o, 64);
#pragma unroll
      for (int k = 0; k < 9; k++) v.e[k] = make_double2(cf * f.e[k].x + cpm * pm.e[k].x, cf * f.e[k].y + cpm * pm.e[k].y);
    } else {
#pragma unroll
      for (int k = 0; k < 9; k++) v.e[k] = make_double2(cf * f.e[k].x, cf * f.e[k].y);
    }
    m3_store(Pm + o, 64, v);
    // fused second half of the RK3 stage (wflow.nim:40-43): U <- exp(v) U into the other buffer, so the
    // compute-bound exp overlaps the L2-bound staple gathers of other waves and v, U are not re-read
    if (Uout) m3_store(Uout + o, 64, m3_mul(CLOSED ? m3_exp_tah(v) : m3_exp(v), m3_load(G + o, 64)));
  } else {
    m3_store(F + o, 64, f);
  }
}

// The plaquette + rectangle derivative (gaugeAction.nim:195-241,275-331) with shared factors instead of 18 independent
// 5-link walks per link.  With the double links D_a(y) = U_a(y) U_a(y+a) (k_double_links: one product per link and stage)
// the eight paths from x to x+mu of a plane (mu, nu) are
//   P1 = U_nu(x) U_mu(x+nu) U_nu(x+mu)^+                 R2 = U_nu(x) D_mu(x+nu) U_nu(x+2mu)^+ U_mu(x+mu)^+
//   P2 = U_nu(x-nu)^+ U_mu(x-nu) U_nu(x+mu-nu)           R5 = U_nu(x-nu)^+ D_mu(x-nu) U_nu(x+2mu-nu) U_mu(x+mu)^+
//   R3 = U_mu(x-mu)^+ U_nu(x-mu) D_mu(x-mu+nu) U_nu(x+mu)^+
//   R6 = U_mu(x-mu)^+ U_nu(x-mu-nu)^+ D_mu(x-mu-nu) U_nu(x+mu-nu)
//   R1 = D_nu(x) U_mu(x+2nu) D_nu(x+mu)^+                 R4 = D_nu(x-2nu)^+ U_mu(x-2nu) D_nu(x+mu-2nu)
// (R1..R6 = RECT_STEPS rows 0, 1, 2, 3, 4, 5) and P1 + R2, P2 + R5, R3 + R6 share their leading link: 17 matrix products
// and 22 gathers per plane instead of 28 and 36 -- 54 instead of 86 products per link with the two of the finish.
// t-sharded fields: the double links of the ghost slices are formed locally from the ghost links (depth 2), bit for bit
// what the neighbour rank forms for its body.
template <bool HALO>   // HALO: also the ghost tiles (virtual slices), whose double links the boundary sites gather
__global__ void __launch_bounds__(256) k_double_links(Geom g, const double2 *__restrict__ G, double2 *D) {
  const int T = HALO ? g.etile : g.ntile;
  const int p = blockIdx.x >= T, tile = blockIdx.x - p * T;
  const int a = threadIdx.x >> 6;
  const int c = tile * 64 + (threadIdx.x & 63);
  if (c >= (HALO ? T * 64 : g.Vh)) return;
  int x[4], y[4];
  coords_of(g, c, p, x);                                   // HALO: x[3] runs over the virtual slices 0 .. Xt+5
  if (HALO) {
    if (x[3] >= g.X[3] + 3) x[3] -= g.X[3] + 6;            // Xt+3 .. Xt+5 are the slices -3 .. -1 (link_off_t)
    if (a == 3 && x[3] + 1 > g.X[3] + 2) return;           // U_t(x+t) beyond the last ghost slice: nobody reads this one
  }
  shifted_t<HALO>(g, x, a, 1, y);
  const size_t o = link_off_t<HALO>(g, x, a);
  m3_store(D + o, 64, m3_mul(m3_load(G + o, 64), m3_load(G + link_off_t<HALO>(g, y, a), 64)));
}
__device__ __forceinline__ void m3_scale(M3 &a, double s) {
#pragma unroll
  for (int k = 0; k < 9; k++) { a.e[k].x *= s; a.e[k].y *= s; }
}
template <bool CLOSED, bool HALO>
__global__ void __launch_bounds__(256, 2) k_force_rect(Geom g, const double2 *__restrict__ G, const double2 *__restrict__ D, double2 *F,
                                                    double cp, double c2, double2 *Pm, double cf, double cpm, int raw, double2 *Uout,
                                                    const int *order, int chunk) {
  const int e = order[(blockIdx.x & 7) * chunk + (blockIdx.x >> 3)];   // tile_order_table
  if (e < 0) return;
  const int mu = threadIdx.x >> 6;
  const int p = e & 1;
  const int c = (e >> 1) * 64 + (threadIdx.x & 63);
  if (c >= g.Vh) return;
  int x[4], xpm[4], xmm[4], y[4], z[4], w[4];
  coords_of(g, c, p, x);
  shifted_t<HALO>(g, x, mu, 1, xpm);
  shifted_t<HALO>(g, x, mu, -1, xmm);
  const size_t o = link_off_t<HALO>(g, x, mu);
#define LDG(f, s, a) m3_load((f) + link_off_t<HALO>(g, s, a), 64)
  M3 acc = m3_zero();
#pragma unroll 1
  for (int nu = 0; nu < 4; nu++) {
    if (nu == mu) continue;
    {   // P1 + R2 = U_nu(x) [cp U_mu(x+nu) U_nu(x+mu)^+ + c2 D_mu(x+nu) U_nu(x+2mu)^+ U_mu(x+mu)^+]
      shifted_t<HALO>(g, x, nu, 1, y);
      M3 s = m3_mul_na(LDG(G, y, mu), LDG(G, xpm, nu));
      m3_scale(s, cp);
      shifted_t<HALO>(g, xpm, mu, 1, z);
      M3 t = m3_mul_na(LDG(D, y, mu), LDG(G, z, nu));
      m3_scale(t, c2);
      m3_mac_na(s, t, LDG(G, xpm, mu));
      m3_mac(acc, LDG(G, x, nu), s);
    }
    __builtin_amdgcn_sched_barrier(0);
    {   // P2 + R5 = U_nu(x-nu)^+ [cp U_mu(x-nu) U_nu(x+mu-nu) + c2 D_mu(x-nu) U_nu(x+2mu-nu) U_mu(x+mu)^+]
      shifted_t<HALO>(g, x, nu, -1, y);
      shifted_t<HALO>(g, y, mu, 1, z);
      M3 s = m3_mul(LDG(G, y, mu), LDG(G, z, nu));
      m3_scale(s, cp);
      shifted_t<HALO>(g, z, mu, 1, w);
      M3 t = m3_mul(LDG(D, y, mu), LDG(G, w, nu));
      m3_scale(t, c2);
      m3_mac_na(s, t, LDG(G, xpm, mu));
      m3_mac_an(acc, LDG(G, y, nu), s);
    }
    __builtin_amdgcn_sched_barrier(0);
    {   // R3 + R6 = U_mu(x-mu)^+ [U_nu(x-mu) D_mu(x-mu+nu) U_nu(x+mu)^+ + U_nu(x-mu-nu)^+ D_mu(x-mu-nu) U_nu(x+mu-nu)]
      shifted_t<HALO>(g, xmm, nu, 1, y);
      M3 t = m3_mul(LDG(G, xmm, nu), LDG(D, y, mu));
      M3 s = m3_mul_na(t, LDG(G, xpm, nu));
      shifted_t<HALO>(g, xmm, nu, -1, y);
      t = m3_mul_an(LDG(G, y, nu), LDG(D, y, mu));
      shifted_t<HALO>(g, xpm, nu, -1, z);
      m3_mac(s, t, LDG(G, z, nu));
      m3_scale(s, c2);
      m3_mac_an(acc, LDG(G, xmm, mu), s);
    }
    __builtin_amdgcn_sched_barrier(0);
    {   // R1 + R4 = D_nu(x) U_mu(x+2nu) D_nu(x+mu)^+ + D_nu(x-2nu)^+ U_mu(x-2nu) D_nu(x+mu-2nu)
      shifted_t<HALO>(g, x, nu, 1, y);
      shifted_t<HALO>(g, y, nu, 1, z);
      M3 t = m3_mul(LDG(D, x, nu), LDG(G, z, mu));
      m3_scale(t, c2);
      m3_mac_na(acc, t, LDG(D, xpm, nu));
      shifted_t<HALO>(g, x, nu, -1, y);
      shifted_t<HALO>(g, y, nu, -1, z);
      shifted_t<HALO>(g, z, mu, 1, w);
      t = m3_mul_an(LDG(D, z, nu), LDG(G, z, mu));
      m3_scale(t, c2);
      m3_mac(acc, t, LDG(D, w, nu));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#undef LDG
  if (raw) {   // the derivative itself (gaugeActionDeriv / gaugeForceCust), no projection
    m3_store(F + o, 64, acc);
    return;
  }
  force_finish<CLOSED>(m3_load(G + o, 64), acc, true, o, F, 1.0, Pm, cf, cpm, Uout, 0);
}

// gaugeForceCust / forceACust of the fork (stagg_pv_hmc/staghmc_spv_gforce.nim:17-253) = the action's
// derivative without the projection, on arbitrary device gauge fields in the natural layout
int gauge_deriv_dev(qexhip_ctx *c, const double2 *G, double2 *F, double cplaq, double c2, int kind) {
  if (kind == 0 && c2 != 0.0) for (int d = 0; d < 4; d++) if (c->g.X[d] < 4) { qexhip_set_error("rectangle action needs extents >= 4"); return -1; }
  const double k2 = kind == 0 ? c2 / 3.0 : 2.0 * c2 / 9.0;
  ScopedTimer tm(c, "staple", c->stream);
  const int *order = nullptr; int chunk = 0;
  CHK(tile_order_table(c, &order, &chunk));
  if (kind == 0 && c2 != 0.0 && c->gn) {
    // rectangle action: the shared-factor kernel on the double links of G (the context's D2 buffer has the layout of any natural field)
    if (!c->gn->D2) HIPCHK(hipMalloc((void **)&c->gn->D2, c->gn->n2 * sizeof(double2)));
    if (c->g.halo) {
      k_double_links<true><<<2 * c->g.etile, 256, 0, c->stream>>>(c->g, G, c->gn->D2);
      k_force_rect<false, true><<<8 * chunk, 256, 0, c->stream>>>(c->g, G, c->gn->D2, F, cplaq / 3.0, k2, nullptr, 0.0, 0.0, 1, nullptr, order, chunk);
    } else {
      k_double_links<false><<<2 * c->g.ntile, 256, 0, c->stream>>>(c->g, G, c->gn->D2);
      k_force_rect<false, false><<<8 * chunk, 256, 0, c->stream>>>(c->g, G, c->gn->D2, F, cplaq / 3.0, k2, nullptr, 0.0, 0.0, 1, nullptr, order, chunk);
    }
  } else {
    k_force_gen<false><<<8 * chunk, 256, 0, c->stream>>>(c->g, G, F, cplaq / 3.0, k2, kind, nullptr, 0.0, 0.0, 1, nullptr, order, chunk);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

static int gn_alloc(qexhip_ctx *c) {
  if (c->gn) return 0;
  c->gn = new GaugeNat();
  c->gn->n2 = (size_t)2 * c->g.etile * 4 * 576;
  size_t bytes = c->gn->n2 * sizeof(double2);
  HIPCHK(hipMalloc((void **)&c->gn->U, bytes));
  HIPCHK(hipMemsetAsync(c->gn->U, 0, bytes, c->stream));
  return 0;
}
static int gn_alloc_fp(qexhip_ctx *c) {
  size_t bytes = c->gn->n2 * sizeof(double2);
  if (!c->gn->F) { HIPCHK(hipMalloc((void **)&c->gn->F, bytes)); HIPCHK(hipMemsetAsync(c->gn->F, 0, bytes, c->stream)); }
  if (!c->gn->P) { HIPCHK(hipMalloc((void **)&c->gn->P, bytes)); HIPCHK(hipMemsetAsync(c->gn->P, 0, bytes, c->stream)); }
  return 0;
}
// refresh the ghost slices of the resident links to at least `depth` (1..3); no-op unless t is sharded
static int gauge_ghosts(qexhip_ctx *c, int depth) {
  const Geom &g = c->g;
  if (!g.halo || c->gn->ghost_valid >= depth) return 0;
  if (g.X[3] < depth) { qexhip_set_error("local t extent %d < ghost depth %d needed by this kernel", g.X[3], depth); return -1; }
  const size_t tile2 = (size_t)4 * 576 * 2;                 // doubles per link-tile
  const size_t ft = (size_t)g.F / 64;                       // tiles per t-slice
  double *bottom[2], *top[2], *ghi[2], *glo[2];
  for (int p = 0; p < 2; p++) {
    double *base = (double *)c->gn->U + (size_t)p * g.etile * tile2;
    bottom[p] = base;
    top[p] = base + ((size_t)g.ntile - depth * ft) * tile2;
    ghi[p] = base + (size_t)g.ntile * tile2;
    glo[p] = base + ((size_t)g.ntile + 3 * ft + (3 - depth) * ft) * tile2;
  }
  ScopedTimer tm(c, "gauge_halo", c->stream);
  CHK(comm_faces_exchange(c, 2, bottom, top, ghi, glo, (size_t)depth * ft * tile2));
  c->gn->ghost_valid = depth;
  return 0;
}
// rank-sum of n device scalars, then read back
static int read_global(qexhip_ctx *c, double *dev, int n, double *host) {
  if (multi_rank(c)) CHK(comm_allreduce(c, dev, n));
  return read_scalars(c, dev, n, host);
}
static inline int ghost_depth_for(double c2, int kind) { return (kind == 0 && c2 != 0.0) ? 2 : 1; }

const double2 *gauge_links_dev(qexhip_ctx *c) { return c->gn ? c->gn->U : nullptr; }

// scratch of the gauge sector that can be rebuilt on demand: the double links of the rectangle force (one more field the size
// of the links; qexhip_release_workspace)
void gauge_release_scratch(qexhip_ctx *c) {
  if (c->gn && c->gn->D2) { (void)hipFree(c->gn->D2); c->gn->D2 = nullptr; }
}
void gauge_free(qexhip_ctx *c) {
  if (!c->gn) return;
  if (c->gn->U) (void)hipFree(c->gn->U);
  if (c->gn->F) (void)hipFree(c->gn->F);
  if (c->gn->P) (void)hipFree(c->gn->P);
  if (c->gn->U2) (void)hipFree(c->gn->U2);
  if (c->gn->D2) (void)hipFree(c->gn->D2);
  if (c->gn->pp) (void)hipFree(c->gn->pp);
  if (c->gn->M) (void)hipFree(c->gn->M);
  if (c->gn->Usave) (void)hipFree(c->gn->Usave);
  delete c->gn;
  c->gn = nullptr;
}

int gauge_set(qexhip_ctx *c, const double *g) {
  HIPCHK(hipSetDevice(c->device));
  CHK(gn_alloc(c));
  size_t bytes = (size_t)c->g.V * 72 * sizeof(double);
  CHK(ensure_stage(c, bytes));
  HIPCHK(hipMemcpyAsync(c->stage, g, bytes, hipMemcpyHostToDevice, c->stream));
  k_gauge_to_tiles<<<(c->g.V + 255) / 256, 256, 0, c->stream>>>(c->g, (const double2 *)c->stage, c->gn->U);
  HIPCHK(hipGetLastError());
  c->gn->ghost_valid = 0;
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}
static int download_nat(qexhip_ctx *c, const double2 *G, double *host) {
  size_t bytes = (size_t)c->g.V * 72 * sizeof(double);
  CHK(ensure_stage(c, bytes));
  k_gauge_from_tiles<<<(c->g.V + 255) / 256, 256, 0, c->stream>>>(c->g, (double2 *)c->stage, G);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(host, c->stage, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}
int gauge_get(qexhip_ctx *c, double *g) {
  if (!c->gn) { qexhip_set_error("gauge field not set (qexhip_gauge_set)"); return -3; }
  return download_nat(c, c->gn->U, g);
}

// one lane per site in the blocked visiting order, 4 table slots per workgroup; `part` holds 6 doubles per workgroup
static int ordered_sites(qexhip_ctx *c, const int **order, int *chunk, int *nb, double **part) {
  CHK(tile_order_table(c, order, chunk));
  *nb = 8 * ((*chunk + 3) / 4);
  const int need = 9 * 8 * *chunk;                      // six plaquette sums and / or three observables per tile-workgroup
  if (c->gn->npp < need) {
    if (c->gn->pp) (void)hipFree(c->gn->pp);
    c->gn->pp = nullptr; c->gn->npp = 0;
    HIPCHK(hipMalloc((void **)&c->gn->pp, sizeof(double) * need));
    c->gn->npp = need;
  }
  *part = c->gn->pp;
  return 0;
}
int gauge_plaq(qexhip_ctx *c, double out[6]) {
  if (!c->gn) { qexhip_set_error("gauge field not set (qexhip_gauge_set)"); return -3; }
  CHK(gauge_ghosts(c, 1));
  const int *order = nullptr; int chunk = 0, nb = 0;
  double *part = nullptr;
  CHK(ordered_sites(c, &order, &chunk, &nb, &part));
  {
    ScopedTimer tm(c, "plaq", c->stream);
    if (c->g.halo) k_plaq<true><<<nb, 256, 0, c->stream>>>(c->g, c->gn->U, part, order, chunk);
    else k_plaq<false><<<nb, 256, 0, c->stream>>>(c->g, c->gn->U, part, order, chunk);
    HIPCHK(hipGetLastError());
  }
  // pl[i]/(physVol*np*nc)  (gaugeUtils.nim:277); rankSum before the normalisation (:275-279)
  k_plaq_final<<<6, 256, 0, c->stream>>>(part, nb, 1.0, &c->dscal[16]);
  HIPCHK(hipGetLastError());
  CHK(read_global(c, &c->dscal[16], 6, out));
  const double norm = (double)c->g.V * (double)c->nranks * 18.0;
  for (int k = 0; k < 6; k++) out[k] = out[k] / norm;
  return 0;
}

// s4_gauge (stagg_pv_hmc/staghmc_spv_meas.nim:25-65): out[2 d + eo], normalised by physVol * 0.5 * (nd - 1) * nc
int gauge_plaq_s4(qexhip_ctx *c, double out[8]) {
  if (!c->gn) { qexhip_set_error("gauge field not set (qexhip_gauge_set)"); return -3; }
  CHK(gauge_ghosts(c, 1));
  const int *order = nullptr; int chunk = 0, nb = 0;
  double *part = nullptr;
  CHK(ordered_sites(c, &order, &chunk, &nb, &part));
  {
    ScopedTimer tm(c, "plaq", c->stream);
    if (c->g.halo) k_plaq<true, true><<<nb, 256, 0, c->stream>>>(c->g, c->gn->U, part, order, chunk);
    else k_plaq<false, true><<<nb, 256, 0, c->stream>>>(c->g, c->gn->U, part, order, chunk);
    HIPCHK(hipGetLastError());
  }
  k_plaq_final<<<8, 256, 0, c->stream>>>(part, nb, 1.0, &c->dscal[40]);
  HIPCHK(hipGetLastError());
  CHK(read_global(c, &c->dscal[40], 8, out));
  const double norm = (double)c->g.V * (double)c->nranks * 0.5 * 3.0 * 3.0;
  for (int k = 0; k < 8; k++) out[k] = out[k] / norm;
  return 0;
}

static int force_dev(qexhip_ctx *c, double cplaq, int flow = 0, double cf = 0, double cpm = 0, double c2 = 0, int kind = 0,
                     double2 *Uout = nullptr) {
  CHK(gn_alloc_fp(c));
  CHK(gauge_ghosts(c, ghost_depth_for(c2, kind)));
  ScopedTimer tm(c, "staple", c->stream);
  const int *order = nullptr; int chunk = 0;
  CHK(tile_order_table(c, &order, &chunk));
  const int nb = 8 * chunk;
  const bool closed = Uout && c->opt_flow_exp;
  if (c2 != 0.0) {
    // kind 0: cr = c.rect/nc ; kind 1: ca = 2 c.adjplaq/nc^2
    const double k2 = kind == 0 ? c2 / 3.0 : 2.0 * c2 / 9.0;
    if (kind == 0) {
      // rectangle action: double links once per call, then the shared-factor kernel
      if (!c->gn->D2) HIPCHK(hipMalloc((void **)&c->gn->D2, c->gn->n2 * sizeof(double2)));
      if (c->g.halo) k_double_links<true><<<2 * c->g.etile, 256, 0, c->stream>>>(c->g, c->gn->U, c->gn->D2);
      else k_double_links<false><<<2 * c->g.ntile, 256, 0, c->stream>>>(c->g, c->gn->U, c->gn->D2);
      double2 *Pf = (flow || Uout) ? c->gn->P : nullptr;
#define QX_FRECT(CL, HL) k_force_rect<CL, HL><<<nb, 256, 0, c->stream>>>(c->g, c->gn->U, c->gn->D2, c->gn->F, cplaq / 3.0, k2, Pf, cf, cpm, 0, Uout, order, chunk)
      if (closed) { if (c->g.halo) QX_FRECT(true, true); else QX_FRECT(true, false); }
      else { if (c->g.halo) QX_FRECT(false, true); else QX_FRECT(false, false); }
#undef QX_FRECT
    } else if (closed)
      k_force_gen<true><<<nb, 256, 0, c->stream>>>(c->g, c->gn->U, c->gn->F, cplaq / 3.0, k2, kind, c->gn->P, cf, cpm, 0, Uout, order, chunk);
    else
      k_force_gen<false><<<nb, 256, 0, c->stream>>>(c->g, c->gn->U, c->gn->F, cplaq / 3.0, k2, kind,
                                                    flow ? c->gn->P : nullptr, cf, cpm, 0, Uout, order, chunk);
  } else {
    // (capping the registers for 3 or 4 waves/SIMD spills: 1460 / 2370 us against 1310 us fused at 2 waves/SIMD)
    const size_t shb = (size_t)8 * 576 * sizeof(double2);         // 72 KiB per parity of a tile position (qexhip_init checks the device offers 144)
    double2 *Pf = (closed || flow) ? c->gn->P : nullptr;
    if (c->opt_force_pair && c->tile_pairs_ok) {
      // both parities of a tile position per workgroup (k_force_lds2): 144 KiB of LDS, one workgroup of 8 wavefronts per CU
      if (!(c->lds_attr_done & 8)) {
        HIPCHK(hipFuncSetAttribute((const void *)k_force_lds2<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * shb)));
        HIPCHK(hipFuncSetAttribute((const void *)k_force_lds2<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * shb)));
        HIPCHK(hipFuncSetAttribute((const void *)k_force_lds2<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * shb)));
        HIPCHK(hipFuncSetAttribute((const void *)k_force_lds2<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * shb)));
        c->lds_attr_done |= 8;
      }
#define QX_FLDS2(CL, HL) k_force_lds2<CL, HL><<<nb / 2, 512, 2 * shb, c->stream>>>(c->g, c->gn->U, c->gn->F, cplaq / 3.0, Pf, cf, cpm, Uout, order, chunk)
      if (closed) { if (c->g.halo) QX_FLDS2(true, true); else QX_FLDS2(true, false); }
      else { if (c->g.halo) QX_FLDS2(false, true); else QX_FLDS2(false, false); }
#undef QX_FLDS2
    } else {
      // lattice shapes whose visiting order cannot pair the parities of a tile position: one (tile, parity) per workgroup
      if (!(c->lds_attr_done & 1)) {
        HIPCHK(hipFuncSetAttribute((const void *)k_force_lds<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
        HIPCHK(hipFuncSetAttribute((const void *)k_force_lds<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
        HIPCHK(hipFuncSetAttribute((const void *)k_force_lds<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
        HIPCHK(hipFuncSetAttribute((const void *)k_force_lds<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
        c->lds_attr_done |= 1;
      }
#define QX_FLDS(CL, HL) k_force_lds<CL, HL><<<nb, 256, shb, c->stream>>>(c->g, c->gn->U, c->gn->F, cplaq / 3.0, Pf, cf, cpm, Uout, order, chunk)
      if (closed) { if (c->g.halo) QX_FLDS(true, true); else QX_FLDS(true, false); }
      else { if (c->g.halo) QX_FLDS(false, true); else QX_FLDS(false, false); }
#undef QX_FLDS
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}
int gauge_force(qexhip_ctx *c, double *f_host, double cplaq, double c2, int kind) {
  if (!c->gn) { qexhip_set_error("gauge field not set (qexhip_gauge_set)"); return -3; }
  if (kind == 0 && c2 != 0.0) for (int d = 0; d < 4; d++) if (c->g.X[d] < 4) { qexhip_set_error("rectangle action needs extents >= 4"); return -1; }
  CHK(force_dev(c, cplaq, 0, 0, 0, c2, kind));
  return download_nat(c, c->gn->F, f_host);
}

int gauge_wflow(qexhip_ctx *c, int nsteps, double eps, double cplaq, double c2, int kind) {
  if (!c->gn) { qexhip_set_error("gauge field not set (qexhip_gauge_set)"); return -3; }
  if (kind == 0 && c2 != 0.0) for (int d = 0; d < 4; d++) if (c->g.X[d] < 4) { qexhip_set_error("rectangle action needs extents >= 4"); return -1; }
  CHK(gn_alloc_fp(c));
  const double epsnc = eps * 3.0;
  const double cf[3] = {(-1.0 / 4.0) * epsnc, (-8.0 / 9.0) * epsnc, (-3.0 / 4.0) * epsnc};
  const double cpm[3] = {0.0, -17.0 / 9.0, -1.0};
  // one kernel per RK3 stage (staples -> v -> U' = exp(v) U into the second link buffer), then the buffers swap
  if (!c->gn->U2) {
    HIPCHK(hipMalloc((void **)&c->gn->U2, c->gn->n2 * sizeof(double2)));
    HIPCHK(hipMemsetAsync(c->gn->U2, 0, c->gn->n2 * sizeof(double2), c->stream));
  }
  for (int s = 0; s < nsteps; s++)
    for (int st = 0; st < 3; st++) {
      CHK(force_dev(c, cplaq, 1, cf[st], cpm[st], c2, kind, c->gn->U2));
      std::swap(c->gn->U, c->gn->U2);
      c->gn->ghost_valid = 0;
    }
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// ---------------- MD building blocks on the resident gauge field (the HMC examples' mdt, reunit, gaction, ploop) ----
// gaugeAction1 / actionA sums (src/gauge/gaugeAction.nim:61-142,614-681): sum ReTr P, sum |tr P|^2, sum ReTr R
__global__ void __launch_bounds__(256) k_action(Geom g, const double2 *__restrict__ G, int rect, double *partials) {
  double sp = 0, sa = 0, sr = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < g.V; i += gridDim.x * 256) {
    int p = i >= g.Vh, c = i - p * g.Vh;
    int x[4];
    coords_of(g, c, p, x);
#pragma unroll 1
    for (int mu = 1; mu < 4; mu++)
#pragma unroll 1
      for (int nu = 0; nu < mu; nu++) {
        ObsPath P;
        P.len = 4; P.coef = 1.0;
        P.step[0] = 1; P.step[1] = 2; P.step[2] = -1; P.step[3] = -2;
        M3 m = path_prod(g, G, x, P, mu, nu);
        const double tr = m.e[0].x + m.e[4].x + m.e[8].x, ti = m.e[0].y + m.e[4].y + m.e[8].y;
        sp += tr;
        sa += tr * tr + ti * ti;
        if (rect) {
          P.len = 6;
          P.step[0] = 1; P.step[1] = 1; P.step[2] = 2; P.step[3] = -1; P.step[4] = -1; P.step[5] = -2;
          m = path_prod(g, G, x, P, mu, nu);
          sr += m.e[0].x + m.e[4].x + m.e[8].x;
          P.step[0] = 1; P.step[1] = 2; P.step[2] = 2; P.step[3] = -1; P.step[4] = -2; P.step[5] = -2;
          m = path_prod(g, G, x, P, mu, nu);
          sr += m.e[0].x + m.e[4].x + m.e[8].x;
        }
      }
  }
  double r;
  r = block_sum_256(sp); if (threadIdx.x == 0) partials[blockIdx.x] = r;
  r = block_sum_256(sa); if (threadIdx.x == 0) partials[gridDim.x + blockIdx.x] = r;
  r = block_sum_256(sr); if (threadIdx.x == 0) partials[2 * gridDim.x + blockIdx.x] = r;
}
__global__ void __launch_bounds__(256) k_sum3(const double *partials, int nb, double *out) {
  for (int k = 0; k < 3; k++) {
    double acc = 0;
    for (int i = threadIdx.x; i < nb; i += 256) acc += partials[(size_t)k * nb + i];
    double r = block_sum_256(acc);
    if (threadIdx.x == 0) out[k] = r;
  }
}
int gauge_action(qexhip_ctx *c, double cplaq, double c2, int kind, double *out) {
  if (!c->gn) { qexhip_set_error("gauge field not set (qexhip_gauge_set)"); return -3; }
  const int rect = (kind == 0 && c2 != 0.0);
  if (rect) for (int d = 0; d < 4; d++) if (c->g.X[d] < 4) { qexhip_set_error("rectangle action needs extents >= 4"); return -1; }
  CHK(gauge_ghosts(c, rect ? 2 : 1));
  int nb = (c->g.V + 255) / 256;
  if (nb > 1024) nb = 1024;
  k_action<<<nb, 256, 0, c->stream>>>(c->g, c->gn->U, rect, c->partials);
  k_sum3<<<1, 256, 0, c->stream>>>(c->partials, nb, &c->dscal[24]);
  HIPCHK(hipGetLastError());
  double s[3];
  CHK(read_global(c, &c->dscal[24], 3, s));
  if (kind == 0) *out = (-1.0 / 3.0) * (cplaq * s[0] + c2 * s[2]);
  else {
    const double a0 = 0.5 * 12.0 * (double)c->g.V * (double)c->nranks;
    *out = cplaq * (a0 - s[0] / 3.0) + c2 * (a0 - s[1] / 9.0);
  }
  return 0;
}
// mdt (src/examples/staghmc_sh.nim:429-435): U <- exp(t p) U on the resident field
int gauge_md_update(qexhip_ctx *c, const double *p_host, double t) {
  if (!c->gn) { qexhip_set_error("gauge field not set (qexhip_gauge_set)"); return -3; }
  CHK(gn_alloc_fp(c));
  const size_t bytes = (size_t)c->g.V * 72 * sizeof(double);
  CHK(ensure_stage(c, bytes));
  HIPCHK(hipMemcpyAsync(c->stage, p_host, bytes, hipMemcpyHostToDevice, c->stream));
  k_gauge_to_tiles<<<(c->g.V + 255) / 256, 256, 0, c->stream>>>(c->g, (const double2 *)c->stage, c->gn->P);
  const size_t ltiles = (size_t)2 * c->g.ntile * 4;
  ScopedTimer tm(c, "expupdate", c->stream);
  k_exp_update<<<(unsigned)((ltiles * 64 + 255) / 256), 256, 0, c->stream>>>(ltiles, c->gn->U, c->gn->P, t, (size_t)c->g.ntile * 4, (size_t)c->g.etile * 4);
  HIPCHK(hipGetLastError());
  c->gn->ghost_valid = 0;
  return 0;
}
// ---------------- resident molecular dynamics ----------------
// The MD loop of QEX's HMC drivers (mdt / mdv / mdvAllfga, src/examples/staghmc_sh.nim:429-640,
// src/stagg_pv_hmc/staghmc_spv.nim:873-1061) with links and momenta left on the device between the updates: through
// the host-pointer entry points every update moves two or three 72-double-per-site fields over PCIe, which at 32^4 is
// 8 of the 9.4 s of a trajectory.  Forces are left in a device buffer ("source": 0 = md_gauge_force, 1 = the nHYP
// closure's last force) and applied by md_kick (p += t f) or md_shift_links (U <- exp(t f) U, the force-gradient shift).
__global__ void __launch_bounds__(256) k_links_axpy(size_t nlinks_tiles, double2 *P, const double2 *__restrict__ F, double t, size_t ntile4, size_t etile4) {
  size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t tile = j >> 6;
  if (tile >= nlinks_tiles) return;
  size_t o = body_tile_off(tile, ntile4, etile4) + (j & 63);
#pragma unroll
  for (int k = 0; k < 9; k++) {
    double2 a = P[o + k * 64];
    const double2 f = F[o + k * 64];
    a.x = fma(t, f.x, a.x); a.y = fma(t, f.y, a.y);
    P[o + k * 64] = a;
  }
}
__global__ void __launch_bounds__(256) k_links_norm2(size_t nlinks_tiles, const double2 *__restrict__ P, size_t ntile4, size_t etile4, double *partials) {
  double acc = 0;
  for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; (j >> 6) < nlinks_tiles; j += (size_t)gridDim.x * 256) {
    size_t o = body_tile_off(j >> 6, ntile4, etile4) + (j & 63);
#pragma unroll
    for (int k = 0; k < 9; k++) { const double2 a = P[o + k * 64]; acc = fma(a.x, a.x, fma(a.y, a.y, acc)); }
  }
  double r = block_sum_256(acc);
  if (threadIdx.x == 0) { partials[blockIdx.x] = r; partials[gridDim.x + blockIdx.x] = 0; partials[2 * gridDim.x + blockIdx.x] = 0; }
}
double2 *nhyp_force_buffer(qexhip_ctx *c);      // smear.hip: the closure's force field (null without a closure)
const double2 *gauge_resident_links(qexhip_ctx *c) { return c->gn ? c->gn->U : nullptr; }
static int md_source(qexhip_ctx *c, int source, const double2 **F) {
  *F = nullptr;
  if (source == 0) *F = c->gn ? c->gn->F : nullptr;
  else if (source == 1) *F = nhyp_force_buffer(c);
  if (!*F) { qexhip_set_error("md: force source %d holds nothing yet", source); return -3; }
  return 0;
}
static int md_ready(qexhip_ctx *c) {
  if (!c->gn || !c->gn->M) { qexhip_set_error("md: call qexhip_md_begin first"); return -3; }
  return 0;
}
static unsigned link_blocks(qexhip_ctx *c) { return (unsigned)(((size_t)2 * c->g.ntile * 4 * 64 + 255) / 256); }
int md_begin(qexhip_ctx *c, const double *g, const double *p) {
  if (g) CHK(gauge_set(c, g));
  if (!c->gn) { qexhip_set_error("md_begin: no resident gauge field (pass g or call qexhip_gauge_set)"); return -3; }
  CHK(gn_alloc_fp(c));
  const size_t nbytes = c->gn->n2 * sizeof(double2);
  if (!c->gn->M) { HIPCHK(hipMalloc((void **)&c->gn->M, nbytes)); HIPCHK(hipMemsetAsync(c->gn->M, 0, nbytes, c->stream)); }
  if (p) {          // p == NULL: keep the resident momenta (qexhip_md_refresh_momenta draws them on the device)
    const size_t bytes = (size_t)c->g.V * 72 * sizeof(double);
    CHK(ensure_stage(c, bytes));
    HIPCHK(hipMemcpyAsync(c->stage, p, bytes, hipMemcpyHostToDevice, c->stream));
    k_gauge_to_tiles<<<(c->g.V + 255) / 256, 256, 0, c->stream>>>(c->g, (const double2 *)c->stage, c->gn->M);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}
int md_momenta_dev(qexhip_ctx *c, double2 **M) {
  if (!c->gn) { qexhip_set_error("no resident gauge field (qexhip_gauge_set / qexhip_md_begin)"); return -3; }
  const size_t nbytes = c->gn->n2 * sizeof(double2);
  if (!c->gn->M) { HIPCHK(hipMalloc((void **)&c->gn->M, nbytes)); HIPCHK(hipMemsetAsync(c->gn->M, 0, nbytes, c->stream)); }
  *M = c->gn->M;
  return 0;
}
int md_end(qexhip_ctx *c, double *g, double *p) {
  CHK(md_ready(c));
  if (g) CHK(download_nat(c, c->gn->U, g));
  if (p) CHK(download_nat(c, c->gn->M, p));
  return 0;
}
int md_momentum_norm2(qexhip_ctx *c, double *out) {
  CHK(md_ready(c));
  const size_t ltiles = (size_t)2 * c->g.ntile * 4;
  const int nb = (int)std::min<size_t>(1024, (ltiles * 64 + 255) / 256);
  k_links_norm2<<<nb, 256, 0, c->stream>>>(ltiles, c->gn->M, (size_t)c->g.ntile * 4, (size_t)c->g.etile * 4, c->partials);
  k_sum3<<<1, 256, 0, c->stream>>>(c->partials, nb, &c->dscal[24]);
  HIPCHK(hipGetLastError());
  double s[3];
  CHK(read_global(c, &c->dscal[24], 3, s));
  *out = s[0];
  return 0;
}
int md_update_links(qexhip_ctx *c, double t) {
  CHK(md_ready(c));
  const size_t ltiles = (size_t)2 * c->g.ntile * 4;
  ScopedTimer tm(c, "expupdate", c->stream);
  k_exp_update<<<link_blocks(c), 256, 0, c->stream>>>(ltiles, c->gn->U, c->gn->M, t, (size_t)c->g.ntile * 4, (size_t)c->g.etile * 4);
  HIPCHK(hipGetLastError());
  c->gn->ghost_valid = 0;
  return 0;
}
int md_gauge_force(qexhip_ctx *c, double cplaq, double c2, int kind) {
  if (!c->gn) { qexhip_set_error("gauge field not set (qexhip_gauge_set)"); return -3; }
  if (kind == 0 && c2 != 0.0) for (int d = 0; d < 4; d++) if (c->g.X[d] < 4) { qexhip_set_error("rectangle action needs extents >= 4"); return -1; }
  return force_dev(c, cplaq, 0, 0, 0, c2, kind);
}
int md_kick(qexhip_ctx *c, int source, double t) {
  CHK(md_ready(c));
  const double2 *F;
  CHK(md_source(c, source, &F));
  const size_t ltiles = (size_t)2 * c->g.ntile * 4;
  k_links_axpy<<<link_blocks(c), 256, 0, c->stream>>>(ltiles, c->gn->M, F, t, (size_t)c->g.ntile * 4, (size_t)c->g.etile * 4);
  HIPCHK(hipGetLastError());
  return 0;
}
int md_shift_links(qexhip_ctx *c, int source, double t) {
  if (!c->gn) { qexhip_set_error("gauge field not set (qexhip_gauge_set)"); return -3; }
  const double2 *F;
  CHK(md_source(c, source, &F));
  const size_t ltiles = (size_t)2 * c->g.ntile * 4;
  ScopedTimer tm(c, "expupdate", c->stream);
  k_exp_update<<<link_blocks(c), 256, 0, c->stream>>>(ltiles, c->gn->U, F, t, (size_t)c->g.ntile * 4, (size_t)c->g.etile * 4);
  HIPCHK(hipGetLastError());
  c->gn->ghost_valid = 0;
  return 0;
}
int md_save_links(qexhip_ctx *c) {
  if (!c->gn) { qexhip_set_error("gauge field not set (qexhip_gauge_set)"); return -3; }
  const size_t nbytes = c->gn->n2 * sizeof(double2);
  if (!c->gn->Usave) HIPCHK(hipMalloc((void **)&c->gn->Usave, nbytes));
  HIPCHK(hipMemcpyAsync(c->gn->Usave, c->gn->U, nbytes, hipMemcpyDeviceToDevice, c->stream));
  c->gn->save_ghost_valid = c->gn->ghost_valid;
  return 0;
}
int md_restore_links(qexhip_ctx *c) {
  if (!c->gn || !c->gn->Usave) { qexhip_set_error("md_restore_links: nothing saved"); return -3; }
  HIPCHK(hipMemcpyAsync(c->gn->U, c->gn->Usave, c->gn->n2 * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
  c->gn->ghost_valid = c->gn->save_ghost_valid;
  return 0;
}
// reunit: g.projectSU (gaugeUtils.nim:1333-1334)
__global__ void __launch_bounds__(256) k_reunit(size_t nlinks_tiles, double2 *G, size_t ntile4, size_t etile4) {
  size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t tile = j >> 6;
  if (tile >= nlinks_tiles) return;
  size_t o = body_tile_off(tile, ntile4, etile4) + (j & 63);
  m3_store(G + o, 64, m3_projectSU(m3_load(G + o, 64)));
}
int gauge_reunit(qexhip_ctx *c) {
  if (!c->gn) { qexhip_set_error("gauge field not set (qexhip_gauge_set)"); return -3; }
  const size_t ltiles = (size_t)2 * c->g.ntile * 4;
  k_reunit<<<(unsigned)((ltiles * 64 + 255) / 256), 256, 0, c->stream>>>(ltiles, c->gn->U, (size_t)c->g.ntile * 4, (size_t)c->g.etile * 4);
  HIPCHK(hipGetLastError());
  c->gn->ghost_valid = 0;
  return 0;
}
// wline (gaugeUtils.nim:1079-1112): volume- and colour-averaged trace of a path product; path entries
// +-(mu+1), any length (Polyakov loops: [mu+1] * L_mu, src/examples/staghmc_sh.nim:281-291)
__global__ void __launch_bounds__(256) k_wline(Geom g, const double2 *__restrict__ G, const int *path, int n, double *partials) {
  double sr = 0, si = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < g.V; i += gridDim.x * 256) {
    int p = i >= g.Vh, c = i - p * g.Vh;
    int x[4];
    coords_of(g, c, p, x);
    M3 m;
    for (int k = 0; k < n; k++) {
      const int s = path[k];
      const int d = (s > 0 ? s : -s) - 1;
      if (s > 0) {
        M3 u = m3_load(G + link_off(g, x, d), 64);
        x[d] = (g.halo && d == 3) ? x[d] + 1 : (x[d] + 1 >= g.X[d] ? 0 : x[d] + 1);
        m = k == 0 ? u : m3_mul(m, u);
      } else {
        x[d] = (g.halo && d == 3) ? x[d] - 1 : (x[d] == 0 ? g.X[d] - 1 : x[d] - 1);
        M3 u = m3_load(G + link_off(g, x, d), 64);
        if (k == 0) { m = m3_zero(); m3_add_diag(m, 1.0); }
        m = m3_mul_na(m, u);
      }
    }
    sr += m.e[0].x + m.e[4].x + m.e[8].x;
    si += m.e[0].y + m.e[4].y + m.e[8].y;
  }
  double r;
  r = block_sum_256(sr); if (threadIdx.x == 0) partials[blockIdx.x] = r;
  r = block_sum_256(si); if (threadIdx.x == 0) partials[gridDim.x + blockIdx.x] = r;
  if (threadIdx.x == 0) partials[2 * gridDim.x + blockIdx.x] = 0.0;
}
// t-sharded Polyakov line: every rank multiplies its own t-links, M_r(xs) = prod_{t local} U_t(xs, t) for the spatial
// sites xs (both parities of the local t = 0 slice), the segments are all-gathered in rank order and every rank forms
// tr(M_0 M_1 ... M_{N-1}) -- the trace is cyclic, so the line through (xs, t) has the same trace for every t.
__global__ void __launch_bounds__(256) k_tline_segment(Geom g, const double2 *__restrict__ G, double2 *M) {
  const int j = blockIdx.x * 256 + threadIdx.x;            // (parity, site of the t = 0 slice)
  if (j >= 2 * g.F) return;
  const int p = j >= g.F, c = j - p * g.F;
  int x[4];
  coords_of(g, c, p, x);
  M3 m = m3_load(G + link_off(g, x, 3), 64);
  for (int t = 1; t < g.X[3]; t++) {
    x[3] = t;
    m = m3_mul(m, m3_load(G + link_off(g, x, 3), 64));
  }
  for (int k = 0; k < 9; k++) M[(size_t)j * 9 + k] = m.e[k];
}
__global__ void __launch_bounds__(256) k_tline_trace(int nsites, int nranks, const double2 *__restrict__ M, double *partials) {
  double sr = 0, si = 0;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < nsites; j += gridDim.x * 256) {
    M3 m;
    for (int k = 0; k < 9; k++) m.e[k] = M[(size_t)j * 9 + k];
    for (int r = 1; r < nranks; r++) {
      M3 u;
      for (int k = 0; k < 9; k++) u.e[k] = M[((size_t)r * nsites + j) * 9 + k];
      m = m3_mul(m, u);
    }
    sr += m.e[0].x + m.e[4].x + m.e[8].x;
    si += m.e[0].y + m.e[4].y + m.e[8].y;
  }
  double r;
  r = block_sum_256(sr); if (threadIdx.x == 0) partials[blockIdx.x] = r;
  r = block_sum_256(si); if (threadIdx.x == 0) partials[gridDim.x + blockIdx.x] = r;
  if (threadIdx.x == 0) partials[2 * gridDim.x + blockIdx.x] = 0.0;
}

// Straight line that closes on itself through the (local) lattice, direction d < 3 or an unsharded t: the Polyakov loop
// (src/flow/gauge_flow.nim:137-156, staghmc_sh.nim:281-291).  The trace is cyclic, so every site of a line carries the same
// value: ONE lane per line multiplies the X[d] links of its line (the field is read once, V / X[d] products of X[d] links
// instead of V) and the mean over the lines is the mean over the sites the reference takes (gaugeUtils.nim:1079-1112).
// dsel >= 0: that direction, partials at `partials`; dsel = -1 - d0: direction d = d0 + blockIdx.y, partials at partials + 1536 d
// (gauge_polyakov: the four directions side by side in one launch -- a line is a chain of X[d] dependent products and
// V / X[d] lanes do not fill the chip, so the directions overlap instead of queueing)
__global__ void __launch_bounds__(256) k_line_trace(Geom g, const double2 *__restrict__ G, int dsel, double *partials) {
  // workgroup = 64 lines x 4 segments: wavefront k multiplies the k-th quarter of its 64 lines (a chain of X[d] / 4
  // dependent products instead of X[d]), wavefronts 1-3 hand their segment products to wavefront 0 through LDS
  __shared__ double2 seg[3][9][64];
  const int d = dsel >= 0 ? dsel : (int)blockIdx.y - dsel - 1;     // dsel = -1 - d0: directions d0 + blockIdx.y
  if (dsel < 0) partials += (size_t)1536 * d;
  double sr = 0, si = 0;
  const int Xd = d == 0 ? g.X[0] : (d == 1 ? g.X[1] : (d == 2 ? g.X[2] : g.X[3]));
  const int nl = g.V / Xd;
  const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int L = (Xd + 3) / 4, t0 = k * L, t1 = min(Xd, t0 + L);
  for (int j0 = blockIdx.x * 64; j0 < nl; j0 += gridDim.x * 64) {
    const int j = j0 + lane;
    M3 m = m3_zero();
    m3_add_diag(m, 1.0);
    if (j < nl && t0 < t1) {
      // the line's site in the hyperplane x[d] = 0: j runs over the other three coordinates, x fastest
      int x[4], r = j;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (q == d) { x[q] = t0; continue; }
        x[q] = r % g.X[q]; r /= g.X[q];
      }
      m = m3_load(G + link_off(g, x, d), 64);
      for (int t = t0 + 1; t < t1; t++) {
        x[0] = d == 0 ? t : x[0]; x[1] = d == 1 ? t : x[1]; x[2] = d == 2 ? t : x[2]; x[3] = d == 3 ? t : x[3];
        m = m3_mul(m, m3_load(G + link_off(g, x, d), 64));
      }
    }
    if (k > 0) {
#pragma unroll
      for (int q = 0; q < 9; q++) seg[k - 1][q][lane] = m.e[q];
    }
    __syncthreads();
    if (k == 0) {
#pragma unroll 1
      for (int q3 = 0; q3 < 3; q3++) {
        M3 o;
#pragma unroll
        for (int q = 0; q < 9; q++) o.e[q] = seg[q3][q][lane];
        m = m3_mul(m, o);
      }
      if (j < nl) {
        sr += m.e[0].x + m.e[4].x + m.e[8].x;
        si += m.e[0].y + m.e[4].y + m.e[8].y;
      }
    }
    __syncthreads();
  }
  double r;
  r = block_sum_256(sr); if (threadIdx.x == 0) partials[blockIdx.x] = r;
  r = block_sum_256(si); if (threadIdx.x == 0) partials[gridDim.x + blockIdx.x] = r;
  if (threadIdx.x == 0) partials[2 * gridDim.x + blockIdx.x] = 0.0;
}
// Lines along x: a lane per line would gather 16 B from 64 different rows per load (the other directions read whole rows).
// Here a line occupies W = 2^k consecutive lanes of a wavefront, every lane multiplies S = 4 (or 2) consecutive links of the
// line and the ordered product of the lanes is formed by a shuffle tree: lane i <- M_i M_{i+s}, s = 1, 2, 4 ... W/2 (lanes
// beyond the line hold the unit matrix): S - 1 + log2(W) products per lane instead of X[0] dependent ones per line.
__device__ __forceinline__ M3 m3_shfl_down(const M3 &a, int off) {
  M3 r;
#pragma unroll
  for (int k = 0; k < 9; k++) r.e[k] = make_double2(__shfl_down(a.e[k].x, off, 64), __shfl_down(a.e[k].y, off, 64));
  return r;
}
template <int S>   // S consecutive links per lane (sequential), then the shuffle tree over the W = 2^k >= X[0] / S lanes of a line
__global__ void __launch_bounds__(256) k_xline_trace(Geom g, const double2 *__restrict__ G, int W, double *partials) {
  double sr = 0, si = 0;
  const int nl = g.V / g.X[0];                       // lines
  const int per = 256 / W;                           // lines per workgroup
  for (int l0 = blockIdx.x * per; l0 < nl; l0 += gridDim.x * per) {
    const int j = l0 + (int)threadIdx.x / W, xs = ((int)threadIdx.x % W) * S;
    M3 m = m3_zero();
    m3_add_diag(m, 1.0);
    if (j < nl && xs < g.X[0]) {
      int x[4], r = j;
      x[0] = xs;
      x[1] = r % g.X[1]; r /= g.X[1];
      x[2] = r % g.X[2]; x[3] = r / g.X[2];
      m = m3_load(G + link_off(g, x, 0), 64);
#pragma unroll
      for (int k = 1; k < S; k++) {
        x[0] = xs + k;
        m = m3_mul(m, m3_load(G + link_off(g, x, 0), 64));
      }
    }
    for (int sft = 1; sft < W; sft <<= 1) {
      const M3 o = m3_shfl_down(m, sft);             // lanes whose partner is out of the line keep garbage nobody reads
      m = m3_mul(m, o);
    }
    if (j < nl && xs == 0) {
      sr += m.e[0].x + m.e[4].x + m.e[8].x;
      si += m.e[0].y + m.e[4].y + m.e[8].y;
    }
  }
  double r;
  r = block_sum_256(sr); if (threadIdx.x == 0) partials[blockIdx.x] = r;
  r = block_sum_256(si); if (threadIdx.x == 0) partials[gridDim.x + blockIdx.x] = r;
  if (threadIdx.x == 0) partials[2 * gridDim.x + blockIdx.x] = 0.0;
}
// the x lines into `partials` (3 x nb entries): cooperative form while a line fits a wavefront
static void launch_xlines(qexhip_ctx *c, int nb, double *partials) {
  const Geom &g = c->g;
  const int S = (g.X[0] % 4 == 0) ? 4 : 2;
  if (g.X[0] / S <= 64) {
    int W = 1;
    while (W < g.X[0] / S) W <<= 1;
    if (S == 4) k_xline_trace<4><<<nb, 256, 0, c->stream>>>(g, c->gn->U, W, partials);
    else k_xline_trace<2><<<nb, 256, 0, c->stream>>>(g, c->gn->U, W, partials);
  } else {
    k_line_trace<<<nb, 256, 0, c->stream>>>(g, c->gn->U, 0, partials);
  }
}
// group b (blockIdx.x): out[3 b + k] = sum of partials[1536 b + k nb + (0..nb)]
__global__ void __launch_bounds__(256) k_sum3_groups(const double *partials, int nb, double *out) {
  const double *pp = partials + (size_t)1536 * blockIdx.x;
  for (int k = 0; k < 3; k++) {
    double acc = 0;
    for (int i = threadIdx.x; i < nb; i += 256) acc += pp[(size_t)k * nb + i];
    double r = block_sum_256(acc);
    if (threadIdx.x == 0) out[3 * blockIdx.x + k] = r;
  }
}

int gauge_wline(qexhip_ctx *c, const int *path, int n, double out[2]) {
  if (!c->gn) { qexhip_set_error("gauge field not set (qexhip_gauge_set)"); return -3; }
  if (n < 1 || n > 4096) { qexhip_set_error("wline: path length out of range"); return -1; }
  for (int k = 0; k < n; k++) if (path[k] == 0 || path[k] > 4 || path[k] < -4) { qexhip_set_error("wline: path entries are +-(mu+1)"); return -1; }
  const Geom &g = c->g;
  int nb = (g.V + 255) / 256;
  if (nb > 1024) nb = 1024;
  double s[3];
  if (g.halo) {
    // how far the path strays in t: within the three ghost slices the generic walk works on the sharded field
    int t = 0, tmin = 0, tmax = 0;
    bool straight = true;
    for (int k = 0; k < n; k++) {
      if (path[k] == 4) t++; else if (path[k] == -4) t--;
      if (path[k] != path[0]) straight = false;
      tmin = std::min(tmin, t); tmax = std::max(tmax, t);
    }
    const int reach = std::max(-tmin, tmax);
    if (reach > 3) {
      const int Lt = g.X[3] * c->nranks;
      if (!(straight && (path[0] == 4 || path[0] == -4) && n == Lt)) {
        qexhip_set_error("wline on a t-sharded field: the path may stray at most 3 slices in t, or be the straight Polyakov line");
        return -3;
      }
      const int ns = 2 * g.F;                                   // spatial sites
      const size_t seg = (size_t)ns * 9 * 2;                    // doubles per rank
      CHK(ensure_stage(c, seg * sizeof(double) * (1 + (size_t)c->nranks)));
      double *mine = c->stage, *all = c->stage + seg;
      k_tline_segment<<<(ns + 255) / 256, 256, 0, c->stream>>>(g, c->gn->U, (double2 *)mine);
      HIPCHK(hipGetLastError());
      CHK(comm_allgather(c, mine, all, seg));
      int nbt = std::min((ns + 255) / 256, 1024);
      k_tline_trace<<<nbt, 256, 0, c->stream>>>(ns, c->nranks, (const double2 *)all, c->partials);
      k_sum3<<<1, 256, 0, c->stream>>>(c->partials, nbt, &c->dscal[24]);
      HIPCHK(hipGetLastError());
      CHK(read_scalars(c, &c->dscal[24], 3, s));               // identical on every rank: no reduction
      const double fac = 1.0 / ((double)ns * 3.0);
      out[0] = s[0] * fac;
      out[1] = (path[0] == 4 ? 1.0 : -1.0) * s[1] * fac;       // the reversed line is the adjoint
      return 0;
    }
    CHK(gauge_ghosts(c, std::max(reach, 1)));
  }
  {
    // a straight line once around the lattice in a direction this rank holds whole: one lane per line
    bool straight = true;
    for (int k = 1; k < n; k++) if (path[k] != path[0]) straight = false;
    const int d = std::abs(path[0]) - 1;
    if (straight && n == g.X[d] && !(g.halo && d == 3)) {
      const int nl = g.V / g.X[d];
      const int nbl = std::min((nl + 63) / 64, 1024);
      if (d == 0) launch_xlines(c, nbl, c->partials);
      else k_line_trace<<<nbl, 256, 0, c->stream>>>(g, c->gn->U, d, c->partials);
      k_sum3<<<1, 256, 0, c->stream>>>(c->partials, nbl, &c->dscal[24]);
      HIPCHK(hipGetLastError());
      CHK(read_global(c, &c->dscal[24], 3, s));
      const double fl = 1.0 / ((double)nl * (double)c->nranks * 3.0);
      out[0] = s[0] * fl;
      out[1] = (path[0] > 0 ? 1.0 : -1.0) * s[1] * fl;          // the reversed line is the adjoint
      return 0;
    }
  }
  CHK(ensure_stage(c, 4096 * sizeof(int)));
  HIPCHK(hipMemcpyAsync(c->stage, path, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
  k_wline<<<nb, 256, 0, c->stream>>>(g, c->gn->U, (const int *)c->stage, n, c->partials);
  k_sum3<<<1, 256, 0, c->stream>>>(c->partials, nb, &c->dscal[24]);
  HIPCHK(hipGetLastError());
  CHK(read_global(c, &c->dscal[24], 3, s));
  const double fac = 1.0 / ((double)g.V * (double)c->nranks * 3.0);
  out[0] = s[0] * fac; out[1] = s[1] * fac;
  return 0;
}

// The four Polyakov loops of a flow-loop / HMC measurement (src/flow/gauge_flow.nim:137-156 `meas_ploop`,
// src/examples/staghmc_sh.nim:281-291 `ploop`): wline([mu+1] * L_mu) for mu = 0..3 in four launches and ONE read-back.
// out[2 mu], out[2 mu + 1] = Re, Im.  On a t-sharded field the t line takes gauge_wline's segment path.
int gauge_polyakov(qexhip_ctx *c, double out[8]) {
  if (!c->gn) { qexhip_set_error("gauge field not set (qexhip_gauge_set)"); return -3; }
  const Geom &g = c->g;
  const int nd = g.halo ? 3 : 4;
  int nlmax = 0;
  for (int d = 0; d < nd; d++) nlmax = std::max(nlmax, g.V / g.X[d]);
  const int nbl = std::min((nlmax + 63) / 64, 512);           // 3 x 512 partials per direction, inside the first 6144 of the buffer
  launch_xlines(c, nbl, c->partials);                                                   // direction 0 -> group 0
  k_line_trace<<<dim3(nbl, nd - 1), 256, 0, c->stream>>>(g, c->gn->U, -2, c->partials);  // directions 1 .. nd-1 -> groups 1 ..
  k_sum3_groups<<<nd, 256, 0, c->stream>>>(c->partials, nbl, &c->dscal[40]);
  HIPCHK(hipGetLastError());
  double s[12];
  CHK(read_global(c, &c->dscal[40], 3 * nd, s));
  for (int d = 0; d < nd; d++) {
    const double fl = 1.0 / ((double)(g.V / g.X[d]) * (double)c->nranks * 3.0);
    out[2 * d] = s[3 * d] * fl;
    out[2 * d + 1] = s[3 * d + 1] * fl;
  }
  if (nd == 3) {
    const int Lt = g.X[3] * c->nranks;
    std::vector<int> path((size_t)Lt, 4);
    CHK(gauge_wline(c, path.data(), Lt, out + 6));
  }
  return 0;
}
