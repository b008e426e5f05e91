// flow_stage.hip -- one RK3 stage of the Wilson flow as a loader / consumer kernel (round 3).
//
// Restates (file:line in ctpeterson/qex), fused into one pass over the links:
//   makeStaples (plaquette staples)       src/gauge/staples.nim:153-238
//   gaugeActionDeriv / gaugeForce         src/gauge/gaugeAction.nim:195-204,334-350
//   contractProjectTAH                    src/gauge/gaugeUtils.nim:389-398
//   one stage of gaugeFlow's RK3          src/gauge/wflow.nim:36-62   (v = cf f + cpm p;  p = v;  U <- exp(v) U)
//
// STATUS: a measured ALTERNATIVE to k_force_lds (gauge.hip), selected by option "flow_ring" / QEXHIP_FLOW_RING=1, NOT the
// default: 835-911 us per stage against 707-790 on the same MI355X (profiles/r03_flow_stage_experiments.md has the numbers,
// the counters and the reasons).  It is kept because it is the form the round-2 review asked for, is tested to the oracle
// like the default (tests/test_gpu_parity.py, tests/test_gauge_actions.py), and documents what the idea costs on this chip.
//
// The idea.  k_force_lds gives a wavefront one direction mu of a 64-site tile; every lane gathers its 14 operand matrices
// itself, multiplies, then waits for the next gathers.  Its time is roughly the SUM of its gather phase (the CU's L2->L1
// path, ~64 GB/s per CU) and its fp64 phase (profiles/r02_kforce_experiments.md): with 256 registers per lane only two
// wavefronts fit a SIMD and both tend to sit in the same phase.  Here the two resources get their own wavefronts
// (cdna_hip_programming.md 5, "glds": LDS-DMA loader + consumers):
//
//   * a persistent workgroup per CU = 4 LOADER wavefronts + 4 CONSUMER wavefronts, walking its XCD's tiles in the
//     blocked order of tile_order_table;
//   * loaders never compute.  Form RS = 0: they issue `global_load_lds_dwordx4` (a 1 KiB wave-instruction, no VGPR
//     destination) for groups of four operand matrices (one 9 KiB slot per loader wavefront) into a ring of three group
//     buffers, two groups ahead of the consumers, and retire them with COUNTED `s_waitcnt vmcnt(9)`.  Form RS = 1 (the
//     default of this kernel): each loader keeps six matrices in flight in its REGISTERS and hands them to a
//     double-buffered ring with ds_write_b128 (the register file is the larger staging buffer: 216 KiB in flight per CU);
//   * consumers never touch global memory for operands: they read slots with ds_read_b128 and multiply.  Consumer w owns
//     the staple DIRECTION w: it keeps U_w(x) and U_w(x-w) in registers for the whole tile and forms, in three rounds
//     over the perfect matchings {w, w^r} (r = 1, 2, 3) of the four directions, the staples of link b = w^r in the plane
//     (b, w).  Inside a round the two wavefronts of a plane share the forward corner operands U_b(x+w), U_w(x+b), so a
//     site costs 44 matrix gathers (8 own/back + 6 planes x 6) instead of 56, plus the momenta;
//   * the staple sums of a link meet in an LDS accumulator (one 9 KiB tile per link; in every round the four consumers
//     write four different links, rounds are separated by the phase barriers: a fixed summation order, bit-reproducible);
//   * consumer w then finishes link w: f = TAH(U acc^+), v = cf cp f + cpm p, p <- v, U' = exp(v) U (non-temporal stores).
//
// One phase = one group: 12 per tile (own links, back links, 3 x {forward corners, U_b(x-w), U_w(x-w+b)}, momenta).
// LDS: ring 3 x 36 KiB (RS = 1 uses two of the three buffers) + accumulator 36 KiB = 144 KiB of the CU's 160.
#include "qexhip_internal.h"
#include "su3.h"
#include "gauge_index.h"

#define FS_SLOT 576                      // double2 elements per slot: [9][64]
#define FS_GROUP (4 * FS_SLOT)
#define FS_NBUF 3
#define FS_LDS_BYTES ((FS_NBUF * FS_GROUP + 4 * FS_SLOT) * sizeof(double2))
#define FS_PHASES 12

typedef __attribute__((address_space(3))) void fs_lds_void;
typedef const __attribute__((address_space(1))) void fs_glob_void;

// wait for the LDS traffic of this wavefront, then the workgroup barrier (raw: no vmcnt drain, the loaders' DMAs stay in flight)
#define FS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// one operand matrix of 64 sites -> slot: nine 1 KiB LDS-DMA wave-instructions (element k of all lanes)
// src: this lane's element 0; slot: the slot's base (wave-uniform; the hardware adds lane * 16)
template <int AUX>   // 0: default cache policy (links: neighbouring tiles re-read them), 2: nt (momenta: read once)
__device__ __forceinline__ void fs_fill_slot(const double2 *src, double2 *slot) {
#pragma unroll
  for (int k = 0; k < 9; k++)
    __builtin_amdgcn_global_load_lds((fs_glob_void *)(src + (size_t)k * 64), (fs_lds_void *)(slot + k * 64), 16, 0, AUX);
}

// matrix -> LDS slot, element-wise as 16-byte vectors (a double2 STRUCT assignment into an LDS pointer becomes a memcpy from
// a stack copy of the matrix, which keeps all the loader's register sets in scratch)
__device__ __forceinline__ void fs_store_lds(double2 *p, const M3 &m) {
#pragma unroll
  for (int k = 0; k < 9; k++) {
    m3_d2v t = {m.e[k].x, m.e[k].y};
    *(m3_d2v *)&p[k * 64] = t;
  }
}

struct FlowStageArgs {
  Geom g;
  const double2 *U;      // links in
  double2 *P;            // RK momentum: read when cpm != 0, always written (v)
  double2 *Uout;         // exp(v) U
  double cp, cf, cpm;    // cp = c.plaq / nc;  v = cf cp f + cpm p
  int chunk;             // the order table is [8][chunk]
};

// per-lane source of loader w's matrix in group kind gi (0 own, 1 back, 2 + 3 (r - 1) + {0, 1, 2}: round r, 11 momenta)
template <bool HALO>
__device__ __forceinline__ const double2 *fs_src(const FlowStageArgs &a, const FsSite &s, int w, int gi) {
  const Geom &g = a.g;
  int lex = s.lex, par = s.par;
  if (gi == 11) return a.P + fs_link_off(g, lex, par, w);
  if (gi == 0) return a.U + fs_link_off(g, lex, par, w);
  if (gi == 1) { fs_hop<HALO>(g, s, w, -1, lex, par); return a.U + fs_link_off(g, lex, par, w); }
  const int r = (gi - 2) / 3 + 1, k = (gi - 2) % 3, q = w ^ r;
  if (k == 0) { fs_hop<HALO>(g, s, q, 1, lex, par); return a.U + fs_link_off(g, lex, par, w); }                 // U_w(x+q)
  if (k == 1) { fs_hop<HALO>(g, s, w, -1, lex, par); return a.U + fs_link_off(g, lex, par, q); }                // U_q(x-w)
  fs_hop<HALO>(g, s, w, -1, lex, par);                                                                          // U_w(x-w+q): the two hops are in
  fs_hop<HALO>(g, s, q, 1, lex, par);                                                                           // different directions (q != w)
  return a.U + fs_link_off(g, lex, par, w);
}

// RS = 0: LDS-DMA loaders (ring of three group buffers, two groups = 72 KiB in flight per CU at most)
// RS = 1: register-staged loaders: every loader wavefront keeps FS_Q matrices in flight in its REGISTERS (the register file,
//         512 KiB per CU, is the larger staging buffer: FS_Q x 36 KiB in flight per CU) and hands them to the consumers through
//         a double-buffered LDS ring with ds_write_b128
#define FS_Q 6
template <bool CLOSED, bool HALO, int DBG, int RS>
__global__ void __launch_bounds__(512) k_flow_stage(FlowStageArgs a, const int *__restrict__ order) {
  // order: tile_order_table, [8][chunk] entries 2 tile + parity, -1 padding at the tail; a kernel argument of its own and
  // __restrict__ so that its reads are scalar loads (a vector load per tile would make the loader drain its DMAs: vmcnt(0))
  extern __shared__ double2 fs_lds[];
  double2 *ring = fs_lds;                                 // [FS_NBUF][4][576]
  constexpr int NBUF = RS ? 2 : FS_NBUF;
  double2 *accs = fs_lds + NBUF * FS_GROUP;               // [4][576]
  const Geom &g = a.g;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3, jstride = gridDim.x >> 3;
  const int *ord = order + (size_t)xcd * a.chunk;
  // tiles of this workgroup: j0, j0 + jstride, ... while the entry is valid (padding sits at the tail of an XCD's list)
  int ntile = 0;
  for (int j = j0; j < a.chunk; j += jstride) {
    if (ord[j] < 0) break;
    ntile++;
  }
  const int nphase = ntile * FS_PHASES;
  const bool with_mom = a.cpm != 0.0;

  if (wave >= 4) {
    // ------------------------------------------------ loader ------------------------------------------------
    if (RS) {
      const int w = wave - 4;
      // FS_Q register sets, named (an array indexed inside a late-unrolled loop stays in scratch)
      M3 st0, st1, st2, st3, st4, st5;
      FsSite xa;                                          // this lane's site of the tile being issued
      // Branch-free issue on purpose: hipcc counts the outstanding loads exactly (s_waitcnt vmcnt(N) with N = the loads issued
      // after the set being written) only when every phase issues its loads; a conditional issue makes it drain everything
      // (vmcnt(0)) at the merge.  Groups past the end of the workgroup's list are therefore loaded (from the last tile's valid
      // addresses) and written like the others -- nobody reads them -- and the momenta are loaded in the first stage too
      // (cpm = 0: ignored by the consumers).
      auto tile_coords = [&](int ti, FsSite &x) {
        const int tc = ti < ntile ? ti : ntile - 1;
        const int e = ord[j0 + tc * jstride];
        int c = (e >> 1) * 64 + lane;
        if (c >= g.Vh) c = g.Vh - 1;                      // padding lanes of the last tile gather a valid site
        fs_site(g, c, e & 1, x);
      };
      if (ntile == 0) { FS_BARRIER(); return; }
      tile_coords(0, xa);
      // prologue: groups 0 .. FS_Q-1 of the first tile into the register sets, group 0 on to buffer 0, group FS_Q behind it
      st0 = m3_load(fs_src<HALO>(a, xa, w, 0), 64); __builtin_amdgcn_sched_barrier(0);
      st1 = m3_load(fs_src<HALO>(a, xa, w, 1), 64); __builtin_amdgcn_sched_barrier(0);
      st2 = m3_load(fs_src<HALO>(a, xa, w, 2), 64); __builtin_amdgcn_sched_barrier(0);
      st3 = m3_load(fs_src<HALO>(a, xa, w, 3), 64); __builtin_amdgcn_sched_barrier(0);
      st4 = m3_load(fs_src<HALO>(a, xa, w, 4), 64); __builtin_amdgcn_sched_barrier(0);
      st5 = m3_load(fs_src<HALO>(a, xa, w, 5), 64); __builtin_amdgcn_sched_barrier(0);
      fs_store_lds(ring + (size_t)w * FS_SLOT + lane, st0);
      __builtin_amdgcn_sched_barrier(0);
      st0 = m3_load(fs_src<HALO>(a, xa, w, FS_Q), 64);
      __builtin_amdgcn_sched_barrier(0);
      FS_BARRIER();
      // steady state, FS_Q phases per trip (FS_Q divides 12, so the register set of a phase is static); the kind of the group
      // issued is a run-time, wavefront-uniform value: its branches hold address arithmetic only, no memory operation
      int ti_issue = 0;                                   // tile whose sites xa holds
      double2 *const myslot = ring + (size_t)w * FS_SLOT + lane;
      // phase n0 + Q: group gw = n0 + Q + 1, loaded FS_Q phases ago into SET, moves on to buffer gw % 2, and SET takes group gw + FS_Q
#define FS_LOADER_PHASE(Q, SET)                                                          \
      {                                                                                    \
        const int gw = n0 + (Q) + 1, gi = gw + FS_Q;                                       \
        if (DBG < 4) fs_store_lds(myslot + (size_t)(gw & 1) * FS_GROUP, SET);              \
        else { double sx = 0; for (int k_ = 0; k_ < 9; k_++) sx += SET.e[k_].x + SET.e[k_].y;    \
               if (sx == 1.2345e300) fs_store_lds(myslot, SET); }                            \
        const int ti = gi / FS_PHASES, ki = gi - ti * FS_PHASES;                           \
        if (ti != ti_issue) { tile_coords(ti, xa); ti_issue = ti; }                        \
        if (ki == 11) SET = m3_load_nt(fs_src<HALO>(a, xa, w, ki), 64);                    \
        else SET = m3_load(fs_src<HALO>(a, xa, w, ki), 64);                                \
        FS_BARRIER();                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                 \
      }
      for (int n0 = 0; n0 < nphase; n0 += FS_Q) {
        FS_LOADER_PHASE(0, st1)
        FS_LOADER_PHASE(1, st2)
        FS_LOADER_PHASE(2, st3)
        FS_LOADER_PHASE(3, st4)
        FS_LOADER_PHASE(4, st5)
        FS_LOADER_PHASE(5, st0)
      }
#undef FS_LOADER_PHASE
      return;
    }
    const int w = wave - 4;                               // slot of every group; direction of the matrix it holds
    FsSite xs;
    int cur_tile = -1;
    auto issue = [&](int n) -> bool {                     // group n (global phase number) -> buffer n % 3; false: nothing issued
      if (n >= nphase) return false;
      const int ti = n / FS_PHASES, gi = n - ti * FS_PHASES;
      if (ti != cur_tile) {
        const int e = ord[j0 + ti * jstride];
        int c = (e >> 1) * 64 + lane;
        if (c >= g.Vh) c = g.Vh - 1;                      // padding lanes of the last tile gather a valid site
        fs_site(g, c, e & 1, xs);
        cur_tile = ti;
      }
      if (gi == 11 && !with_mom) return false;
      if (DBG == 2) return false;
      double2 *slot = ring + (size_t)(n % NBUF) * FS_GROUP + (size_t)w * FS_SLOT;
      // the SOURCE is per lane (the lanes of a shifted tile are not always consecutive -- x wraps, rows end); only the LDS
      // side is "uniform base + lane * 16"
      if (gi == 11) fs_fill_slot<2>(fs_src<HALO>(a, xs, w, gi), slot);     // momenta: streamed once
      else fs_fill_slot<0>(fs_src<HALO>(a, xs, w, gi), slot);
      return true;
    };
    // prologue: groups 0 and 1
    bool i0 = issue(0);
    bool i1 = issue(1);
    (void)i0;
    if (i1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FS_BARRIER();
    for (int n = 0; n < nphase; n++) {
      const bool is = issue(n + 2);
      if (is) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      FS_BARRIER();
    }
    return;
  }

  // ------------------------------------------------ consumer ------------------------------------------------
  const int w = wave;                                     // staple direction; finishes link w
  FS_BARRIER();                                           // prologue: group 0 has landed
  M3 A, D, S = m3_zero(), t = m3_zero();
  int n = 0;
  for (int ti = 0; ti < ntile; ti++) {
    const int e = ord[j0 + ti * jstride];
    const int p = e & 1;
    const int c0 = (e >> 1) * 64 + lane;
    const bool live = c0 < g.Vh;
    // phase 0: own link, phase 1: back link
    A = m3_load(ring + (size_t)(n % NBUF) * FS_GROUP + (size_t)w * FS_SLOT + lane, 64);
    FS_BARRIER(); n++;
    D = m3_load(ring + (size_t)(n % NBUF) * FS_GROUP + (size_t)w * FS_SLOT + lane, 64);
    FS_BARRIER(); n++;
#pragma unroll 1
    for (int r = 1; r <= 3; r++) {
      const int b = w ^ r;
      {                                                   // forward: S = U_w(x) U_b(x+w) U_w(x+b)^+      (staples.nim:181-183)
        const double2 *buf = ring + (size_t)(n % NBUF) * FS_GROUP + lane;
        if (DBG == 0 || DBG == 2) {
          t = m3_mul_na(m3_load(buf + (size_t)b * FS_SLOT, 64), m3_load(buf + (size_t)w * FS_SLOT, 64));
          S = m3_mul(A, t);
        }
        FS_BARRIER(); n++;
      }
      {                                                   // backward: t = U_w(x-w)^+ U_b(x-w)              (staples.nim:184-186)
        const double2 *buf = ring + (size_t)(n % NBUF) * FS_GROUP + lane;
        if (DBG == 0 || DBG == 2) t = m3_mul_an(D, m3_load(buf + (size_t)w * FS_SLOT, 64));
        FS_BARRIER(); n++;
      }
      {                                                   // S += t U_w(x-w+b); hand S to link b's accumulator
        const double2 *buf = ring + (size_t)(n % NBUF) * FS_GROUP + lane;
        if (DBG == 0 || DBG == 2) m3_mac(S, t, m3_load(buf + (size_t)w * FS_SLOT, 64));
        double2 *ac = accs + (size_t)b * FS_SLOT + lane;
        if (r == 1) {
#pragma unroll
          for (int k = 0; k < 9; k++) ac[k * 64] = S.e[k];
        } else {
#pragma unroll
          for (int k = 0; k < 9; k++) { double2 v = ac[k * 64]; ac[k * 64] = make_double2(v.x + S.e[k].x, v.y + S.e[k].y); }
        }
        FS_BARRIER(); n++;
      }
    }
    {                                                     // phase 11: finish link w
      const M3 acc = m3_load(accs + (size_t)w * FS_SLOT + lane, 64);
      M3 f = m3_tah(m3_mul_na(A, acc));
      const double cfp = a.cf * a.cp;
      M3 v;
      if (with_mom) {
        const M3 pm = m3_load(ring + (size_t)(n % NBUF) * FS_GROUP + (size_t)w * FS_SLOT + lane, 64);
#pragma unroll
        for (int k = 0; k < 9; k++) v.e[k] = make_double2(cfp * f.e[k].x + a.cpm * pm.e[k].x, cfp * f.e[k].y + a.cpm * pm.e[k].y);
      } else {
#pragma unroll
        for (int k = 0; k < 9; k++) v.e[k] = make_double2(cfp * f.e[k].x, cfp * f.e[k].y);
      }
      if (live && (DBG < 3 || v.e[0].x == 1.2345e300)) {     // DBG >= 3: no stores
        int x[4];
        coords_of(g, c0, p, x);
        const size_t o = link_off_t<HALO>(g, x, w);
        m3_store_nt(a.P + o, 64, v);
        if (DBG == 0 || DBG == 2) {
          const M3 un = m3_mul(CLOSED ? m3_exp_tah(v) : m3_exp(v), A);
          m3_store_nt(a.Uout + o, 64, un);
        } else {
          m3_store_nt(a.Uout + o, 64, v);
        }
      }
      FS_BARRIER(); n++;
    }
  }
}

// ---- host ----
// The measurement variants of profiles/r03_flow_stage_experiments.md (DBG = 1: consumers skip the arithmetic, 2: loaders skip the
// loads, 3: 1 without the result stores, 4: 3 without the LDS hand-off; wrong results by design) are compiled only with
// -DQEXHIP_FLOW_STAGE_DEBUG (make CXXFLAGS+=-DQEXHIP_FLOW_STAGE_DEBUG) and then selected by QEXHIP_FLOW_STAGE_DBG.
int flow_stage_launch(qexhip_ctx *c, const double2 *U, double2 *P, double2 *Uout, double cp, double cf, double cpm,
                      const int *order, int chunk, bool closed) {
  static const int wgs = [] { const char *e = getenv("QEXHIP_FLOW_STAGE_WGS"); return e ? atoi(e) : 256; }();   // one per CU
  static const int rs = [] { const char *e = getenv("QEXHIP_FLOW_STAGE_RS"); return e ? atoi(e) : 1; }();
#define QX_ATTR(K) HIPCHK(hipFuncSetAttribute((const void *)K, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FS_LDS_BYTES))
  if (!(c->lds_attr_done & 4)) {
    QX_ATTR((k_flow_stage<true, false, 0, 0>)); QX_ATTR((k_flow_stage<false, false, 0, 0>));
    QX_ATTR((k_flow_stage<true, true, 0, 0>));  QX_ATTR((k_flow_stage<false, true, 0, 0>));
    QX_ATTR((k_flow_stage<true, false, 0, 1>)); QX_ATTR((k_flow_stage<false, false, 0, 1>));
    QX_ATTR((k_flow_stage<true, true, 0, 1>));  QX_ATTR((k_flow_stage<false, true, 0, 1>));
#ifdef QEXHIP_FLOW_STAGE_DEBUG
    QX_ATTR((k_flow_stage<true, false, 1, 0>)); QX_ATTR((k_flow_stage<true, false, 2, 0>));
    QX_ATTR((k_flow_stage<true, false, 1, 1>)); QX_ATTR((k_flow_stage<true, false, 3, 1>)); QX_ATTR((k_flow_stage<true, false, 4, 1>));
#endif
    c->lds_attr_done |= 4;
  }
#undef QX_ATTR
  FlowStageArgs a;
  a.g = c->g; a.U = U; a.P = P; a.Uout = Uout; a.cp = cp; a.cf = cf; a.cpm = cpm; a.chunk = chunk;
  int nb = wgs & ~7;
  if (nb < 8) nb = 8;
  if (nb > 8 * chunk) nb = 8 * chunk;
#define QX_FS(CL, HL) do { if (rs) k_flow_stage<CL, HL, 0, 1><<<nb, 512, FS_LDS_BYTES, c->stream>>>(a, order); \
                           else k_flow_stage<CL, HL, 0, 0><<<nb, 512, FS_LDS_BYTES, c->stream>>>(a, order); } while (0)
#ifdef QEXHIP_FLOW_STAGE_DEBUG
  static const int dbg = [] { const char *e = getenv("QEXHIP_FLOW_STAGE_DBG"); return e ? atoi(e) : 0; }();
  if (dbg == 1) { if (rs) k_flow_stage<true, false, 1, 1><<<nb, 512, FS_LDS_BYTES, c->stream>>>(a, order); else k_flow_stage<true, false, 1, 0><<<nb, 512, FS_LDS_BYTES, c->stream>>>(a, order); }
  else if (dbg == 2) k_flow_stage<true, false, 2, 0><<<nb, 512, FS_LDS_BYTES, c->stream>>>(a, order);
  else if (dbg == 3) k_flow_stage<true, false, 3, 1><<<nb, 512, FS_LDS_BYTES, c->stream>>>(a, order);
  else if (dbg == 4) k_flow_stage<true, false, 4, 1><<<nb, 512, FS_LDS_BYTES, c->stream>>>(a, order);
  else
#endif
  if (closed) { if (c->g.halo) QX_FS(true, true); else QX_FS(true, false); }
  else { if (c->g.halo) QX_FS(false, true); else QX_FS(false, false); }
#undef QX_FS
  HIPCHK(hipGetLastError());
  return 0;
}
