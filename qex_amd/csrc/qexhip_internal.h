// qexhip_internal.h -- shared declarations of libqexhip.so (gfx950 only).
//
// Device data layout (DESIGN.md "Data layout in HBM"):
//   * sites of one parity are numbered by the checkerboard index c = lex/2 of the rank-local
//     lattice (x fastest, t slowest), so a t-slice is the contiguous range [t*F,(t+1)*F);
//   * sites are grouped in TILES of 64 = one wavefront.  A colour vector of one parity is
//       double2 v[tile][3][64]           (re,im packed: one 16-byte load per lane and colour)
//     and the Dslash-ready gauge field of one parity is
//       double2 W[tile][ndir][9][64]     ndir = 8 (fat) or 16 (fat + Naik)
//     with dir d = 2*mu (+8 for the 3-hop links): forward link U_mu(s); d = 2*mu+1: the
//     backward link already shifted and adjointed, U_mu(s-mu)^+.  A wavefront therefore streams
//     one contiguous 72 KiB (144 KiB) block of links per sweep;
//   * with the t dimension sharded, `depth` ghost t-slices follow the body tiles:
//       [ body: Vh sites | ghost_hi: depth*F (upper neighbour's t=0..depth-1) |
//         ghost_lo: depth*F (lower neighbour's t=Xt-depth..Xt-1) ]
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include <map>

#define QEXHIP_TILE 64

struct Geom {
  int X[4];
  int Xh;      // X[0]/2
  int V, Vh;
  int F;       // sites of one parity per t-slice
  int ntile;   // body tiles per parity
  int depth;   // ghost depth in t (0: no ghost zones)
  int gtile;   // ghost tiles per side per parity
  int etile;   // tiles per parity incl. ghosts
  int halo;    // 1: t-hops across the local boundary read the ghost zones
};

// position (site number within one parity, possibly in a ghost zone) -> double2 offset
__host__ __device__ inline size_t vec_off(int pos, int colour) {
  return (size_t)(pos >> 6) * 192 + (size_t)colour * 64 + (size_t)(pos & 63);
}

struct DevField {  // full-volume colour vector: parity halves, each with its ghost tiles
  double2 *d = nullptr;
  size_t half = 0;  // double2 elements per parity half (= etile*192)
  double2 *par(int p) const { return d + (size_t)p * half; }
};

// CG scalars resident on the device (no host round trip per iteration)
struct CgScal {
  double b2, r2, rzo, pAp, r2stop, tmp;
  int itn, maxits, done, pad;
  // plain CG (blas.hip): state after k iterations in slot k&1, so that the kernel which closes iteration k-1 can
  // write slot k&1 while its other workgroups still read slot (k-1)&1
  double r2s[2];
  int itns[2], dones[2];
  // sharded: {r2, -r2, itn, -itn} of the state the host is about to read, max-reduced over the ranks at the end of a chunk:
  // the ranks decide on `done` each for itself, so their residuals must agree bit for bit (comm_agree)
  double agree[4];
};

struct TimerSlot {
  std::vector<hipEvent_t> ev;  // pairs
  size_t used = 0;
  long count = 0;
  double total_ms = 0;
};

// Device-side stream join and the one error word of every bounded device-side wait (peer.hip).  `signal` raises a counter behind what a
// stream has posted (a one-lane kernel), `wait` makes another stream wait for it (a one-wave kernel with a bounded poll) -- instead of
// the runtime's cross-queue event dependency, which costs ~20-28 us of dead time per join here (profiles/r05_timeline_*.txt).  Both
// transports use it since round 6.
struct DevJoin {
  unsigned long long *ready = nullptr;   // device: word k*16 = joins FROM stream k (0 compute, 1 comm); word 56: start time of the fused sweep's push (emulation)
  unsigned long long *err = nullptr;     // pinned host word: code of the first wait that gave up
  long long ticks = 0;                   // QEXHIP_PEER_TIMEOUT in wall_clock64 ticks
  double timeout_s = 30.0;
  unsigned long long seq[2]{0, 0};
  int deferred = 0;                      // the compute stream still owes a wait for seq[1]: the next mailbox all-reduce of partials polls for it in its
                                         // prologue, anything else posts it first (devjoin_flush)
};

struct GaugeNat;  // natural-layout gauge field for plaquette / flow (gauge.hip)
struct PeerComm;  // peer-memory transport (peer.hip)

enum { WK_SLOTS = 16 };   // >= WK_N (below)

struct qexhip_ctx {
  int device = 0;
  Geom g{};
  int rankGeom[4]{1, 1, 1, 1}, rankCoord[4]{0, 0, 0, 0};
  hipStream_t stream = nullptr, cstream = nullptr;
  hipEvent_t ev_ready = nullptr;      // compute -> comm stream: "the producer of the faces is done" (the way back is a device-side join: DevJoin)
  // communicator
  void *comm = nullptr;  // ncclComm_t: everything posted on the compute stream (all-reduces, ghost refreshes, non-overlapped faces)
  void *comm2 = nullptr; // ncclComm_t split off comm: the face exchanges posted on cstream beside the interior sweep
  PeerComm *peer = nullptr; // peer-mapped memory (peer.hip): control block with the mailboxes of the rank sums, and -- when `comm` is null -- the
                            // receive arenas of the faces too.  comm && peer: RCCL faces + mailbox sums (hybrid_sums)
  DevJoin dj;               // device-side stream joins + the error word of the bounded waits (allocated by qexhip_init)
  int opt_transport = -1;   // option "transport" (before comm_init): -1 QEXHIP_TRANSPORT decides, 0 auto, 1 rccl, 2 peer
  int nranks = 1, rank = 0;
  int force_halo = 0;
  // staggered links
  double2 *W = nullptr;  // [parity][tile][ndir][9][64]
  int ndir = 0;
  // device fields
  std::map<int, DevField> fields;
  int next_field = 1;
  // scratch
  double *stage = nullptr; size_t stage_bytes = 0;   // host-format staging on device
  double *partials = nullptr; int npartials = 0;     // block partial sums: [0,part2_off) Dslash / redot, then 2048 for the CG update
  int part2_off = 0;
  double *dscal = nullptr;                           // 64 device scalars (reductions), by owner: [0..4] solvers (b2, r2, norms of
                                                     // the full solve), [8..9] the BLAS entry points, [16..21] plaquettes, [24..32]
                                                     // flow observables / action / line sums, [40..51]
                                                     // s4 / Polyakov sums, [56..59] comm_allreduce_max, [60] comm_init's agreement,
                                                     // [62] link-compression test
  CgScal *cg = nullptr;                              // device CG state
  struct { int valid = 0; const double2 *x = nullptr; int par_even = 0; double m2 = 0; int k = 0; } cg_resume;   // what solve_xx_continue_dev needs
  int cg_r2parts = 0;                                // values the |r|^2 partial buffer holds after the last cg_update (1 once a peer all-reduce has summed them)
  double *hist = nullptr; int histcap = 0;           // device residual history
  void *pinned = nullptr;                            // pinned host scratch (4 KiB)
  // work vectors (lazily allocated, like the {.global.} temp of stagD.nim:437-442)
  int wk[WK_SLOTS]{0};
  // timers
  int timers_on = 0;
  std::map<std::string, TimerSlot> timers;
  // compressed links (recon = 1: rows 0,1 + sign mask; 2: rows 0,1 + det; row 2 rebuilt in the kernel)
  double2 *Wc = nullptr; unsigned long long *Ws = nullptr; size_t Wc_rows = 0; int recon = 0; double recon_dev = 0;
  int opt_batch_multi = 0; // test hook: take the multi-rank reduction branch of the batched CG on one rank
  int opt_multi_reduce = 0; // test hook: take the multi-rank reduction branches of CG / multi-shift CG / norms on one rank
                            // (with a one-rank RCCL communicator the all-reduces are real collectives)
  int opt_force_pair = 1; // option "force_pair" (test hook): 0 takes k_force_lds, the form lattice shapes without paired tile positions get, on any shape
  int opt_obs_clover = 1; // option "obs_clover" (test hook): 0 takes the generic path walker, the form fmunu loops 3-5 get, for loop 1 as well
  int opt_hop_split = -1;    // option "hop_split" / QEXHIP_HOP_SPLIT: how an OVERLAPPED sweep is laid out on the peer transport.  2 = the fused sweep
                             // (k_dslash_fused: ONE launch on ONE stream pushes the faces, takes the interior, and its boundary workgroups take
                             // the hops that stay inside the slab, wait SHORTLY for the faces, and either finish or park their accumulator for
                             // the cleanup workgroups at the end of the grid), 0 = split by sites (interior launch | exchange + boundary launch
                             // on the comm stream, device-side join; what the RCCL transport always runs), -1 = whichever set_links measured
                             // faster (fused until measured; sweep_autotune)
  int opt_fused_spin_us = -1; // option "fused_spin_us": the short wait of the fused sweep's boundary workgroups; -1 = about the measured (else estimated)
                              // transfer time, >= 0 that many microseconds, -2 = park every boundary block (test hook: cleanup path everywhere)
  int ranks_share_device = 0;   // comm_init's rendezvous saw two ranks of this job on one GPU (informational: no kernel holds more than a few
                                // dozen slots while it waits for another rank any more)
  int hybrid_sums = 0;          // RCCL carries the faces, the mailboxes of the peer control block the CG's rank sums (comm.cpp: comm_init)
  unsigned int *fz_buf = nullptr; int fz_cap = 0;   // FusedCtl words + parked-block list of the fused sweep (dslash.hip)
  const double2 *bnd_out_on_cstream = nullptr;      // the parity half whose t-faces the LAST sweep's boundary launch wrote on the comm stream (split by sites), else null
  double xchg_us[2]{0, 0};      // measured at set_links (collective, max over ranks): one face exchange of the 8- / 16-link operator, us (0: not measured)
  int form_auto[2]{-1, -1};     // measured at set_links: 2 fused / 0 by sites for 8- and 16-link operators (-1: not measured)
  int opt_chain_overlap = 1; // option "chain_overlap" (A/B, test hook): 1 = the nHYP force chain's staple derivatives of a t-sharded field run in two passes,
                          // the ghost-free slices beside the exchange of the level's chain fields, the boundary slices behind it
  int opt_smear_ca = 1;   // option "smear_ca" (A/B, test hook): 1 = the nHYP levels of a t-sharded field are computed on shrinking ghost slices from
                          // ONE depth-3 thin-link exchange; 0 = every projected level field refreshes its ghost slices (rounds 1-4)
  int opt_flow_exp = 1;   // QEXHIP_FLOW_EXP / option "flow_exp": 1 = closed-form exp(v) in the fused Wilson-flow stage (same function,
                          // another algorithm than the reference's; agrees with it to ~1e-15 per element; the default); 0 = the reference's
                          // Taylor + 20 squarings (matexp.nim), which the MD link updates always use
  int opt_recon = 2;      // QEXHIP_RECON: 0 keeps the 18-real links always, 1 sign format only, 2 also the U(3) format
  int opt_overlap = -1;  // QEXHIP_OVERLAP / option "overlap": 1 always use the comm stream, 0 never, -1 measured once per operator shape when the
                         // communicator has more than one rank (sweep_autotune), by interior / face size otherwise; -2: measure on one rank too
  int emu_link_gbs = 0;                            // option emu_link_gbs: > 0 adds bytes / (GB/s of one xGMI direction) to every emulated exchange
  int emu_exchange_us = 0, emu_allreduce_us = 0;   // options of the same names (test / rehearsal hooks): delay posted in front of every face
                                                   // exchange / all-reduce, as long as the transfer would take between distinct GPUs
  int overlap_auto[2]{-1, -1};          // the measured decision for 8- and 16-link operators (-1: not measured)
  double overlap_tune_us[2][3]{};       // us per sweep the measurement saw: [8 | 16 links][exchange first | overlapped by sites | fused], max over ranks (0: form not available)
  // natural gauge (flow)
  GaugeNat *gn = nullptr;
  void *nhyp = nullptr;   // NhypState (smear.hip): the smearGetForce closure
  void *hisq = nullptr;   // HisqState (smear.hip): HisqCoefs.smearGetForce's closure
  // small per-context device scratch owned by single kernels' host wrappers
  double2 *outer_F = nullptr; size_t outer_Fn = 0;   // force field of stag_outer_host (force.hip)
  int tile_pairs_ok = 0;   // tile_order_table: every even slot holds (tile, 0) and the next one (tile, 1) (or both are padding): k_force_lds2
  int *tile_order = nullptr; int tile_order_n = 0;   // blocked (tile, parity) visiting order of the gather kernels (gauge.hip)
  int *tile_order_pl[16]{};                          // the same for kernels that shift in the (mu, nu) plane only (layout.hip)
  void *obs_table = nullptr;                         // ObsTable of gauge_flow_obs (gauge.hip)
  void *batch = nullptr;                             // BatchState of the lock-step multi-system CG (batch.hip)
  int lds_attr_done = 0;                             // per context (= per device): which kernels had MaxDynamicSharedMemorySize raised (bit 0 k_force_lds, 1 k_flow_obs_clover, 2 k_flow_obs_clover2, 3 k_force_lds2, 4 k_projUderiv_batch)
  void *cgm_scal = nullptr;                          // CgmScal of the multi-shift solver (multishift.hip)
};

// work-field slots (get_work)
enum { WK_T = 0, WK_R, WK_P, WK_AP, WK_Y, WK_D, WK_R2, WK_XT, WK_IN, WK_OUT, WK_IN2, WK_N };

// ---- error handling ----
void qexhip_set_error(const char *fmt, ...);
#define HIPCHK(expr)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      qexhip_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
      return -2;                                                                         \
    }                                                                                    \
  } while (0)
#define CHK(expr) do { int r_ = (expr); if (r_ != 0) return r_; } while (0)

// ---- timers ----
struct ScopedTimer {
  qexhip_ctx *c; TimerSlot *s = nullptr; hipStream_t st;
  ScopedTimer(qexhip_ctx *c_, const char *name, hipStream_t st_);
  ~ScopedTimer();
};
int timers_collect(qexhip_ctx *c);
// reserve an event pair of timer class `name` (false when timing is off for that class)
bool timer_event_pair(qexhip_ctx *c, const char *name, hipEvent_t *e0, hipEvent_t *e1);

// ---- layout.hip ----
int geom_init(Geom &g, const int X[4], int depth, int halo);
int field_alloc(qexhip_ctx *c, DevField &f);
int field_upload(qexhip_ctx *c, DevField &f, const double *host);      // host MILC order -> tiles
int field_download(qexhip_ctx *c, const DevField &f, double *host);
int links_upload(qexhip_ctx *c, const double *fat, const double *lng);
int ensure_stage(qexhip_ctx *c, size_t bytes);
int links_from_natural(qexhip_ctx *c, const double2 *fat, const double2 *lng);
int links_compress(qexhip_ctx *c);
int op_eo_reconstruct_pub(qexhip_ctx *c, DevField &r, DevField &b, double m);
int op_eo_reduce_pub(qexhip_ctx *c, DevField &r, DevField &b, double m);
int op_stagD_pub(qexhip_ctx *c, DevField &r, DevField &x, int parity, double m, double sc, double a);   // one subset of stagD
void batch_state_free(qexhip_ctx *c);
int batch_io_fields(qexhip_ctx *c, int n, DevField **xs, DevField **bs);
int solve_full_batch_dev(qexhip_ctx *c, int n, DevField **x, DevField **b, const double *mass, const double *r2req,
                         int maxits, int *iters, double *r2_final);
int nhyp_fforce(qexhip_ctx *c, double *f_host, int n, const double *const *phi, const double *mass, const double *scale,
                const double *r2req, int maxits, int bcmask, const int ph[4], int *iters, DevField *const *phi_dev = nullptr);
int solve_batch_host(qexhip_ctx *c, int n, double *const *x, const double *const *b, const double *mass,
                     const double *r2req, int maxits, int xx_parity, int *iters, double *r2);

// reductions end in an all-reduce: more than one rank, or the one-rank rehearsal of that code path
inline bool multi_rank(const qexhip_ctx *c) { return c->nranks > 1 || c->opt_multi_reduce; }

// ---- comm.cpp ----
int comm_halo_push_only(qexhip_ctx *c, DevField &f, int parity, const double2 **gh_hi, const double2 **gh_lo, struct PeerPush *push);   // peer faces: the fused sweep
                                                                              // pushes the faces of f itself and reads what arrives in the receive arena
int comm_halo_push_only_multi(qexhip_ctx *c, int n, DevField *const *f, int parity, const double2 **gh_hi, const double2 **gh_lo, struct PeerPush *push);   // n <= 4 fields (lock-step batch)
int comm_halo_exchange(qexhip_ctx *c, DevField &f, int parity, int overlap, bool wait_ready = true);  // overlap: on cstream after ev_ready (wait_ready);
                                                                              // the caller joins behind what it posts next
int comm_halo_exchange_multi(qexhip_ctx *c, int n, DevField *const *f, int parity, int overlap);   // n fields, one RCCL group
int comm_allreduce(qexhip_ctx *c, double *dptr, int n);          // on stream
int comm_allreduce_parts(qexhip_ctx *c, double *parts, int n, int *n_out);   // workgroup partials -> *n_out values whose sum is the rank-global dot product
int comm_allreduce_max(qexhip_ctx *c, double *host, int n);      // host values -> max over the ranks, back on the host (n <= 4, synchronous)
int comm_agree_post(qexhip_ctx *c);                               // max-reduce c->cg->agree over the ranks (on stream)
int comm_agree_check(qexhip_ctx *c, const CgScal &host);          // after the state was read back: all ranks hold the same residual and count
int comm_allgather(qexhip_ctx *c, const double *send, double *recv, size_t n);
int comm_faces_exchange(qexhip_ctx *c, int nbuf, double *const bottom[], double *const top[], double *const ghost_hi[],
                        double *const ghost_lo[], size_t ndoubles, int async = 0);   // async: on the comm stream after ev_ready, no join
int comm_exchange_raw(qexhip_ctx *c, const void *send_up, void *recv_from_down, size_t bytes, hipStream_t st);
void comm_destroy(qexhip_ctx *c);
inline bool comm_ready(const qexhip_ctx *c) { return c->comm != nullptr || c->peer != nullptr; }
inline bool peer_faces(const qexhip_ctx *c) { return c->peer != nullptr && c->comm == nullptr; }    // the faces travel through peer-mapped arenas (else RCCL / copies)

// ---- peer.hip: device-side joins, and the peer-memory transport behind the same comm_* entry points ----
int devjoin_init(qexhip_ctx *c);                                                  // qexhip_init
void devjoin_destroy(qexhip_ctx *c);
int devjoin_check(qexhip_ctx *c);                       // a bounded device-side wait gave up since the last check -> QEXHIP_ERR_COMM
int devjoin_signal(qexhip_ctx *c, hipStream_t from);                              // device-side event: record ...
int devjoin_wait(qexhip_ctx *c, hipStream_t waiter, hipStream_t from);            // ... and wait, without the runtime's cross-queue dependency
int devjoin_defer(qexhip_ctx *c);                                                 // the compute stream's wait rides in the next peer_allreduce_parts ...
int devjoin_flush(qexhip_ctx *c);                                                 // ... or is posted now (no-op when nothing is deferred)
struct PeerHost;
int peer_init(qexhip_ctx *c, PeerHost &host);          // after the host rendezvous (collective): control block exported, everybody's mapped
int peer_selftest(qexhip_ctx *c);                       // a few mailbox all-reduces with known answers (collective; bounded): 0 = this rank saw the right sums
void peer_destroy(qexhip_ctx *c);
inline int peer_check(qexhip_ctx *c) { return devjoin_check(c); }
int peer_exchange(qexhip_ctx *c, hipStream_t st, int ns_dn, const void *const *src_dn, int ns_up, const void *const *src_up,
                  void *const *dst_from_up, void *const *dst_from_dn, size_t bytes, double emu_us = 0.0, const void **zc_from_up = nullptr,
                  const void **zc_from_dn = nullptr, struct PeerPush *push_only = nullptr);   // push_only (+ zc_*): nothing is launched, the caller's kernel
                                                                                              // pushes and reads the arena (credits owed: peer_ghost_args)
int peer_allreduce_parts(qexhip_ctx *c, double *parts, int n);       // parts[0] := sum over ranks of (sum of parts[0..n) in cg_sum_parts order)
int peer_allreduce(qexhip_ctx *c, double *dptr, int n, int op);      // on the compute stream; op 0 sum, 1 max; rank order
int peer_host_reduce(qexhip_ctx *c, double *host, int n, int op);    // host operands (op 0 max, 1 min, 2 sum), synchronous
int peer_allgather(qexhip_ctx *c, const double *send, double *recv, size_t n);
void peer_info(const qexhip_ctx *c, long out[4]);                    // exchanges, all-reduces, arena growths, arena bytes

// ---- dslash.hip ----
// out[parity] = ca*rin + cb*xs + sgn * sum_mu [ U x(+) - U^+ x(-) ]; optional dot = Re<xs,out> partials
struct DslashOpts {
  double ca = 0, cb = 0;
  const DevField *rin = nullptr;   // a-term source (same parity as out)
  const DevField *xs = nullptr;    // b-term source (same parity as out)
  int neg = 0;                     // 1: subtract the hop sum (stagDM)
  double post = 1.0;               // out *= post  (stagD's 0.5*sc)
  int dot = 0;                     // 1: write block partials of Re<xs,out> and reduce into dot_out
                                   // 2: leave the partials in c->partials[0..*nparts_out) (deferred)
  int *nparts_out = nullptr;
  double *dot_out = nullptr;       // device scalar
  const int *done = nullptr;       // device flag: skip when set
  int pair2 = 0;                   // second sweep of a back-to-back pair out2 = D (D in) (op_xx): its input is the first sweep's output and nothing came
                                   // between -- where the first sweep's boundary launch ran on the comm stream, the faces this sweep sends were
                                   // produced THERE: its exchange is posted at once, only its boundary launch waits for the first sweep's interior
  int defer_join = 0;              // the caller's next operation on the compute stream is comm_allreduce_parts (or devjoin_flush): a sweep split by
                                   // sites leaves its join from the comm stream to that kernel's prologue where the mailboxes carry the sum
};
int dslash_sweep(qexhip_ctx *c, DevField &out, DevField &in, int parity, const DslashOpts &o);
int sweep_autotune(qexhip_ctx *c);      // measure the sweep's forms once per operator shape (collective)
void sweep_plan(const qexhip_ctx *c, int *lo_end, int *hi_beg, int *overlap);   // boundary / interior ranges and the overlap decision
int sweep_form(const qexhip_ctx *c, int overlap);                              // 2 fused / 0 by sites: what an overlapped sweep runs as
double sweep_push_fraction(const qexhip_ctx *c, int interior_sites, int nrhs = 1);   // where in the dispatch order the fused sweep's boundary workgroups go
struct FusedCtl;
int sweep_fused_ctl(qexhip_ctx *c, int nbnd, FusedCtl *F, int nrhs = 1);               // the bookkeeping words + short wait of the next fused launch

// ---- blas.hip ----
int blas_zero(qexhip_ctx *c, DevField &f, int parity);
int blas_copy(qexhip_ctx *c, DevField &dst, const DevField &src, int parity);
int blas_axpy(qexhip_ctx *c, double a, const DevField &x, DevField &y, int parity);
int blas_xpay(qexhip_ctx *c, const DevField &x, double a, DevField &y, int parity);
int blas_scale(qexhip_ctx *c, double a, DevField &y, int parity);
int blas_axpby(qexhip_ctx *c, double a, const DevField &x, double b, const DevField &y, DevField &z, int parity);  // z = a x + b y
int blas_norm2(qexhip_ctx *c, const DevField &x, int parity, double *dev_out);   // rank-global
int blas_redot(qexhip_ctx *c, const DevField &x, const DevField &y, int parity, double *dev_out);
int blas_cdot(qexhip_ctx *c, const DevField &x, const DevField &y, int parity, double *dev_out);    // dev_out[0..1] = Re, Im of <x, y>
int tile_order_table(qexhip_ctx *c, const int **tab, int *chunk);   // blocked (tile, parity) visiting order, layout.hip
int tile_order_plane(qexhip_ctx *c, int mu, int nu, const int **tab, int *chunk);   // order for kernels whose gathers stay in the (mu, nu) plane
int reduce_partials(qexhip_ctx *c, int n, double *dev_out);  // sum partials[0..n) -> dev_out (+ allreduce)
int read_scalars(qexhip_ctx *c, const double *dev, int n, double *host);  // sync readback
int blas_grid(const qexhip_ctx *c, int parity_count);
int blas_delay(hipStream_t st, int us);      // transport emulation: a one-lane kernel that waits `us` microseconds on st
// CG fused kernels
int cg_xpay(qexhip_ctx *c, DevField &p, const DevField &r, int parity, int k, int rolled);
int cg_update(qexhip_ctx *c, DevField &x, DevField &r, const DevField &p, const DevField &Ap, int parity, int k, int ndot);
int cg_close(qexhip_ctx *c, int k);
int cg_init(qexhip_ctx *c, double r2req, int maxits);  // after b2 (dscal[0]) and r2 (dscal[1]) are known
int cg_resume(qexhip_ctx *c, int k, double r2req, int maxits);   // CgState re-entry: new stopping criterion on the kept state

// ---- solver.cpp ----
int get_work(qexhip_ctx *c, int slot, DevField **f);
int op_xx(qexhip_ctx *c, DevField &r, DevField &x, double m2, int par_even, int dot, const int *done, int *ndot = nullptr);
int op_D(qexhip_ctx *c, DevField &r, DevField &x, double m, double sc, double a = 0.0);
int solve_xx_dev(qexhip_ctx *c, DevField &x, DevField &b, double mass, double r2req, int maxits,
                 int par_even, int *iters, double *r2_over_b2, double *hist, int histcap);
int solve_xx_continue_dev(qexhip_ctx *c, DevField &x, double r2req, int maxits, int *iters, double *r2_over_b2, double *hist, int histcap);
int solve_full_dev(qexhip_ctx *c, DevField &x, DevField &b, double mass, double r2req, int maxits,
                   int *iters, double *r2_final, int use_prev = 0);
int solve_xx_multi_dev(qexhip_ctx *c, std::vector<DevField *> &xs, DevField &b, const double *shifts,
                       int nmass, double r2req, int maxits, int par_even, int *iters, double *hist, int histcap);
int solve_multi_dev(qexhip_ctx *c, std::vector<DevField *> &xs, DevField &b, const double *masses,
                    int nmass, double r2req, int maxits, int *iters, double *r2_final);
// persistent multi-shift workspace (multishift.hip): search directions, per-parity solutions, host-entry solutions
enum { POOL_PS = 0, POOL_YS = 32, POOL_XS = 64 };
int pool_field(qexhip_ctx *c, int idx, DevField **f);

// ---- force.hip ----
int stag_outer_host(qexhip_ctx *c, double *f_host, const double *x_host, double se, double so, int accumulate);

// ---- smear.hip ----
int smear_fat7_host(qexhip_ctx *c, const double *g_host, const double coef[5], double *fl_host, double *ll_host, double naik);
int smear_hisq_host(qexhip_ctx *c, const double *g_host, double *fl_host, double *ll_host);
int smear_nhyp_host(qexhip_ctx *c, const double *g_host, double *fl_host, double a1, double a2, double a3);
int smear_set_links_hisq(qexhip_ctx *c, const double *g_host);
int smear_hisq_force_host(qexhip_ctx *c, const double *g_host, const double *dfl_host, const double *dll_host, double *f_host);
int smear_fat7_deriv_host(qexhip_ctx *c, const double *g_host, const double *dfl_host, const double coef[5], const double *dll_host,
                          double naik, double *d_host);
void nhyp_state_free(qexhip_ctx *c);
void hisq_state_free(qexhip_ctx *c);
int hisq_prepare(qexhip_ctx *c, const double *g_host, double *fl_host, double *ll_host);
int hisq_closure_force(qexhip_ctx *c, const double *dfl_host, const double *dll_host, double *f_host);
int gauge_deriv_dev(qexhip_ctx *c, const double2 *G, double2 *F, double cplaq, double c2, int kind);
int stag_outer_dev(qexhip_ctx *c, DevField &fx, double2 *F, double se, double so, int accumulate, int hop = 1);
int hisq_fermion_force(qexhip_ctx *c, double *f_host, const double *const *psi, const double *scale, int n);
int nhyp_gauge_force(qexhip_ctx *c, double *f_host, double cplaq, double c2, int kind);
int nhyp_fermion_force(qexhip_ctx *c, double *f_host, const double *const *psi, const double *scale, int n, int bcmask, const int ph[4]);
int nhyp_prepare(qexhip_ctx *c, const double *g_host, double a1, double a2, double a3, double *fl_host);
int nhyp_force_host(qexhip_ctx *c, double *f_host, const double *chain_host);
int smear_set_links_nhyp(qexhip_ctx *c, const double *g_host, double a1, double a2, double a3, int bcmask, const int ph[4]);

// ---- gauge.hip ----
int gauge_set(qexhip_ctx *c, const double *g);
int gauge_get(qexhip_ctx *c, double *g);
int gauge_plaq(qexhip_ctx *c, double out[6]);
int gauge_force(qexhip_ctx *c, double *f_host, double cplaq, double c2 = 0.0, int kind = 0);
int gauge_wflow(qexhip_ctx *c, int nsteps, double eps, double cplaq = 1.0, double c2 = 0.0, int kind = 0);
int gauge_flow_obs(qexhip_ctx *c, int loop, double out[3], double *plaq6 = nullptr);
int gauge_action(qexhip_ctx *c, double cplaq, double c2, int kind, double *out);
int gauge_md_update(qexhip_ctx *c, const double *p_host, double t);
int gauge_reunit(qexhip_ctx *c);
int md_begin(qexhip_ctx *c, const double *g, const double *p);
int md_end(qexhip_ctx *c, double *g, double *p);
int md_momentum_norm2(qexhip_ctx *c, double *out);
int md_update_links(qexhip_ctx *c, double t);
int md_gauge_force(qexhip_ctx *c, double cplaq, double c2, int kind);
int md_kick(qexhip_ctx *c, int source, double t);
int md_shift_links(qexhip_ctx *c, int source, double t);
int md_save_links(qexhip_ctx *c);
int md_restore_links(qexhip_ctx *c);
int gauge_wline(qexhip_ctx *c, const int *path, int n, double out[2]);
int gauge_polyakov(qexhip_ctx *c, double out[8]);
int gauge_plaq_s4(qexhip_ctx *c, double out[8]);
void gauge_free(qexhip_ctx *c);
void gauge_release_scratch(qexhip_ctx *c);
const double2 *gauge_links_dev(qexhip_ctx *c);   // resident natural-layout links (nullptr before qexhip_gauge_set)
// ---- rng.hip (device-side generation) ----
struct qexhip_rng;
int rng_dev_generate(qexhip_ctx *c, qexhip_rng *R, int what, DevField *f, double2 *P);   // what: 0 gaussian vector, 1 u1 vector, 2 randomTAH -> P
int md_momenta_dev(qexhip_ctx *c, double2 **M);                                          // resident MD momenta (allocated on demand)
