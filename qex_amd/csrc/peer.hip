// peer.hip -- the peer-memory transport: faces pushed straight into the neighbour's HBM through hipIpc-mapped memory,
// sequence-numbered flags instead of a collective library, rank-ordered mailbox all-reduce.
//
// Replaces, like comm.cpp's RCCL arm, the QMP send/recv pairs of the shifts (src/layout/qshifts.nim:51-131,
// src/layout/shifts.nim:67-94,254-285) and the rank sum of threadRankSum (src/comms/commsUtils.nim:195-204).  Two reasons
// for a second transport next to RCCL:
//   * RCCL refuses two ranks on one device; this transport does not care which device a peer's memory lives on, so
//     N processes sharing ONE MI355X run the real multi-rank protocol (rank > 0 kernels, real neighbours) on the one-GPU
//     boxes the test suite gets;
//   * latency: a CG iteration on a thin slab is dominated by fixed costs.  A mailbox all-reduce is one
//     single-workgroup launch (3.5-4.5 us measured against ~21 us for ncclAllReduce of the same scalar,
//     profiles/r05_ipc_probe.log), a face exchange one launch with no proxy thread in the path.
//
// Protocol (every rank runs the same sequence of calls, as with any collective library):
//   channel  = (stream class s: 0 compute stream, 1 comm stream; direction d).  Each rank owns, per class, an inbound
//              ARENA with two halves (messages from the lower / from the upper neighbour) and, in its CONTROL block, one
//              `data` word per inbound channel and one `credit` word per outbound channel.
//   exchange n on a channel, ONE kernel per rank (k_peer_exchange):
//     push    wait until my credit word says the receiver has unpacked message n-1 (its arena half is free), copy the
//             faces into the receiver's arena half, system-scope release, then data := n in the receiver's control block;
//     unpack  wait until my data word says n, system-scope acquire, copy my arena half into the ghost tiles (or any other
//             destination: nothing has to be registered), then credit := n in the sender's control block.
//     The push of exchange n needs only the peer's unpack n-1, which needs only my push n-1 (an earlier kernel on the same
//     stream): no cycle, whatever the residency of the workgroups.
//   all-reduce k, ONE single-workgroup kernel per rank: my operand goes into slot k&3 of EVERY rank's mailbox, then I wait for the
//     N operands in my own mailbox and sum them in rank order -- the same bits on every rank, whatever the arrival order.  A
//     rank can be at most one all-reduce ahead of the slowest one, so four slots never collide.  Up to 32 doubles (the CG's
//     scalars: k_peer_allreduce_small) travel INSIDE sequence-tagged 8-byte words, no fence and no separate flag; longer
//     vectors (k_peer_allreduce) as payload + release + flag.
//   stream join (devjoin_*, used by BOTH transports since round 6): a one-lane kernel raises a device counter behind what a stream has
//     posted, a one-wave kernel makes another stream wait for it -- instead of the runtime's cross-queue event dependency (~20-28 us of
//     dead time per sweep here).
// Every device-side wait is bounded (timeout -> error word in pinned host memory -> QEXHIP_ERR_COMM at the next host
// sync): both processes may share the CUs, and a wave that never exits would take the box down.
#include "qexhip_internal.h"
#include "peer_shm.h"
#include "cg_device.h"
#include "peer_device.h"
#include "../../include/qexhip.h"
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <vector>

static_assert(sizeof(hipIpcMemHandle_t) <= PEER_HANDLE_BYTES, "ipc handle size");

enum { PEER_MAXSEG = 32, PEER_MBOX_N = 4096 };
// control block, in 8-byte words: every polled word on a 128-byte line of its own
enum { CW_DATA = 0 /* + (s*2+d)*16 */, CW_CREDIT = 64 /* + (s*2+dir)*16 */, CW_MFLAG = 256 /* + slot*PEER_MAXR + src */,
       CW_GRAN = 1024 /* + ((slot*PEER_MAXR + src)*PEER_GRAN_N + i)*2 */,
       CTRL_MBOX_BYTE = 65536, CTRL_BYTES = CTRL_MBOX_BYTE + PEER_NSLOT * PEER_MAXR * PEER_MBOX_N * 8 };
static_assert((CW_GRAN + PEER_NSLOT * PEER_MAXR * PEER_GRAN_N * 2) * 8 <= CTRL_MBOX_BYTE, "control block layout");

struct PeerComm {
  PeerHost host;
  int nranks = 1, rank = 0;
  char *ctrl = nullptr;
  char *arena[2]{nullptr, nullptr};      // per stream class: [from lower: cap | from upper: cap]
  size_t cap[2]{0, 0};                   // bytes of ONE half
  char *pctrl[PEER_MAXR]{};              // every rank's control block as mapped here (own pointer at [rank])
  char *parena[2][PEER_MAXR]{};          // arenas of the two neighbours as mapped here
  unsigned int *done = nullptr;          // device: [class][push | unpack] completion counters
  u64 *err = nullptr;                    // = the context's error word (DevJoin): the kernels write it on a timeout
  u64 seq_out[2][2]{}, seq_in[2][2]{}, seq_red = 0;
  struct { int live = 0; u64 *credit_out[2]; u64 seq_in[2]; const u64 *in_flag[2]; long long emu_ticks; } zc;   // the push-only exchange (comm stream class) whose credits are still owed
  long long ticks = 0;                   // timeout in wall_clock64 ticks
  double timeout_s = 30.0;
  long exchanges = 0, allreduces = 0, grows = 0;
  std::vector<char *> retired;           // outgrown arenas: freed at destroy only (see peer_ensure_arena)
};

static inline int upper(const PeerComm *p) { return (p->rank + 1) % p->nranks; }
static inline int lower(const PeerComm *p) { return (p->rank - 1 + p->nranks) % p->nranks; }

// ---------------- device side ----------------
struct PeerXfer {
  const uint4 *src[2][PEER_MAXSEG];   // pieces to [lower | upper]
  uint4 *dst[2][PEER_MAXSEG];         // destinations of the pieces from [lower | upper]
  uint4 *out_arena[2];                // peer-mapped: [lower's from-upper half | upper's from-lower half]
  const uint4 *in_arena[2];           // own: [from-lower half | from-upper half]
  u64 *out_flag[2];                   // peer-mapped data words
  const u64 *credit[2];               // own credit words of the two outbound channels
  const u64 *in_flag[2];              // own data words
  u64 *credit_out[2];                 // peer-mapped credit words of the two senders
  u64 seq_out[2], seq_in[2];
  u64 *err;
  unsigned int *done;                 // [0] push, [1] unpack
  long long ticks;
  long long emu_ticks;                // transport emulation: the inbound data counts as arrived no earlier than this long after the kernel started
  unsigned n16;                       // 16-byte units per piece (< 2^32: pieces below 64 GiB)
  int ns[2], nr[2];
};

__global__ void __launch_bounds__(256) k_peer_exchange(const PeerXfer X) {
  __shared__ int ok;
  const long long t_start = X.emu_ticks > 0 ? wall_clock64() : 0;
  // NOTE all chunk arithmetic is 32-bit on purpose: with a 64-bit `n16 - off < PEER_CHUNK ? n16 - off : PEER_CHUNK` hipcc
  // (ROCm 7.2, gfx950) selected the tail length on a stale SCC (s_cselect_b32 behind a VALU v_cmp_lt_u64): piece 0 of a
  // multi-piece message copied a whole chunk and ran over its neighbour in the arena (profiles/r05_notes.md)
  const unsigned cpp = (X.n16 + PEER_CHUNK - 1) / PEER_CHUNK;   // chunks per piece
  // ---- push ----
  const unsigned nout = (unsigned)(X.ns[0] + X.ns[1]) * cpp;
  if (threadIdx.x == 0) {
    int good = 1;
    for (int dir = 0; dir < 2; dir++)
      if (X.ns[dir] > 0 && !peer_poll_ge(X.credit[dir], X.seq_out[dir] - 1, X.err, X.ticks, 0x100 + dir)) good = 0;
    ok = good;
  }
  __syncthreads();
  if (!ok) return;
  for (unsigned ch = blockIdx.x; ch < nout; ch += gridDim.x) {
    const unsigned q = ch / cpp, j = ch - q * cpp;
    const int dir = q >= (unsigned)X.ns[0];
    const unsigned k = dir ? q - X.ns[0] : q;
    const unsigned off = j * PEER_CHUNK;
    const unsigned n = min(X.n16 - off, (unsigned)PEER_CHUNK);
    peer_copy_chunk(X.out_arena[dir] + (u64)k * X.n16 + off, X.src[dir][k] + off, n);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave: its stores have reached L2 / the fabric
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");        // system scope: this XCD's dirty lines are written back ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // ... before the counter / flag (MI355X_MICROARCH: the compiler may drop this wait)
    const unsigned a = __hip_atomic_fetch_add(&X.done[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (a == gridDim.x - 1) {
      __hip_atomic_store(&X.done[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int dir = 0; dir < 2; dir++)
        if (X.ns[dir] > 0) __hip_atomic_store(X.out_flag[dir], X.seq_out[dir], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // ---- unpack ----
  const unsigned nin = (unsigned)(X.nr[0] + X.nr[1]) * cpp;
  if (threadIdx.x == 0) {
    int good = 1;
    // rehearsal on one GPU: between distinct GPUs the push above would have taken bytes / link bandwidth, and so would the
    // neighbour's -- nothing arrives before that (the local copy that stands in for the push runs inside this window)
    if (X.emu_ticks > 0) while (wall_clock64() - t_start < X.emu_ticks) __builtin_amdgcn_s_sleep(8);
    for (int d = 0; d < 2; d++)
      if (X.nr[d] > 0 && !peer_poll_ge(X.in_flag[d], X.seq_in[d], X.err, X.ticks, 0x200 + d)) good = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ok = good;
  }
  __syncthreads();
  if (!ok) return;
  for (unsigned ch = blockIdx.x; ch < nin; ch += gridDim.x) {
    const unsigned q = ch / cpp, j = ch - q * cpp;
    const int d = q >= (unsigned)X.nr[0];
    const unsigned k = d ? q - X.nr[0] : q;
    const unsigned off = j * PEER_CHUNK;
    const unsigned n = min(X.n16 - off, (unsigned)PEER_CHUNK);
    peer_copy_chunk(X.dst[d][k] + off, X.in_arena[d] + (u64)k * X.n16 + off, n);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every load of the arena has returned before the credit goes out
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned a = __hip_atomic_fetch_add(&X.done[1], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (a == gridDim.x - 1) {
      __hip_atomic_store(&X.done[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int d = 0; d < 2; d++)
        if (X.nr[d] > 0) __hip_atomic_store(X.credit_out[d], X.seq_in[d], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// One wave on the consumer's stream: returns once a device counter has reached `want` (peer_stream_join).  Replaces the
// cross-stream event join (record on the comm stream, wait on the compute stream: ~20 us of dead time per sweep on this
// runtime, profiles/r05_timeline_*.txt); the kernel boundary behind it is the consumer's acquire.
__global__ void k_peer_wait(const u64 *ready, u64 want, u64 *err, long long ticks) {
  if (threadIdx.x == 0) (void)peer_poll_ge(ready, want, err, ticks, 0x400);
}

struct PeerMbox {
  double *box[PEER_MAXR];     // mailbox payload of every rank as mapped here: [slot][src][PEER_MBOX_N]
  u64 *flag[PEER_MAXR];       // mailbox flags of every rank: [slot][src]
  u64 *err;
  long long ticks;
  int nranks, me;
};

// OP 0: sum, 1: max.  x[0..n) := reduction over the ranks, in rank order
template <int OP>
__global__ void __launch_bounds__(512) k_peer_allreduce(double *x, int n, const PeerMbox M, u64 seq) {
  __shared__ int ok;
  const int slot = (int)(seq & (PEER_NSLOT - 1));
  const size_t mine = ((size_t)slot * PEER_MAXR + M.me) * PEER_MBOX_N;
  for (int i = threadIdx.x; i < n; i += 512) {
    const double v = x[i];
    for (int r = 0; r < M.nranks; r++) M.box[r][mine + i] = v;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) ok = 1;
  __syncthreads();
  if ((int)threadIdx.x < M.nranks) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&M.flag[threadIdx.x][slot * PEER_MAXR + M.me], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!peer_poll_ge(&M.flag[M.me][slot * PEER_MAXR + threadIdx.x], seq, M.err, M.ticks, 0x300 + threadIdx.x)) ok = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (!ok) return;
  const double *my = M.box[M.me] + (size_t)slot * PEER_MAXR * PEER_MBOX_N;
  for (int i = threadIdx.x; i < n; i += 512) {
    double acc = my[i];
    for (int r = 1; r < M.nranks; r++) {
      const double v = my[(size_t)r * PEER_MBOX_N + i];
      if (OP == 0) acc += v;
      else acc = (v > acc || v != v) ? v : acc;
    }
    x[i] = acc;
  }
}

// Small all-reduces (the CG's scalars): tagged 8-byte granules, peer_device.h.  Lane (r, i) sends value i to rank r and
// collects value i of rank r; the sums run over the ranks in rank order.
// x[0..n) := reduction over the ranks (OP 0 sum, 1 max), n <= PEER_GRAN_N.  PARTS: x[0..nparts) are workgroup partials whose sum
// (in cg_sum_parts order, the order in which the consumers would have summed the vector themselves) is this rank's ONE operand.
// join != nullptr: the stream this kernel is on has not waited yet for what the other stream posted (a deferred peer_stream_join:
// the partials of the boundary launch come from there) -- one lane polls the join counter first, bounded like every wait here.
template <int OP, bool PARTS>
__global__ void __launch_bounds__(256) k_peer_allreduce_small(double *x, int n, int nparts, const PeerGran G, u64 seq, const u64 *join, u64 joinval) {
  __shared__ double val[PEER_MAXR * PEER_GRAN_N];
  __shared__ int ok;
  if (threadIdx.x == 0) ok = 1;
  if (join) {
    if (threadIdx.x == 0) {
      if (!peer_poll_ge(join, joinval, G.err, G.ticks, 0x400)) ok = 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (!ok) return;
  }
  const long long t_start = G.emu_ticks > 0 ? wall_clock64() : 0;
  double local = 0;
  if (PARTS) { local = cg_sum_parts(x, nparts); n = 1; }
  const int slot = (int)(seq & (PEER_NSLOT - 1));
  const unsigned tag = (unsigned)seq;
  __syncthreads();
  const int r = threadIdx.x / PEER_GRAN_N, i = threadIdx.x % PEER_GRAN_N;      // 256 lanes >= 8 ranks x 32 values; more ranks loop
  for (int rr = r; rr < G.nranks; rr += 256 / PEER_GRAN_N) {
    if (i < n) {
      const double mine = PARTS ? local : x[i];
      gran_send(G.gran[rr] + ((size_t)(slot * PEER_MAXR + G.me) * PEER_GRAN_N + i) * 2, mine, tag);
    }
  }
  if (G.emu_ticks > 0) while (wall_clock64() - t_start < G.emu_ticks) __builtin_amdgcn_s_sleep(2);    // rehearsal: the peers' granules cross xGMI
  for (int rr = r; rr < G.nranks; rr += 256 / PEER_GRAN_N) {
    if (i < n) {
      double v = 0;
      if (!gran_recv(G.gran[G.me] + ((size_t)(slot * PEER_MAXR + rr) * PEER_GRAN_N + i) * 2, tag, &v, G.err, G.ticks, 0x300 + rr)) ok = 0;
      val[rr * PEER_GRAN_N + i] = v;
    }
  }
  __syncthreads();
  if (!ok) return;
  if ((int)threadIdx.x < n) {
    double acc = val[threadIdx.x];
    for (int q = 1; q < G.nranks; q++) {
      const double v = val[q * PEER_GRAN_N + threadIdx.x];
      if (OP == 0) acc += v;
      else acc = (v > acc || v != v) ? v : acc;
    }
    x[threadIdx.x] = acc;
  }
}

// ---------------- host side ----------------
static u64 *ctrl_word(char *ctrl, int w) { return (u64 *)ctrl + w; }

// ---- device-side joins and the error word: context-level, both transports ----
int devjoin_init(qexhip_ctx *c) {
  DevJoin &J = c->dj;
  double tmo = 30.0;
  if (const char *e = getenv("QEXHIP_PEER_TIMEOUT")) { const double v = atof(e); if (v > 0) tmo = v; }
  int khz = 0;
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device);
  if (khz <= 0) khz = 100000;
  J.ticks = (long long)(tmo * 1000.0 * khz);
  J.timeout_s = tmo;
  HIPCHK(hipMalloc((void **)&J.ready, 512));
  HIPCHK(hipMemset(J.ready, 0, 512));
  HIPCHK(hipHostMalloc((void **)&J.err, 64, hipHostMallocDefault));
  *J.err = 0;
  HIPCHK(hipDeviceSynchronize());          // (the context's streams are non-blocking: nothing orders them behind the memset by itself)
  return 0;
}
void devjoin_destroy(qexhip_ctx *c) {
  if (c->dj.ready) (void)hipFree(c->dj.ready);
  if (c->dj.err) (void)hipHostFree(c->dj.err);
  c->dj.ready = nullptr; c->dj.err = nullptr;
}
int devjoin_check(qexhip_ctx *c) {
  if (!c->dj.err) return 0;
  const u64 e = __atomic_load_n(c->dj.err, __ATOMIC_ACQUIRE);
  if (e) {
    const char *what = (e & 0xF00) == 0x100 ? "credit of an outbound channel" : (e & 0xF00) == 0x200 ? "data of an inbound channel"
                       : (e & 0xF00) == 0x400 ? "boundary launch on the comm stream (stream join)"
                       : (e & 0xFF0) == 0x510 ? "neighbours' faces in a fused sweep (its cleanup workgroups: the LONG wait)"
                       : (e & 0xFF0) == 0x520 ? "boundary workgroups of a fused sweep to finish or park" : "all-reduce contribution";
    qexhip_set_error("rank %d timed out on the device waiting for the %s (code 0x%llx): a neighbour is gone, or the ranks "
                     "did not issue the same sequence of exchanges", c->rank, what, e);
    if (c->peer) peer_host_fail(&c->peer->host);
    return QEXHIP_ERR_COMM;
  }
  return 0;
}
static int peer_check_err(qexhip_ctx *c) { return devjoin_check(c); }

static void fill_mbox(const PeerComm *p, PeerMbox &M) {
  for (int r = 0; r < p->nranks; r++) {
    M.box[r] = (double *)(p->pctrl[r] + CTRL_MBOX_BYTE);
    M.flag[r] = ctrl_word(p->pctrl[r], CW_MFLAG);
  }
  M.err = p->err; M.ticks = p->ticks; M.nranks = p->nranks; M.me = p->rank;
}

static void fill_gran(const PeerComm *p, PeerGran &G) {
  for (int r = 0; r < p->nranks; r++) G.gran[r] = ctrl_word(p->pctrl[r], CW_GRAN);
  G.err = p->err; G.ticks = p->ticks; G.nranks = p->nranks; G.me = p->rank;
  G.emu_ticks = 0;
}

static int peer_init_body(qexhip_ctx *c, PeerHost &host);
int peer_init(qexhip_ctx *c, PeerHost &host) {
  const int e = peer_init_body(c, host);
  if (e && c->peer) peer_host_fail(&c->peer->host);      // the other ranks stand in a barrier of the handle exchange: they fail at once instead of timing out
  return e;
}
static int peer_init_body(qexhip_ctx *c, PeerHost &host) {
  PeerComm *p = new PeerComm();
  p->host = host;
  p->nranks = host.nranks; p->rank = host.rank;
  c->peer = p;                                   // from here comm_destroy owns it
  p->ticks = c->dj.ticks;
  p->timeout_s = c->dj.timeout_s;
  p->err = c->dj.err;
  // Everything a PEER writes lives in fine-grained device memory: between distinct GPUs the writes arrive over xGMI behind the
  // local L2's back, and only fine-grained (MTYPE NC) lines are guaranteed to be dropped by a system-scope acquire -- coarse-grained
  // hipMalloc memory is kept coherent for the owning agent only.  (Between processes on ONE device both kinds pass every check of
  // scratch/ipc_probe.cpp; what RCCL allocates for its own receive buffers is the precedent.)
  HIPCHK(hipExtMallocWithFlags((void **)&p->ctrl, CTRL_BYTES, hipDeviceMallocFinegrained));
  HIPCHK(hipMemset(p->ctrl, 0, CTRL_BYTES));
  HIPCHK(hipMalloc((void **)&p->done, 64));
  HIPCHK(hipMemset(p->done, 0, 64));
  HIPCHK(hipDeviceSynchronize());
  PeerShmSlot &me = p->host.shm->s[p->rank];
  if (p->nranks > 1) {
    hipIpcMemHandle_t h;
    HIPCHK(hipIpcGetMemHandle(&h, p->ctrl));
    memcpy(me.handle[0], &h, sizeof h);
    me.cap[0] = CTRL_BYTES;
  }
  CHK(peer_host_barrier(&p->host));
  for (int r = 0; r < p->nranks; r++) {
    if (r == p->rank) { p->pctrl[r] = p->ctrl; continue; }
    hipIpcMemHandle_t h;
    memcpy(&h, (const void *)p->host.shm->s[r].handle[0], sizeof h);
    HIPCHK(hipIpcOpenMemHandle((void **)&p->pctrl[r], h, hipIpcMemLazyEnablePeerAccess));
  }
  CHK(peer_host_barrier(&p->host));
  peer_host_unlink(&p->host);          // every rank has mapped the segment: the name can go (nothing is left behind if the job dies)
  return 0;
}

// the inbound arena of stream class s must hold `bytes` per half.  Growing is COLLECTIVE: by symmetry (equal local
// volumes, same call sequence) every rank is in this call with the same size
static int peer_ensure_arena(qexhip_ctx *c, int s, size_t bytes) {
  PeerComm *p = c->peer;
  if (bytes <= p->cap[s]) return 0;
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipStreamSynchronize(c->cstream));
  CHK(peer_check_err(c));
  CHK(peer_host_barrier(&p->host));               // nobody is writing into anybody's old arena any more
  const int nb[2] = {lower(p), upper(p)};
  for (int k = 0; k < 2; k++) {
    const int r = nb[k];
    if (r != p->rank && p->parena[s][r]) { (void)hipIpcCloseMemHandle(p->parena[s][r]); }
    p->parena[s][r] = nullptr;
  }
  CHK(peer_host_barrier(&p->host));               // nobody maps my old arena any more
  // An outgrown arena is RETIRED, not freed: on this driver (dmabuf IPC) hipIpcGetMemHandle fails with "invalid argument" on
  // a fresh allocation once exported memory has been freed and its address range re-used -- 4 processes, 10th growth,
  // reproduced outside the library (scratch/ipc_probe.cpp regrow mode 1 against mode 2, profiles/r05_ipc_probe.log).
  // Capacities at least double, so everything retired together is smaller than the arena in use.
  if (p->arena[s]) p->retired.push_back(p->arena[s]);
  p->arena[s] = nullptr;
  size_t cap = std::max<size_t>(std::max<size_t>(bytes + bytes / 4, 2 * p->cap[s]), (size_t)1 << 20);
  cap = (cap + 4095) & ~(size_t)4095;
  HIPCHK(hipExtMallocWithFlags((void **)&p->arena[s], 2 * cap, hipDeviceMallocFinegrained));    // written by the neighbours: see peer_init
  PeerShmSlot &me = p->host.shm->s[p->rank];
  if (p->nranks > 1) {
    hipIpcMemHandle_t h;
    HIPCHK(hipIpcGetMemHandle(&h, p->arena[s]));
    memcpy(me.handle[1 + s], &h, sizeof h);
  }
  me.cap[1 + s] = cap;
  CHK(peer_host_barrier(&p->host));
  for (int k = 0; k < 2; k++) {
    const int r = nb[k];
    if (p->host.shm->s[r].cap[1 + s] != cap) {
      qexhip_set_error("peer transport: rank %d sized its arena %llu bytes, rank %d %zu: the ranks' messages differ in size", r,
                       (unsigned long long)p->host.shm->s[r].cap[1 + s], p->rank, cap);
      peer_host_fail(&p->host);
      return QEXHIP_ERR_COMM;
    }
    if (r == p->rank) { p->parena[s][r] = p->arena[s]; continue; }
    if (p->parena[s][r]) continue;               // two ranks: lower == upper, opened once
    hipIpcMemHandle_t h;
    memcpy(&h, (const void *)p->host.shm->s[r].handle[1 + s], sizeof h);
    HIPCHK(hipIpcOpenMemHandle((void **)&p->parena[s][r], h, hipIpcMemLazyEnablePeerAccess));
  }
  CHK(peer_host_barrier(&p->host));
  p->cap[s] = cap;
  p->grows++;
  return 0;
}

// One exchange on stream st: ns_dn pieces to the lower neighbour (they arrive in ITS from-upper half), ns_up pieces to the upper
// one; by symmetry ns_up pieces arrive from the lower neighbour (-> dst_from_dn) and ns_dn from the upper one (-> dst_from_up).
// Every piece is `bytes` long (a multiple of 16).
// push_only (1..4 pieces per direction, as many up as down; comm stream class): NOTHING is launched -- the caller's own kernel pushes (its
// first push_only->nblocks workgroups run peer_push_block) and reads what arrives in the receive arena itself: zc_from_up / zc_from_dn
// return where the faces from the upper / lower neighbour lie there (piece k at k * bytes), and the credits stay owed until that kernel
// returns them (peer_ghost_args).
int peer_exchange(qexhip_ctx *c, hipStream_t st, int ns_dn, const void *const *src_dn, int ns_up, const void *const *src_up,
                  void *const *dst_from_up, void *const *dst_from_dn, size_t bytes, double emu_us, const void **zc_from_up,
                  const void **zc_from_dn, PeerPush *push_only) {
  PeerComm *p = c->peer;
  const bool zc = push_only != nullptr;
  if (zc && (ns_dn != ns_up || ns_dn < 1 || ns_dn > PEER_PUSH_MAXPIECE || !zc_from_up || !zc_from_dn)) {
    qexhip_set_error("peer transport: a push-only exchange takes 1..%d pieces, as many up as down", (int)PEER_PUSH_MAXPIECE);
    return QEXHIP_ERR_ARG;
  }
  if (bytes % 16 != 0) { qexhip_set_error("peer transport: message of %zu bytes is not a multiple of 16", bytes); return QEXHIP_ERR_ARG; }
  if (ns_dn < 0 || ns_up < 0 || (ns_dn == 0 && ns_up == 0) || bytes == 0) return 0;
  const int s = (st == c->cstream) ? 1 : 0;
  CHK(peer_check_err(c));
  for (int k0 = 0; k0 < std::max(ns_dn, ns_up); k0 += (int)PEER_MAXSEG) {
    const int nd = std::max(0, std::min((int)PEER_MAXSEG, ns_dn - k0)), nu = std::max(0, std::min((int)PEER_MAXSEG, ns_up - k0));
    CHK(peer_ensure_arena(c, s, (size_t)std::max(nd, nu) * bytes));
    PeerXfer X;
    memset(&X, 0, sizeof X);
    const int lo = lower(p), up = upper(p);
    if (s == 1 && p->zc.live) { qexhip_set_error("peer transport: an exchange was posted while the credits of a fused sweep's receive are still owed"); return QEXHIP_ERR_STATE; }
    for (int k = 0; k < nd; k++) { X.src[0][k] = (const uint4 *)src_dn[k0 + k]; X.dst[1][k] = zc ? nullptr : (uint4 *)dst_from_up[k0 + k]; }
    for (int k = 0; k < nu; k++) { X.src[1][k] = (const uint4 *)src_up[k0 + k]; X.dst[0][k] = zc ? nullptr : (uint4 *)dst_from_dn[k0 + k]; }
    X.ns[0] = nd; X.ns[1] = nu; X.nr[0] = nu; X.nr[1] = nd;
    if (bytes / 16 >= ((size_t)1 << 32) - PEER_CHUNK) { qexhip_set_error("peer transport: a piece of %zu bytes is too large", bytes); return QEXHIP_ERR_ARG; }
    X.n16 = (unsigned)(bytes / 16);
    X.out_arena[0] = (uint4 *)(p->parena[s][lo] + p->cap[s]);       // the lower neighbour's from-upper half
    X.out_arena[1] = (uint4 *)(p->parena[s][up]);                   // the upper neighbour's from-lower half
    X.in_arena[0] = (const uint4 *)p->arena[s];
    X.in_arena[1] = (const uint4 *)(p->arena[s] + p->cap[s]);
    X.out_flag[0] = ctrl_word(p->pctrl[lo], CW_DATA + (s * 2 + 1) * 16);
    X.out_flag[1] = ctrl_word(p->pctrl[up], CW_DATA + (s * 2 + 0) * 16);
    X.credit[0] = ctrl_word(p->ctrl, CW_CREDIT + (s * 2 + 0) * 16);
    X.credit[1] = ctrl_word(p->ctrl, CW_CREDIT + (s * 2 + 1) * 16);
    X.in_flag[0] = ctrl_word(p->ctrl, CW_DATA + (s * 2 + 0) * 16);
    X.in_flag[1] = ctrl_word(p->ctrl, CW_DATA + (s * 2 + 1) * 16);
    X.credit_out[0] = ctrl_word(p->pctrl[lo], CW_CREDIT + (s * 2 + 1) * 16);   // the lower neighbour's to-upper channel
    X.credit_out[1] = ctrl_word(p->pctrl[up], CW_CREDIT + (s * 2 + 0) * 16);   // the upper neighbour's to-lower channel
    for (int d = 0; d < 2; d++) {
      if (X.ns[d] > 0) X.seq_out[d] = ++p->seq_out[s][d];
      if (X.nr[d] > 0) X.seq_in[d] = ++p->seq_in[s][d];
    }
    X.err = p->err; X.done = p->done + s * 2; X.ticks = p->ticks;
    X.emu_ticks = (long long)(emu_us * 1e-6 * (double)p->ticks / p->timeout_s);
    const size_t cpp = ((size_t)X.n16 + PEER_CHUNK - 1) / PEER_CHUNK;
    const size_t nch = (size_t)(nd + nu) * cpp;
    // Few, fat workgroups: between distinct GPUs the push is bound by one xGMI direction (~45 GB/s: a couple of workgroups
    // saturate it), and beside an interior sweep every workgroup here queues behind that sweep's and slows it down
    // (162 workgroups for two 2.65 MB faces cost 30 us per iteration at 48^3 x 12, profiles/r05_emulated_scaling.log).
    // One workgroup per 64 KiB chunk of either phase, 2..128.
    const size_t tot = (size_t)(nd + nu) * bytes;
    // Residency: a workgroup that has pushed spins in its unpack phase until the neighbour's LAST pushing workgroup is through, so the
    // workgroups of one exchange kernel must all be able to become resident while their siblings spin.  128 of the chip's ~2000
    // slots for this kernel's register budget, on streams that own the whole chip (other kernels' workgroups retire on their own):
    // never put this kernel on a CU-masked stream with fewer slots than its grid -- the round-5 experiment that gave the comm
    // stream 8 CUs of its own deadlocked until the grid was capped (and lost anyway: profiles/r05_comm_cus_experiment.log).
    const int grid = (int)std::min<size_t>(nch, std::max<size_t>(2, std::min<size_t>(128, tot >> 16)));
    if (push_only) {
      PeerPush &P = *push_only;
      for (int d = 0; d < 2; d++) {
        for (int k = 0; k < PEER_PUSH_MAXPIECE; k++) P.src[d][k] = X.src[d][k < nd ? k : 0];
        P.out_arena[d] = X.out_arena[d]; P.out_flag[d] = X.out_flag[d]; P.credit[d] = X.credit[d]; P.seq_out[d] = X.seq_out[d];
      }
      P.npiece = nd;
      P.n16 = X.n16; P.done = X.done; P.t_start_out = (long long *)(c->dj.ready + 56); P.err = X.err; P.ticks = X.ticks; P.nblocks = grid;
      p->zc.live = 1;
      for (int d = 0; d < 2; d++) { p->zc.credit_out[d] = X.credit_out[d]; p->zc.seq_in[d] = X.seq_in[d]; p->zc.in_flag[d] = X.in_flag[d]; }
      p->zc.emu_ticks = X.emu_ticks;
      *zc_from_dn = X.in_arena[0];
      *zc_from_up = X.in_arena[1];
    } else {
      hipLaunchKernelGGL(k_peer_exchange, dim3(grid), dim3(256), 0, st, X);
      HIPCHK(hipGetLastError());
    }
    p->exchanges++;
  }
  return 0;
}

// what the fused sweep's boundary / cleanup workgroups need for the push-only exchange just prepared: the inbound data words, the
// emulated transport time, the credits they owe
int peer_ghost_args(qexhip_ctx *c, PeerGhost *G) {
  PeerComm *p = c->peer;
  memset(G, 0, sizeof *G);
  if (!p || !p->zc.live) { qexhip_set_error("peer transport: no push-only exchange to consume"); return QEXHIP_ERR_STATE; }
  G->err = p->err; G->ticks = p->ticks;
  for (int d = 0; d < 2; d++) {
    G->flag[d] = p->zc.in_flag[d]; G->flagval[d] = p->zc.seq_in[d];
    G->credit[d] = p->zc.credit_out[d]; G->credit_val[d] = p->zc.seq_in[d];
  }
  G->t_start = (const long long *)(c->dj.ready + 56);
  G->emu_ticks = p->zc.emu_ticks;
  p->zc.live = 0;
  return 0;
}

__global__ void k_peer_set(u64 *flag, u64 val) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// Device-side join of two streams of this context without an event: `signal` raises a counter behind everything posted on
// `from` so far (a one-lane kernel: the kernel boundary in front of it is the release), `wait` makes `waiter` wait for the
// latest value (a one-wave kernel with a bounded poll: the boundary behind it is the acquire).  One counter per direction.
int devjoin_signal(qexhip_ctx *c, hipStream_t from) {
  const int k = (from == c->cstream) ? 1 : 0;
  hipLaunchKernelGGL(k_peer_set, dim3(1), dim3(64), 0, from, c->dj.ready + k * 16, ++c->dj.seq[k]);
  HIPCHK(hipGetLastError());
  return 0;
}
// the compute stream's wait for what the comm stream has signalled so far is postponed into the next peer_allreduce_parts
// (the CG's <p,Ap> right behind the second sweep); devjoin_flush posts it as a kernel of its own if something else comes first
int devjoin_defer(qexhip_ctx *c) { c->dj.deferred = 1; return 0; }
int devjoin_flush(qexhip_ctx *c) {
  if (!c->dj.deferred) return 0;
  return devjoin_wait(c, c->stream, c->cstream);
}
int devjoin_wait(qexhip_ctx *c, hipStream_t waiter, hipStream_t from) {
  const int k = (from == c->cstream) ? 1 : 0;
  if (waiter == c->stream && k == 1) c->dj.deferred = 0;
  hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(64), 0, waiter, c->dj.ready + k * 16, c->dj.seq[k], c->dj.err, c->dj.ticks);
  HIPCHK(hipGetLastError());
  return 0;
}

int peer_allreduce(qexhip_ctx *c, double *dptr, int n, int op) {
  PeerComm *p = c->peer;
  CHK(peer_check_err(c));
  CHK(devjoin_flush(c));
  if (n <= PEER_GRAN_N) {
    PeerGran G;
    fill_gran(p, G);
    G.emu_ticks = (long long)(c->emu_allreduce_us * 1e-6 * (double)p->ticks / p->timeout_s);
    if (op == 0) hipLaunchKernelGGL((k_peer_allreduce_small<0, false>), dim3(1), dim3(256), 0, c->stream, dptr, n, 0, G, ++p->seq_red, (const u64 *)nullptr, (u64)0);
    else hipLaunchKernelGGL((k_peer_allreduce_small<1, false>), dim3(1), dim3(256), 0, c->stream, dptr, n, 0, G, ++p->seq_red, (const u64 *)nullptr, (u64)0);
    p->allreduces++;
    HIPCHK(hipGetLastError());
    return 0;
  }
  PeerMbox M;
  fill_mbox(p, M);
  for (int i0 = 0; i0 < n; i0 += PEER_MBOX_N) {
    const int m = std::min<int>(PEER_MBOX_N, n - i0);
    const u64 seq = ++p->seq_red;
    if (op == 0) hipLaunchKernelGGL(k_peer_allreduce<0>, dim3(1), dim3(512), 0, c->stream, dptr + i0, m, M, seq);
    else hipLaunchKernelGGL(k_peer_allreduce<1>, dim3(1), dim3(512), 0, c->stream, dptr + i0, m, M, seq);
    p->allreduces++;
  }
  HIPCHK(hipGetLastError());
  return 0;
}

int peer_allreduce_parts(qexhip_ctx *c, double *parts, int n) {
  PeerComm *p = c->peer;
  CHK(peer_check_err(c));
  PeerGran G;
  fill_gran(p, G);
  G.emu_ticks = (long long)(c->emu_allreduce_us * 1e-6 * (double)p->ticks / p->timeout_s);
  const u64 *join = c->dj.deferred ? c->dj.ready + 16 : nullptr;      // the deferred join from the comm stream rides in this kernel's prologue
  c->dj.deferred = 0;
  hipLaunchKernelGGL((k_peer_allreduce_small<0, true>), dim3(1), dim3(256), 0, c->stream, parts, 1, n, G, ++p->seq_red, join, c->dj.seq[1]);
  p->allreduces++;
  HIPCHK(hipGetLastError());
  return 0;
}

// A few mailbox all-reduces with known answers, right after the control blocks were mapped (collective): rank r contributes (r + 1) * k,
// every rank must read N (N + 1) / 2 * k.  Between distinct GPUs this is the first time a granule crosses xGMI: a mapping that "works" but
// is not coherent shows up here -- as a wrong sum or as a wait that runs out (bounded: 5 s instead of QEXHIP_PEER_TIMEOUT) -- and
// comm_init then leaves the rank sums to RCCL instead of finding out in the middle of a solve.  Returns 0 when THIS rank saw every
// sum right; the caller agrees the outcome over the ranks.
int peer_selftest(qexhip_ctx *c) {
  PeerComm *p = c->peer;
  double *d = &c->dscal[60];
  const long long saved = p->ticks;
  p->ticks = (long long)(5.0 * (double)saved / p->timeout_s);
  int bad = 0;
  for (int k = 1; k <= 8 && !bad; k++) {
    const double mine = (double)(p->rank + 1) * k;
    if (hipMemcpyAsync(d, &mine, sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) { bad = 1; break; }
    if (peer_allreduce(c, d, 1, 0)) { bad = 1; break; }
    double got = 0;
    if (hipMemcpyAsync(&got, d, sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { bad = 1; break; }
    if (__atomic_load_n(p->err, __ATOMIC_ACQUIRE) != 0 || got != 0.5 * p->nranks * (p->nranks + 1) * k) bad = 1;
  }
  p->ticks = saved;
  if (bad) (void)hipGetLastError();
  return bad;
}

int peer_host_reduce(qexhip_ctx *c, double *host, int n, int op) { return peer_host_allreduce(&c->peer->host, host, n, op); }

// rank-ordered concatenation by a ring of N-1 one-way exchanges: step k forwards the segment of rank (me - k) upwards
int peer_allgather(qexhip_ctx *c, const double *send, double *recv, size_t n) {
  PeerComm *p = c->peer;
  HIPCHK(hipMemcpyAsync(recv + (size_t)p->rank * n, send, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  for (int k = 0; k + 1 < p->nranks; k++) {
    const int have = (p->rank - k + p->nranks) % p->nranks, get = (p->rank - k - 1 + p->nranks) % p->nranks;
    const void *src = recv + (size_t)have * n;
    void *dst = recv + (size_t)get * n;
    CHK(peer_exchange(c, c->stream, 0, nullptr, 1, &src, nullptr, &dst, n * sizeof(double), 0.0));
  }
  return 0;
}

void peer_info(const qexhip_ctx *c, long out[4]) {
  const PeerComm *p = c->peer;
  out[0] = p ? p->exchanges : 0; out[1] = p ? p->allreduces : 0; out[2] = p ? p->grows : 0;
  out[3] = p ? (long)(p->cap[0] + p->cap[1]) * 2 : 0;
}

void peer_destroy(qexhip_ctx *c) {
  PeerComm *p = c->peer;
  if (!p) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  // nobody unmaps or frees while a neighbour may still be writing; a short bounded wait -- a peer that died is not waited for
  p->host.timeout_s = std::min(p->host.timeout_s, 10.0);
  const bool all_here = p->host.shm && peer_host_barrier(&p->host) == 0;
  for (int r = 0; r < p->nranks; r++) {
    if (r == p->rank) continue;
    if (p->pctrl[r]) (void)hipIpcCloseMemHandle(p->pctrl[r]);
    for (int s = 0; s < 2; s++) if (p->parena[s][r]) (void)hipIpcCloseMemHandle(p->parena[s][r]);
  }
  if (all_here) (void)peer_host_barrier(&p->host);
  for (int s = 0; s < 2; s++) if (p->arena[s]) (void)hipFree(p->arena[s]);
  for (char *a : p->retired) (void)hipFree(a);
  if (p->ctrl) (void)hipFree(p->ctrl);
  if (p->done) (void)hipFree(p->done);
  peer_host_close(&p->host);
  delete p;
  c->peer = nullptr;
}
