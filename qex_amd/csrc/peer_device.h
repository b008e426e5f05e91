// peer_device.h -- the device side of the peer transport's granule all-reduce (peer.hip), shared with the CG kernel that carries
// one in its tail (blas.hip: k_cg_update).
//
// Small all-reduces (the CG's scalars): the payload travels INSIDE the flags.  A double is cut into two 8-byte granules
// {32 data bits, 32-bit tag = low half of the sequence number}; an aligned 8-byte store is one transaction, so a granule is
// either the old one (tag of all-reduce k-4 in this slot) or the new one, never a mixture -- no release before, no acquire
// after, no separate flag (MI355X_MICROARCH "handoff-1to1": data-tagged granules).
#pragma once
#include "qexhip_internal.h"
#include "peer_shm.h"

typedef unsigned long long u64;
enum { PEER_CHUNK = 4096,         // 16-byte units per copy chunk (one workgroup pass: 16 loads in flight per lane)
       PEER_NSLOT = 4,            // all-reduce k uses mailbox slot k & 3: a rank is at most one all-reduce ahead of the slowest one
       PEER_GRAN_N = 32 };        // doubles per small all-reduce: 2 tagged 8-byte granules each

// bounded wait for a monotonic word (flags, credits, join counters): false after `ticks` of the wall clock or as soon as another
// wait of this rank has given up; the error word (pinned host memory) carries the code of the first failure
__device__ inline bool peer_poll_ge(const u64 *p, u64 want, u64 *err, long long ticks, u64 code) {
  if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= want) return true;
  const long long t0 = wall_clock64();
  for (unsigned it = 1;; it++) {
    __builtin_amdgcn_s_sleep(4);
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= want) return true;
    if ((it & 255) == 0) {
      if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return false;    // somebody gave up already
      if (wall_clock64() - t0 > ticks) {
        __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return false;
      }
    }
  }
}

__device__ inline void gran_send(u64 *dst, double v, unsigned tag) {
  const u64 b = (u64)__double_as_longlong(v);
  __hip_atomic_store(dst, (b & 0xffffffff00000000ULL) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(dst + 1, (b << 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ inline bool gran_recv(const u64 *src, unsigned tag, double *v, u64 *err, long long ticks, u64 code) {
  u64 hi = 0, lo = 0;
  const long long t0 = wall_clock64();
  for (unsigned it = 1;; it++) {
    hi = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    lo = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((unsigned)hi == tag && (unsigned)lo == tag) break;
    __builtin_amdgcn_s_sleep(2);
    if ((it & 255) == 0) {
      if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return false;
      if (wall_clock64() - t0 > ticks) { __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return false; }
    }
  }
  *v = __longlong_as_double((long long)((hi & 0xffffffff00000000ULL) | (lo >> 32)));
  return true;
}

struct PeerGran {
  u64 *gran[PEER_MAXR];       // granule area of every rank as mapped here: [slot][src][PEER_GRAN_N][2]
  u64 *err;
  long long ticks, emu_ticks;
  int nranks, me;
};

// One double, called by all 256 threads of ONE workgroup: `local` (the same value in every thread) goes into slot seq & 3 of every
// rank's mailbox, the N operands are collected from the own mailbox and summed in RANK ORDER -- the same bits on every rank,
// whatever the arrival order.  Returns false (and leaves *out alone) after a timeout; the error word is set.
// emu_ticks > 0 (one-rank rehearsal): the peers' granules count as arrived no earlier than that long after the send.
__device__ inline bool gran_allreduce_block(double local, const PeerGran &G, u64 seq, double *out) {
  __shared__ double gr_val[PEER_MAXR];
  __shared__ int gr_ok;
  const int slot = (int)(seq & (PEER_NSLOT - 1));
  const unsigned tag = (unsigned)seq;
  if (threadIdx.x == 0) gr_ok = 1;
  __syncthreads();
  const int r = threadIdx.x;
  const long long t0 = G.emu_ticks > 0 ? wall_clock64() : 0;
  if (r < G.nranks) gran_send(G.gran[r] + ((size_t)(slot * PEER_MAXR + G.me) * PEER_GRAN_N) * 2, local, tag);
  if (G.emu_ticks > 0) while (wall_clock64() - t0 < G.emu_ticks) __builtin_amdgcn_s_sleep(2);
  if (r < G.nranks) {
    double v = 0;
    if (!gran_recv(G.gran[G.me] + ((size_t)(slot * PEER_MAXR + r) * PEER_GRAN_N) * 2, tag, &v, G.err, G.ticks, 0x300 + r)) gr_ok = 0;
    gr_val[r] = v;
  }
  __syncthreads();
  if (!gr_ok) return false;
  double acc = gr_val[0];
  for (int q = 1; q < G.nranks; q++) acc += gr_val[q];
  *out = acc;
  return true;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ inline void peer_copy_chunk(uint4 *__restrict__ d4, const uint4 *__restrict__ s4, unsigned n) {
  // n <= PEER_CHUNK units, 256 lanes.  Whole chunks: 16 loads (256 B) in flight per lane before the first store -- beside an
  // HBM-saturating sweep a copy gets bandwidth in proportion to the requests it keeps outstanding
  u32x4 *__restrict__ d = (u32x4 *)d4;
  const u32x4 *__restrict__ s = (const u32x4 *)s4;
  if (n == PEER_CHUNK) {
    u32x4 v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = s[threadIdx.x + 256 * j];
    // all 16 loads issued before the first store: hipcc otherwise sinks every load to its store (one 16-byte request in flight
    // per lane; seen in the ISA).  An empty asm that "modifies" the values pins them in registers at this point.
#pragma unroll
    for (int j = 0; j < 16; j++) asm volatile("" : "+v"(v[j]));
#pragma unroll
    for (int j = 0; j < 16; j++) d[threadIdx.x + 256 * j] = v[j];
  } else {
    for (unsigned i = threadIdx.x; i < n; i += 256) d[i] = s[i];
  }
}

// The push half of a face exchange as workgroups of SOMEBODY ELSE's launch (the fused hop-split sweep: dslash.hip): one piece per
// direction; workgroup blk of nblocks copies its chunks into the neighbours' arenas, the last one through raises the data words.
struct PeerPush {
  const uint4 *src[2];                // my faces: [to lower | to upper]
  uint4 *out_arena[2];                // peer-mapped halves they go to
  u64 *out_flag[2];                   // peer-mapped data words
  const u64 *credit[2];               // my credit words of the two outbound channels
  u64 seq_out[2];
  unsigned n16;                       // 16-byte units per piece
  unsigned int *done;                 // completion counter of the pushing workgroups
  long long *t_start_out;             // when the push started (emulated transport time counts from here)
  u64 *err; long long ticks;
  int nblocks;                        // 0: no push in this launch
};
__device__ inline void peer_push_block(const PeerPush &P, const unsigned blk) {
  __shared__ int push_ok;
  if (blk == 0 && threadIdx.x == 0) __hip_atomic_store(P.t_start_out, wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0) {
    int good = 1;
    for (int dir = 0; dir < 2; dir++)
      if (!peer_poll_ge(P.credit[dir], P.seq_out[dir] - 1, P.err, P.ticks, 0x100 + dir)) good = 0;
    push_ok = good;
  }
  __syncthreads();
  if (!push_ok) return;
  const unsigned cpp = (P.n16 + PEER_CHUNK - 1) / PEER_CHUNK;       // 32-bit chunk arithmetic on purpose: see k_peer_exchange
  for (unsigned ch = blk; ch < 2 * cpp; ch += (unsigned)P.nblocks) {
    const int dir = ch >= cpp;
    const unsigned off = (dir ? ch - cpp : ch) * PEER_CHUNK;
    const unsigned n = min(P.n16 - off, (unsigned)PEER_CHUNK);
    peer_copy_chunk(P.out_arena[dir] + off, P.src[dir] + off, n);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave: its stores have reached L2 / the fabric
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");        // system scope: this XCD's dirty lines are written back ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // ... before the counter / flag
    const unsigned a = __hip_atomic_fetch_add(P.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (a == (unsigned)P.nblocks - 1) {
      __hip_atomic_store(P.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int dir = 0; dir < 2; dir++) __hip_atomic_store(P.out_flag[dir], P.seq_out[dir], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// The same reduction in the PROLOGUE of a many-workgroup consumer (k_cg_update for <p,Ap>, k_cg_xpay / k_cg_close for |r|^2): every
// workgroup holds the same `local` (it summed the same partials in the same order); workgroup 0 sends it, every workgroup collects
// the N operands from the own mailbox and sums them in rank order -- no all-reduce launch, no launch boundary.  Every workgroup of
// the launch spins until the slowest rank has sent: only where the ranks have a GPU each, or the launch is small (blas.hip decides).
struct PeerFold {
  int on;
  u64 seq;
  long long *t_send;          // under emulation: when workgroup 0 sent (the transport time counts from there)
  PeerGran G;
};
__device__ inline bool gran_allreduce_grid(double local, const PeerFold &F, double *out) {
  __shared__ double gf_val[PEER_MAXR];
  __shared__ int gf_ok;
  const int slot = (int)(F.seq & (PEER_NSLOT - 1));
  const unsigned tag = (unsigned)F.seq;
  const int r = threadIdx.x;
  if (threadIdx.x == 0) gf_ok = 1;
  __syncthreads();
  if (blockIdx.x == 0) {
    if (r == 0 && F.G.emu_ticks > 0) __hip_atomic_store(F.t_send, wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (r < F.G.nranks) gran_send(F.G.gran[r] + ((size_t)(slot * PEER_MAXR + F.G.me) * PEER_GRAN_N) * 2, local, tag);
  }
  if (r < F.G.nranks) {
    double v = 0;
    if (!gran_recv(F.G.gran[F.G.me] + ((size_t)(slot * PEER_MAXR + r) * PEER_GRAN_N) * 2, tag, &v, F.G.err, F.G.ticks, 0x300 + r)) gf_ok = 0;
    gf_val[r] = v;
  }
  if (F.G.emu_ticks > 0 && r == 0) {       // rehearsal: the peers' granules cross xGMI (all operands are in: workgroup 0 has sent)
    const long long t0 = __hip_atomic_load(F.t_send, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (wall_clock64() - t0 < F.G.emu_ticks) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
  if (!gf_ok) return false;
  double acc = gf_val[0];
  for (int q = 1; q < F.G.nranks; q++) acc += gf_val[q];
  *out = acc;
  return true;
}
int peer_fold_args(qexhip_ctx *c, PeerFold *F);     // fills F for ONE all-reduce (sequence number taken); peer.hip

// The |r|^2 all-reduce of a sharded CG iteration inside the tail of k_cg_update (blas.hip): the workgroup whose arrival ticket comes
// last sums the partials and runs gran_allreduce_block -- one launch and one launch boundary less per iteration, and still only
// ONE spinning workgroup per rank (ranks may share a device).  on == 0: the kernel leaves its partials for comm_allreduce_parts.
struct PeerTail {
  int on;
  unsigned int *count;        // arrival tickets (device, zero between launches)
  u64 seq;
  PeerGran G;
};
int peer_tail_args(qexhip_ctx *c, PeerTail *T);     // fills T for ONE all-reduce on the compute stream (sequence number taken); peer.hip

// What the hop-split boundary launch of an overlapped sweep (dslash.hip: k_dslash<..., PART = 2>) needs from the transport: the
// counter the comm stream raises behind its exchange kernel (join == nullptr: ordered by an event instead, RCCL arm), and -- zero-copy
// receive -- the two credit words its last workgroup writes once every workgroup has read the arena (ticket == nullptr: unpacked).
struct PeerGhost {
  const u64 *join; u64 joinval;
  const u64 *flag[2]; u64 flagval[2];            // instead of join: the two inbound data words of a zero-copy exchange whose kernel did not stay
  const long long *t_start; long long emu_ticks; // ... and, under emulation, the transport time counted from that kernel's start
  u64 *err; long long ticks;
  unsigned int *ticket;
  u64 *credit[2]; u64 credit_val[2];
};
int peer_ghost_args(qexhip_ctx *c, PeerGhost *G, bool zc, bool direct = false);   // after peer_stream_signal(c, c->cstream), or direct; zc: takes over the owed credits; peer.hip

// the wait of a consumer for "the faces are in", one lane: true when they are
__device__ inline bool peer_ghost_wait(const PeerGhost &G) {
  if (G.flag[0]) {
    if (!peer_poll_ge(G.flag[0], G.flagval[0], G.err, G.ticks, 0x510) || !peer_poll_ge(G.flag[1], G.flagval[1], G.err, G.ticks, 0x511)) return false;
    if (G.emu_ticks > 0) {
      const long long t0 = __hip_atomic_load(G.t_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (wall_clock64() - t0 < G.emu_ticks) __builtin_amdgcn_s_sleep(8);
    }
    return true;
  }
  return !G.join || peer_poll_ge(G.join, G.joinval, G.err, G.ticks, 0x500);
}
