// peer_device.h -- the device side of the peer transport that OTHER kernels carry (peer.hip has the transport's own kernels): bounded
// polls, the tagged granules of the small all-reduce, and the push / wait / park pieces of the fused sweep (dslash.hip).
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

// The push half of a face exchange as workgroups of SOMEBODY ELSE's launch (the fused sweeps: dslash.hip, batch.hip): npiece pieces per
// direction (one per system of a lock-step batch), piece k at k * n16 of the neighbour's arena half; workgroup blk of nblocks copies
// its chunks, the last one through raises the data words.
enum { PEER_PUSH_MAXPIECE = 4 };
struct PeerPush {
  const uint4 *src[2][PEER_PUSH_MAXPIECE];   // my faces: [to lower | to upper][piece]
  uint4 *out_arena[2];                // peer-mapped halves they go to
  u64 *out_flag[2];                   // peer-mapped data words
  const u64 *credit[2];               // my credit words of the two outbound channels
  u64 seq_out[2];
  unsigned n16;                       // 16-byte units per piece
  int npiece;
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
  const unsigned per_dir = (unsigned)P.npiece * cpp;
  for (unsigned ch = blk; ch < 2 * per_dir; ch += (unsigned)P.nblocks) {
    const int dir = ch >= per_dir;
    const unsigned q = dir ? ch - per_dir : ch;
    const unsigned k = q / cpp, off = (q - k * cpp) * PEER_CHUNK;
    const unsigned n = min(P.n16 - off, (unsigned)PEER_CHUNK);
    const uint4 *sp = P.src[dir][0];
#pragma unroll
    for (int j = 1; j < PEER_PUSH_MAXPIECE; j++) sp = (unsigned)j == k ? P.src[dir][j] : sp;     // (selects, not an indexed read of the kernel arguments)
    peer_copy_chunk(P.out_arena[dir] + (u64)k * P.n16 + off, sp + off, n);
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

// What the boundary workgroups of the fused sweep (dslash.hip: k_dslash_fused) need from the transport: the two inbound data words they
// poll themselves (nothing is posted behind the push, nobody stays for it), under emulation the transport time counted from the
// push's start, and the two credit words that go back to the senders once every reader of the arena is through.
struct PeerGhost {
  const u64 *flag[2]; u64 flagval[2];
  const long long *t_start; long long emu_ticks;
  u64 *err; long long ticks;
  u64 *credit[2]; u64 credit_val[2];
};
int peer_ghost_args(qexhip_ctx *c, PeerGhost *G);   // for the push-only exchange just prepared on the comm stream class: takes over the owed credits; peer.hip

// one look: the flags; once they are in (remembered in `seen`), under emulation only the clock -- the transfer between two GPUs would
// take emu_ticks from the push's start -- so that a rehearsal's waiting lanes do not keep hammering the flags' memory channel
__device__ inline bool peer_ghost_ready(const PeerGhost &G, bool &seen, long long &t0e) {
  if (!seen) {
    if (__hip_atomic_load(G.flag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < G.flagval[0]) return false;
    if (__hip_atomic_load(G.flag[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < G.flagval[1]) return false;
    seen = true;
    if (G.emu_ticks > 0) t0e = __hip_atomic_load(G.t_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return G.emu_ticks <= 0 || wall_clock64() - t0e >= G.emu_ticks;
}
// The SHORT wait of a boundary workgroup between its local and its slab-leaving hops, one lane: true when the faces are in.  False
// after `spin_ticks` (about the estimated transfer time), or at once when another workgroup of this launch has already waited that
// long in vain (`late`): the caller then parks its accumulator and leaves the slot to others -- on a chip shared with the neighbour's
// process, to the very kernel it is waiting for.  Never an error: a lost peer is found by the cleanup workgroups' long wait.
__device__ inline bool peer_ghost_try(const PeerGhost &G, long long spin_ticks, unsigned int *late) {
  bool seen = false;
  long long t0e = 0;
  if (peer_ghost_ready(G, seen, t0e)) return true;
  if (spin_ticks <= 0 || __hip_atomic_load(late, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
  const long long t0 = wall_clock64();
  for (unsigned it = 1;; it++) {
    __builtin_amdgcn_s_sleep(4);
    if (peer_ghost_ready(G, seen, t0e)) return true;
    if ((it & 15) == 0) {
      if (__hip_atomic_load(late, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
      if (wall_clock64() - t0 > spin_ticks) {
        __hip_atomic_store(late, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
}
// the LONG wait (cleanup workgroups; bounded by QEXHIP_PEER_TIMEOUT): false = the neighbour is gone, error word set
__device__ inline bool peer_ghost_wait(const PeerGhost &G) {
  if (!peer_poll_ge(G.flag[0], G.flagval[0], G.err, G.ticks, 0x510) || !peer_poll_ge(G.flag[1], G.flagval[1], G.err, G.ticks, 0x511)) return false;
  if (G.emu_ticks > 0) {
    const long long t0 = __hip_atomic_load(G.t_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (wall_clock64() - t0 < G.emu_ticks) __builtin_amdgcn_s_sleep(8);
  }
  return true;
}

// Bookkeeping of ONE fused launch in device memory of the context (dslash.hip): every boundary workgroup counts itself in `dec` once
// it has finished or parked; a parked one appends its block to `list` first.  All words are zero between launches (the last cleanup
// workgroup resets them; launches of one stream do not overlap).
struct FusedCtl {
  unsigned int *dec, *ndef, *cl_done, *late;     // each on a 128-byte line of its own
  unsigned int *list;                            // parked blocks (logical workgroup numbers), capacity >= boundary workgroups of the launch
  long long spin_ticks;                          // the short wait's bound; < 0: park unconditionally (test hook: the cleanup path on every block)
  int ncl;                                       // cleanup workgroups at the end of the grid (>= 1)
};
// (coarse on purpose, ~1 us between looks: up to 32 cleanup workgroups poll ONE word for as long as the sweep runs when the whole
// launch is resident at once, and every look is an L2 atomic on the channel the boundary workgroups' counts go through)
__device__ inline bool peer_poll_u32(const unsigned int *p, unsigned int want, u64 *err, long long ticks, u64 code) {
  if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
  const long long t0 = wall_clock64();
  for (unsigned it = 1;; it++) {
    __builtin_amdgcn_s_sleep(32);
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
    if ((it & 255) == 0) {
      if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return false;
      if (wall_clock64() - t0 > ticks) { __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return false; }
    }
  }
}
