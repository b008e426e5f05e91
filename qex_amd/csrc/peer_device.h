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
enum { PEER_NSLOT = 4,            // all-reduce k uses mailbox slot k & 3: a rank is at most one all-reduce ahead of the slowest one
       PEER_GRAN_N = 32 };        // doubles per small all-reduce: 2 tagged 8-byte granules each

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
