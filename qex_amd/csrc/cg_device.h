// cg_device.h -- device-side bookkeeping of the CG iteration shared by the kernel that opens an iteration (k_cg_xpay) and
// the one that closes a chunk (k_cg_close), both in blas.hip.
#pragma once
#include "qexhip_internal.h"
#include "reduce.h"

__device__ __forceinline__ double cg_sum_parts(const double *parts, int n) {
  double a = 0;
  for (int i = threadIdx.x; i < n; i += 256) a += parts[i];
  return block_sum_256_all(a);
}
__device__ __forceinline__ void cg_roll(CgScal *s, int k, double r2k, double *hist, int histcap) {
  const int cur = k & 1;
  // the last LIVE bookkeeping is what a re-entry continues from (cg.nim:256-261: state.r2 / rzold / iterations); the carries
  // that follow a finished solve copy slot to slot and would lose rzold
  s->rzo = s->r2s[cur ^ 1]; s->r2 = r2k; s->itn = k;
  s->r2s[cur] = r2k;
  s->itns[cur] = k;
  s->dones[cur] = !(k < s->maxits && r2k > s->r2stop);
  if (k < histcap) hist[k] = r2k / s->b2;
  s->agree[0] = r2k; s->agree[1] = -r2k; s->agree[2] = (double)k; s->agree[3] = -(double)k;
}
__device__ __forceinline__ void cg_carry(CgScal *s, int k) {   // finished earlier: carry the final state forward
  const int cur = k & 1, prv = cur ^ 1;
  s->r2s[cur] = s->r2s[prv];
  s->itns[cur] = s->itns[prv];
  s->dones[cur] = 1;
  s->agree[0] = s->r2s[prv]; s->agree[1] = -s->r2s[prv]; s->agree[2] = (double)s->itns[prv]; s->agree[3] = -(double)s->itns[prv];
}
