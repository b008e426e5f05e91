// reduce.h -- wavefront (64 lanes) + workgroup reductions in fp64.
// Replaces threadSum/simdSum of src/base/threading.nim:291-316: lane partials are combined by
// DPP/shuffle inside the wavefront, then across the 4 wavefronts of a 256-thread workgroup
// through LDS; one value per workgroup is written and summed in a fixed order by k_reduce_final
// (blas.hip), so results are run-to-run deterministic.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}

// sum over a 256-thread workgroup; result valid in thread 0
__device__ __forceinline__ double block_sum_256(double v) {
  __shared__ double sm[4];
  v = wave_sum(v);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sm[w] = v;
  __syncthreads();
  double r = 0;
  if (threadIdx.x == 0) r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  __syncthreads();
  return r;
}

// same, result broadcast to every thread of the workgroup
__device__ __forceinline__ double block_sum_256_all(double v) {
  __shared__ double smb[4];
  v = wave_sum(v);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) smb[w] = v;
  __syncthreads();
  double r = (smb[0] + smb[1]) + (smb[2] + smb[3]);
  __syncthreads();
  return r;
}
