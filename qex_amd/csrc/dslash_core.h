// dslash_core.h -- device pieces shared by the Dslash kernels (dslash.hip, batch.hip)
#pragma once
#include <hip/hip_runtime.h>

typedef double d2v __attribute__((ext_vector_type(2)));

// acc += U v  (SUB = false)  /  acc -= U v  (SUB = true); every term one v_fma_f64
template <bool SUB>
__device__ __forceinline__ void mv3(double2 acc[3], const double2 U[9], const double2 v[3]) {
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      if (!SUB) {
        acc[i].x += U[3 * i + j].x * v[j].x;
        acc[i].x -= U[3 * i + j].y * v[j].y;
        acc[i].y += U[3 * i + j].x * v[j].y;
        acc[i].y += U[3 * i + j].y * v[j].x;
      } else {
        acc[i].x -= U[3 * i + j].x * v[j].x;
        acc[i].x += U[3 * i + j].y * v[j].y;
        acc[i].y -= U[3 * i + j].x * v[j].y;
        acc[i].y -= U[3 * i + j].y * v[j].x;
      }
    }
  }
}

// rows 0,1 of a link -> full link: row 2 = det * conj(row0 x row1); det = +-1 (format 1) or U[6] (format 2)
template <int FMT>
__device__ __forceinline__ void recon_row2(double2 U[9], bool neg) {
  const double2 ph = U[6];
  double2 r2[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int a = (k + 1) % 3, b = (k + 2) % 3;
    double rx = U[a].x * U[3 + b].x;
    rx -= U[a].y * U[3 + b].y;
    rx -= U[b].x * U[3 + a].x;
    rx += U[b].y * U[3 + a].y;
    double ry = U[b].x * U[3 + a].y;
    ry += U[b].y * U[3 + a].x;
    ry -= U[a].x * U[3 + b].y;
    ry -= U[a].y * U[3 + b].x;
    if (FMT == 1) r2[k] = make_double2(neg ? -rx : rx, neg ? -ry : ry);
    // explicit fma: "a*b - c*d" leaves the compiler two ways to contract, and it chose differently in different kernels -- the
    // single-system and the lock-step batched sweep must reconstruct the same bits (tests/test_gpu_batch.py)
    else r2[k] = make_double2(fma(ph.x, rx, -(ph.y * ry)), fma(ph.x, ry, ph.y * rx));
  }
#pragma unroll
  for (int k = 0; k < 3; k++) U[6 + k] = r2[k];
}

