// su3.h -- 3x3 complex fp64 matrix helpers for the gauge kernels (device only).
// Restates the pieces of src/maths that the flow path uses:
//   mul / adj                      src/maths/matrixOps.nim
//   projectTAH                     src/maths/matrixFunctions.nim:375-380
//   exp = expm1Poly4(m/2^20), 20 squarings r <- r(r+2), +1
//                                  src/maths/matrixFunctions.nim:436-445, src/maths/matexp.nim:80-85,634-649,707-710
#pragma once
#include <hip/hip_runtime.h>

struct M3 {
  double2 e[9];
};

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 cmulc(double2 a, double2 b) { /* a * conj(b) */ return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
__device__ __forceinline__ double2 ccmul(double2 a, double2 b) { /* conj(a) * b */ return make_double2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x); }

__device__ __forceinline__ M3 m3_load(const double2 *p, int stride) {
  M3 r;
#pragma unroll
  for (int k = 0; k < 9; k++) r.e[k] = p[(size_t)k * stride];
  return r;
}
__device__ __forceinline__ void m3_store(double2 *p, int stride, const M3 &a) {
#pragma unroll
  for (int k = 0; k < 9; k++) p[(size_t)k * stride] = a.e[k];
}
// The products are written as chained multiply-adds (s += a*b; s -= c*d; ...) so that hipcc
// contracts every term into one v_fma_f64: 4 FMAs per complex multiply-accumulate, 108 per 3x3
// product (a `cmul` temporary followed by an add costs 6 instructions per term).
#define M3_MAC(sx, sy, ax, ay, bx, by) \
  do { sx += (ax) * (bx); sx -= (ay) * (by); sy += (ax) * (by); sy += (ay) * (bx); } while (0)
// a*b
__device__ __forceinline__ M3 m3_mul(const M3 &a, const M3 &b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double sx = 0.0, sy = 0.0;
#pragma unroll
      for (int k = 0; k < 3; k++) M3_MAC(sx, sy, a.e[3 * i + k].x, a.e[3 * i + k].y, b.e[3 * k + j].x, b.e[3 * k + j].y);
      r.e[3 * i + j] = make_double2(sx, sy);
    }
  return r;
}
// a*b^dagger
__device__ __forceinline__ M3 m3_mul_na(const M3 &a, const M3 &b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double sx = 0.0, sy = 0.0;
#pragma unroll
      for (int k = 0; k < 3; k++) M3_MAC(sx, sy, a.e[3 * i + k].x, a.e[3 * i + k].y, b.e[3 * j + k].x, -b.e[3 * j + k].y);
      r.e[3 * i + j] = make_double2(sx, sy);
    }
  return r;
}
// a^dagger*b
__device__ __forceinline__ M3 m3_mul_an(const M3 &a, const M3 &b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double sx = 0.0, sy = 0.0;
#pragma unroll
      for (int k = 0; k < 3; k++) M3_MAC(sx, sy, a.e[3 * k + i].x, -a.e[3 * k + i].y, b.e[3 * k + j].x, b.e[3 * k + j].y);
      r.e[3 * i + j] = make_double2(sx, sy);
    }
  return r;
}
// r += a*b, r += a*b^dagger, r += a^dagger*b  (accumulating forms: no temporary product matrix)
__device__ __forceinline__ void m3_mac(M3 &r, const M3 &a, const M3 &b) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double sx = r.e[3 * i + j].x, sy = r.e[3 * i + j].y;
#pragma unroll
      for (int k = 0; k < 3; k++) M3_MAC(sx, sy, a.e[3 * i + k].x, a.e[3 * i + k].y, b.e[3 * k + j].x, b.e[3 * k + j].y);
      r.e[3 * i + j] = make_double2(sx, sy);
    }
}
__device__ __forceinline__ void m3_mac_na(M3 &r, const M3 &a, const M3 &b) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double sx = r.e[3 * i + j].x, sy = r.e[3 * i + j].y;
#pragma unroll
      for (int k = 0; k < 3; k++) M3_MAC(sx, sy, a.e[3 * i + k].x, a.e[3 * i + k].y, b.e[3 * j + k].x, -b.e[3 * j + k].y);
      r.e[3 * i + j] = make_double2(sx, sy);
    }
}
__device__ __forceinline__ void m3_mac_an(M3 &r, const M3 &a, const M3 &b) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double sx = r.e[3 * i + j].x, sy = r.e[3 * i + j].y;
#pragma unroll
      for (int k = 0; k < 3; k++) M3_MAC(sx, sy, a.e[3 * k + i].x, -a.e[3 * k + i].y, b.e[3 * k + j].x, b.e[3 * k + j].y);
      r.e[3 * i + j] = make_double2(sx, sy);
    }
}
__device__ __forceinline__ void m3_axpy(M3 &r, double a, const M3 &x) {
#pragma unroll
  for (int k = 0; k < 9; k++) { r.e[k].x += a * x.e[k].x; r.e[k].y += a * x.e[k].y; }
}
__device__ __forceinline__ void m3_add_diag(M3 &r, double s) { r.e[0].x += s; r.e[4].x += s; r.e[8].x += s; }
__device__ __forceinline__ M3 m3_zero() {
  M3 r;
#pragma unroll
  for (int k = 0; k < 9; k++) r.e[k] = make_double2(0, 0);
  return r;
}
// Re tr(a^dagger b)
__device__ __forceinline__ double m3_redot(const M3 &a, const M3 &b) {
  double s = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) s += a.e[k].x * b.e[k].x + a.e[k].y * b.e[k].y;
  return s;
}
// traceless anti-Hermitian part
__device__ __forceinline__ M3 m3_tah(const M3 &x) {
  M3 t;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      t.e[3 * i + j] = make_double2(0.5 * (x.e[3 * i + j].x - x.e[3 * j + i].x), 0.5 * (x.e[3 * i + j].y + x.e[3 * j + i].y));
  double dr = (t.e[0].x + t.e[4].x + t.e[8].x) / 3.0;
  double di = (t.e[0].y + t.e[4].y + t.e[8].y) / 3.0;
  t.e[0].x -= dr; t.e[0].y -= di;
  t.e[4].x -= dr; t.e[4].y -= di;
  t.e[8].x -= dr; t.e[8].y -= di;
  return t;
}
__device__ __forceinline__ M3 m3_exp(const M3 &m) {
  const double s = 1.0 / (double)(1 << 20);
  M3 ms, a;
#pragma unroll
  for (int k = 0; k < 9; k++) ms.e[k] = make_double2(s * m.e[k].x, s * m.e[k].y);
  M3 m2 = m3_mul(ms, ms);
#pragma unroll
  for (int k = 0; k < 9; k++) a.e[k] = make_double2((1.0 / 24.0) * m2.e[k].x, (1.0 / 24.0) * m2.e[k].y);
  m3_axpy(a, 1.0 / 6.0, ms);
  m3_add_diag(a, 0.5);
  M3 e = m3_mul(a, m2);
#pragma unroll
  for (int k = 0; k < 9; k++) { e.e[k].x += ms.e[k].x; e.e[k].y += ms.e[k].y; }
#pragma unroll 1
  for (int it = 0; it < 20; it++) {
    M3 t = e;
    m3_add_diag(t, 2.0);
    e = m3_mul(e, t);
  }
  m3_add_diag(e, 1.0);
  return e;
}
