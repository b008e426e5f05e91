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

__host__ __device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__host__ __device__ __forceinline__ double2 cmulc(double2 a, double2 b) { /* a * conj(b) */ return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
__host__ __device__ __forceinline__ double2 ccmul(double2 a, double2 b) { /* conj(a) * b */ return make_double2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x); }

__host__ __device__ __forceinline__ M3 m3_load(const double2 *p, int stride) {
  M3 r;
#pragma unroll
  for (int k = 0; k < 9; k++) r.e[k] = p[(size_t)k * stride];
  return r;
}
__host__ __device__ __forceinline__ void m3_store(double2 *p, int stride, const M3 &a) {
#pragma unroll
  for (int k = 0; k < 9; k++) p[(size_t)k * stride] = a.e[k];
}
// The products are written as chained multiply-adds (s += a*b; s -= c*d; ...) so that hipcc
// contracts every term into one v_fma_f64: 4 FMAs per complex multiply-accumulate, 108 per 3x3
// product (a `cmul` temporary followed by an add costs 6 instructions per term).
#define M3_MAC(sx, sy, ax, ay, bx, by) \
  do { sx += (ax) * (bx); sx -= (ay) * (by); sy += (ax) * (by); sy += (ay) * (bx); } while (0)
// a*b
__host__ __device__ __forceinline__ M3 m3_mul(const M3 &a, const M3 &b) {
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
__host__ __device__ __forceinline__ M3 m3_mul_na(const M3 &a, const M3 &b) {
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
__host__ __device__ __forceinline__ M3 m3_mul_an(const M3 &a, const M3 &b) {
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
__host__ __device__ __forceinline__ void m3_mac(M3 &r, const M3 &a, const M3 &b) {
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
__host__ __device__ __forceinline__ void m3_mac_na(M3 &r, const M3 &a, const M3 &b) {
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
__host__ __device__ __forceinline__ void m3_mac_an(M3 &r, const M3 &a, const M3 &b) {
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
__host__ __device__ __forceinline__ void m3_axpy(M3 &r, double a, const M3 &x) {
#pragma unroll
  for (int k = 0; k < 9; k++) { r.e[k].x += a * x.e[k].x; r.e[k].y += a * x.e[k].y; }
}
__host__ __device__ __forceinline__ void m3_add_diag(M3 &r, double s) { r.e[0].x += s; r.e[4].x += s; r.e[8].x += s; }
__host__ __device__ __forceinline__ M3 m3_zero() {
  M3 r;
#pragma unroll
  for (int k = 0; k < 9; k++) r.e[k] = make_double2(0, 0);
  return r;
}
// Re tr(a^dagger b)
__host__ __device__ __forceinline__ double m3_redot(const M3 &a, const M3 &b) {
  double s = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) s += a.e[k].x * b.e[k].x + a.e[k].y * b.e[k].y;
  return s;
}
// traceless anti-Hermitian part
__host__ __device__ __forceinline__ M3 m3_tah(const M3 &x) {
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
// r (r + 2) = r r + 2 r, with the "+ 2" folded into the start values of the accumulation chains
__host__ __device__ __forceinline__ M3 m3_sq_p2(const M3 &a) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double sx = 0.0, sy = 0.0;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const double bx = a.e[3 * k + j].x + (k == j ? 2.0 : 0.0), by = a.e[3 * k + j].y;
        M3_MAC(sx, sy, a.e[3 * i + k].x, a.e[3 * i + k].y, bx, by);
      }
      r.e[3 * i + j] = make_double2(sx, sy);
    }
  return r;
}
__host__ __device__ __forceinline__ M3 m3_exp(const M3 &m) {
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
  // 20 squarings r <- r (r + 2), two per trip so that the result of one is the operand of the next without a copy
  // (a rolled single-step loop costs 18 register moves per 108 FMAs)
#pragma unroll 1
  for (int it = 0; it < 10; it++) {
    M3 t = m3_sq_p2(e);
    e = m3_sq_p2(t);
  }
  m3_add_diag(e, 1.0);
  return e;
}

// eigs3 + rsqrtPHM3f + rsqrtPHM3 + projectU (matrixFunctions.nim:79-182,279-313)
// z = (x^+ x + 1e-20)^(-1/2)   (projectUrsqrt, matrixFunctions.nim:301-306)
__host__ __device__ __forceinline__ M3 m3_rsqrt_xdx(const M3 &x) {
  M3 t = m3_mul_an(x, x);
  m3_add_diag(t, 1e-20);
  const double tr = t.e[0].x + t.e[4].x + t.e[8].x;
  M3 t2 = m3_mul(t, t);
  const double p2 = t2.e[0].x + t2.e[4].x + t2.e[8].x;
  // Re det (matrixFunctions.nim:72-75)
  const double2 d01 = make_double2(t.e[0].x * t.e[4].x - t.e[0].y * t.e[4].y - (t.e[1].x * t.e[3].x - t.e[1].y * t.e[3].y),
                                   t.e[0].x * t.e[4].y + t.e[0].y * t.e[4].x - (t.e[1].x * t.e[3].y + t.e[1].y * t.e[3].x));
  const double2 d20 = make_double2(t.e[2].x * t.e[3].x - t.e[2].y * t.e[3].y - (t.e[0].x * t.e[5].x - t.e[0].y * t.e[5].y),
                                   t.e[2].x * t.e[3].y + t.e[2].y * t.e[3].x - (t.e[0].x * t.e[5].y + t.e[0].y * t.e[5].x));
  const double2 d12 = make_double2(t.e[1].x * t.e[5].x - t.e[1].y * t.e[5].y - (t.e[2].x * t.e[4].x - t.e[2].y * t.e[4].y),
                                   t.e[1].x * t.e[5].y + t.e[1].y * t.e[5].x - (t.e[2].x * t.e[4].y + t.e[2].y * t.e[4].x));
  const double det = (d01.x * t.e[8].x - d01.y * t.e[8].y) + (d20.x * t.e[7].x - d20.y * t.e[7].y) + (d12.x * t.e[6].x - d12.y * t.e[6].y);
  // eigs3
  const double tr3 = (1.0 / 3.0) * tr, p23 = (1.0 / 3.0) * p2, tr32 = tr3 * tr3;
  const double q = fabs(0.5 * (p23 - tr32));
  const double r = 0.25 * tr3 * (5 * tr32 - p2) - 0.5 * det;
  const double sq = sqrt(q), sq3 = q * sq;
  const double isq3c = fmin(3e38, fmax(-3e38, 1.0 / sq3));
  const double rsq3 = fmin(1.0, fmax(-1.0, r * isq3c));
  const double th = (1.0 / 3.0) * acos(rsq3);
  const double st = sin(th), ct = cos(th);
  const double sqc = sq * ct, sqs = 1.73205080756887729352 * sq * st;
  const double ll = tr3 + sqc;
  const double l0 = tr3 - 2 * sqc, l1 = ll + sqs, l2 = ll - sqs;
  // rsqrtPHM3f
  const double sl0 = sqrt(fabs(l0)), sl1 = sqrt(fabs(l1)), sl2 = sqrt(fabs(l2));
  const double u = sl0 + sl1 + sl2, w = sl0 * sl1 * sl2;
  const double d = w * (sl0 + sl1) * (sl0 + sl2) * (sl1 + sl2);
  const double di = 1 / d;
  const double c0 = (w * u * u + l0 * sl0 * (l1 + l2) + l1 * sl1 * (l0 + l2) + l2 * sl2 * (l0 + l1)) * di;
  const double c1 = -(tr * u + w) * di;
  const double c2 = u * di;
  M3 rs;
#pragma unroll
  for (int k = 0; k < 9; k++) rs.e[k] = make_double2(c1 * t.e[k].x + c2 * t2.e[k].x, c1 * t.e[k].y + c2 * t2.e[k].y);
  m3_add_diag(rs, c0);
  return rs;
}
__host__ __device__ __forceinline__ M3 m3_projectU(const M3 &x) { return m3_mul(x, m3_rsqrt_xdx(x)); }

// projectSU (matrixFunctions.nim:359-370): projectU, then remove the determinant's phase
__host__ __device__ __forceinline__ M3 m3_projectSU(const M3 &x) {
  const M3 m = m3_projectU(x);
  const double2 *e = m.e;
  const double2 d01 = make_double2(e[0].x * e[4].x - e[0].y * e[4].y - (e[1].x * e[3].x - e[1].y * e[3].y),
                                   e[0].x * e[4].y + e[0].y * e[4].x - (e[1].x * e[3].y + e[1].y * e[3].x));
  const double2 d20 = make_double2(e[2].x * e[3].x - e[2].y * e[3].y - (e[0].x * e[5].x - e[0].y * e[5].y),
                                   e[2].x * e[3].y + e[2].y * e[3].x - (e[0].x * e[5].y + e[0].y * e[5].x));
  const double2 d12 = make_double2(e[1].x * e[5].x - e[1].y * e[5].y - (e[2].x * e[4].x - e[2].y * e[4].y),
                                   e[1].x * e[5].y + e[1].y * e[5].x - (e[2].x * e[4].y + e[2].y * e[4].x));
  const double dr = (d01.x * e[8].x - d01.y * e[8].y) + (d20.x * e[7].x - d20.y * e[7].y) + (d12.x * e[6].x - d12.y * e[6].y);
  const double di = (d01.x * e[8].y + d01.y * e[8].x) + (d20.x * e[7].y + d20.y * e[7].x) + (d12.x * e[6].y + d12.y * e[6].x);
  const double p = (1.0 / (double)(-3)) * atan2(di, dr);
  const double cr = cos(p), ci = sin(p);
  M3 r;
#pragma unroll
  for (int k = 0; k < 9; k++) r.e[k] = make_double2(cr * e[k].x - ci * e[k].y, cr * e[k].y + ci * e[k].x);
  return r;
}
