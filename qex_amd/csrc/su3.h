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
__host__ __device__ __forceinline__ M3 m3_adj(const M3 &u) {
  M3 m;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int q = 0; q < 3; q++) m.e[3 * r + q] = make_double2(u.e[3 * q + r].x, -u.e[3 * q + r].y);
  return m;
}
#if defined(__HIPCC__)
// streaming forms for data nobody reads again soon (an accumulator's read-modify-write): they leave the L2 to the operands
// that neighbouring sites share
typedef double m3_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ M3 m3_load_nt(const double2 *p, int stride) {
  M3 r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    m3_d2v t = __builtin_nontemporal_load((const m3_d2v *)&p[(size_t)k * stride]);
    r.e[k] = make_double2(t.x, t.y);
  }
  return r;
}
__device__ __forceinline__ void m3_store_nt(double2 *p, int stride, const M3 &a) {
#pragma unroll
  for (int k = 0; k < 9; k++) {
    m3_d2v t = {a.e[k].x, a.e[k].y};
    __builtin_nontemporal_store(t, (m3_d2v *)&p[(size_t)k * stride]);
  }
}
#endif
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

// exp of a TRACELESS ANTI-HERMITIAN matrix through Cayley-Hamilton: v = iQ, exp(iQ) = f0 + f1 Q + f2 Q^2 with f_j from
// c0 = det Q, c1 = tr Q^2 / 2 -- by the Taylor series reduced with the characteristic polynomial while c1 <= 0.75 (round 3),
// in closed form beyond (Morningstar & Peardon, Phys. Rev. D 69, 054501, eqs. 19-33).  One matrix product,
// one trace and a handful of scalar functions instead of the 22 products of m3_exp: the same matrix function, NOT the
// reference's algorithm (matexp.nim: order-4 Taylor at v/2^20, 20 squarings), so it is an opt-in of the flow only
// (option "flow_exp" = 1); it agrees with m3_exp to the rounding error of m3_exp's squarings (~2e-15 absolute,
// tests/cpp/test_exp_ch.cpp) and is unitary to 4e-16 where m3_exp is to 2e-15.
__host__ __device__ __forceinline__ M3 m3_exp_tah(const M3 &v) {
  // Q = -i v  (Hermitian, traceless);  Q2 = Q Q = -(v v)
  M3 Q, Q2;
#pragma unroll
  for (int k = 0; k < 9; k++) Q.e[k] = make_double2(v.e[k].y, -v.e[k].x);
  // Q Hermitian => Q2 Hermitian: three off-diagonal products and the three row norms instead of nine products
#pragma unroll
  for (int i = 0; i < 3; i++) {
    double d = 0.0;
#pragma unroll
    for (int k = 0; k < 3; k++) d = fma(Q.e[3 * i + k].x, Q.e[3 * i + k].x, fma(Q.e[3 * i + k].y, Q.e[3 * i + k].y, d));
    Q2.e[4 * i] = make_double2(d, 0.0);
#pragma unroll
    for (int j = i + 1; j < 3; j++) {
      double sx = 0.0, sy = 0.0;
#pragma unroll
      for (int k = 0; k < 3; k++) M3_MAC(sx, sy, Q.e[3 * i + k].x, Q.e[3 * i + k].y, Q.e[3 * k + j].x, Q.e[3 * k + j].y);
      Q2.e[3 * i + j] = make_double2(sx, sy);
      Q2.e[3 * j + i] = make_double2(sx, -sy);
    }
  }
  const double c1 = 0.5 * (Q2.e[0].x + Q2.e[4].x + Q2.e[8].x);
  // c0 = det Q = tr(Q^3)/3 = Re tr(Q Q2)/3
  double t3 = 0.0;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int k = 0; k < 3; k++) t3 += Q.e[3 * i + k].x * Q2.e[3 * k + i].x - Q.e[3 * i + k].y * Q2.e[3 * k + i].y;
  double c0 = t3 * (1.0 / 3.0);
  M3 r;
  double f0r, f0i, f1r, f1i, f2r, f2i;
  if (c1 <= 0.75) {
    // Every eigenvalue of Q is below sqrt(4 c1 / 3) <= 1 in magnitude (the flow: |v| ~ 0.1): the Taylor series of exp(v) to
    // order 20, summed by Horner's rule in the quotient ring of the characteristic polynomial v^3 = -c1 v - i c0 --
    // with P = a0 + a1 v + a2 v^2,  1/n! + v P = (1/n! - i c0 a2) + (a0 - c1 a2) v + a1 v^2: four FMAs per order and no
    // acos / sincos / sqrt / division (those were ~500 of the ~2750 instructions a lane of the flow stage executes).
    // exp(v) = a0 + a1 (iQ) - a2 Q^2.
    constexpr double ifact[21] = {1.0, 1.0, 1.0 / 2, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320, 1.0 / 362880,
                                  1.0 / 3628800, 1.0 / 39916800, 1.0 / 479001600, 1.0 / 6227020800.0, 1.0 / 87178291200.0,
                                  1.0 / 1307674368000.0, 1.0 / 20922789888000.0, 1.0 / 355687428096000.0,
                                  1.0 / 6402373705728000.0, 1.0 / 121645100408832000.0, 1.0 / 2432902008176640000.0};
    double a0x = ifact[20], a0y = 0.0, a1x = 0.0, a1y = 0.0, a2x = 0.0, a2y = 0.0;
#pragma unroll
    for (int n = 19; n >= 0; n--) {
      const double n0x = fma(c0, a2y, ifact[n]), n0y = -c0 * a2x;
      const double n1x = fma(-c1, a2x, a0x), n1y = fma(-c1, a2y, a0y);
      a2x = a1x; a2y = a1y; a1x = n1x; a1y = n1y; a0x = n0x; a0y = n0y;
    }
    f0r = a0x; f0i = a0y; f1r = -a1y; f1i = a1x; f2r = -a2x; f2i = -a2y;
  } else {
    const bool neg = c0 < 0.0;                     // f_j(-c0) = (-1)^j conj f_j(c0): evaluate at |c0| (eq. 34)
    c0 = fabs(c0);
    const double c13 = c1 * (1.0 / 3.0);
    const double c0max = 2.0 * c13 * sqrt(c13);
    const double th = acos(fmin(1.0, c0 / c0max));
    double st3, ct3, sw, cw, su, cu;                 // one range reduction per angle
    sincos(th * (1.0 / 3.0), &st3, &ct3);
    const double u = sqrt(c13) * ct3;
    const double w = sqrt(c1) * st3;
    const double w2 = w * w, u2 = u * u;
    sincos(w, &sw, &cw);
    const double xi0 = fabs(w) < 0.05 ? 1.0 - w2 * (1.0 / 6.0) * (1.0 - w2 * (1.0 / 20.0) * (1.0 - w2 * (1.0 / 42.0))) : sw / w;
    sincos(u, &su, &cu);
    const double c2u = cu * cu - su * su, s2u = 2.0 * su * cu;     // e^{2iu}
    // h_j = A_j e^{2iu} + e^{-iu} (B_j + i C_j)
    const double b0 = 8.0 * u2 * cw, d0 = 2.0 * u * (3.0 * u2 + w2) * xi0;
    const double b1 = -2.0 * u * cw, d1 = (3.0 * u2 - w2) * xi0;
    const double b2 = -cw, d2 = -3.0 * u * xi0;
    const double a0 = u2 - w2, a1 = 2.0 * u;
    // e^{-iu} (b + i d) = (b cu + d su) + i (d cu - b su)
    const double den = 1.0 / (9.0 * u2 - w2);
    f0r = (a0 * c2u + b0 * cu + d0 * su) * den; f0i = (a0 * s2u + d0 * cu - b0 * su) * den;
    f1r = (a1 * c2u + b1 * cu + d1 * su) * den; f1i = (a1 * s2u + d1 * cu - b1 * su) * den;
    f2r = (c2u + b2 * cu + d2 * su) * den; f2i = (s2u + d2 * cu - b2 * su) * den;
    if (neg) { f0i = -f0i; f1r = -f1r; f2i = -f2i; }
  }
#pragma unroll
  for (int k = 0; k < 9; k++) {
    r.e[k].x = f1r * Q.e[k].x - f1i * Q.e[k].y + f2r * Q2.e[k].x - f2i * Q2.e[k].y;
    r.e[k].y = f1r * Q.e[k].y + f1i * Q.e[k].x + f2r * Q2.e[k].y + f2i * Q2.e[k].x;
  }
  r.e[0].x += f0r; r.e[0].y += f0i;
  r.e[4].x += f0r; r.e[4].y += f0i;
  r.e[8].x += f0r; r.e[8].y += f0i;
  return r;
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
