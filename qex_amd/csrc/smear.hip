// smear.hip -- link construction upstream of the solver: fat7 / HISQ and nHYP smearing
// (forward direction).
//
// Restates (file:line in ctpeterson/qex):
//   computeGenStaple / makeImpLinks     src/gauge/fat7l.nim:24-161
//   HisqCoefs.init / smear              src/physics/hisqLinks.nim:9-43
//   symStaple                           src/gauge/smearutil.nim:3-20
//   nHYP forward smearing               src/gauge/hypsmear.nim:49-144
//   projectU = x (x^+x + 1e-20)^(-1/2)  src/maths/matrixFunctions.nim:79-182,279-313
// Everything is built from ONE field-level kernel, the generic staple
//   st(x) = A(x) B(x+nu) A(x+mu)^+ + A(x-nu)^+ B(x-nu) A(x-nu+mu)
// (A: side links of direction nu, B: any matrix field standing for the mu link), plus scale,
// projectU and the 3-link Naik product.  One lane per site; matrix fields use the tile layout
// [parity][tile][9][64] (a gauge field is four of them interleaved: [parity][tile][mu][9][64]).
// Single GPU (periodic wrap), like the flow kernels.
#include "qexhip_internal.h"
#include "su3.h"
#include <cmath>
#include <algorithm>

struct MView {        // read-only matrix field view
  const double2 *p;
  int tstride;        // double2 between consecutive tiles (576 for a single field, 4*576 inside a gauge field)
};
struct MViewW {
  double2 *p;
  int tstride;
};

__device__ __forceinline__ void coords_sm(const Geom &g, int c, int p, int x[4]) {
  unsigned r = (unsigned)c;
  int xh = r % (unsigned)g.Xh; r /= (unsigned)g.Xh;
  x[1] = r % (unsigned)g.X[1]; r /= (unsigned)g.X[1];
  x[2] = r % (unsigned)g.X[2];
  x[3] = r / (unsigned)g.X[2];
  x[0] = 2 * xh + ((x[1] + x[2] + x[3] + p) & 1);
}
// t-sharded runs: every parity half of a matrix field is [body ntile | ghost_hi 3F/64 tiles | ghost_lo 3F/64 tiles]
// (the layout of gauge.hip); a t-hop never wraps, virtual slices Xt..Xt+2 / -3..-1 map to the ghost tiles.  HALO is a
// template parameter for the two register-tight staple kernels, a runtime flag (g.halo) everywhere else.
template <bool HALO>
__device__ __forceinline__ size_t site_off_t(const Geom &g, const int x[4], int tstride) {
  int t = x[3];
  if (HALO) t = t < 0 ? t + g.X[3] + 6 : t;
  int lex = x[0] + g.X[0] * (x[1] + g.X[1] * (x[2] + g.X[2] * t));
  int p = (x[0] + x[1] + x[2] + x[3]) & 1;
  int c = lex >> 1;
  return ((size_t)p * g.etile + (c >> 6)) * tstride + (c & 63);
}
template <bool HALO>
__device__ __forceinline__ void shift_sm_t(const Geom &g, const int x[4], int mu, int d, int y[4]) {
  y[0] = x[0]; y[1] = x[1]; y[2] = x[2]; y[3] = x[3];
  int v = y[mu] + d;
  if (HALO && mu == 3) { y[3] = v; return; }
  y[mu] = v >= g.X[mu] ? v - g.X[mu] : (v < 0 ? v + g.X[mu] : v);
}
__device__ __forceinline__ size_t site_off(const Geom &g, const int x[4], int tstride) {
  return g.halo ? site_off_t<true>(g, x, tstride) : site_off_t<false>(g, x, tstride);
}
__device__ __forceinline__ void shift_sm(const Geom &g, const int x[4], int mu, int d, int y[4]) {
  if (g.halo) shift_sm_t<true>(g, x, mu, d, y); else shift_sm_t<false>(g, x, mu, d, y);
}

// blocked visiting order for the gather kernels (tile_order_table / tile_order_plane, layout.hip)
static int smear_order(qexhip_ctx *c, const Geom &g, const int **order, int *chunk, int *nblk, int mu = -1, int nu = -1) {
  *order = nullptr; *chunk = 0; *nblk = (g.V + 255) / 256;
  CHK(tile_order_plane(c, mu, nu, order, chunk));     // mu < 0: the generic blocked order
  *nblk = 8 * ((*chunk + 3) / 4);
  return 0;
}

// staple field (optional) and acc (+)= coef * staple (optional).  Two fusions for the nHYP levels
// (hypsmear.nim:98-143): with `init` the accumulator STARTS as cinit * init(x) instead of being read
// (the `l := ma * g[mu]` assignment), with `proj` the finished sum is also projected, proj(x) = projectU(acc(x))
// (the `l.proj lx` that follows the last staple of a level).
template <bool HALO>
__global__ void __launch_bounds__(256) k_gen_staple(Geom g, MView A, MView B, int mu, int nu, MViewW st, MViewW acc, double coef, int swz,
                                                    MView init, double cinit, MViewW proj, const int *order, int chunk, int nt,
                                                    int gc0 = 0, int gc1 = 0) {
  int p, c;
  if (gc1 > gc0) {   // t-sharded, communication-avoiding levels: the positions [gc0, gc1) of both parities -- ghost slices
    const int n = gc1 - gc0, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * n) return;
    p = i >= n; c = gc0 + i - p * n;
  } else if (order) {   // blocked visiting order (tile_order_table): wavefront w takes slot 4*(b>>3)+w of XCD b&7
    const int slot = 4 * (blockIdx.x >> 3) + (threadIdx.x >> 6);
    const int e = slot < chunk ? order[(blockIdx.x & 7) * chunk + slot] : -1;
    if (e < 0) return;
    p = e & 1; c = (e >> 1) * 64 + (threadIdx.x & 63);
    if (c >= g.Vh) return;
  } else {
    int bid = blockIdx.x;
    if (swz && (gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);   // contiguous site range per XCD
    int i = bid * 256 + threadIdx.x;
    if (i >= g.V) return;
    p = i >= g.Vh; c = i - p * g.Vh;
  }
  int x[4], xpn[4], xpm[4], xmn[4], xmnpm[4];
  coords_sm(g, c, p, x);
  if (HALO && x[3] >= g.X[3] + 3) x[3] -= g.X[3] + 6;     // a position in ghost_lo: virtual slice -3..-1 (Xt + 6 is even: same parity)
  shift_sm_t<HALO>(g, x, nu, 1, xpn);
  shift_sm_t<HALO>(g, x, mu, 1, xpm);
  shift_sm_t<HALO>(g, x, nu, -1, xmn);
  shift_sm_t<HALO>(g, xmn, mu, 1, xmnpm);
  M3 t = m3_mul_na(m3_load(B.p + site_off_t<HALO>(g, xpn, B.tstride), 64), m3_load(A.p + site_off_t<HALO>(g, xpm, A.tstride), 64));
  M3 s = m3_mul(m3_load(A.p + site_off_t<HALO>(g, x, A.tstride), 64), t);
  t = m3_mul_an(m3_load(A.p + site_off_t<HALO>(g, xmn, A.tstride), 64), m3_load(B.p + site_off_t<HALO>(g, xmn, B.tstride), 64));
  M3 u = m3_mul(t, m3_load(A.p + site_off_t<HALO>(g, xmnpm, A.tstride), 64));
#pragma unroll
  for (int k = 0; k < 9; k++) { s.e[k].x += u.e[k].x; s.e[k].y += u.e[k].y; }
  if (st.p) { if (nt) m3_store_nt(st.p + site_off_t<HALO>(g, x, st.tstride), 64, s); else m3_store(st.p + site_off_t<HALO>(g, x, st.tstride), 64, s); }
  if (acc.p) {
    double2 *a = acc.p + site_off_t<HALO>(g, x, acc.tstride);
    M3 o;
    if (init.p) {
      o = m3_load(init.p + site_off_t<HALO>(g, x, init.tstride), 64);
#pragma unroll
      for (int k = 0; k < 9; k++) { o.e[k].x *= cinit; o.e[k].y *= cinit; }
    } else {
      o = nt ? m3_load_nt(a, 64) : m3_load(a, 64);
    }
    m3_axpy(o, coef, s);
    if (nt) m3_store_nt(a, 64, o); else m3_store(a, 64, o);
    if (proj.p) {
      const M3 pr = m3_projectU(o);
      if (nt) m3_store_nt(proj.p + site_off_t<HALO>(g, x, proj.tstride), 64, pr); else m3_store(proj.p + site_off_t<HALO>(g, x, proj.tstride), 64, pr);
    }
  }
}
// dst += coef * src
__global__ void __launch_bounds__(256) k_maxpy(Geom g, MViewW dst, double coef, MView src) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.V) return;
  const int p = i >= g.Vh, c = i - p * g.Vh;
  const size_t od = ((size_t)p * g.etile + (c >> 6)) * dst.tstride + (c & 63);
  const size_t os = ((size_t)p * g.etile + (c >> 6)) * src.tstride + (c & 63);
  M3 o = m3_load(dst.p + od, 64);
  m3_axpy(o, coef, m3_load(src.p + os, 64));
  m3_store(dst.p + od, 64, o);
}
// dst = coef * src
__global__ void __launch_bounds__(256) k_mscale(Geom g, MViewW dst, double coef, MView src) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.V) return;
  const int p = i >= g.Vh, c = i - p * g.Vh;
  const size_t od = ((size_t)p * g.etile + (c >> 6)) * dst.tstride + (c & 63);
  const size_t os = ((size_t)p * g.etile + (c >> 6)) * src.tstride + (c & 63);
  M3 m = m3_load(src.p + os, 64);
#pragma unroll
  for (int k = 0; k < 9; k++) { m.e[k].x *= coef; m.e[k].y *= coef; }
  m3_store(dst.p + od, 64, m);
}

// ---- projectUderiv and its pieces (matrixFunctions.nim:329-357, projUderiv.nim:8-38,96-147, matinv.nim:90-115)
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cinv(double2 a) { const double d = 1.0 / (a.x * a.x + a.y * a.y); return make_double2(a.x * d, -a.y * d); }
__device__ __forceinline__ double2 cm2(double2 a, double2 b, double2 c, double2 d) { return csub(cmul(a, b), cmul(c, d)); }  // ab - cd
__device__ __forceinline__ M3 m3_adjugate(const M3 &m) {
  const double2 *x = m.e;
  M3 r;
  r.e[0] = cm2(x[4], x[8], x[5], x[7]); r.e[1] = cm2(x[7], x[2], x[8], x[1]); r.e[2] = cm2(x[1], x[5], x[2], x[4]);
  r.e[3] = cm2(x[5], x[6], x[3], x[8]); r.e[4] = cm2(x[8], x[0], x[6], x[2]); r.e[5] = cm2(x[2], x[3], x[0], x[5]);
  r.e[6] = cm2(x[3], x[7], x[4], x[6]); r.e[7] = cm2(x[6], x[1], x[7], x[0]); r.e[8] = cm2(x[0], x[4], x[1], x[3]);
  return r;
}
__device__ __forceinline__ M3 m3_inverse(const M3 &m) {
  const double2 *x = m.e;
  const double2 det0 = cm2(x[0], x[4], x[1], x[3]), det1 = cm2(x[2], x[3], x[0], x[5]), det2 = cm2(x[1], x[5], x[2], x[4]);
  const double2 det = cadd(cadd(cmul(det0, x[8]), cmul(det1, x[7])), cmul(det2, x[6]));
  const double2 idet = cinv(det);
  M3 r;
  r.e[0] = cmul(idet, cm2(x[4], x[8], x[5], x[7])); r.e[1] = cmul(idet, cm2(x[7], x[2], x[8], x[1])); r.e[2] = cmul(idet, det2);
  r.e[3] = cmul(idet, cm2(x[5], x[6], x[3], x[8])); r.e[4] = cmul(idet, cm2(x[8], x[0], x[6], x[2])); r.e[5] = cmul(idet, det1);
  r.e[6] = cmul(idet, cm2(x[3], x[7], x[4], x[6])); r.e[7] = cmul(idet, cm2(x[6], x[1], x[7], x[0])); r.e[8] = cmul(idet, det0);
  return r;
}
// A X + X A = C
__device__ __forceinline__ M3 m3_sylsolve(const M3 &a, const M3 &c) {
  const M3 ad = m3_adjugate(a);
  const double2 t = cadd(cadd(a.e[0], a.e[4]), a.e[8]), s = cadd(cadd(ad.e[0], ad.e[4]), ad.e[8]);
  const double2 r = cadd(cadd(cmul(a.e[0], ad.e[0]), cmul(a.e[1], ad.e[3])), cmul(a.e[2], ad.e[6]));
  const double2 two_d = csub(cmul(s, t), r);
  const double2 c2 = cinv(make_double2(2.0 * two_d.x, 2.0 * two_d.y));
  const double2 c0 = cmul(c2, cadd(s, cmul(t, t))), c1 = cmul(c2, cmul(t, cinv(r))), c4 = cmul(c2, t);
  const M3 ac = m3_mul(a, c), ca = m3_mul(c, a);
  M3 x;
  {
    const M3 aca = m3_mul(ac, a);
#pragma unroll
    for (int k = 0; k < 9; k++) x.e[k] = cadd(cmul(c0, c.e[k]), csub(cmul(c2, aca.e[k]), cmul(c4, cadd(ac.e[k], ca.e[k]))));
  }
  const M3 adc = m3_mul(ad, c), cad = m3_mul(c, ad);
  const M3 adcad = m3_mul(adc, ad);
#pragma unroll
  for (int k = 0; k < 9; k++) x.e[k] = cadd(x.e[k], csub(cmul(c1, adcad.e[k]), cmul(c2, cadd(adc.e[k], cad.e[k]))));
  return x;
}
// a b for factors whose product is known to be Hermitian: the diagonal and the upper triangle (6 of 9 entries), mirrored
__device__ __forceinline__ M3 m3_mul_herm(const M3 &a, const M3 &b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = i; j < 3; j++) {
      double sx = 0.0, sy = 0.0;
#pragma unroll
      for (int k = 0; k < 3; k++) M3_MAC(sx, sy, a.e[3 * i + k].x, a.e[3 * i + k].y, b.e[3 * k + j].x, b.e[3 * k + j].y);
      r.e[3 * i + j] = make_double2(sx, i == j ? 0.0 : sy);
      if (j > i) r.e[3 * j + i] = make_double2(sx, -sy);
    }
  return r;
}
// A X + X A = C for Hermitian A and Hermitian C (then X is Hermitian too); ad = adjugate(A).  The closed form of
// m3_sylsolve with what Hermiticity gives away: C A = (A C)^+, C ad = (ad C)^+, and A C A, ad C ad are Hermitian
// (two thirds of a product each): 3.3 products instead of 6.7.
__device__ __forceinline__ M3 m3_sylsolve_herm(const M3 &a, const M3 &ad, const M3 &c) {
  const double2 t = cadd(cadd(a.e[0], a.e[4]), a.e[8]), s = cadd(cadd(ad.e[0], ad.e[4]), ad.e[8]);
  const double2 r = cadd(cadd(cmul(a.e[0], ad.e[0]), cmul(a.e[1], ad.e[3])), cmul(a.e[2], ad.e[6]));
  const double2 two_d = csub(cmul(s, t), r);
  const double2 c2 = cinv(make_double2(2.0 * two_d.x, 2.0 * two_d.y));
  const double2 c0 = cmul(c2, cadd(s, cmul(t, t))), c1 = cmul(c2, cmul(t, cinv(r))), c4 = cmul(c2, t);
  M3 x;
  {
    const M3 ac = m3_mul(a, c);
    const M3 aca = m3_mul_herm(ac, a);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const double2 sym = make_double2(ac.e[3 * i + j].x + ac.e[3 * j + i].x, ac.e[3 * i + j].y - ac.e[3 * j + i].y);   // A C + C A
        x.e[3 * i + j] = cadd(cmul(c0, c.e[3 * i + j]), csub(cmul(c2, aca.e[3 * i + j]), cmul(c4, sym)));
      }
  }
  const M3 adc = m3_mul(ad, c);
  const M3 adcad = m3_mul_herm(adc, ad);
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const double2 sym = make_double2(adc.e[3 * i + j].x + adc.e[3 * j + i].x, adc.e[3 * i + j].y - adc.e[3 * j + i].y);
      x.e[3 * i + j] = cadd(x.e[3 * i + j], csub(cmul(c1, adcad.e[3 * i + j]), cmul(c2, sym)));
    }
  return x;
}
// F with d Re tr(C^+ U(X)) = Re tr(dX^+ F)                      (matrixFunctions.nim:329-357, projUderiv.nim:8-38,96-147)
// The reference solves Y T + T Y = U^+ R for T and then uses T + T^+ only; Y = (X^+X)^(1/2) is Hermitian, so the solve
// commutes with taking the Hermitian part (solve(Y, C)^+ = solve(Y, C^+)): the right-hand side is made Hermitian FIRST
// and the Hermitian solver above does the rest.  adjugate(Y) = det(Y) Y^-1 = Z / det(Z) comes for free (Z is at hand).
// Round 3: 23 % fewer fp64 operations than the literal form (m3_sylsolve, kept above for reference); same function to
// rounding (tests hold the chain to the oracle's literal form at 1e-11).
__device__ __forceinline__ M3 m3_projectUderiv(const M3 &x, const M3 &chain) {
  const M3 z = m3_rsqrt_xdx(x);
  const M3 u = m3_mul(x, z);          // = projectU(x), bit for bit what k_projectU / k_gen_staple stored: never re-read
  // y = z^-1 and adjugate(y) = z / det z
  const double2 *q = z.e;
  const double2 det0 = cm2(q[0], q[4], q[1], q[3]), det1 = cm2(q[2], q[3], q[0], q[5]), det2 = cm2(q[1], q[5], q[2], q[4]);
  const double2 idet = cinv(cadd(cadd(cmul(det0, q[8]), cmul(det1, q[7])), cmul(det2, q[6])));
  M3 y, ady;
  y.e[0] = cmul(idet, cm2(q[4], q[8], q[5], q[7])); y.e[1] = cmul(idet, cm2(q[7], q[2], q[8], q[1])); y.e[2] = cmul(idet, det2);
  y.e[3] = cmul(idet, cm2(q[5], q[6], q[3], q[8])); y.e[4] = cmul(idet, cm2(q[8], q[0], q[6], q[2])); y.e[5] = cmul(idet, det1);
  y.e[6] = cmul(idet, cm2(q[3], q[7], q[4], q[6])); y.e[7] = cmul(idet, cm2(q[6], q[1], q[7], q[0])); y.e[8] = cmul(idet, det0);
#pragma unroll
  for (int k = 0; k < 9; k++) ady.e[k] = cmul(idet, z.e[k]);
  M3 r = m3_mul(chain, z);
  const M3 t1 = m3_mul_an(u, r);
  M3 ch;                               // the Hermitian part (times two) of the right-hand side
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      ch.e[3 * i + j] = make_double2(t1.e[3 * i + j].x + t1.e[3 * j + i].x, t1.e[3 * i + j].y - t1.e[3 * j + i].y);
  const M3 t2 = m3_sylsolve_herm(y, ady, ch);
  const M3 xt = m3_mul(x, t2);
#pragma unroll
  for (int k = 0; k < 9; k++) r.e[k] = csub(r.e[k], xt.e[k]);
  return r;
}
// the same with x parked in LDS (xl: this lane's column, stride 64): x is needed at the start (z, u) and for the very last
// product only, and 36 VGPRs is what separates the function from two wavefronts per SIMD
__device__ __forceinline__ M3 m3_projectUderiv_parked(const double2 *xl, const M3 &chain) {
  M3 z, u;
  {
    const M3 x = m3_load(xl, 64);
    z = m3_rsqrt_xdx(x);
    u = m3_mul(x, z);
  }
  const double2 *q = z.e;
  const double2 det0 = cm2(q[0], q[4], q[1], q[3]), det1 = cm2(q[2], q[3], q[0], q[5]), det2 = cm2(q[1], q[5], q[2], q[4]);
  const double2 idet = cinv(cadd(cadd(cmul(det0, q[8]), cmul(det1, q[7])), cmul(det2, q[6])));
  M3 y, ady;
  y.e[0] = cmul(idet, cm2(q[4], q[8], q[5], q[7])); y.e[1] = cmul(idet, cm2(q[7], q[2], q[8], q[1])); y.e[2] = cmul(idet, det2);
  y.e[3] = cmul(idet, cm2(q[5], q[6], q[3], q[8])); y.e[4] = cmul(idet, cm2(q[8], q[0], q[6], q[2])); y.e[5] = cmul(idet, det1);
  y.e[6] = cmul(idet, cm2(q[3], q[7], q[4], q[6])); y.e[7] = cmul(idet, cm2(q[6], q[1], q[7], q[0])); y.e[8] = cmul(idet, det0);
#pragma unroll
  for (int k = 0; k < 9; k++) ady.e[k] = cmul(idet, z.e[k]);
  M3 r = m3_mul(chain, z);
  const M3 t1 = m3_mul_an(u, r);
  M3 ch;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      ch.e[3 * i + j] = make_double2(t1.e[3 * i + j].x + t1.e[3 * j + i].x, t1.e[3 * i + j].y - t1.e[3 * j + i].y);
  const M3 t2 = m3_sylsolve_herm(y, ady, ch);
  __builtin_amdgcn_sched_barrier(0);
  const M3 xt = m3_mul(m3_load(xl, 64), t2);
#pragma unroll
  for (int k = 0; k < 9; k++) r.e[k] = csub(r.e[k], xt.e[k]);
  return r;
}
// r = projectUderiv(U or projectU(X), X, C);  f (=|+=) ma * r;  dst = alp * r      (dst may alias C)
// -- the projection's chain rule fused with the bookkeeping that follows it at every level
// (hypsmear.nim:166-173,196-205,224-233: `f[mu] += ma*fl; fl *= alp`)
__global__ void __launch_bounds__(256) k_projUderiv(Geom g, MViewW dst, MView U, MView X, MView C, MViewW f, double ma, double alp,
                                                    int accumulate) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.V) return;
  const int p = i >= g.Vh, c = i - p * g.Vh;
  const size_t t = (size_t)p * g.etile + (c >> 6);
  const int l = c & 63;
  // U (the stored projection of X) is accepted for the call sites' sake and not read: x (x^+x)^(-1/2) is rebuilt from the
  // rsqrt the derivative needs anyway, which saves a 144 B/site read of a kernel that streams 864
  (void)U;
  const M3 x = m3_load(X.p + t * X.tstride + l, 64);
  const M3 ch = m3_load(C.p + t * C.tstride + l, 64);
  M3 r = m3_projectUderiv(x, ch);
  if (f.p) {
    M3 o = accumulate ? m3_load(f.p + t * f.tstride + l, 64) : m3_zero();
    m3_axpy(o, ma, r);
    m3_store(f.p + t * f.tstride + l, 64, o);
  }
#pragma unroll
  for (int k = 0; k < 9; k++) { r.e[k].x *= alp; r.e[k].y *= alp; }
  m3_store(dst.p + t * dst.tstride + l, 64, r);
}
// The projUderiv calls of one level in ONE launch: blockIdx.y = mu, and a lane walks the nn constituents (mu; nu_j) of its
// link in the order the separate launches had (so f[mu] accumulates bit for bit the same) -- f[mu] is read and written
// once instead of nn times (1584 instead of 2160 B per site and direction at nn = 3), 3 launches instead of 28 per chain
struct ProjBatch {
  MViewW dst[4][3];
  MView X[4][3], C[4][3];
  MViewW f[4];
  int nn, accumulate;
  double ma, alp;
};
// The accumulator o and the operand x of a lane are PARKED in LDS (2 x 9 KiB per wavefront, 72 KiB per workgroup): x is
// needed at the start (rsqrt, projection) and for the very last product only, o once per constituent, and their 72 VGPRs are
// what separates the function (256 + 56 registers with everything live) from two wavefronts per SIMD.  Round 4, A/B on
// one GPU (profiles/r04_projuderiv_parked_ab.log): chain 11.64-11.71 ms -> 10.90-10.92; capping the registers at 256
// instead (55 spilled to scratch) made it 12.2-12.3.  (Round 3: prefetching the operands of constituent j + 1 by LDS-DMA
// changed nothing and was removed again.)
__global__ void __launch_bounds__(256, 2) k_projUderiv_batch(Geom g, ProjBatch B) {
  extern __shared__ double2 smPB[];                 // [wavefront][o | x][9][64]
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.V) return;
  const int mu = blockIdx.y;
  const int p = i >= g.Vh, c = i - p * g.Vh;
  const size_t t = (size_t)p * g.etile + (c >> 6);
  const int l = c & 63;
  // LDS column = the LANE, not the site's position in its tile: where the parity halves meet inside a wavefront (Vh not a
  // multiple of 64) two lanes hold sites with the same tile position
  double2 *ol = smPB + (size_t)(threadIdx.x >> 6) * 2 * 576 + (threadIdx.x & 63), *xl = ol + 576;
  const MViewW f = B.f[mu];
  {
    const M3 o = B.accumulate ? m3_load_nt(f.p + t * f.tstride + l, 64) : m3_zero();
    m3_store(ol, 64, o);
  }
#pragma unroll 1
  for (int j = 0; j < B.nn; j++) {
    const MView X = B.X[mu][j], C = B.C[mu][j];
    const MViewW dst = B.dst[mu][j];
    m3_store(xl, 64, m3_load_nt(X.p + t * X.tstride + l, 64));
    const M3 ch = m3_load_nt(C.p + t * C.tstride + l, 64);
    M3 r = m3_projectUderiv_parked(xl, ch);
    {
      M3 o = m3_load(ol, 64);
      m3_axpy(o, B.ma, r);
      m3_store(ol, 64, o);
    }
#pragma unroll
    for (int k = 0; k < 9; k++) { r.e[k].x *= B.alp; r.e[k].y *= B.alp; }
    m3_store_nt(dst.p + t * dst.tstride + l, 64, r);
  }
  m3_store_nt(f.p + t * f.tstride + l, 64, m3_load(ol, 64));
}
// symStapleDeriv (smearutil.nim:22-50) gathered per site:
//   f1(x) += g2(x) g1(x+mu) c(x+nu)^+ + c(x) g1(x+mu) g2(x+nu)^+ + [g2^+ g1 c(+nu) + c^+ g1 g2(+nu)](x-mu)
//   f2(x) += g1(x) c(x+nu) g1(x+mu)^+ + [g1^+ c g1(+mu)](x-nu)
template <bool SCALED, bool HALO, bool SB, int WPE = 2>   // SCALED: f += coef * (derivative) instead of accumulating in place; SB: scheduling fences; WPE: wavefronts per SIMD the registers are capped for
__global__ void __launch_bounds__(256, WPE) k_staple_deriv(Geom g, MViewW f1, MViewW f2, MView g1, MView g2, MView cf, int mu, int nu, int swz,
                                                      int z1, int z2,      // z1 / z2: f1 / f2 start from zero (first contribution)
                                                      double coef, const int *order, int chunk) {
  int p, c;
  if (order) {   // blocked visiting order (tile_order_table): wavefront w takes slot 4*(b>>3)+w of XCD b&7
    const int slot = 4 * (blockIdx.x >> 3) + (threadIdx.x >> 6);
    const int e = slot < chunk ? order[(blockIdx.x & 7) * chunk + slot] : -1;
    if (e < 0) return;
    p = e & 1; c = (e >> 1) * 64 + (threadIdx.x & 63);
    if (c >= g.Vh) return;
  } else {
    int bid = blockIdx.x;
    if (swz && (gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);   // contiguous site range per XCD
    int i = bid * 256 + threadIdx.x;
    if (i >= g.V) return;
    p = i >= g.Vh; c = i - p * g.Vh;
  }
  int x[4], xpm[4], xpn[4], xmm[4], xmn[4], xmmpn[4], xmnpm[4];
  coords_sm(g, c, p, x);
  shift_sm_t<HALO>(g, x, mu, 1, xpm);
  shift_sm_t<HALO>(g, x, nu, 1, xpn);
  shift_sm_t<HALO>(g, x, mu, -1, xmm);
  shift_sm_t<HALO>(g, x, nu, -1, xmn);
  shift_sm_t<HALO>(g, xmm, nu, 1, xmmpn);
  shift_sm_t<HALO>(g, xmn, mu, 1, xmnpm);
  const size_t o0 = ((size_t)p * g.etile + (c >> 6));
  const int l = c & 63;
  {
    M3 a = (z1 || SCALED) ? m3_zero() : m3_load(f1.p + o0 * f1.tstride + l, 64);
    {
      const M3 g1pm = m3_load(g1.p + site_off_t<HALO>(g, xpm, g1.tstride), 64);
      M3 t = m3_mul_na(g1pm, m3_load(cf.p + site_off_t<HALO>(g, xpn, cf.tstride), 64));
      m3_mac(a, m3_load(g2.p + o0 * g2.tstride + l, 64), t);
      if (SB) __builtin_amdgcn_sched_barrier(0);   // keep the loads of the next product from being hoisted above this one
      t = m3_mul_na(g1pm, m3_load(g2.p + site_off_t<HALO>(g, xpn, g2.tstride), 64));
      m3_mac(a, m3_load(cf.p + o0 * cf.tstride + l, 64), t);
      if (SB) __builtin_amdgcn_sched_barrier(0);   // keep the loads of the next product from being hoisted above this one
    }
    {
      const M3 g1mm = m3_load(g1.p + site_off_t<HALO>(g, xmm, g1.tstride), 64);
      M3 t = m3_mul(g1mm, m3_load(cf.p + site_off_t<HALO>(g, xmmpn, cf.tstride), 64));
      m3_mac_an(a, m3_load(g2.p + site_off_t<HALO>(g, xmm, g2.tstride), 64), t);
      if (SB) __builtin_amdgcn_sched_barrier(0);   // keep the loads of the next product from being hoisted above this one
      t = m3_mul(g1mm, m3_load(g2.p + site_off_t<HALO>(g, xmmpn, g2.tstride), 64));
      m3_mac_an(a, m3_load(cf.p + site_off_t<HALO>(g, xmm, cf.tstride), 64), t);
      if (SB) __builtin_amdgcn_sched_barrier(0);   // keep the loads of the next product from being hoisted above this one
    }
    if (SCALED) {
      M3 o = z1 ? m3_zero() : m3_load(f1.p + o0 * f1.tstride + l, 64);
      m3_axpy(o, coef, a);
      a = o;
    }
    m3_store(f1.p + o0 * f1.tstride + l, 64, a);
  }
  {
    M3 a = (z2 || SCALED) ? m3_zero() : m3_load(f2.p + o0 * f2.tstride + l, 64);
    M3 t = m3_mul_na(m3_load(cf.p + site_off_t<HALO>(g, xpn, cf.tstride), 64), m3_load(g1.p + site_off_t<HALO>(g, xpm, g1.tstride), 64));
    m3_mac(a, m3_load(g1.p + o0 * g1.tstride + l, 64), t);
    if (SB) __builtin_amdgcn_sched_barrier(0);   // keep the loads of the next product from being hoisted above this one
    t = m3_mul(m3_load(cf.p + site_off_t<HALO>(g, xmn, cf.tstride), 64), m3_load(g1.p + site_off_t<HALO>(g, xmnpm, g1.tstride), 64));
    m3_mac_an(a, m3_load(g1.p + site_off_t<HALO>(g, xmn, g1.tstride), 64), t);
    if (SB) __builtin_amdgcn_sched_barrier(0);   // keep the loads of the next product from being hoisted above this one
    if (SCALED) {
      M3 o = z2 ? m3_zero() : m3_load(f2.p + o0 * f2.tstride + l, 64);
      m3_axpy(o, coef, a);
      a = o;
    }
    m3_store(f2.p + o0 * f2.tstride + l, 64, a);
  }
}
// Two symStapleDerivs in one pass.  Every call of the nHYP back-propagation has a partner with the roles of its fields
// exchanged: (g1, g2, cA, mu, nu) adds into (f1, f2) and (g2, g1, cB, nu, mu) adds into (f2, f1) -- hypsmear.nim:175-245
// visits both (mu, nu) and (nu, mu) at every level.  Summed, the pair is
//   F1(x) += g2 T + cA S + g2(-mu)^+ [g1(-mu) cA(-mu+nu) + cB(-mu) g2(-mu+nu)] + cA(-mu)^+ [g1(-mu) g2(-mu+nu)]
//   F2(x) += g1 T^+ + cB S^+ + g1(-nu)^+ [g2(-nu) cB(-nu+mu) + cA(-nu) g1(-nu+mu)] + cB(-nu)^+ [g2(-nu) g1(-nu+mu)]
//   with T = g1(+mu) cA(+nu)^+ + cB(+mu) g2(+nu)^+,  S = g1(+mu) g2(+nu)^+        (fields without argument: at x)
// and F2 is F1 with (g1, cA, mu) <-> (g2, cB, nu).  So ONE code path serves both: wavefront 2k of a workgroup forms F1
// of tile k, wavefront 2k+1 forms F2 of the same tile with the roles exchanged (wavefront-uniform pointer selects).
// Against two single calls: 6 field reads + 2 writes instead of 10 + 4 (the partner's operands are fetched at the same
// moment by the same CU: L1 / L2 hits), products that are adjoints of each other or share a factor formed once per
// wavefront, and only HALF as many tiles in flight per XCD for the same number of resident wavefronts -- the kernel is
// bound by what misses the 4 MB L2 (PMC: 33 % hit rate, 2.7x its unique bytes fetched, for the one-tile-per-wavefront
// form), so the smaller working set is worth more than the duplicated corner products cost.
// (Round 3: FOUR wavefronts per tile -- role x {forward, backward} half, the backward half handing its partial sum over through
// LDS, 64 instead of 128 tiles in flight per XCD -- measured 12.5 against 11.75 ms for the chain and was not kept: the halves wait
// for each other at the hand-off, which costs more than the smaller working set saves.)
template <bool HALO>
__global__ void __launch_bounds__(256) k_staple_deriv_pair(Geom g, MViewW F1, MViewW F2, MView g1, MView g2, MView cA, MView cB, int mu, int nu,
                                                          int z1, int z2, const int *order, int chunk, int nt, int tsel = 0) {
  const int wv = threadIdx.x >> 6;
  const int slot = 2 * (blockIdx.x >> 3) + (wv >> 1);
  const int e = slot < chunk ? order[(blockIdx.x & 7) * chunk + slot] : -1;
  if (e < 0) return;
  const int p = e & 1, c = (e >> 1) * 64 + (threadIdx.x & 63);
  if (c >= g.Vh) return;
  if (HALO && tsel) {
    // t-sharded: a tile lies in ONE t-slice (64 | F), so whole wavefronts take part or leave.  tsel 1: the slices that read
    // no ghost data (0 < t < Xt-1: they run while the chain fields' ghost slices travel), 2: the two boundary slices
    const int t = (e >> 1) * 64 / g.F;
    const bool bnd = t == 0 || t == g.X[3] - 1;
    if (bnd != (tsel == 2)) return;
  }
  const bool second = __builtin_amdgcn_readfirstlane(wv & 1) != 0;      // wavefront-uniform role
  const MViewW Fo = second ? F2 : F1;
  const MView ga = second ? g2 : g1, gb = second ? g1 : g2, ca = second ? cB : cA, cb = second ? cA : cB;
  const int m = second ? nu : mu, n = second ? mu : nu;
  const int zo = second ? z2 : z1;
  int x[4], xpm[4], xpn[4], xmm[4], xmmpn[4];
  coords_sm(g, c, p, x);
  shift_sm_t<HALO>(g, x, m, 1, xpm);
  shift_sm_t<HALO>(g, x, n, 1, xpn);
  shift_sm_t<HALO>(g, x, m, -1, xmm);
  shift_sm_t<HALO>(g, xmm, n, 1, xmmpn);
  const size_t o0 = ((size_t)p * g.etile + (c >> 6));
  const int l = c & 63;
#define LD(F, X) m3_load((F).p + site_off_t<HALO>(g, X, (F).tstride), 64)
#define LD0(F) m3_load((F).p + o0 * (F).tstride + l, 64)
#define FENCE() __builtin_amdgcn_sched_barrier(0)   /* keep the loads of the next group from being hoisted above this one */
  M3 a = zo ? m3_zero() : (nt ? m3_load_nt(Fo.p + o0 * Fo.tstride + l, 64) : LD0(Fo));
  {
    M3 T, S;
    {
      const M3 gap = LD(ga, xpm);
      T = m3_mul_na(gap, LD(ca, xpn));
      FENCE();
      const M3 gbn = LD(gb, xpn);
      S = m3_mul_na(gap, gbn);
      FENCE();
      m3_mac_na(T, LD(cb, xpm), gbn);
    }
    FENCE();
    m3_mac(a, LD0(gb), T);
    FENCE();
    m3_mac(a, LD0(ca), S);
    FENCE();
  }
  {
    M3 T, S;
    {
      const M3 gam = LD(ga, xmm);
      T = m3_mul(gam, LD(ca, xmmpn));
      FENCE();
      const M3 gbq = LD(gb, xmmpn);
      S = m3_mul(gam, gbq);
      FENCE();
      m3_mac(T, LD(cb, xmm), gbq);
    }
    FENCE();
    m3_mac_an(a, LD(gb, xmm), T);
    FENCE();
    m3_mac_an(a, LD(ca, xmm), S);
  }
  if (nt) m3_store_nt(Fo.p + o0 * Fo.tstride + l, 64, a);
  else m3_store(Fo.p + o0 * Fo.tstride + l, 64, a);
#undef LD
#undef LD0
#undef FENCE
}
__global__ void __launch_bounds__(256) k_projectU(Geom g, MViewW dst, MView src) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.V) return;
  const int p = i >= g.Vh, c = i - p * g.Vh;
  const size_t od = ((size_t)p * g.etile + (c >> 6)) * dst.tstride + (c & 63);
  const size_t os = ((size_t)p * g.etile + (c >> 6)) * src.tstride + (c & 63);
  m3_store(dst.p + od, 64, m3_projectU(m3_load(src.p + os, 64)));
}
// ll(x) = naik * U(x) U(x+d) U(x+2d)   (fat7l.nim:146-156)
__global__ void __launch_bounds__(256) k_naik(Geom g, MViewW dst, MView U, int dir, double naik) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.V) return;
  const int p = i >= g.Vh, c = i - p * g.Vh;
  int x[4], x1[4], x2[4];
  coords_sm(g, c, p, x);
  shift_sm(g, x, dir, 1, x1);
  shift_sm(g, x1, dir, 1, x2);
  M3 t = m3_mul(m3_load(U.p + site_off(g, x1, U.tstride), 64), m3_load(U.p + site_off(g, x2, U.tstride), 64));
  M3 m = m3_mul(m3_load(U.p + site_off(g, x, U.tstride), 64), t);
#pragma unroll
  for (int k = 0; k < 9; k++) { m.e[k].x *= naik; m.e[k].y *= naik; }
  m3_store(dst.p + site_off(g, x, dst.tstride), 64, m);
}

// host [idx][mu][9] <-> gauge tiles (same as gauge.hip; local copies keep this file self-contained)
__global__ void __launch_bounds__(256) k_sm_to_tiles(Geom g, const double2 *__restrict__ host, double2 *G) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.V) return;
  int p = i >= g.Vh, c = i - p * g.Vh;
  for (int mu = 0; mu < 4; mu++) {
    double2 *w = G + (((size_t)p * g.etile + (c >> 6)) * 4 + mu) * 576 + (c & 63);
    for (int k = 0; k < 9; k++) w[k * 64] = host[((size_t)i * 4 + mu) * 9 + k];
  }
}
__global__ void __launch_bounds__(256) k_sm_from_tiles(Geom g, double2 *__restrict__ host, const double2 *G) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.V) return;
  int p = i >= g.Vh, c = i - p * g.Vh;
  for (int mu = 0; mu < 4; mu++) {
    const double2 *w = G + (((size_t)p * g.etile + (c >> 6)) * 4 + mu) * 576 + (c & 63);
    for (int k = 0; k < 9; k++) host[((size_t)i * 4 + mu) * 9 + k] = w[k * 64];
  }
}

// derivative of ll(x) = naik U(x) U(x+d) U(x+2d) (fat7l.nim:146-156) w.r.t. the three places of a link:
//   F(x) += naik [ C(x) U(x+2d)^+ U(x+d)^+ + U(x-d)^+ C(x-d) U(x+d)^+ + U(x-d)^+ U(x-2d)^+ C(x-2d) ]
__global__ void __launch_bounds__(256) k_naik_deriv(Geom g, MViewW F, MView U, MView Cl, int dir, double naik) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.V) return;
  const int p = i >= g.Vh, c = i - p * g.Vh;
  int x[4], xp[4], xpp[4], xm[4], xmm[4];
  coords_sm(g, c, p, x);
  shift_sm(g, x, dir, 1, xp);
  shift_sm(g, xp, dir, 1, xpp);
  shift_sm(g, x, dir, -1, xm);
  shift_sm(g, xm, dir, -1, xmm);
  const M3 up = m3_load(U.p + site_off(g, xp, U.tstride), 64), um = m3_load(U.p + site_off(g, xm, U.tstride), 64);
  M3 acc = m3_mul_na(m3_mul_na(m3_load(Cl.p + site_off(g, x, Cl.tstride), 64), m3_load(U.p + site_off(g, xpp, U.tstride), 64)), up);
  m3_mac_na(acc, m3_mul_an(um, m3_load(Cl.p + site_off(g, xm, Cl.tstride), 64)), up);
  m3_mac_an(acc, um, m3_mul_an(m3_load(U.p + site_off(g, xmm, U.tstride), 64), m3_load(Cl.p + site_off(g, xmm, Cl.tstride), 64)));
  double2 *f = F.p + site_off(g, x, F.tstride);
  M3 o = m3_load(f, 64);
  m3_axpy(o, naik, acc);
  m3_store(f, 64, o);
}

namespace {
struct NhypKeep {
  double2 *l1x[4][4], *l1[4][4], *l2x[4][4], *l2[4][4], *flx = nullptr;
};
// k_links_from_nat (layout.hip) reads ONE thing below the slab: U_t on the lower neighbour's top `depth` slices, for the backward t-links
// of the first slices.  Copy the direction-3 sub-blocks of `ntf` whole tiles per parity between a gauge-shaped field and a packed buffer.
__global__ void __launch_bounds__(256) k_tlink_tiles(double2 *G, double2 *buf, size_t etile, size_t tile0, size_t ntf, int to_field) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;        // (parity, tile, element)
  if (i >= 2 * ntf * 576) return;
  const size_t p = i / (ntf * 576), r = i - p * ntf * 576, j = r / 576, e = r - j * 576;
  double2 *f = G + ((p * etile + tile0 + j) * 4 + 3) * 576 + e;
  if (to_field) *f = buf[i]; else buf[i] = *f;
}

struct Smear {
  qexhip_ctx *c;
  Geom g;
  size_t gsz, fsz;  // double2 per gauge field / per single matrix field
  std::vector<double2 *> owned;
  explicit Smear(qexhip_ctx *c_) : c(c_), g(c_->g) {
    fsz = (size_t)2 * g.etile * 576;    // incl. the ghost tiles of a t-sharded field
    gsz = 4 * fsz;
  }
  ~Smear() { (void)hipStreamSynchronize(c->stream); for (auto p : owned) (void)hipFree(p); }
  // One slab for the fields of a long-lived state (the nHYP closure keeps 92 matrix fields, 14 GB at 32^4): a hipMalloc
  // costs milliseconds whatever its size, 77 of them made the first smearGetForce of a context take 240 ms.
  double2 *slab = nullptr;
  size_t slab_n = 0, slab_used = 0;
  int reserve(size_t n) {
    if (slab) return 0;
    HIPCHK(hipMalloc((void **)&slab, n * sizeof(double2)));
    HIPCHK(hipMemsetAsync(slab, 0, n * sizeof(double2), c->stream));
    owned.push_back(slab);
    slab_n = n; slab_used = 0;
    return 0;
  }
  int alloc(double2 **p, size_t n) {
    const size_t n16 = (n + 15) & ~(size_t)15;            // keep every field 256-byte aligned inside the slab
    if (slab && slab_used + n16 <= slab_n) {
      *p = slab + slab_used;                              // already zeroed by reserve()
      slab_used += n16;
      return 0;
    }
    // scratch of a long-lived closure (scratch_keep): the k-th allocation of a call takes the k-th buffer of the previous call again
    // instead of a hipMalloc / hipFree pair per field and call -- milliseconds each, and after a large free (the nHYP closure's 14 GB
    // slab) ONE such call was seen to stall for 0.95 s in the runtime (profiles/r06_notes.md section 6)
    if (scratch_keep && scratch_next < scratch.size() && scratch[scratch_next].second >= n) {
      *p = scratch[scratch_next++].first;
      HIPCHK(hipMemsetAsync(*p, 0, n * sizeof(double2), c->stream));
      return 0;
    }
    HIPCHK(hipMalloc((void **)p, n * sizeof(double2)));
    HIPCHK(hipMemsetAsync(*p, 0, n * sizeof(double2), c->stream));
    owned.push_back(*p);
    if (scratch_keep) {
      if (scratch_next < scratch.size()) scratch[scratch_next] = {*p, n};      // (a larger request than last time: the old buffer stays owned, unused)
      else scratch.push_back({*p, n});
      scratch_next++;
    }
    return 0;
  }
  // per-call scratch kept between calls: scratch_begin() before the first per-call alloc of a call (everything allocated earlier is state)
  bool scratch_keep = false;
  std::vector<std::pair<double2 *, size_t>> scratch;
  size_t scratch_next = 0;
  void scratch_begin() { scratch_keep = true; scratch_next = 0; }
  int nb() const { return (g.V + 255) / 256; }
  // t-sharded: refresh the ghost slices of a field (tstride 576: one matrix field, 2304: gauge-shaped) to `depth`;
  // the caller does this for every field that is about to be read at shifted sites.  No-op on one GPU without ghosts.
  int ghosts(const double2 *field, int tstride, int depth = 1) { return ghosts_many(&field, 1, tstride, depth, 0); }
  // the same for n fields of one shape in ONE RCCL group.  async = 1: posted on the comm stream behind what the compute
  // stream has produced so far, and NOT waited for: the compute stream goes on with kernels that do not read these ghosts
  // (the next field of the same smearing level), ghosts_join() comes before the first kernel that does.
  int ghosts_many(const double2 *const *fields, int n, int tstride, int depth, int async) {
    if (!g.halo) return 0;
    if (n < 1 || n > 12) { qexhip_set_error("internal: ghosts_many takes 1..12 fields"); return -3; }
    const size_t tile2 = (size_t)tstride * 2, ft = (size_t)g.F / 64;
    double *bottom[24], *top[24], *ghi[24], *glo[24];
    for (int k = 0; k < n; k++)
      for (int p = 0; p < 2; p++) {
        double *base = (double *)fields[k] + (size_t)p * g.etile * tile2;
        bottom[2 * k + p] = base;
        top[2 * k + p] = base + ((size_t)g.ntile - depth * ft) * tile2;
        ghi[2 * k + p] = base + (size_t)g.ntile * tile2;
        glo[2 * k + p] = base + ((size_t)g.ntile + 3 * ft + (3 - depth) * ft) * tile2;
      }
    if (async) HIPCHK(hipEventRecord(c->ev_ready, c->stream));
    ScopedTimer tm(c, "smear_halo", async ? c->cstream : c->stream);
    return comm_faces_exchange(c, 2 * n, bottom, top, ghi, glo, (size_t)depth * ft * tile2, async);
  }
  int ghosts_f_async(const double2 *f) { return ghosts_many(&f, 1, 576, 1, 1); }
  int ghosts_join() {
    if (!g.halo) return 0;
    CHK(devjoin_signal(c, c->cstream));               // device-side join (peer.hip), as the sweeps: no cross-queue event dependency
    CHK(devjoin_wait(c, c->stream, c->cstream));
    return 0;
  }
  int ghosts_f(const double2 *f, int depth = 1) { return ghosts(f, 576, depth); }
  int ghosts_g(const double2 *G, int depth = 1) { return ghosts(G, 4 * 576, depth); }
  // What links_from_natural needs of a gauge field's ghost slices and no more: the t-links of the lower neighbour's top `depth` slices in
  // MY ghost_lo slices (k_links_from_nat: the backward t-links of the first slices are U_t(x - hop t)^+).  One direction of four, one way:
  // a quarter of the bytes of ghosts_g (48^3 x 12: 48 instead of 191 MB for the Naik links' three slices).  Packed through the staging buffer.
  int ghosts_tlinks_lo(double2 *G, int depth) {
    if (!g.halo) return 0;
    const size_t ft = (size_t)g.F / 64, ntf = (size_t)depth * ft, n2 = 2 * ntf * 576;
    CHK(ensure_stage(c, 2 * n2 * sizeof(double2)));
    double2 *snd = (double2 *)c->stage, *rcv = snd + n2;
    const unsigned nb = (unsigned)((n2 + 255) / 256);
    ScopedTimer tm(c, "smear_halo", c->stream);
    k_tlink_tiles<<<nb, 256, 0, c->stream>>>(G, snd, (size_t)g.etile, (size_t)g.ntile - ntf, ntf, 0);
    HIPCHK(hipGetLastError());
    CHK(comm_exchange_raw(c, snd, rcv, n2 * sizeof(double2), c->stream));       // my top slices -> the upper neighbour; the lower one's -> me
    k_tlink_tiles<<<nb, 256, 0, c->stream>>>(G, rcv, (size_t)g.etile, (size_t)g.ntile + 3 * ft + (3 - depth) * ft, ntf, 1);
    HIPCHK(hipGetLastError());
    return 0;
  }
  MView gv(const double2 *G, int mu) const { return MView{G + (size_t)mu * 576, 4 * 576}; }
  MViewW gvw(double2 *G, int mu) const { return MViewW{G + (size_t)mu * 576, 4 * 576}; }
  MView fv(const double2 *F) const { return MView{F, 576}; }
  MViewW fvw(double2 *F) const { return MViewW{F, 576}; }
  int upload(double2 *G, const double *host) {
    const size_t bytes = (size_t)g.V * 72 * sizeof(double);
    CHK(ensure_stage(c, bytes));
    HIPCHK(hipMemcpyAsync(c->stage, host, bytes, hipMemcpyHostToDevice, c->stream));
    k_sm_to_tiles<<<nb(), 256, 0, c->stream>>>(g, (const double2 *)c->stage, G);
    HIPCHK(hipGetLastError());
    return 0;
  }
  int download(double *host, const double2 *G) {
    const size_t bytes = (size_t)g.V * 72 * sizeof(double);
    CHK(ensure_stage(c, bytes));
    k_sm_from_tiles<<<nb(), 256, 0, c->stream>>>(g, (double2 *)c->stage, G);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host, c->stage, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
  }
  // ext > 0 (t-sharded only): also on the `ext` ghost slices either side of the slab -- the operands must be valid `ext + 1`
  // slices out (communication-avoiding smearing levels: nhyp() below)
  // ext_acc = false: on the ghost slices only the staple field `st` is wanted (fat7: the accumulator is read on the slab only)
  // part: 0 = the body launch and (ext > 0) the two ghost-slice launches; 1 = the body launch only; 2 = the ghost-slice launches only
  int staple(MView A, MView B, int mu, int nu, MViewW st, MViewW acc, double coef, MView init = MView{nullptr, 0}, double cinit = 0.0,
             MViewW proj = MViewW{nullptr, 0}, int ext = 0, bool ext_acc = true, int part = 0) {
    ScopedTimer tm(c, "smear", c->stream);
    constexpr int swz = 1;     // XCD-aware block remap of the gather kernels (measured winner, profiles/r02_pmc_staple_kernels_order.log)
    const int *order; int chunk, nblk;
    CHK(smear_order(c, g, &order, &chunk, &nblk, mu, nu));
    // Dynamic LDS the kernel never touches, as an occupancy limiter: 60 KB per workgroup admits two workgroups (8
    // wavefronts = 8 tiles) per CU instead of the 3-4 its registers allow.  The kernel is bound by what misses the XCD's
    // 4 MB L2, and fewer tiles in flight keep the lines its wavefronts share alive: nHYP smearing 7.27 -> 6.70 ms at 32^4
    // (45 KB: 7.14, 100 KB = one workgroup per CU: 7.41; none: 7.27).
    constexpr int ldsb = 60000;
    constexpr int gnt = 1;     // non-temporal stores of the staple / accumulator (written once, read by a later kernel)
    if (part != 2) {
      if (g.halo) k_gen_staple<true><<<nblk, 256, ldsb, c->stream>>>(g, A, B, mu, nu, st, acc, coef, swz, init, cinit, proj, order, chunk, gnt);
      else k_gen_staple<false><<<nblk, 256, ldsb, c->stream>>>(g, A, B, mu, nu, st, acc, coef, swz, init, cinit, proj, order, chunk, gnt);
    }
    if (g.halo && ext > 0 && part != 1) {
      // ghost_hi holds the virtual slices Xt .. Xt+2 at [Vh, Vh + 3F), ghost_lo the slices -3 .. -1 at [Vh + 3F, Vh + 6F)
      const int F = g.F, hi0 = g.Vh, lo1 = g.Vh + 6 * F;
      const int nb = (2 * ext * F + 255) / 256;
      const MViewW ga = ext_acc ? acc : MViewW{nullptr, 0}, gp = ext_acc ? proj : MViewW{nullptr, 0};
      k_gen_staple<true><<<nb, 256, 0, c->stream>>>(g, A, B, mu, nu, st, ga, coef, 0, init, cinit, gp, nullptr, 0, gnt, hi0, hi0 + ext * F);
      k_gen_staple<true><<<nb, 256, 0, c->stream>>>(g, A, B, mu, nu, st, ga, coef, 0, init, cinit, gp, nullptr, 0, gnt, lo1 - ext * F, lo1);
    }
    HIPCHK(hipGetLastError());
    return 0;
  }
  // makeImpLinks (fat7l.nim:77-161) on device gauge fields
  int fat7(double2 *fl, const double2 *gf, const double coef[5], double2 *ll, const double2 *gfLong, double naik) {
    const double c3 = coef[1], c5 = coef[2], c7 = coef[3], cL = coef[4];
    const double c1 = coef[0] - 6.0 * cL;
    const bool have5 = (c5 != 0.0) || (c7 != 0.0) || (cL != 0.0);
    const bool have3 = (c3 != 0.0) || have5;
    double2 *stp = nullptr, *tmp = nullptr;
    CHK(alloc(&stp, fsz));
    CHK(alloc(&tmp, fsz));
    const MViewW none{nullptr, 0};
    // t-sharded, communication-avoiding (round 5, as the nHYP levels): the thin links arrive three slices deep ONCE; the
    // 3-staple field is computed on the slab plus two ghost slices either side (one without 7-staples), the 5-staple field on the
    // slab plus one -- 36 single-field refreshes per pass (12 + 24 x 16 MB per direction at 48^3 x 12) become none.
    const bool ca = g.halo && c->opt_smear_ca;
    const int ext5 = ca ? (c7 != 0.0 ? 1 : 0) : 0, ext3 = ca ? (have5 ? ext5 + 1 : 0) : 0;
    if (ca) CHK(ghosts_g(gf, 3));
    for (int dir = 0; dir < 4; dir++) {
      k_mscale<<<nb(), 256, 0, c->stream>>>(g, gvw(fl, dir), c1, gv(gf, dir));
      HIPCHK(hipGetLastError());
      if (!have3) continue;
      for (int nu = 0; nu < 4; nu++) {
        if (nu == dir) continue;
        CHK(staple(gv(gf, nu), gv(gf, dir), dir, nu, fvw(stp), gvw(fl, dir), c3, MView{nullptr, 0}, 0.0, none, ext3, false));
        if (have5 && !ca) CHK(ghosts_f(stp));               // the staple is the middle link of the next level
        if (cL != 0.0) CHK(staple(gv(gf, nu), fv(stp), dir, nu, none, gvw(fl, dir), cL));
        if (c5 != 0.0 || c7 != 0.0)
          for (int rho = 0; rho < 4; rho++) {
            if (rho == dir || rho == nu) continue;
            CHK(staple(gv(gf, rho), fv(stp), dir, rho, fvw(tmp), gvw(fl, dir), c5, MView{nullptr, 0}, 0.0, none, ext5, false));
            if (c7 != 0.0 && !ca) CHK(ghosts_f(tmp));
            if (c7 != 0.0)
              for (int sig = 0; sig < 4; sig++) {
                if (sig == dir || sig == nu || sig == rho) continue;
                CHK(staple(gv(gf, sig), fv(tmp), dir, sig, none, gvw(fl, dir), c7));
              }
          }
      }
    }
    if (naik != 0.0 && ll)
      for (int dir = 0; dir < 4; dir++) {
        k_naik<<<nb(), 256, 0, c->stream>>>(g, gvw(ll, dir), gv(gfLong, dir), dir, naik);
        HIPCHK(hipGetLastError());
      }
    return 0;
  }
  int sderiv(MViewW f1, MViewW f2, MView g1, MView g2, MView cf, int mu, int nu, double coef) {
    constexpr int swz = 1;     // XCD-aware block remap of the gather kernels (measured winner, profiles/r02_pmc_staple_kernels_order.log)
    ScopedTimer tm(c, "smear_deriv", c->stream);
    // (scheduling fences cost 3 % here: 48.5 vs 46.9 ms per HISQ force, A/B on one GPU)
    const int *order; int chunk, nblk;
    CHK(smear_order(c, g, &order, &chunk, &nblk, mu, nu));
    if (g.halo) k_staple_deriv<true, true, false><<<nblk, 256, 0, c->stream>>>(g, f1, f2, g1, g2, cf, mu, nu, swz, 0, 0, coef, order, chunk);
    else k_staple_deriv<true, false, false><<<nblk, 256, 0, c->stream>>>(g, f1, f2, g1, g2, cf, mu, nu, swz, 0, 0, coef, order, chunk);
    HIPCHK(hipGetLastError());
    return 0;
  }
  // reverse of fat7 (gauge/fat7lderiv.nim): d += d/dU^+ of sum Re tr(cfl^+ fl) + sum Re tr(cll^+ ll), over the same
  // staple graph as fat7() above, every generic staple differentiated by k_staple_deriv (symStapleDeriv)
  int fat7_deriv(double2 *d, const double2 *gf, const double2 *cfl, const double coef[5], const double2 *cll, double naik) {
    const double c3 = coef[1], c5 = coef[2], c7 = coef[3], cL = coef[4];
    const double c1 = coef[0] - 6.0 * cL;
    const bool have5 = (c5 != 0.0) || (c7 != 0.0) || (cL != 0.0);
    const bool have3 = (c3 != 0.0) || have5;
    double2 *st1, *tmp, *ast1, *atmp;
    CHK(alloc(&st1, fsz)); CHK(alloc(&tmp, fsz)); CHK(alloc(&ast1, fsz)); CHK(alloc(&atmp, fsz));
    const MViewW none{nullptr, 0};
    for (int dir = 0; dir < 4; dir++) {
      const MView ch = gv(cfl, dir);
      // d[dir] += c1 * chain: an accumulate with the generic scale kernel's twin
      k_maxpy<<<nb(), 256, 0, c->stream>>>(g, gvw(d, dir), c1, ch);
      HIPCHK(hipGetLastError());
      if (!have3) continue;
      for (int nu = 0; nu < 4; nu++) {
        if (nu == dir) continue;
        CHK(staple(gv(gf, nu), gv(gf, dir), dir, nu, fvw(st1), none, 0.0));
        if (have5) CHK(ghosts_f(st1));
        k_mscale<<<nb(), 256, 0, c->stream>>>(g, fvw(ast1), c3, ch);
        if (cL != 0.0) CHK(sderiv(gvw(d, nu), fvw(ast1), gv(gf, nu), fv(st1), ch, dir, nu, cL));
        if (c5 != 0.0 || c7 != 0.0)
          for (int rho = 0; rho < 4; rho++) {
            if (rho == dir || rho == nu) continue;
            CHK(staple(gv(gf, rho), fv(st1), dir, rho, fvw(tmp), none, 0.0));
            if (c7 != 0.0) CHK(ghosts_f(tmp));
            k_mscale<<<nb(), 256, 0, c->stream>>>(g, fvw(atmp), c5, ch);
            if (c7 != 0.0)
              for (int sig = 0; sig < 4; sig++) {
                if (sig == dir || sig == nu || sig == rho) continue;
                CHK(sderiv(gvw(d, sig), fvw(atmp), gv(gf, sig), fv(tmp), ch, dir, sig, c7));
              }
            CHK(ghosts_f(atmp));                            // adjoints are chains of the next derivative: read shifted
            CHK(sderiv(gvw(d, rho), fvw(ast1), gv(gf, rho), fv(st1), fv(atmp), dir, rho, 1.0));
          }
        CHK(ghosts_f(ast1));
        CHK(sderiv(gvw(d, nu), gvw(d, dir), gv(gf, nu), gv(gf, dir), fv(ast1), dir, nu, 1.0));
      }
    }
    if (naik != 0.0 && cll)
      for (int dir = 0; dir < 4; dir++) {
        k_naik_deriv<<<nb(), 256, 0, c->stream>>>(g, gvw(d, dir), gv(gf, dir), gv(cll, dir), dir, naik);
        HIPCHK(hipGetLastError());
      }
    return 0;
  }
  // HisqCoefs.smearGetForce's smearedForce (gauge/hisqsmear.nim:55-90): second fat7 + Naik, projectU, first fat7, in reverse
  int hisq_force(const double2 *G, const double2 *CF, const double2 *CL, double2 *F) {
    double2 *V, *W;
    CHK(alloc(&V, gsz)); CHK(alloc(&W, gsz));
    CHK(hisq_first(G, V, W));
    return hisq_reverse(G, V, W, CF, CL, F);
  }
  // first half of the smearing, the part the reverse pass needs: V = fat7_1(G), W = projectU(V) (ghosts of G, W refreshed)
  int hisq_first(const double2 *G, double2 *V, double2 *W, bool refresh_w = true) {
    const double f7lf = 0.0;
    const double c_first[5] = {(1.0 + 3.0 * f7lf + 0.0) / 8.0, -1.0 / 16.0, 1.0 / 64.0, -1.0 / 384.0, -f7lf / 16.0};
    if (!(g.halo && c->opt_smear_ca)) CHK(ghosts_g(G));    // (communication-avoiding: fat7 fetches its input three slices deep itself)
    CHK(fat7(V, G, c_first, nullptr, G, 0.0));
    for (int mu = 0; mu < 4; mu++) k_projectU<<<nb(), 256, 0, c->stream>>>(g, gvw(W, mu), gv(V, mu));
    HIPCHK(hipGetLastError());
    // refresh_w = false: the caller's next step is the communication-avoiding second fat7 pass, which fetches W three slices deep itself
    // (and with it what a later reverse pass needs): the two-slice refresh here would be a second exchange of the same field
    // (2.85 ms at 45 GB/s on a 48^3 face; round 6).  hisq_force, which goes straight into the reverse pass, keeps it.
    return refresh_w ? ghosts_g(W, 2) : 0;
  }
  int hisq_reverse(const double2 *G, const double2 *V, const double2 *W, const double2 *CF, const double2 *CL, double2 *F) {
    const double f7lf = 0.0, naik = 1.0, f2 = 2.0 - f7lf;
    const double c_first[5] = {(1.0 + 3.0 * f7lf + 0.0) / 8.0, -1.0 / 16.0, 1.0 / 64.0, -1.0 / 384.0, -f7lf / 16.0};
    const double c_second[5] = {(1.0 + 3.0 * f2 + naik) / 8.0, -1.0 / 16.0, 1.0 / 64.0, -1.0 / 384.0, -f2 / 16.0};
    double2 *T;
    CHK(alloc(&T, gsz));                                               // alloc zero-fills
    CHK(ghosts_g(CF)); CHK(ghosts_g(CL, 2));
    CHK(fat7_deriv(T, W, CF, c_second, CL, -naik / 24.0));
    for (int mu = 0; mu < 4; mu++)
      k_projUderiv<<<nb(), 256, 0, c->stream>>>(g, gvw(T, mu), gv(W, mu), gv(V, mu), gv(T, mu), MViewW{nullptr, 0}, 0.0, 1.0, 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(F, 0, gsz * sizeof(double2), c->stream));
    CHK(ghosts_g(T));
    return fat7_deriv(F, G, T, c_first, nullptr, 0.0);
  }
  // HisqCoefs.init + smear (hisqLinks.nim:9-43) on device fields
  int hisq(const double2 *G, double2 *FL, double2 *LL) {
    const double f7lf = 0.0, naik = 1.0, f2 = 2.0 - f7lf;
    const double c_first[5] = {(1.0 + 3.0 * f7lf + 0.0) / 8.0, -1.0 / 16.0, 1.0 / 64.0, -1.0 / 384.0, -f7lf / 16.0};
    const double c_second[5] = {(1.0 + 3.0 * f2 + naik) / 8.0, -1.0 / 16.0, 1.0 / 64.0, -1.0 / 384.0, -f2 / 16.0};
    double2 *T1, *T2;
    CHK(alloc(&T1, gsz)); CHK(alloc(&T2, gsz));
    if (!(g.halo && c->opt_smear_ca)) CHK(ghosts_g(G));
    CHK(fat7(T1, G, c_first, nullptr, G, 0.0));
    for (int mu = 0; mu < 4; mu++) {
      k_projectU<<<nb(), 256, 0, c->stream>>>(g, gvw(T2, mu), gv(T1, mu));
      HIPCHK(hipGetLastError());
    }
    if (!(g.halo && c->opt_smear_ca)) CHK(ghosts_g(T2, 2)); // Naik: x+d, x+2d
    return fat7(FL, T2, c_second, LL, T2, -naik / 24.0);
  }
  // nHYP forward smearing (hypsmear.nim:49-144) on device fields; with `keep` the unprojected and
  // projected level-1/2 fields and the unprojected level-3 sum stay alive for the force chain
  int nhyp(const double2 *G, double2 *FL, double a1, double a2, double a3, NhypKeep *keep = nullptr, bool reuse = false) {
    double2 *tmp = nullptr;
    NhypKeep loc;
    NhypKeep &K = keep ? *keep : loc;
    if (!keep) CHK(alloc(&tmp, fsz));
    for (int mu = 0; mu < 4 && !reuse; mu++)
      for (int nu = 0; nu < 4; nu++) {
        K.l1[mu][nu] = K.l2[mu][nu] = K.l1x[mu][nu] = K.l2x[mu][nu] = nullptr;
        if (mu == nu) continue;
        CHK(alloc(&K.l1[mu][nu], fsz)); CHK(alloc(&K.l2[mu][nu], fsz));
        if (keep) { CHK(alloc(&K.l1x[mu][nu], fsz)); CHK(alloc(&K.l2x[mu][nu], fsz)); }
        else K.l1x[mu][nu] = K.l2x[mu][nu] = tmp;
      }
    if (keep && !reuse) CHK(alloc(&K.flx, gsz));
    const double alp1 = a1 / 2.0, alp2 = a2 / 4.0, alp3 = a3 / 6.0;
    const MViewW none{nullptr, 0};
    const MView noinit{nullptr, 0};
    // every level: first staple starts the sum from ma * U_mu, last staple projects it (fused, see k_gen_staple).
    // t-sharded, communication-avoiding (round 5): ONE exchange -- the thin links three slices deep -- and then level 1 is
    // computed on the slab plus two ghost slices either side, level 2 on the slab plus one, level 3 on the slab: the ghost
    // slices of the projected level-1/2 fields are computed here, redundantly with the neighbour that owns them, instead of
    // being fetched (rounds 1-4: 12 + 12 matrix-field refreshes of 16 MB per direction at 48^3 x 12 against 3 x 64 MB once,
    // i.e. 382 -> 191 MB per direction, and no exchange left inside the levels; profiles/r05_notes.md).  Extra arithmetic:
    // 4 / T of level 1 and 2 / T of level 2.
    const bool ca = g.halo && c->opt_smear_ca;      // option "smear_ca" = 0: the per-field refreshes of rounds 1-4 (A/B, test hook)
    if (ca) {
      // Round 6: the one exchange travels beside what does not need it.  A level-1 staple gathers in its (mu, nu) plane only, so the BODY
      // launches of the six purely spatial planes read no ghost slice: they run while the thin links' three ghost slices are on the
      // way; behind the join come the body launches of the six planes that contain t and every plane's launches ON the ghost slices.
      // Same launches, another order: bit-identical.  (+4.2 -> +2.7 ms exposed per smear at 48^3 x 12, 45 GB/s; profiles/r06_notes.md section 6)
      const double2 *gp = G;
      CHK(ghosts_many(&gp, 1, 4 * 576, 3, 1));
      for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) CHK(ghosts_join());
        for (int mu = 0; mu < 4; mu++)
          for (int nu = 0; nu < 4; nu++) {
            if (nu == mu) continue;
            const bool spatial = mu != 3 && nu != 3;
            const int part = pass == 0 ? (spatial ? 1 : -1) : (spatial ? 2 : 0);
            if (part < 0) continue;
            CHK(staple(gv(G, nu), gv(G, mu), mu, nu, none, fvw(K.l1x[mu][nu]), alp1, gv(G, mu), 1 - a1, fvw(K.l1[mu][nu]), 2, true, part));
          }
      }
    } else {
      CHK(ghosts_g(G, 1));
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++) {
          if (nu == mu) continue;
          CHK(staple(gv(G, nu), gv(G, mu), mu, nu, none, fvw(K.l1x[mu][nu]), alp1, gv(G, mu), 1 - a1, fvw(K.l1[mu][nu]), 0));
          if (g.halo) CHK(ghosts_f_async(K.l1[mu][nu]));                  // travels while the next (mu, nu) is computed
        }
      if (g.halo) CHK(ghosts_join());
    }
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) {
        if (nu == mu) continue;
        int cnt = 0;
        for (int a = 0; a < 4; a++) {
          if (a == mu || a == nu) continue;
          const int b = 6 - mu - nu - a;
          const bool first = cnt == 0, last = cnt == 1;
          CHK(staple(fv(K.l1[a][b]), fv(K.l1[mu][b]), mu, a, none, fvw(K.l2x[mu][nu]), alp2, first ? gv(G, mu) : noinit, 1 - a2,
                     last ? fvw(K.l2[mu][nu]) : none, ca ? 1 : 0));
          cnt++;
        }
        if (g.halo && !ca) CHK(ghosts_f_async(K.l2[mu][nu]));
      }
    if (g.halo && !ca) CHK(ghosts_join());
    for (int mu = 0; mu < 4; mu++) {
      const MViewW x3 = keep ? gvw(K.flx, mu) : fvw(tmp);
      int cnt = 0;
      for (int nu = 0; nu < 4; nu++) {
        if (nu == mu) continue;
        CHK(staple(fv(K.l2[nu][mu]), fv(K.l2[mu][nu]), mu, nu, none, x3, alp3, cnt == 0 ? gv(G, mu) : noinit, 1 - a3,
                   cnt == 2 ? gvw(FL, mu) : none));
        cnt++;
      }
    }
    return 0;
  }
};
}  // namespace

// smearGetForce's closure (hypsmear.nim:49-247): everything smearedForce needs, resident on the device
struct NhypState {
  Smear S;
  NhypKeep K;
  double2 *G = nullptr, *FL = nullptr, *F = nullptr, *fc = nullptr, *fl1[4][4], *fl2[4][4];
  double a1 = 0, a2 = 0, a3 = 0;
  explicit NhypState(qexhip_ctx *c) : S(c) {}
};
const double2 *gauge_resident_links(qexhip_ctx *c);   // gauge.hip
void nhyp_state_free(qexhip_ctx *c) {
  if (c->nhyp) { delete (NhypState *)c->nhyp; c->nhyp = nullptr; }
}
int nhyp_prepare(qexhip_ctx *c, const double *g_host, double a1, double a2, double a3, double *fl_host) {
  if (c->g.halo && c->g.X[3] < 4) { qexhip_set_error("t-sharded smearing needs a local t extent >= 4"); return -1; }
  for (int i = 0; i < 4; i++) if (c->g.X[i] < 2) { qexhip_set_error("nhyp force chain needs local extents >= 2"); return -1; }
  NhypState *st = (NhypState *)c->nhyp;
  const bool fresh = !st;       // a second smearGetForce on this context reuses the ~70 device fields
  if (fresh) {
    st = new NhypState(c);
    c->nhyp = st;
  }
  Smear &S = st->S;
  st->a1 = a1; st->a2 = a2; st->a3 = a3;
  if (fresh) {
    // G, FL, F, fc, flx: 5 gauge-shaped fields; fl1, fl2: 24 matrix fields; l1, l2, l1x, l2x: 48 (nhyp() below)
    CHK(S.reserve(5 * S.gsz + 72 * S.fsz + 128 * 16));
    CHK(S.alloc(&st->G, S.gsz)); CHK(S.alloc(&st->FL, S.gsz)); CHK(S.alloc(&st->F, S.gsz)); CHK(S.alloc(&st->fc, S.gsz));
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) {
        st->fl1[mu][nu] = st->fl2[mu][nu] = nullptr;
        if (mu != nu) { CHK(S.alloc(&st->fl1[mu][nu], S.fsz)); CHK(S.alloc(&st->fl2[mu][nu], S.fsz)); }
      }
  }
  if (g_host) {
    CHK(S.upload(st->G, g_host));
  } else {                                       // resident MD: the thin links are the device field of qexhip_gauge_set / md_*
    const double2 *U = gauge_resident_links(c);
    if (!U) { qexhip_set_error("nhyp_prepare(g = NULL) needs a resident gauge field (qexhip_gauge_set / qexhip_md_begin)"); return -3; }
    HIPCHK(hipMemcpyAsync(st->G, U, S.gsz * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
  }
  CHK(S.nhyp(st->G, st->FL, a1, a2, a3, &st->K, !fresh));
  if (fl_host) CHK(S.download(fl_host, st->FL));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}
// smearedForce(f, chain) on the device field st->F (in: chain, out: f)   (hypsmear.nim:146-245)
// the (mu, nu) and (nu, mu) symStapleDerivs of a level in one pass (k_staple_deriv_pair)
static int staple_deriv_pair(qexhip_ctx *c, const Geom &g, MViewW F1, MViewW F2, MView g1, MView g2, MView cA, MView cB, int mu, int nu,
                             int z1, int z2, int tsel = 0) {
  const int *order; int chunk, nblk;
  CHK(smear_order(c, g, &order, &chunk, &nblk, mu, nu));
  const int nb2 = 8 * ((chunk + 1) / 2);         // two wavefronts per tile: 2 table slots per 256-thread workgroup
  constexpr int nt = 1;
  if (g.halo) k_staple_deriv_pair<true><<<nb2, 256, 0, c->stream>>>(g, F1, F2, g1, g2, cA, cB, mu, nu, z1, z2, order, chunk, nt, tsel);
  else k_staple_deriv_pair<false><<<nb2, 256, 0, c->stream>>>(g, F1, F2, g1, g2, cA, cB, mu, nu, z1, z2, order, chunk, nt);
  HIPCHK(hipGetLastError());
  return 0;
}
static int nhyp_backward_dev(qexhip_ctx *c, NhypState *st) {
  Smear &S = st->S;
  const Geom &g = S.g;
  const int nblk = S.nb();
  const double alp1 = st->a1 / 2.0, alp2 = st->a2 / 4.0, alp3 = st->a3 / 6.0;
  const double ma1 = 1 - st->a1, ma2 = 1 - st->a2, ma3 = 1 - st->a3;
  ScopedTimer tm(c, "nhyp_force", c->stream);
  if (!(c->lds_attr_done & 16)) {
    HIPCHK(hipFuncSetAttribute((const void *)k_projUderiv_batch, hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
    c->lds_attr_done |= 16;
  }
#define QX_PB_LAUNCH k_projUderiv_batch<<<dim3(nblk, 4), 256, 73728, c->stream>>>(g, PB)
  // the projectUderiv calls of one level in ONE launch (k_projUderiv_batch: grid.y = direction, up to three fields each)
  ProjBatch PB;
  {
    for (int mu = 0; mu < 4; mu++) {
      PB.dst[mu][0] = S.gvw(st->fc, mu); PB.X[mu][0] = S.gv(st->K.flx, mu); PB.C[mu][0] = S.gv(st->F, mu);
      PB.f[mu] = S.gvw(st->F, mu);
    }
    PB.nn = 1; PB.accumulate = 0; PB.ma = ma3; PB.alp = alp3;
    QX_PB_LAUNCH;
  }
  HIPCHK(hipGetLastError());
  // t-sharded: the chain fields of a level are read at shifted sites by the next staple derivative, so their ghost slices have
  // to arrive first -- 4 / 12 / 12 matrix fields per level, 64 / 191 / 191 MB per direction at 48^3 x 12.  Round 5: the exchange is
  // posted on the comm stream and the derivative runs in two passes, first the slices that read no ghost data (0 < t < Xt-1)
  // beside it, then, after the join, the two boundary slices.  Every site still receives its contributions in the same call
  // order, so the result is bit-identical to the one-pass form (option "chain_overlap" = 0).
  const bool split = g.halo && c->opt_chain_overlap && g.X[3] >= 4;
  auto refresh = [&](const double2 *const *fs, int nf, int tstride) -> int {
    if (!g.halo) return 0;
    return S.ghosts_many(fs, nf, tstride, 1, split ? 1 : 0);
  };
  auto passes = [&](auto &&level) -> int {       // level(tsel): all staple-derivative launches of one level
    if (!split) return level(0);
    CHK(level(1));
    CHK(S.ghosts_join());
    return level(2);
  };
  {
    const double2 *fs[1] = {st->fc};
    CHK(refresh(fs, 1, 4 * 576));
  }
  CHK(passes([&](int tsel) -> int {
    bool w[4][4] = {};          // fl2 is a sum of several staple derivatives: the first contribution of a PASS to each field writes, the rest accumulate
    for (int mu = 0; mu < 4; mu++)
      for (int nu = mu + 1; nu < 4; nu++) {       // (mu, nu) and (nu, mu) together
        CHK(staple_deriv_pair(c, g, S.fvw(st->fl2[nu][mu]), S.fvw(st->fl2[mu][nu]), S.fv(st->K.l2[nu][mu]), S.fv(st->K.l2[mu][nu]),
                              S.gv(st->fc, mu), S.gv(st->fc, nu), mu, nu, !w[nu][mu], !w[mu][nu], tsel));
        w[nu][mu] = w[mu][nu] = true;
      }
    return 0;
  }));
  HIPCHK(hipGetLastError());
  {
    for (int mu = 0; mu < 4; mu++) {
      int j = 0;
      for (int nu = 0; nu < 4; nu++) {
        if (nu == mu) continue;
        PB.dst[mu][j] = S.fvw(st->fl2[mu][nu]); PB.X[mu][j] = S.fv(st->K.l2x[mu][nu]); PB.C[mu][j] = S.fv(st->fl2[mu][nu]);
        j++;
      }
      PB.f[mu] = S.gvw(st->F, mu);
    }
    PB.nn = 3; PB.accumulate = 1; PB.ma = ma2; PB.alp = alp2;
    QX_PB_LAUNCH;
    HIPCHK(hipGetLastError());
    {
      // The chain fields of the level in ONE group -- those the next level reads across the t boundary.  A staple derivative gathers in
      // its (mu, a) plane only (k_staple_deriv_pair: x +- mu, x +- a and the two corners), and fl2[m][n] enters the calls whose plane
      // contains m and not n: fl2[m][3] (m != 3) is never read at a t-shifted site.  9 of 12 fields travel (round 6).
      const double2 *fs[12]; int nf = 0;
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++)
          if (nu != mu && !(nu == 3 && mu != 3)) fs[nf++] = st->fl2[mu][nu];
      CHK(refresh(fs, nf, 576));
    }
  }
  HIPCHK(hipGetLastError());
  // the call (mu, nu, a) and its partner (a, nu, mu) share b = 6 - mu - nu - a and exchange the roles of their fields
  CHK(passes([&](int tsel) -> int {
    bool w[4][4] = {};
    for (int mu = 0; mu < 4; mu++)
      for (int a = mu + 1; a < 4; a++)
        for (int nu = 0; nu < 4; nu++) {
          if (nu == mu || nu == a) continue;
          const int b = 6 - mu - nu - a;
          CHK(staple_deriv_pair(c, g, S.fvw(st->fl1[a][b]), S.fvw(st->fl1[mu][b]), S.fv(st->K.l1[a][b]), S.fv(st->K.l1[mu][b]),
                                S.fv(st->fl2[mu][nu]), S.fv(st->fl2[a][nu]), mu, a, !w[a][b], !w[mu][b], tsel));
          w[a][b] = w[mu][b] = true;
        }
    return 0;
  }));
  HIPCHK(hipGetLastError());
  {
    for (int mu = 0; mu < 4; mu++) {
      int j = 0;
      for (int nu = 0; nu < 4; nu++) {
        if (nu == mu) continue;
        PB.dst[mu][j] = S.fvw(st->fl1[mu][nu]); PB.X[mu][j] = S.fv(st->K.l1x[mu][nu]); PB.C[mu][j] = S.fv(st->fl1[mu][nu]);
        j++;
      }
      PB.f[mu] = S.gvw(st->F, mu);
    }
    PB.nn = 3; PB.accumulate = 1; PB.ma = ma1; PB.alp = alp1;
    QX_PB_LAUNCH;
    HIPCHK(hipGetLastError());
    {
      // ... and here fl1[m][n] is read by the ONE call whose plane is (m, n): only fl1[m][3] and fl1[3][m] cross the t boundary --
      // 6 of 12 fields, 2.1 instead of 4.25 ms at 45 GB/s on a 48^3 face, beside 1.9 ms of interior kernels (profiles/r06_notes.md section 6)
      const double2 *fs[12]; int nf = 0;
      for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++)
          if (nu != mu && (mu == 3 || nu == 3)) fs[nf++] = st->fl1[mu][nu];
      CHK(refresh(fs, nf, 576));
    }
  }
  HIPCHK(hipGetLastError());
  CHK(passes([&](int tsel) -> int {
    for (int mu = 0; mu < 4; mu++)
      for (int nu = mu + 1; nu < 4; nu++)
        CHK(staple_deriv_pair(c, g, S.gvw(st->F, nu), S.gvw(st->F, mu), S.gv(st->G, nu), S.gv(st->G, mu),
                              S.fv(st->fl1[mu][nu]), S.fv(st->fl1[nu][mu]), mu, nu, 0, 0, tsel));
    return 0;
  }));
  HIPCHK(hipGetLastError());
  return 0;
}
int nhyp_force_host(qexhip_ctx *c, double *f_host, const double *chain_host) {
  NhypState *st = (NhypState *)c->nhyp;
  if (!st) { qexhip_set_error("nhyp_force: call qexhip_nhyp_prepare first (smearGetForce)"); return -1; }
  CHK(st->S.upload(st->F, chain_host));
  CHK(nhyp_backward_dev(c, st));
  return st->S.download(f_host, st->F);
}

__global__ void k_force_projtah(size_t nlinks_tiles, double2 *F, const double2 *__restrict__ G, int adj);

// HisqCoefs.smearGetForce's closure (gauge/hisqsmear.nim:55-90): u, the intermediate v, w and the smeared su, sul stay on
// the device; smearedForce(dsdu, dsdsu, dsdsul) only runs the reverse pass
struct HisqState {
  Smear S;
  double2 *G = nullptr, *V = nullptr, *W = nullptr, *FL = nullptr, *LL = nullptr, *CF = nullptr, *CL = nullptr, *F = nullptr;
  explicit HisqState(qexhip_ctx *c) : S(c) {}
};
void hisq_state_free(qexhip_ctx *c) {
  if (c->hisq) { delete (HisqState *)c->hisq; c->hisq = nullptr; }
}
int hisq_prepare(qexhip_ctx *c, const double *g_host, double *fl_host, double *ll_host) {
  for (int i = 0; i < 4; i++) if (c->g.X[i] < 4) { qexhip_set_error("HISQ smearing needs lattice extents >= 4"); return -1; }
  HisqState *st = (HisqState *)c->hisq;
  if (!st) {
    st = new HisqState(c);
    c->hisq = st;
    Smear &S = st->S;
    for (double2 **p : {&st->G, &st->V, &st->W, &st->FL, &st->LL, &st->CF, &st->CL, &st->F}) CHK(S.alloc(p, S.gsz));
  }
  Smear &S = st->S;
  S.scratch_begin();                   // the scratch of the two fat7 passes is kept from call to call (qexhip_hisq_release / finalize free it)
  const double naik = 1.0, f2 = 2.0;
  const double c_second[5] = {(1.0 + 3.0 * f2 + naik) / 8.0, -1.0 / 16.0, 1.0 / 64.0, -1.0 / 384.0, -f2 / 16.0};
  CHK(S.upload(st->G, g_host));
  CHK(S.hisq_first(st->G, st->V, st->W, !(c->g.halo && c->opt_smear_ca)));
  CHK(S.fat7(st->FL, st->W, c_second, st->LL, st->W, -naik / 24.0));
  if (fl_host) CHK(S.download(fl_host, st->FL));
  if (ll_host) CHK(S.download(ll_host, st->LL));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}
int hisq_closure_force(qexhip_ctx *c, const double *dfl_host, const double *dll_host, double *f_host) {
  HisqState *st = (HisqState *)c->hisq;
  if (!st) { qexhip_set_error("hisq force: call qexhip_hisq_prepare first (smearGetForce)"); return -1; }
  Smear &S = st->S;
  S.scratch_begin();                   // the reverse pass's scratch: the buffers of the previous call again
  CHK(S.upload(st->CF, dfl_host)); CHK(S.upload(st->CL, dll_host));
  CHK(S.hisq_reverse(st->G, st->V, st->W, st->CF, st->CL, st->F));
  CHK(S.download(f_host, st->F));
  return 0;
}
// fermionForce of the HISQ HMC (src/examples/hisqhmc.nim:496-541) on the closure's fields: f1 = sum_k s_k p_k(x) (x) p_k(x+mu)^+,
// f3 the same with x + 3 mu, odd sites *= -1, smearedForce, TAH(ff u^+) with the (phased) u of the closure
int hisq_fermion_force(qexhip_ctx *c, double *f_host, const double *const *psi, const double *scale, int n) {
  HisqState *st = (HisqState *)c->hisq;
  if (!st) { qexhip_set_error("hisq fermion force: call qexhip_hisq_prepare first (smearGetForce)"); return -1; }
  if (n < 1) { qexhip_set_error("hisq fermion force: n < 1"); return -1; }
  Smear &S = st->S;
  S.scratch_begin();
  DevField *fx;
  CHK(get_work(c, WK_IN, &fx));
  for (int k = 0; k < n; k++) {
    CHK(field_upload(c, *fx, psi[k]));
    CHK(stag_outer_dev(c, *fx, st->CF, scale[k], -scale[k], k > 0, 1));
    CHK(stag_outer_dev(c, *fx, st->CL, scale[k], -scale[k], k > 0, 3));
  }
  CHK(S.hisq_reverse(st->G, st->V, st->W, st->CF, st->CL, st->F));
  const size_t ltiles = (size_t)2 * c->g.etile * 4;
  k_force_projtah<<<(unsigned)((ltiles * 64 + 255) / 256), 256, 0, c->stream>>>(ltiles, st->F, st->G, 0);
  HIPCHK(hipGetLastError());
  CHK(S.download(f_host, st->F));
  return 0;
}
int hisq_set_links_from_closure(qexhip_ctx *c) {
  HisqState *st = (HisqState *)c->hisq;
  if (!st) { qexhip_set_error("set_links_hisq(g = NULL) needs qexhip_hisq_prepare first"); return -1; }
  CHK(st->S.ghosts_tlinks_lo(st->FL, 1)); CHK(st->S.ghosts_tlinks_lo(st->LL, 3));
  return links_from_natural(c, st->FL, st->LL);
}

// setBC_cust + stagPhase on a device gauge field (stagg_pv_hmc/staghmc_spv.nim:367-401,
// gauge/gaugeUtils.nim:124-131, physics/stagD.nim:509-520): sign flips only
__global__ void __launch_bounds__(256) k_rephase(Geom g, double2 *G, int bcmask, int ph0, int ph1, int ph2, int ph3, int tlast) {
  // tlast: this rank holds the last global t slice (the t boundary condition lives there; local t parity = global)
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.V) return;
  const int p = i >= g.Vh, c = i - p * g.Vh;
  int x[4];
  coords_sm(g, c, p, x);
  const int ph[4] = {ph0, ph1, ph2, ph3};
  for (int mu = 0; mu < 4; mu++) {
    int s = 0;
    for (int k = 0; k < 4; k++) s += (ph[mu] >> k) & x[k];
    if (((bcmask >> mu) & 1) && x[mu] == g.X[mu] - 1 && (mu != 3 || tlast)) s += 1;
    if (s & 1) {
      double2 *w = G + (((size_t)p * g.etile + (c >> 6)) * 4 + mu) * 576 + (c & 63);
      for (int k = 0; k < 9; k++) { double2 v = w[k * 64]; w[k * 64] = make_double2(-v.x, -v.y); }
    }
  }
}


static int smear_check(qexhip_ctx *c, int min_extent) {
  if (c->g.halo && c->g.X[3] < 4) { qexhip_set_error("t-sharded smearing needs a local t extent >= 4"); return -1; }
  for (int d = 0; d < 4; d++)
    if (c->g.X[d] < min_extent) { qexhip_set_error("smearing needs lattice extents >= %d", min_extent); return -1; }
  return 0;
}

int smear_fat7_host(qexhip_ctx *c, const double *g_host, const double coef[5], double *fl_host, double *ll_host, double naik) {
  CHK(smear_check(c, 4));
  Smear S(c);
  double2 *G, *FL, *LL = nullptr;
  CHK(S.alloc(&G, S.gsz));
  CHK(S.alloc(&FL, S.gsz));
  if (ll_host && naik != 0.0) CHK(S.alloc(&LL, S.gsz));
  CHK(S.upload(G, g_host));
  CHK(S.ghosts_g(G, 2));
  CHK(S.fat7(FL, G, coef, LL, G, naik));
  CHK(S.download(fl_host, FL));
  if (LL) CHK(S.download(ll_host, LL));
  return 0;
}

int smear_hisq_host(qexhip_ctx *c, const double *g_host, double *fl_host, double *ll_host) {
  CHK(smear_check(c, 4));
  Smear S(c);
  double2 *G, *FL, *LL;
  CHK(S.alloc(&G, S.gsz)); CHK(S.alloc(&FL, S.gsz)); CHK(S.alloc(&LL, S.gsz));
  CHK(S.upload(G, g_host));
  CHK(S.hisq(G, FL, LL));
  CHK(S.download(fl_host, FL));
  return S.download(ll_host, LL);
}

int smear_hisq_force_host(qexhip_ctx *c, const double *g_host, const double *dfl_host, const double *dll_host, double *f_host) {
  CHK(smear_check(c, 4));
  Smear S(c);
  double2 *G, *CF, *CL, *F;
  CHK(S.alloc(&G, S.gsz)); CHK(S.alloc(&CF, S.gsz)); CHK(S.alloc(&CL, S.gsz)); CHK(S.alloc(&F, S.gsz));
  CHK(S.upload(G, g_host)); CHK(S.upload(CF, dfl_host)); CHK(S.upload(CL, dll_host));
  CHK(S.hisq_force(G, CF, CL, F));
  return S.download(f_host, F);
}
int smear_fat7_deriv_host(qexhip_ctx *c, const double *g_host, const double *dfl_host, const double coef[5], const double *dll_host,
                          double naik, double *d_host) {
  CHK(smear_check(c, dll_host && naik != 0.0 ? 4 : 2));
  Smear S(c);
  double2 *G, *CF, *CL = nullptr, *D;
  CHK(S.alloc(&G, S.gsz)); CHK(S.alloc(&CF, S.gsz)); CHK(S.alloc(&D, S.gsz));
  CHK(S.upload(G, g_host)); CHK(S.upload(CF, dfl_host));
  if (dll_host) { CHK(S.alloc(&CL, S.gsz)); CHK(S.upload(CL, dll_host)); CHK(S.ghosts_g(CL, 2)); }
  CHK(S.ghosts_g(G, 2)); CHK(S.ghosts_g(CF));
  CHK(S.fat7_deriv(D, G, CF, coef, CL, naik));
  return S.download(d_host, D);
}

int smear_nhyp_host(qexhip_ctx *c, const double *g_host, double *fl_host, double a1, double a2, double a3) {
  CHK(smear_check(c, 2));
  Smear S(c);
  double2 *G, *FL;
  CHK(S.alloc(&G, S.gsz)); CHK(S.alloc(&FL, S.gsz));
  CHK(S.upload(G, g_host));
  CHK(S.nhyp(G, FL, a1, a2, a3));
  return S.download(fl_host, FL);
}

// smear on the device and hand the result straight to the Dslash (no PCIe round trip of the
// smeared links): Staggered.g <- HISQ(g) / rephase(nHYP(g))
int smear_set_links_hisq(qexhip_ctx *c, const double *g_host) {
  CHK(smear_check(c, 4));
  if (!g_host) return hisq_set_links_from_closure(c);
  Smear S(c);
  double2 *G, *FL, *LL;
  CHK(S.alloc(&G, S.gsz)); CHK(S.alloc(&FL, S.gsz)); CHK(S.alloc(&LL, S.gsz));
  CHK(S.upload(G, g_host));
  CHK(S.hisq(G, FL, LL));
  CHK(S.ghosts_tlinks_lo(FL, 1)); CHK(S.ghosts_tlinks_lo(LL, 3));      // backward t-links x - t, x - 3 t below the slab
  return links_from_natural(c, FL, LL);
}
int smear_set_links_nhyp(qexhip_ctx *c, const double *g_host, double a1, double a2, double a3, int bcmask, const int ph[4]) {
  CHK(smear_check(c, 2));
  if (!g_host) {
    // reuse the links the closure already smeared (qexhip_nhyp_prepare): copy, rephase, hand over
    NhypState *st = (NhypState *)c->nhyp;
    if (!st) { qexhip_set_error("set_links_nhyp(g = NULL) needs qexhip_nhyp_prepare first"); return -1; }
    HIPCHK(hipMemcpyAsync(st->F, st->FL, st->S.gsz * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
    k_rephase<<<st->S.nb(), 256, 0, c->stream>>>(st->S.g, st->F, bcmask, ph[0], ph[1], ph[2], ph[3], c->rankCoord[3] == c->rankGeom[3] - 1);
    HIPCHK(hipGetLastError());
    CHK(st->S.ghosts_tlinks_lo(st->F, 1));
    return links_from_natural(c, st->F, nullptr);
  }
  Smear S(c);
  double2 *G, *FL;
  CHK(S.alloc(&G, S.gsz)); CHK(S.alloc(&FL, S.gsz));
  CHK(S.upload(G, g_host));
  CHK(S.nhyp(G, FL, a1, a2, a3));
  k_rephase<<<S.nb(), 256, 0, c->stream>>>(S.g, FL, bcmask, ph[0], ph[1], ph[2], ph[3], c->rankCoord[3] == c->rankGeom[3] - 1);
  HIPCHK(hipGetLastError());
  CHK(S.ghosts_tlinks_lo(FL, 1));
  return links_from_natural(c, FL, nullptr);
}

// projTAH(f, g) of the fork (stagg_pv_hmc/staghmc_spv_gforce.nim:256-291) on device gauge-layout fields:
// adj = 0 (matter):  f <- TAH(f g^+);  adj = 1 (gauge):  f <- TAH(g f^+)
__global__ void __launch_bounds__(256) k_force_projtah(size_t nlinks_tiles, double2 *F, const double2 *__restrict__ G, int adj) {
  size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t tile = j >> 6;
  if (tile >= nlinks_tiles) return;
  size_t o = tile * 576 + (j & 63);
  const M3 f = m3_load(F + o, 64), u = m3_load(G + o, 64);
  m3_store(F + o, 64, m3_tah(adj ? m3_mul_na(u, f) : m3_mul_na(f, u)));
}
static int nhyp_finish(qexhip_ctx *c, NhypState *st, int adj, double *f_host) {
  CHK(nhyp_backward_dev(c, st));
  const size_t ltiles = (size_t)2 * c->g.etile * 4;       // ghost tiles included: harmless, keeps the index linear
  k_force_projtah<<<(unsigned)((ltiles * 64 + 255) / 256), 256, 0, c->stream>>>(ltiles, st->F, st->G, adj);
  HIPCHK(hipGetLastError());
  return f_host ? st->S.download(f_host, st->F) : 0;      // f = NULL: left on the device for qexhip_md_kick / md_shift_links
}
double2 *nhyp_force_buffer(qexhip_ctx *c) { return c->nhyp ? ((NhypState *)c->nhyp)->F : nullptr; }
// gforce(act, g, sg, f, smear_force) (stagg_pv_hmc/staghmc_spv.nim:217-228): derivative of the gauge
// action on the SMEARED links -> smearedForce -> TAH(g f^+) with the thin links
int nhyp_gauge_force(qexhip_ctx *c, double *f_host, double cplaq, double c2, int kind) {
  NhypState *st = (NhypState *)c->nhyp;
  if (!st) { qexhip_set_error("nhyp_gauge_force: call qexhip_nhyp_prepare first (smearGetForce)"); return -1; }
  CHK(st->S.ghosts_g(st->FL, (kind == 0 && c2 != 0.0) ? 2 : 1));
  CHK(gauge_deriv_dev(c, st->FL, st->F, cplaq, c2, kind));
  return nhyp_finish(c, st, 1, f_host);
}
// fforce + smeared_one_link_force (stagg_pv_hmc/staghmc_spv.nim:716-865): sum_k scale_k psi_k (x) psi_k(+mu)^+,
// rephase (BC + staggered phases) and odd-site sign, smearedForce, TAH(f g^+)
int nhyp_fermion_force(qexhip_ctx *c, double *f_host, const double *const *psi, const double *scale, int n, int bcmask, const int ph[4]) {
  NhypState *st = (NhypState *)c->nhyp;
  if (!st) { qexhip_set_error("nhyp_fermion_force: call qexhip_nhyp_prepare first (smearGetForce)"); return -1; }
  if (n < 1) { qexhip_set_error("nhyp_fermion_force: n < 1"); return -1; }
  DevField *fx;
  CHK(get_work(c, WK_IN, &fx));
  for (int k = 0; k < n; k++) {
    CHK(field_upload(c, *fx, psi[k]));
    CHK(stag_outer_dev(c, *fx, st->F, scale[k], -scale[k], k > 0));
  }
  k_rephase<<<st->S.nb(), 256, 0, c->stream>>>(st->S.g, st->F, bcmask, ph[0], ph[1], ph[2], ph[3], c->rankCoord[3] == c->rankGeom[3] - 1);
  HIPCHK(hipGetLastError());
  return nhyp_finish(c, st, 0, f_host);
}

// the whole fforce (src/examples/staghmc_sh.nim:387-427, src/stagg_pv_hmc/staghmc_spv.nim:758-865) on the device:
// solve D(m_k) psi_k = phi_k for the n fields (lock-step batches of four on the operator's current links, which
// the caller has set from this closure: qexhip_stag_set_links_nhyp(g = NULL)), outer products straight from the
// device solutions, rephase, chain, TAH.  Only phi goes in and f comes out over PCIe.
// phi: host sources, or nullptr with phi_dev: sources already resident (pseudofermions born on the device)
int nhyp_fforce(qexhip_ctx *c, double *f_host, int n, const double *const *phi, const double *mass, const double *scale,
                const double *r2req, int maxits, int bcmask, const int ph[4], int *iters, DevField *const *phi_dev) {
  NhypState *st = (NhypState *)c->nhyp;
  if (!st) { qexhip_set_error("nhyp_fforce: call qexhip_nhyp_prepare first (smearGetForce)"); return -1; }
  if (n < 1) { qexhip_set_error("nhyp_fforce: n < 1"); return -1; }
  for (int k0 = 0; k0 < n; k0 += 4) {
    const int k = std::min(4, n - k0);
    DevField *xs[4], *bs[4];
    CHK(batch_io_fields(c, k, xs, bs));
    for (int j = 0; j < k; j++) {
      if (phi_dev) bs[j] = phi_dev[k0 + j];                 // read only (the solver copies its source first)
      else CHK(field_upload(c, *bs[j], phi[k0 + j]));
    }
    CHK(solve_full_batch_dev(c, k, xs, bs, mass + k0, r2req + k0, maxits, iters ? iters + k0 : nullptr, nullptr));
    for (int j = 0; j < k; j++) CHK(stag_outer_dev(c, *xs[j], st->F, scale[k0 + j], -scale[k0 + j], (k0 + j) > 0));
  }
  k_rephase<<<st->S.nb(), 256, 0, c->stream>>>(st->S.g, st->F, bcmask, ph[0], ph[1], ph[2], ph[3], c->rankCoord[3] == c->rankGeom[3] - 1);
  HIPCHK(hipGetLastError());
  return nhyp_finish(c, st, 0, f_host);
}
