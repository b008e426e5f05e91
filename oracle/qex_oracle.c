/* qex_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see qex_oracle.h).
 *
 * Plain-C restatement of the reference algorithms for the staggered Dslash / CG /
 * Wilson-flow path of ctpeterson/qex.  Every function cites the reference
 * file:line (relative to /root/reference) it follows.  Nothing here is used by
 * the product path.
 *
 * Build: see oracle/Makefile  (gcc -O2 -fopenmp, NO fast-math so results are
 * reproducible; the reference itself is -Ofast, hence 1e-12-level gates).
 */
#include "qex_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* layout: V=1 MILC even-odd order  (src/layout/qlayout.nim:110-131)   */
/* ------------------------------------------------------------------ */
struct qo_layout {
  int L[4];
  int vol, volh;
  int *coords;      /* [vol][4] */
  int *nb[4][4];    /* nb[mu][k]: k=0:+1, 1:-1, 2:+3, 3:-3 */
};

int qo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

int qo_index(const qo_layout *lo, const int x[4]) {
  /* lex_i with x[0] fastest (qlayout.nim:85-97), parity from coordinate sum
   * (:125-129): oi2 = oi/2 (even) or (oi+nSites)/2 (odd). */
  int lex = 0, p = 0;
  for (int i = 3; i >= 0; i--) lex = lex * lo->L[i] + x[i];
  for (int i = 0; i < 4; i++) p += x[i];
  if (p & 1) return (lex + lo->vol) / 2;
  return lex / 2;
}

void qo_coord(const qo_layout *lo, int idx, int x[4]) {
  for (int i = 0; i < 4; i++) x[i] = lo->coords[4 * idx + i];
}

qo_layout *qo_layout_new(const int L[4]) {
  qo_layout *lo = (qo_layout *)calloc(1, sizeof(qo_layout));
  lo->vol = 1;
  for (int i = 0; i < 4; i++) { lo->L[i] = L[i]; lo->vol *= L[i]; }
  lo->volh = lo->vol / 2;
  lo->coords = (int *)malloc(sizeof(int) * 4 * (size_t)lo->vol);
  int x[4];
  for (x[3] = 0; x[3] < L[3]; x[3]++)
    for (x[2] = 0; x[2] < L[2]; x[2]++)
      for (x[1] = 0; x[1] < L[1]; x[1]++)
        for (x[0] = 0; x[0] < L[0]; x[0]++) {
          int idx = qo_index(lo, x);
          for (int i = 0; i < 4; i++) lo->coords[4 * idx + i] = x[i];
        }
  static const int lens[4] = {1, -1, 3, -3};
  for (int mu = 0; mu < 4; mu++)
    for (int k = 0; k < 4; k++) {
      lo->nb[mu][k] = (int *)malloc(sizeof(int) * (size_t)lo->vol);
      for (int idx = 0; idx < lo->vol; idx++) {
        int y[4];
        for (int i = 0; i < 4; i++) y[i] = lo->coords[4 * idx + i];
        /* dest s receives source s + len*mu  (shiftX.nim:76-81, qshifts.nim:204) */
        y[mu] = ((y[mu] + lens[k]) % L[mu] + L[mu]) % L[mu];
        lo->nb[mu][k][idx] = qo_index(lo, y);
      }
    }
  return lo;
}

void qo_layout_free(qo_layout *lo) {
  if (!lo) return;
  for (int mu = 0; mu < 4; mu++)
    for (int k = 0; k < 4; k++) free(lo->nb[mu][k]);
  free(lo->coords);
  free(lo);
}

int qo_vol(const qo_layout *lo) { return lo->vol; }

int qo_neighbor(const qo_layout *lo, int idx, int mu, int len) {
  switch (len) {
    case 1: return lo->nb[mu][0][idx];
    case -1: return lo->nb[mu][1][idx];
    case 3: return lo->nb[mu][2][idx];
    case -3: return lo->nb[mu][3][idx];
    default: {
      int y[4];
      for (int i = 0; i < 4; i++) y[i] = lo->coords[4 * idx + i];
      y[mu] = ((y[mu] + len) % lo->L[mu] + lo->L[mu]) % lo->L[mu];
      return qo_index(lo, y);
    }
  }
}

static inline void subset_range(const qo_layout *lo, int parity, int *s0, int *s1) {
  /* layoutSubset (layoutX.nim:285-295): even = [0,nEven), odd = [nEven,nSites) */
  if (parity == 0) { *s0 = 0; *s1 = lo->volh; }
  else if (parity == 1) { *s0 = lo->volh; *s1 = lo->vol; }
  else { *s0 = 0; *s1 = lo->vol; }
}

/* ------------------------------------------------------------------ */
/* 3x3 complex matrix helpers: double m[18], row-major, (re,im)        */
/* ------------------------------------------------------------------ */
#define RE(m, i, j) ((m)[2 * (3 * (i) + (j))])
#define IM(m, i, j) ((m)[2 * (3 * (i) + (j)) + 1])

static inline void m_zero(double *r) { for (int i = 0; i < 18; i++) r[i] = 0.0; }
static inline void m_copy(double *r, const double *a) { for (int i = 0; i < 18; i++) r[i] = a[i]; }
static inline void m_unit(double *r) { m_zero(r); RE(r,0,0) = RE(r,1,1) = RE(r,2,2) = 1.0; }
static inline void m_adj(double *r, const double *a) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { RE(r,i,j) = RE(a,j,i); IM(r,i,j) = -IM(a,j,i); }
}
/* r = a*b */
static inline void m_mul(double *r, const double *a, const double *b) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double sr = 0, si = 0;
    for (int k = 0; k < 3; k++) {
      sr += RE(a,i,k) * RE(b,k,j) - IM(a,i,k) * IM(b,k,j);
      si += RE(a,i,k) * IM(b,k,j) + IM(a,i,k) * RE(b,k,j);
    }
    RE(r,i,j) = sr; IM(r,i,j) = si;
  }
}
/* r = a * b^dagger */
static inline void m_mul_na(double *r, const double *a, const double *b) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double sr = 0, si = 0;
    for (int k = 0; k < 3; k++) {
      sr += RE(a,i,k) * RE(b,j,k) + IM(a,i,k) * IM(b,j,k);
      si += IM(a,i,k) * RE(b,j,k) - RE(a,i,k) * IM(b,j,k);
    }
    RE(r,i,j) = sr; IM(r,i,j) = si;
  }
}
/* r = a^dagger * b */
static inline void m_mul_an(double *r, const double *a, const double *b) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double sr = 0, si = 0;
    for (int k = 0; k < 3; k++) {
      sr += RE(a,k,i) * RE(b,k,j) + IM(a,k,i) * IM(b,k,j);
      si += RE(a,k,i) * IM(b,k,j) - IM(a,k,i) * RE(b,k,j);
    }
    RE(r,i,j) = sr; IM(r,i,j) = si;
  }
}
static inline void m_axpy(double *r, double a, const double *x) { for (int i = 0; i < 18; i++) r[i] += a * x[i]; }
static inline void m_scale(double *r, double a, const double *x) { for (int i = 0; i < 18; i++) r[i] = a * x[i]; }
static inline void m_add_diag(double *r, double s) { RE(r,0,0) += s; RE(r,1,1) += s; RE(r,2,2) += s; }
static inline double m_retr_adj_mul(const double *a, const double *b) {
  /* redot(a,b) = Re tr(a^dagger b) = sum re*re + im*im */
  double s = 0; for (int i = 0; i < 18; i++) s += a[i] * b[i]; return s;
}
/* determinant (matrixFunctions.nim:72-75) */
static inline void m_det(double *dr, double *di, const double *x) {
#define CMUL(ar, ai, br, bi, cr, ci) do { cr = (ar) * (br) - (ai) * (bi); ci = (ar) * (bi) + (ai) * (br); } while (0)
  double t1r, t1i, t2r, t2i, pr, pi, sr = 0, si = 0, qr, qi;
  /* (x00*x11 - x01*x10)*x22 */
  CMUL(RE(x,0,0), IM(x,0,0), RE(x,1,1), IM(x,1,1), t1r, t1i);
  CMUL(RE(x,0,1), IM(x,0,1), RE(x,1,0), IM(x,1,0), t2r, t2i);
  pr = t1r - t2r; pi = t1i - t2i;
  CMUL(pr, pi, RE(x,2,2), IM(x,2,2), qr, qi); sr += qr; si += qi;
  /* (x02*x10 - x00*x12)*x21 */
  CMUL(RE(x,0,2), IM(x,0,2), RE(x,1,0), IM(x,1,0), t1r, t1i);
  CMUL(RE(x,0,0), IM(x,0,0), RE(x,1,2), IM(x,1,2), t2r, t2i);
  pr = t1r - t2r; pi = t1i - t2i;
  CMUL(pr, pi, RE(x,2,1), IM(x,2,1), qr, qi); sr += qr; si += qi;
  /* (x01*x12 - x02*x11)*x20 */
  CMUL(RE(x,0,1), IM(x,0,1), RE(x,1,2), IM(x,1,2), t1r, t1i);
  CMUL(RE(x,0,2), IM(x,0,2), RE(x,1,1), IM(x,1,1), t2r, t2i);
  pr = t1r - t2r; pi = t1i - t2i;
  CMUL(pr, pi, RE(x,2,0), IM(x,2,0), qr, qi); sr += qr; si += qi;
  *dr = sr; *di = si;
}

/* eigs3 (matrixFunctions.nim:79-111) */
static void eigs3(double *e0, double *e1, double *e2, double tr, double p2, double det) {
  double tr3 = (1.0 / 3.0) * tr;
  double p23 = (1.0 / 3.0) * p2;
  double tr32 = tr3 * tr3;
  double q = fabs(0.5 * (p23 - tr32));
  double r = 0.25 * tr3 * (5 * tr32 - p2) - 0.5 * det;
  double sq = sqrt(q);
  double sq3 = q * sq;
  double isq3 = 1.0 / sq3;
  double isq3c = fmin(3e38, fmax(-3e38, isq3));
  double rsq3c = r * isq3c;
  double rsq3 = fmin(1.0, fmax(-1.0, rsq3c));
  double t = (1.0 / 3.0) * acos(rsq3);
  double st = sin(t), ct = cos(t);
  double sqc = sq * ct;
  double sqs = 1.73205080756887729352 * sq * st;
  double ll = tr3 + sqc;
  *e0 = tr3 - 2 * sqc;
  *e1 = ll + sqs;
  *e2 = ll - sqs;
}

/* rsqrtPHM3f + rsqrtPHM3 (matrixFunctions.nim:126-182): r = x^{-1/2} for pos. Hermitian x */
static void rsqrtPHM3(double *r, const double *x) {
  double tr = RE(x,0,0) + RE(x,1,1) + RE(x,2,2);
  double x2[18];
  m_mul(x2, x, x);
  double p2 = RE(x2,0,0) + RE(x2,1,1) + RE(x2,2,2);
  double det, deti;
  m_det(&det, &deti, x);
  double l0, l1, l2;
  eigs3(&l0, &l1, &l2, tr, p2, det);
  double sl0 = sqrt(fabs(l0)), sl1 = sqrt(fabs(l1)), sl2 = sqrt(fabs(l2));
  double u = sl0 + sl1 + sl2;
  double w = sl0 * sl1 * sl2;
  double d = w * (sl0 + sl1) * (sl0 + sl2) * (sl1 + sl2);
  double di = 1 / d;
  double c0 = (w * u * u + l0 * sl0 * (l1 + l2) + l1 * sl1 * (l0 + l2) + l2 * sl2 * (l0 + l1)) * di;
  double c1 = -(tr * u + w) * di;
  double c2 = u * di;
  /* r := c0 + c1*x + c2*x2 */
  for (int i = 0; i < 18; i++) r[i] = c1 * x[i] + c2 * x2[i];
  m_add_diag(r, c0);
}

/* projectU: x (x'x + eps)^{-1/2}  (matrixFunctions.nim:293-313) */
void qo_projectU(double *r, const double *x) {
  double t[18], t2[18];
  m_mul_an(t, x, x);
  m_add_diag(t, 1e-20);
  rsqrtPHM3(t2, t);
  double xx[18];
  m_copy(xx, x); /* allow r==x */
  m_mul(r, xx, t2);
}

/* projectSU (matrixFunctions.nim:359-370) */
void qo_projectSU(double *r, const double *x) {
  double m[18];
  qo_projectU(m, x);
  double dr, di;
  m_det(&dr, &di, m);
  double p = (1.0 / (double)(-3)) * atan2(di, dr);
  double cr = cos(p), ci = sin(p);
  for (int i = 0; i < 9; i++) {
    double a = m[2 * i], b = m[2 * i + 1];
    r[2 * i] = cr * a - ci * b;
    r[2 * i + 1] = cr * b + ci * a;
  }
}

/* projectTAH (matrixFunctions.nim:375-380): r = 0.5(x - x^dag) - trace/nc */
void qo_projectTAH(double *r, const double *x) {
  double t[18];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    RE(t,i,j) = 0.5 * (RE(x,i,j) - RE(x,j,i));
    IM(t,i,j) = 0.5 * (IM(x,i,j) + IM(x,j,i));
  }
  double dr = (RE(t,0,0) + RE(t,1,1) + RE(t,2,2)) / 3.0;
  double di = (IM(t,0,0) + IM(t,1,1) + IM(t,2,2)) / 3.0;
  for (int i = 0; i < 3; i++) { RE(t,i,i) -= dr; IM(t,i,i) -= di; }
  m_copy(r, t);
}

/* exp: ExpParam{scale 20, ekPoly, order 4} (matrixFunctions.nim:436-445);
 * expm1 = expm1Poly4(m/2^20) then 20x r <- r(r+2) (matexp.nim:80-85,634-649); +1 (:707-710) */
void qo_exp(double *r, const double *m) {
  const double s = 1.0 / (double)(1 << 20);
  double ms[18], m2[18], a[18], e[18], t[18];
  m_scale(ms, s, m);
  m_mul(m2, ms, ms);
  /* a = C4*m2 + C3*m + C2 ; r = a*m2 + m   (splitVars order: a=C4*m2; a+=C3*m; a+=C2) */
  for (int i = 0; i < 18; i++) a[i] = (1.0 / 24.0) * m2[i];
  for (int i = 0; i < 18; i++) a[i] += (1.0 / 6.0) * ms[i];
  m_add_diag(a, 0.5);
  m_mul(e, a, m2);
  for (int i = 0; i < 18; i++) e[i] += ms[i];
  for (int k = 0; k < 20; k++) {
    m_copy(t, e);
    m_add_diag(t, 2.0);
    m_mul(a, e, t);
    m_copy(e, a);
  }
  m_add_diag(e, 1.0);
  m_copy(r, e);
}

/* ------------------------------------------------------------------ */
/* RNGs                                                                */
/* ------------------------------------------------------------------ */
/* RngMilc6 (rng/milcrng.nim:12-14,48-55,92-110,120-133,150-154,158-193) */
typedef struct { uint32_t r0, r1, r2, r3, r4, r5, r6, ic, mult; } milc6_t;
#define M6_INDX1 69607u
#define M6_INDX2 8u
#define M6_ADDEND 12345u
#define M6_MASK 0x00FFFFFFu

static void milc6_seed(milc6_t *p, uint32_t seed0, uint32_t index) {
  uint32_t seed = seed0;
#define M6SET(x) do { seed = (M6_INDX1 + M6_INDX2 * index) * seed + M6_ADDEND; x = (seed >> 8) & M6_MASK; } while (0)
  M6SET(p->r0); M6SET(p->r1); M6SET(p->r2); M6SET(p->r3); M6SET(p->r4); M6SET(p->r5); M6SET(p->r6);
  seed = (M6_INDX1 + M6_INDX2 * index) * seed + M6_ADDEND;
  p->ic = seed;
  p->mult = 100005u + 8u * index;
}
static inline uint32_t milc6_next(milc6_t *p) {
  uint32_t t = (((p->r5 >> 7) | (p->r6 << 17)) ^ ((p->r4 >> 1) | (p->r5 << 23))) & M6_MASK;
  p->r6 = p->r5; p->r5 = p->r4; p->r4 = p->r3; p->r3 = p->r2; p->r2 = p->r1; p->r1 = p->r0; p->r0 = t;
  uint32_t s = p->ic * p->mult + M6_ADDEND;
  p->ic = s;
  return t ^ ((s >> 8) & M6_MASK);
}
static inline float milc6_uniform(milc6_t *p) {
  const float SCALE = 1.0f / (float)0x01000000;
  return SCALE * (float)milc6_next(p);
}
static inline double milc6_gaussian(milc6_t *p) {
  /* milcrng.nim:183-193 (non-FUELCompat branch).
   * PINNED BY G1 + G5: the arithmetic is double, but the deviate every caller receives is
   * rounded to float32.  48 precision variants were tried against the six golden plaquettes of
   * tests/reprod/trandgauge.nim:17; only "all-double, result rounded to float32" reproduces them
   * (sum diff^2 = 2.6e-32 vs the test's 1e-30 bound; every other variant >= 1e-20).  The same
   * rounding is required for randTah3 (gaugeUtils.nim:1356-1375) to reproduce
   * tests/base/trngseed.nim:56 (131563.7475902051: 1e-15 with rounding, 6e-11 without). */
  const double TINY = 9.999999999999999e-308;
  double v = (double)milc6_uniform(p);
  double pp = (double)milc6_uniform(p) * 2.0 * 3.14159265358979323846;
  double r = sqrt(-2.0 * log(v + TINY));
  return (double)(float)(r * cos(pp));
}

/* MRG32k3a (rng/mrg32k3a.nim) */
typedef struct { uint32_t s1[3], s2[3]; } mrg_t;
#define MRG_M1 4294967087ull
#define MRG_M2 4294944443ull
static uint32_t mrg_a1sq[190][3][3], mrg_a2sq[190][3][3];
static int mrg_init_done = 0;
static void mrg_squaremod(uint32_t x[3][3], uint32_t a[3][3], uint64_t m) {
  /* mrg32k3a.nim:18-30 */
  for (int i = 0; i < 3; i++) {
    uint64_t t[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++) {
      uint64_t aik = a[i][k];
      t[0] += (aik * a[k][0]) % m; t[1] += (aik * a[k][1]) % m; t[2] += (aik * a[k][2]) % m;
    }
    x[i][0] = (uint32_t)(t[0] % m); x[i][1] = (uint32_t)(t[1] % m); x[i][2] = (uint32_t)(t[2] % m);
  }
}
static void mrg_init(void) {
  if (mrg_init_done) return;
  uint32_t a1[3][3] = {{0, 1, 0}, {0, 0, 1}, {(uint32_t)(MRG_M1 - 810728ull), 1403580u, 0}};
  uint32_t a2[3][3] = {{0, 1, 0}, {0, 0, 1}, {(uint32_t)(MRG_M2 - 1370589ull), 0, 527612u}};
  memcpy(mrg_a1sq[0], a1, sizeof(a1)); memcpy(mrg_a2sq[0], a2, sizeof(a2));
  for (int i = 1; i < 190; i++) { mrg_squaremod(mrg_a1sq[i], mrg_a1sq[i - 1], MRG_M1); mrg_squaremod(mrg_a2sq[i], mrg_a2sq[i - 1], MRG_M2); }
  mrg_init_done = 1;
}
static void mrg_matvecmod(uint32_t a[3][3], uint32_t v[3], uint64_t m) {
  uint64_t v0 = v[0], v1 = v[1], v2 = v[2];
  for (int i = 0; i < 3; i++) v[i] = (uint32_t)((((uint64_t)a[i][0] * v0) % m + ((uint64_t)a[i][1] * v1) % m + ((uint64_t)a[i][2] * v2) % m) % m);
}
static void mrg_skip(mrg_t *p, uint64_t offset, int base) {
  int i = 0; uint64_t s = offset;
  while (s > 0) {
    if (s & 1) { mrg_matvecmod(mrg_a1sq[base + i], p->s1, MRG_M1); mrg_matvecmod(mrg_a2sq[base + i], p->s2, MRG_M2); }
    s >>= 1; i++;
  }
}
static void mrg_seed(mrg_t *p, uint64_t seed, uint64_t subseq) {
  /* seedX, mrg32k3a.nim:103-120 */
  mrg_init();
  if (seed != 0) {
    uint64_t d1 = 12345ull * (uint64_t)((uint32_t)seed ^ 0x55555555u);
    uint64_t d2 = 12345ull * (uint64_t)((uint32_t)(seed >> 32) ^ 0xAAAAAAAAu);
    p->s1[0] = (uint32_t)(d1 % MRG_M1); p->s1[1] = (uint32_t)(d2 % MRG_M1); p->s1[2] = (uint32_t)(d1 % MRG_M1);
    p->s2[0] = (uint32_t)(d2 % MRG_M2); p->s2[1] = (uint32_t)(d1 % MRG_M2); p->s2[2] = (uint32_t)(d2 % MRG_M2);
  } else {
    for (int i = 0; i < 3; i++) { p->s1[i] = 12345u; p->s2[i] = 12345u; }
  }
  mrg_skip(p, subseq, 76);
}
static inline int64_t mrg_next(mrg_t *p) {
  /* nextI, mrg32k3a.nim:158-187 */
  int64_t p1 = 1403580ll * (int64_t)p->s1[1] - 810728ll * (int64_t)p->s1[0];
  p1 = p1 % (int64_t)MRG_M1; if (p1 < 0) p1 += (int64_t)MRG_M1;
  p->s1[0] = p->s1[1]; p->s1[1] = p->s1[2]; p->s1[2] = (uint32_t)p1;
  int64_t p2 = 527612ll * (int64_t)p->s2[2] - 1370589ll * (int64_t)p->s2[0];
  p2 = p2 % (int64_t)MRG_M2; if (p2 < 0) p2 += (int64_t)MRG_M2;
  p->s2[0] = p->s2[1]; p->s2[1] = p->s2[2]; p->s2[2] = (uint32_t)p2;
  return (p1 <= p2) ? p1 - p2 + (int64_t)MRG_M1 : p1 - p2;
}
static inline double mrg_uniform(mrg_t *p) { return 2.328306549295728e-10 * (double)mrg_next(p); }
static inline double mrg_gaussian(mrg_t *p) {
  /* mrg32k3a.nim:225-232: all double, no epsilon */
  double v = mrg_uniform(p);
  double pp = mrg_uniform(p) * 2.0 * 3.14159265358979323846;
  double r = sqrt(-2.0 * log(v));
  return r * cos(pp);
}

struct qo_rngfield {
  int kind, vol;
  milc6_t *m6;
  mrg_t *mrg;
};

qo_rngfield *qo_rngfield_new(const qo_layout *lo, int kind, uint64_t seed) {
  /* newRNGField (distributionUtils.nim:306-331): one generator per site, seeded with
   * (seed, lexicographic index with x fastest). */
  qo_rngfield *rf = (qo_rngfield *)calloc(1, sizeof(qo_rngfield));
  rf->kind = kind; rf->vol = lo->vol;
  if (kind == QO_RNG_MILC6) rf->m6 = (milc6_t *)malloc(sizeof(milc6_t) * (size_t)lo->vol);
  else rf->mrg = (mrg_t *)malloc(sizeof(mrg_t) * (size_t)lo->vol);
  for (int j = 0; j < lo->vol; j++) {
    const int *c = &lo->coords[4 * j];
    long l = c[3];
    for (int i = 2; i >= 0; i--) l = l * lo->L[i] + c[i];
    if (kind == QO_RNG_MILC6) milc6_seed(&rf->m6[j], (uint32_t)seed, (uint32_t)l); /* seedIndep narrows to uint32, milcrng.nim:111-112 */
    else mrg_seed(&rf->mrg[j], seed, (uint64_t)l);
  }
  return rf;
}
void qo_rngfield_free(qo_rngfield *rf) { if (!rf) return; free(rf->m6); free(rf->mrg); free(rf); }

static inline double rf_gaussian(qo_rngfield *rf, int site) {
  return rf->kind == QO_RNG_MILC6 ? milc6_gaussian(&rf->m6[site]) : mrg_gaussian(&rf->mrg[site]);
}

void qo_milc6_test(uint32_t seed, uint32_t index, int n, float *uniforms, double *gaussians) {
  milc6_t p;
  if (uniforms) { milc6_seed(&p, seed, index); for (int i = 0; i < n; i++) uniforms[i] = milc6_uniform(&p); }
  if (gaussians) { milc6_seed(&p, seed, index); for (int i = 0; i < n; i++) gaussians[i] = milc6_gaussian(&p); }
}
void qo_mrg32k3a_test(uint64_t seed, uint64_t index, int n, double *uniforms) {
  mrg_t p; mrg_seed(&p, seed, index);
  for (int i = 0; i < n; i++) uniforms[i] = mrg_uniform(&p);
}

/* gaussian(Field, RNGField): per site, components in storage order, re then im
 * (distributionUtils.nim:64-96) */
#define rf_gaussian_field rf_gaussian
void qo_vector_gaussian(const qo_layout *lo, qo_rngfield *rf, double *v) {
  for (int s = 0; s < lo->vol; s++)
    for (int k = 0; k < 6; k++) v[6 * (size_t)s + k] = rf_gaussian_field(rf, s);
}
/* uniform(Field, RNGField) (distributionUtils.nim:23-44): ncomp reals per site in storage order.
 * Test hook for golden set G4 (tests/base/tmrg32k3a.nim:22-27). */
void qo_field_uniform(const qo_layout *lo, qo_rngfield *rf, int ncomp, double *v, int round_f32) {
  for (int s = 0; s < lo->vol; s++)
    for (int k = 0; k < ncomp; k++) {
      double u = rf->kind == QO_RNG_MILC6 ? (double)milc6_uniform(&rf->m6[s]) : mrg_uniform(&rf->mrg[s]);
      v[(size_t)ncomp * s + k] = round_f32 ? (double)(float)u : u;
    }
}
/* g[mu].gaussian r for mu = 0..3 in turn: field-major, so each site's stream is consumed
 * 18 numbers per direction, direction by direction (gaugeUtils.nim:1424-1429) */
void qo_gauge_gaussian(const qo_layout *lo, qo_rngfield *rf, double *g) {
  for (int mu = 0; mu < 4; mu++)
    for (int s = 0; s < lo->vol; s++) {
      double *m = &g[((size_t)s * 4 + mu) * 18];
      for (int k = 0; k < 18; k++) m[k] = rf_gaussian_field(rf, s);
    }
}
void qo_gauge_random(const qo_layout *lo, qo_rngfield *rf, double *g) {
  /* randomSU = gaussian + projectSU (gaugeUtils.nim:1352-1354) */
  qo_gauge_gaussian(lo, rf, g);
#pragma omp parallel for
  for (int i = 0; i < lo->vol * 4; i++) { double *m = &g[(size_t)i * 18]; qo_projectSU(m, m); }
}
void qo_gauge_unit(const qo_layout *lo, double *g) {
  for (int i = 0; i < lo->vol * 4; i++) m_unit(&g[(size_t)i * 18]);
}
/* randTah3 (gaugeUtils.nim:1356-1375) */
static void rand_tah3(double *m, qo_rngfield *rf, int site) {
  const double s2 = 0.70710678118654752440, s3 = 0.57735026918962576450;
  double r3 = s2 * rf_gaussian(rf, site);
  double r8 = s2 * s3 * rf_gaussian(rf, site);
  m_zero(m);
  IM(m,0,0) = r8 + r3; IM(m,1,1) = r8 - r3; IM(m,2,2) = -2 * r8;
  double r01 = s2 * rf_gaussian(rf, site), r02 = s2 * rf_gaussian(rf, site), r12 = s2 * rf_gaussian(rf, site);
  double i01 = s2 * rf_gaussian(rf, site), i02 = s2 * rf_gaussian(rf, site), i12 = s2 * rf_gaussian(rf, site);
  RE(m,0,1) = r01; IM(m,0,1) = i01; RE(m,1,0) = -r01; IM(m,1,0) = i01;
  RE(m,0,2) = r02; IM(m,0,2) = i02; RE(m,2,0) = -r02; IM(m,2,0) = i02;
  RE(m,1,2) = r12; IM(m,1,2) = i12; RE(m,2,1) = -r12; IM(m,2,1) = i12;
}
void qo_gauge_random_tah(const qo_layout *lo, qo_rngfield *rf, double *g) {
  for (int mu = 0; mu < 4; mu++)
    for (int s = 0; s < lo->vol; s++) rand_tah3(&g[((size_t)s * 4 + mu) * 18], rf, s);
}
void qo_gauge_warm(const qo_layout *lo, qo_rngfield *rf, double s, double *g) {
  /* warmSU: x = exp(s * randomTAH) (gaugeUtils.nim:1384-1388,1431-1441) */
  qo_gauge_random_tah(lo, rf, g);
#pragma omp parallel for
  for (int i = 0; i < lo->vol * 4; i++) {
    double *m = &g[(size_t)i * 18], t[18];
    m_scale(t, s, m);
    qo_exp(m, t);
  }
}

/* ------------------------------------------------------------------ */
/* BC and staggered phases                                             */
/* ------------------------------------------------------------------ */
void qo_setBC(const qo_layout *lo, double *g) {
  /* gaugeUtils.nim:124-131: U_3 *= -1 on the last t slice */
  for (int s = 0; s < lo->vol; s++)
    if (lo->coords[4 * s + 3] == lo->L[3] - 1) {
      double *m = &g[((size_t)s * 4 + 3) * 18];
      for (int k = 0; k < 18; k++) m[k] = -m[k];
    }
}
void qo_stagPhase(const qo_layout *lo, double *g, const int phases[4]) {
  /* stagD.nim:509-518: bit k of phases[mu] selects coordinate k */
  for (int mu = 0; mu < 4; mu++)
    for (int i = 0; i < lo->vol; i++) {
      int s = 0;
      for (int k = 0; k < 4; k++) s += (phases[mu] >> k) & lo->coords[4 * i + k];
      if (s & 1) {
        double *m = &g[((size_t)i * 4 + mu) * 18];
        for (int k = 0; k < 18; k++) m[k] = -m[k];
      }
    }
}

/* ------------------------------------------------------------------ */
/* plaquette (gaugeUtils.nim:213-282)                                  */
/* ------------------------------------------------------------------ */
#define GLINK(g, s, mu) (&(g)[((size_t)(s) * 4 + (mu)) * 18])

void qo_plaq(const qo_layout *lo, const double *g, double out[6]) {
  int nt = qo_num_threads();
  double *part = (double *)calloc((size_t)nt * 6, sizeof(double));
#pragma omp parallel
  {
    double plt[6] = {0, 0, 0, 0, 0, 0};
#pragma omp for schedule(static)
    for (int ir = 0; ir < lo->vol; ir++)
      for (int mu = 1; mu < 4; mu++)
        for (int nu = 0; nu < mu; nu++) {
          double unumu[18], umunu[18];
          /* unumu = U_nu(x) U_mu(x+nu) ; umunu = U_mu(x) U_nu(x+mu) */
          m_mul(unumu, GLINK(g, ir, nu), GLINK(g, lo->nb[nu][0][ir], mu));
          m_mul(umunu, GLINK(g, ir, mu), GLINK(g, lo->nb[mu][0][ir], nu));
          plt[(mu * (mu - 1)) / 2 + nu] += m_retr_adj_mul(umunu, unumu);
        }
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    for (int i = 0; i < 6; i++) part[6 * tid + i] = plt[i];
  }
  /* thread-order sum, as threadSum does (base/threading.nim:291-316) */
  for (int i = 0; i < 6; i++) {
    double s = 0;
    for (int t = 0; t < nt; t++) s += part[6 * t + i];
    out[i] = s / ((double)lo->vol * (double)(6 * 3));
  }
  free(part);
}

/* s4_gauge (src/stagg_pv_hmc/staghmc_spv_meas.nim:25-65): for mu > nu the site plaquette
 * ps = redot(U_mu(x) U_nu(x+mu), U_nu(x) U_mu(x+nu)) is added to peo[mu][x_mu mod 2] and peo[nu][x_nu mod 2] (:46-52);
 * out[2 d + eo] = peo[d][eo] / (physVol * 0.5 * (nd - 1) * nc) (:58-61).  Serial sum (test sizes only). */
void qo_s4_gauge(const qo_layout *lo, const double *g, double out[8]) {
  double peo[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int ir = 0; ir < lo->vol; ir++) {
    int x[4];
    qo_coord(lo, ir, x);
    for (int mu = 1; mu < 4; mu++)
      for (int nu = 0; nu < mu; nu++) {
        double unumu[18], umunu[18];
        m_mul(unumu, GLINK(g, ir, nu), GLINK(g, lo->nb[nu][0][ir], mu));
        m_mul(umunu, GLINK(g, ir, mu), GLINK(g, lo->nb[mu][0][ir], nu));
        const double ps = m_retr_adj_mul(umunu, unumu);
        peo[2 * mu + (x[mu] & 1)] += ps;
        peo[2 * nu + (x[nu] & 1)] += ps;
      }
  }
  const double n = 1.0 / ((double)lo->vol * 0.5 * 3.0 * 3.0);
  for (int k = 0; k < 8; k++) out[k] = peo[k] * n;
}

/* ------------------------------------------------------------------ */
/* gauge force + Wilson flow                                           */
/* ------------------------------------------------------------------ */
/* gaugeActionDeriv, plaquette part (gaugeAction.nim:148-204) with staples from
 * makeStaples (staples.nim:153-238):
 *   stf[mu,nu](x) = U_nu(x) U_mu(x+nu) U_nu(x+mu)^+
 *   stu[mu,nu](x) = U_nu(x)^+ U_mu(x) U_nu(x+mu)   (used at x+nu: "offset up")
 *   f[mu](x) = cp * sum_{nu!=mu} [ stf[mu,nu](x) + stu[mu,nu](x-nu) ],  cp = c.plaq/nc
 * accumulation order follows the (mu>nu) pair loop of :186-194. */
void qo_gauge_deriv(const qo_layout *lo, const double *g, double *f, double cplaq) {
  const double cp = cplaq / 3.0;
#pragma omp parallel for schedule(static)
  for (int ir = 0; ir < lo->vol; ir++) {
    double acc[4][18];
    for (int mu = 0; mu < 4; mu++) m_zero(acc[mu]);
    for (int mu = 1; mu < 4; mu++)
      for (int nu = 0; nu < mu; nu++) {
        double umunu[18], t[18], st[18];
        const double *Umu = GLINK(g, ir, mu), *Unu = GLINK(g, ir, nu);
        const double *umu_n = GLINK(g, lo->nb[nu][0][ir], mu); /* U_mu(x+nu) */
        const double *unu_m = GLINK(g, lo->nb[mu][0][ir], nu); /* U_nu(x+mu) */
        m_mul_na(umunu, umu_n, unu_m);          /* U_mu(x+nu) U_nu(x+mu)^+ */
        m_mul(st, Unu, umunu);                  /* stf[mu,nu] */
        m_axpy(acc[mu], cp, st);
        m_mul_na(st, Umu, umunu);               /* stf[nu,mu] = U_mu(x) umunu^+ */
        m_axpy(acc[nu], cp, st);
        /* backward staples, evaluated at x-nu (for mu) and x-mu (for nu) */
        int xb = lo->nb[nu][1][ir];
        m_mul_an(t, GLINK(g, xb, nu), GLINK(g, xb, mu));       /* U_nu^+ U_mu at x-nu */
        m_mul(st, t, GLINK(g, lo->nb[mu][0][xb], nu));         /* * U_nu(x-nu+mu) */
        m_axpy(acc[mu], cp, st);
        xb = lo->nb[mu][1][ir];
        m_mul_an(t, GLINK(g, xb, mu), GLINK(g, xb, nu));       /* U_mu^+ U_nu at x-mu  (= unumu^+) */
        m_mul(st, t, GLINK(g, lo->nb[nu][0][xb], mu));         /* * U_mu(x-mu+nu) */
        m_axpy(acc[nu], cp, st);
      }
    for (int mu = 0; mu < 4; mu++) m_copy(&f[((size_t)ir * 4 + mu) * 18], acc[mu]);
  }
}

/* gaugeForce (gaugeAction.nim:334-350) = deriv + contractProjectTAH (gaugeUtils.nim:389-398):
 * f <- TAH( U_mu(x) f_mu(x)^+ ) */
void qo_gauge_force(const qo_layout *lo, const double *g, double *f) {
  qo_gauge_deriv(lo, g, f, 1.0);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < lo->vol * 4; i++) {
    double s[18];
    m_mul_na(s, &g[(size_t)i * 18], &f[(size_t)i * 18]);
    qo_projectTAH(&f[(size_t)i * 18], s);
  }
}

/* gaugeFlow (wflow.nim:21-67): Luescher RK3 */
void qo_wflow(const qo_layout *lo, double *g, int nsteps, double eps) {
  size_t n = (size_t)lo->vol * 4 * 18;
  double *p = (double *)malloc(sizeof(double) * n);
  double *f = (double *)malloc(sizeof(double) * n);
  const double epsnc = eps * 3.0;
  for (int step = 0; step < nsteps; step++) {
    for (int stage = 0; stage < 3; stage++) {
      qo_gauge_force(lo, g, f);
#pragma omp parallel for schedule(static)
      for (int i = 0; i < lo->vol * 4; i++) {
        double v[18], e[18], t[18];
        double *gi = &g[(size_t)i * 18], *fi = &f[(size_t)i * 18], *pi = &p[(size_t)i * 18];
        if (stage == 0) for (int k = 0; k < 18; k++) v[k] = (-1.0 / 4.0) * epsnc * fi[k];
        else if (stage == 1) for (int k = 0; k < 18; k++) v[k] = (-8.0 / 9.0) * epsnc * fi[k] + (-17.0 / 9.0) * pi[k];
        else for (int k = 0; k < 18; k++) v[k] = (-3.0 / 4.0) * epsnc * fi[k] - pi[k];
        qo_exp(e, v);
        m_mul(t, e, gi);
        if (stage < 2) m_copy(pi, v);
        m_copy(gi, t);
      }
    }
  }
  free(p); free(f);
}

/* ------------------------------------------------------------------ */
/* field algebra: norm2 / redot with fp64 accumulation                 */
/* (fieldET.nim:605-625,704-724; thread partials summed in thread      */
/*  order, base/threading.nim:291-316)                                 */
/* ------------------------------------------------------------------ */
double qo_norm2(const qo_layout *lo, const double *x, int parity) { return qo_redot(lo, x, x, parity); }

double qo_redot(const qo_layout *lo, const double *x, const double *y, int parity) {
  int s0, s1; subset_range(lo, parity, &s0, &s1);
  int nt = qo_num_threads();
  double *part = (double *)calloc((size_t)nt, sizeof(double));
#pragma omp parallel
  {
    /* per-thread SIMD accumulator of V=8 lanes (globals.nim:33-36), then simdSum (lane order),
     * then thread partials in thread order -- the structure of fieldET.nim:605-625 */
    double lane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma omp for schedule(static)
    for (int s = s0; s < s1; s++) {
      double t = 0;
      for (int k = 0; k < 6; k++) t += x[6 * (size_t)s + k] * y[6 * (size_t)s + k];
      lane[s & 7] += t;
    }
    double acc = ((lane[0] + lane[1]) + (lane[2] + lane[3])) + ((lane[4] + lane[5]) + (lane[6] + lane[7]));
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    part[tid] = acc;
  }
  double r = 0;
  for (int t = 0; t < nt; t++) r += part[t];
  free(part);
  return r;
}

/* ------------------------------------------------------------------ */
/* staggered Dslash                                                    */
/* ------------------------------------------------------------------ */
/* rir += U * v   (imadd) */
static inline void mv_add(double *r, const double *U, const double *v) {
  for (int i = 0; i < 3; i++) {
    double sr = r[2 * i], si = r[2 * i + 1];
    for (int j = 0; j < 3; j++) {
      sr += RE(U,i,j) * v[2 * j] - IM(U,i,j) * v[2 * j + 1];
      si += RE(U,i,j) * v[2 * j + 1] + IM(U,i,j) * v[2 * j];
    }
    r[2 * i] = sr; r[2 * i + 1] = si;
  }
}
/* t = U^+ * v */
static inline void mv_adj(double *t, const double *U, const double *v) {
  for (int i = 0; i < 3; i++) {
    double sr = 0, si = 0;
    for (int j = 0; j < 3; j++) {
      sr += RE(U,j,i) * v[2 * j] + IM(U,j,i) * v[2 * j + 1];
      si += RE(U,j,i) * v[2 * j + 1] - IM(U,j,i) * v[2 * j];
    }
    t[2 * i] = sr; t[2 * i + 1] = si;
  }
}

/* stagD2 (stagD.nim:349-395): r = a*r + b*x + sum_mu [U_mu(s) x(s+mu) - U_mu^+(s-mu) x(s-mu)],
 * with 8 links the odd entries hop +-3 (initStagD3T :38-49; fat = g[2mu], long = g[2mu+1]).
 * Per-site order: for mu { forward fat, backward fat, forward long, backward long }. */
void qo_stagD2(const qo_layout *lo, const double *fat, const double *lng,
               double *r, const double *x, int parity, double a, double b) {
  int s0, s1; subset_range(lo, parity, &s0, &s1);
#pragma omp parallel for schedule(static)
  for (int ir = s0; ir < s1; ir++) {
    double rir[6], t[6];
    for (int k = 0; k < 6; k++) {
      /* a==0 must not propagate NaN/garbage from r (reference evaluates a*r[ir]; with r
       * freshly allocated = 0 this is identical).  We keep the literal expression. */
      rir[k] = (a == 0.0 ? 0.0 : a * r[6 * (size_t)ir + k]) + (b == 0.0 ? 0.0 : b * x[6 * (size_t)ir + k]);
    }
    for (int mu = 0; mu < 4; mu++) {
      int xf = lo->nb[mu][0][ir], xb = lo->nb[mu][1][ir];
      mv_add(rir, GLINK(fat, ir, mu), &x[6 * (size_t)xf]);
      mv_adj(t, GLINK(fat, xb, mu), &x[6 * (size_t)xb]);
      for (int k = 0; k < 6; k++) rir[k] -= t[k];
      if (lng) {
        int xf3 = lo->nb[mu][2][ir], xb3 = lo->nb[mu][3][ir];
        mv_add(rir, GLINK(lng, ir, mu), &x[6 * (size_t)xf3]);
        mv_adj(t, GLINK(lng, xb3, mu), &x[6 * (size_t)xb3]);
        for (int k = 0; k < 6; k++) rir[k] -= t[k];
      }
    }
    for (int k = 0; k < 6; k++) r[6 * (size_t)ir + k] = rir[k];
  }
}

/* stagD (stagD.nim:406-409): stagD2(a/(.5sc), m/(.5sc)); r[subset] := (.5sc)*r */
void qo_stagD(const qo_layout *lo, const double *fat, const double *lng,
              double *r, const double *x, int parity, double m, double sc, double a) {
  qo_stagD2(lo, fat, lng, r, x, parity, a / (0.5 * sc), m / (0.5 * sc));
  int s0, s1; subset_range(lo, parity, &s0, &s1);
  const double h = 0.5 * sc;
#pragma omp parallel for schedule(static)
  for (int s = s0; s < s1; s++) for (int k = 0; k < 6; k++) r[6 * (size_t)s + k] = h * r[6 * (size_t)s + k];
}
/* D / Ddag (stagD.nim:566-571) */
void qo_D(const qo_layout *lo, const double *fat, const double *lng, double *r, const double *x, double m) {
  qo_stagD(lo, fat, lng, r, x, 0, m, 1.0, 0.0);
  qo_stagD(lo, fat, lng, r, x, 1, m, 1.0, 0.0);
}
void qo_Ddag(const qo_layout *lo, const double *fat, const double *lng, double *r, const double *x, double m) {
  qo_stagD(lo, fat, lng, r, x, 0, m, -1.0, 0.0);
  qo_stagD(lo, fat, lng, r, x, 1, m, -1.0, 0.0);
}
/* eoReduce (stagD.nim:575-581): r.even = (D^+ b).even -- stagD on the even subset with sc = -1 */
void qo_eoReduce(const qo_layout *lo, const double *fat, const double *lng, double *r, const double *b, double m) {
  qo_stagD(lo, fat, lng, r, b, 0, m, -1.0, 0.0);
}
/* eoReconstruct (stagD.nim:583-586): r.odd = (b.odd - D_oe r.even)/m */
void qo_eoReconstruct(const qo_layout *lo, const double *fat, const double *lng,
                      double *r, const double *b, double m) {
  qo_stagD(lo, fat, lng, r, r, 1, 0.0, -1.0 / m, 0.0);
#pragma omp parallel for schedule(static)
  for (int s = lo->volh; s < lo->vol; s++) for (int k = 0; k < 6; k++) r[6 * (size_t)s + k] += b[6 * (size_t)s + k] / m;
}

/* stagD2xx (stagD.nim:434-469): t[y] = (2D) x via stagDP (:200-237);
 * r[x-parity] = 4 m2 x - (2D) t via stagDM (:278-313: per mu, -U t(+mu) then +U^+ t(-mu)). */
static void stagD2xx_t(const qo_layout *lo, const double *fat, const double *lng,
                       double *r, const double *x, double m2, int par_even, double *t) {
  int px = par_even ? 0 : 1, py = 1 - px;
  /* stagDP on the other parity: all forward hops, then all backward hops */
  int s0, s1; subset_range(lo, py, &s0, &s1);
  const int nl = lng ? 2 : 1;
#pragma omp parallel for schedule(static)
  for (int ir = s0; ir < s1; ir++) {
    double rir[6] = {0, 0, 0, 0, 0, 0}, u[6];
    for (int mu = 0; mu < 4; mu++)
      for (int l = 0; l < nl; l++) {
        const double *G = l ? lng : fat;
        mv_add(rir, GLINK(G, ir, mu), &x[6 * (size_t)lo->nb[mu][2 * l][ir]]);
      }
    for (int mu = 0; mu < 4; mu++)
      for (int l = 0; l < nl; l++) {
        const double *G = l ? lng : fat;
        int xb = lo->nb[mu][2 * l + 1][ir];
        mv_adj(u, GLINK(G, xb, mu), &x[6 * (size_t)xb]);
        for (int k = 0; k < 6; k++) rir[k] -= u[k];
      }
    for (int k = 0; k < 6; k++) t[6 * (size_t)ir + k] = rir[k];
  }
  subset_range(lo, px, &s0, &s1);
#pragma omp parallel for schedule(static)
  for (int ir = s0; ir < s1; ir++) {
    double rir[6], u[6], nrm[6];
    for (int k = 0; k < 6; k++) rir[k] = (4.0 * m2) * x[6 * (size_t)ir + k];
    for (int mu = 0; mu < 4; mu++)
      for (int l = 0; l < nl; l++) {
        const double *G = l ? lng : fat;
        /* imsub(rir, U, t(+)) */
        for (int k = 0; k < 6; k++) nrm[k] = 0;
        mv_add(nrm, GLINK(G, ir, mu), &t[6 * (size_t)lo->nb[mu][2 * l][ir]]);
        for (int k = 0; k < 6; k++) rir[k] -= nrm[k];
        int xb = lo->nb[mu][2 * l + 1][ir];
        mv_adj(u, GLINK(G, xb, mu), &t[6 * (size_t)xb]);
        for (int k = 0; k < 6; k++) rir[k] += u[k];
      }
    for (int k = 0; k < 6; k++) r[6 * (size_t)ir + k] = rir[k];
  }
}
void qo_stagD2xx(const qo_layout *lo, const double *fat, const double *lng,
                 double *r, const double *x, double m2, int par_even) {
  double *t = (double *)calloc((size_t)lo->vol * 6, sizeof(double));
  stagD2xx_t(lo, fat, lng, r, x, m2, par_even, t);
  free(t);
}

/* ------------------------------------------------------------------ */
/* CG (solvers/cg.nim:55-272, precon = cpNone => z=r, q=p, LAp=Ap)     */
/* ------------------------------------------------------------------ */
int qo_solveXX(const qo_layout *lo, const double *fat, const double *lng,
               double *x, const double *b, double m, double r2req, int maxits, int par_even,
               double *r2hist, int histcap, double *final_r2_over_b2) {
  /* solveXX(s, r, x, ...) (stagSolve.nim:57-132): here `x` is the solution ("r" there), `b` the rhs. */
  const int par = par_even ? 0 : 1;
  size_t n6 = (size_t)lo->vol * 6;
  int s0, s1; subset_range(lo, par, &s0, &s1);
  double *r = (double *)calloc(n6, sizeof(double));
  double *p = (double *)calloc(n6, sizeof(double));
  double *Ap = (double *)calloc(n6, sizeof(double));
  double *t = (double *)calloc(n6, sizeof(double));
  const double m2 = m * m;
  /* threads: r := 0  (stagSolve.nim:63-64) -- whole field */
  memset(x, 0, sizeof(double) * n6);
  double b2 = qo_norm2(lo, b, par);                                  /* cg.nim:134 */
  double r2 = 1.0, rzo = 1.0;
  int itn = 0, nh = 0;
  if (b2 == 0.0) {
    r2 = 0.0;                                                        /* :139-144 */
  } else {
    stagD2xx_t(lo, fat, lng, Ap, x, m2, par_even, t);                /* :147 op.apply(Ap,x) */
#pragma omp parallel for schedule(static)
    for (int s = s0; s < s1; s++) for (int k = 0; k < 6; k++) { r[6 * (size_t)s + k] = b[6 * (size_t)s + k] - Ap[6 * (size_t)s + k]; p[6 * (size_t)s + k] = 0; }
    r2 = qo_norm2(lo, r, par);
  }
  const double r2stop = r2req * b2;                                  /* :155 */
  if (r2hist && nh < histcap) r2hist[nh++] = (b2 != 0 ? r2 / b2 : 0.0);
  while (itn < maxits && r2 > r2stop) {                              /* :174 */
    double rz = r2;                                                  /* getRz, cpNone */
    double beta = rz / rzo;                                          /* :186 */
    rzo = rz;
    if (itn == 0) {
#pragma omp parallel for schedule(static)
      for (int s = s0; s < s1; s++) for (int k = 0; k < 6; k++) p[6 * (size_t)s + k] = r[6 * (size_t)s + k];
    } else {
#pragma omp parallel for schedule(static)
      for (int s = s0; s < s1; s++) for (int k = 0; k < 6; k++) p[6 * (size_t)s + k] = r[6 * (size_t)s + k] + beta * p[6 * (size_t)s + k];
    }
    itn++;
    stagD2xx_t(lo, fat, lng, Ap, p, m2, par_even, t);                /* :200 */
    double pAp = qo_redot(lo, p, Ap, par);                           /* :206 */
    double alpha = rz / pAp;                                         /* :208 */
#pragma omp parallel for schedule(static)
    for (int s = s0; s < s1; s++) for (int k = 0; k < 6; k++) {
      x[6 * (size_t)s + k] += alpha * p[6 * (size_t)s + k];          /* :209 */
      r[6 * (size_t)s + k] -= alpha * Ap[6 * (size_t)s + k];         /* :211 */
    }
    r2 = qo_norm2(lo, r, par);                                       /* :213 */
    if (r2hist && nh < histcap) r2hist[nh++] = r2 / b2;
  }
  if (final_r2_over_b2) *final_r2_over_b2 = (b2 != 0 ? r2 / b2 : 0.0);
  free(r); free(p); free(Ap); free(t);
  return itn;
}

/* ------------------------------------------------------------------ */
/* full solve  D x = b  (stagSolve.nim:141-294)                        */
/* ------------------------------------------------------------------ */
static int solve_inner(const qo_layout *lo, const double *fat, const double *lng,
                       double *x, const double *b, double m, double r2req, int maxits,
                       double b2e, double b2o) {
  size_t n6 = (size_t)lo->vol * 6;
  const double b2 = b2e + b2o;
  const double r2stop = r2req * b2, r2stop2 = 0.5 * r2stop;
  const double r2stope = (b2o <= r2stop2) ? r2stop - b2o : r2stop2;
  const double r2stopo = (b2e <= r2stop2) ? r2stop - b2e : r2stop2;
  int its = 0;
  if (b2e <= r2stope || b2o <= r2stopo || m == 0.0) {
    /* solveReconR (:141-176) */
    double *y = (double *)calloc(n6, sizeof(double));
    if (b2e > r2stope) {
      its = qo_solveXX(lo, fat, lng, y, b, m, r2stope / b2e, maxits, 1, NULL, 0, NULL);
      for (int s = 0; s < lo->volh; s++) for (int k = 0; k < 6; k++) y[6 * (size_t)s + k] *= 4;
      qo_Ddag(lo, fat, lng, x, y, m);
    } else if (b2o > r2stopo) {
      its = qo_solveXX(lo, fat, lng, y, b, m, r2stopo / b2o, maxits, 0, NULL, 0, NULL);
      for (int s = lo->volh; s < lo->vol; s++) for (int k = 0; k < 6; k++) y[6 * (size_t)s + k] *= 4;
      qo_Ddag(lo, fat, lng, x, y, m);
    }
    free(y);
  } else {
    /* solveReconL (:179-208) */
    double *d = (double *)calloc(n6, sizeof(double));
    qo_Ddag(lo, fat, lng, d, b, m);
    memset(x, 0, sizeof(double) * n6);
    double d2e = qo_norm2(lo, d, 0);
    double rr = 0.99 * r2req * (b2e + b2o) * m * m / d2e;
    its = qo_solveXX(lo, fat, lng, x, d, m, rr, maxits, 1, NULL, 0, NULL);
    for (int s = 0; s < lo->volh; s++) for (int k = 0; k < 6; k++) x[6 * (size_t)s + k] *= 4;
    qo_eoReconstruct(lo, fat, lng, x, b, m);
    free(d);
  }
  return its;
}

int qo_solve_prev(const qo_layout *lo, const double *fat, const double *lng,
                  double *x, const double *b, double m, double r2req, int maxits, int use_prev, double *r2_final);
int qo_solve(const qo_layout *lo, const double *fat, const double *lng,
             double *x, const double *b, double m, double r2req, int maxits, double *r2_final) {
  return qo_solve_prev(lo, fat, lng, x, b, m, r2req, maxits, 0, r2_final);
}
/* sp.usePrevSoln (stagSolve.nim:234-243): start from the x handed in, r = b - D x */
int qo_solve_prev(const qo_layout *lo, const double *fat, const double *lng,
                  double *x, const double *b, double m, double r2req, int maxits, int use_prev, double *r2_final) {
  size_t n6 = (size_t)lo->vol * 6;
  double b2 = qo_norm2(lo, b, 2);
  const double r2stop = r2req * b2;
  double *r = (double *)malloc(sizeof(double) * n6);
  double *y = (double *)calloc(n6, sizeof(double));
  if (use_prev) {
    qo_D(lo, fat, lng, r, x, m);
    for (size_t i = 0; i < n6; i++) r[i] = b[i] - r[i];
  } else {
    memset(x, 0, sizeof(double) * n6);
    memcpy(r, b, sizeof(double) * n6);
  }
  double r2e = qo_norm2(lo, r, 0), r2o = qo_norm2(lo, r, 1);
  double r2 = r2e + r2o;
  int its = 0;
  while (r2 > r2stop) {
    int mx = maxits - its;
    if (mx <= 0) break;
    its += solve_inner(lo, fat, lng, y, r, m, r2stop / r2, mx, r2e, r2o);
    for (size_t i = 0; i < n6; i++) x[i] += y[i];
    qo_D(lo, fat, lng, r, x, m);
    for (size_t i = 0; i < n6; i++) r[i] = b[i] - r[i];
    r2e = qo_norm2(lo, r, 0); r2o = qo_norm2(lo, r, 1);
    r2 = r2e + r2o;
  }
  if (r2_final) *r2_final = (b2 != 0 ? r2 / b2 : 0.0);
  free(r); free(y);
  return its;
}

/* ------------------------------------------------------------------ */
/* multi-shift CG (solvers/cgm.nim:84-315, precon = cpNone)            */
/* op = stagD2ee|oo(mass^2 + shift)  (stagSolve.nim:318-325)           */
/* ------------------------------------------------------------------ */
int qo_solveXX_multi(const qo_layout *lo, const double *fat, const double *lng,
                     double **xs, const double *b, const double *shifts, int nmass,
                     double r2req, int maxits, int par_even, double *r2hist, int histcap) {
  const int par = par_even ? 0 : 1;
  size_t n6 = (size_t)lo->vol * 6;
  int s0, s1; subset_range(lo, par, &s0, &s1);
  const double mass = shifts[0];
  const double m2 = mass * mass;
  double *sg = (double *)calloc((size_t)nmass, sizeof(double));
  for (int k = 1; k < nmass; k++) sg[k] = shifts[k];  /* sg[0] = 0 (cgm.nim:108-111) */
  double *r = (double *)calloc(n6, sizeof(double));
  double *Ap = (double *)calloc(n6, sizeof(double));
  double *t = (double *)calloc(n6, sizeof(double));
  double **ps = (double **)calloc((size_t)nmass, sizeof(double *));
  for (int k = 0; k < nmass; k++) ps[k] = (double *)calloc(n6, sizeof(double));
  double *zi = (double *)calloc((size_t)nmass, sizeof(double));
  double *zim1 = (double *)calloc((size_t)nmass, sizeof(double));
  /* new solution branch (:169-175): r := b; xs := 0; b2 = |b|^2; r2 = b2 */
#pragma omp parallel for schedule(static)
  for (int s = s0; s < s1; s++) for (int k = 0; k < 6; k++) r[6 * (size_t)s + k] = b[6 * (size_t)s + k];
  for (int k = 0; k < nmass; k++)
    for (int s = s0; s < s1; s++) for (int c = 0; c < 6; c++) xs[k][6 * (size_t)s + c] = 0.0;
  double b2 = qo_norm2(lo, b, par), r2 = b2;
  const double r2stop = r2req * b2;
  int itn = 0, nh = 0;
  if (r2hist && nh < histcap) r2hist[nh++] = (b2 != 0 ? r2 / b2 : 0.0);
  if (r2 > r2stop) {
    double alphaim1 = -1.0, betaim1 = 0.0, alpha = 0, beta = 0;
    for (int k = 0; k < nmass; k++) { zim1[k] = 1.0; zi[k] = 1.0; }
    double r2i = r2, r2ip1 = 0;
    for (int k = 0; k < nmass; k++)
      for (int s = s0; s < s1; s++) for (int c = 0; c < 6; c++) ps[k][6 * (size_t)s + c] = r[6 * (size_t)s + c];
    int continuing = 1;
    while (continuing) {
      for (int k = 0; k < nmass; k++) {
        if (k == 0) {
          stagD2xx_t(lo, fat, lng, Ap, ps[0], m2, par_even, t);
          itn++;
          double qLAp = qo_redot(lo, ps[0], Ap, par);
          alpha = (qLAp != 0.0) ? r2i / qLAp : 0.0;
#pragma omp parallel for schedule(static)
          for (int s = s0; s < s1; s++) for (int c = 0; c < 6; c++) {
            r[6 * (size_t)s + c] -= alpha * Ap[6 * (size_t)s + c];
            xs[0][6 * (size_t)s + c] += alpha * ps[0][6 * (size_t)s + c];
          }
          r2ip1 = qo_norm2(lo, r, par);
          beta = (r2i != 0.0) ? r2ip1 / r2i : 0.0;
          continuing = (itn < maxits) && (r2ip1 > r2stop);
          if (continuing) {
#pragma omp parallel for schedule(static)
            for (int s = s0; s < s1; s++) for (int c = 0; c < 6; c++)
              ps[0][6 * (size_t)s + c] = r[6 * (size_t)s + c] + beta * ps[0][6 * (size_t)s + c];
          }
        } else {
          double zip1d = alpha * betaim1 * (zim1[k] - zi[k]);
          zip1d += zim1[k] * alphaim1 * (1.0 + sg[k] * alpha);
          double zip1 = (zip1d != 0.0) ? zi[k] * zim1[k] * alphaim1 / zip1d : 0.0;
          double zr = (zi[k] != 0.0) ? zip1 / zi[k] : 0.0;
          const double axz = alpha * zr;
#pragma omp parallel for schedule(static)
          for (int s = s0; s < s1; s++) for (int c = 0; c < 6; c++)
            xs[k][6 * (size_t)s + c] += axz * ps[k][6 * (size_t)s + c];
          if (continuing) {
            const double bzz = beta * zr * zr;
#pragma omp parallel for schedule(static)
            for (int s = s0; s < s1; s++) for (int c = 0; c < 6; c++)
              ps[k][6 * (size_t)s + c] = zip1 * r[6 * (size_t)s + c] + bzz * ps[k][6 * (size_t)s + c];
            zim1[k] = zi[k]; zi[k] = zip1;
          }
        }
      }
      alphaim1 = alpha; betaim1 = beta; r2i = r2ip1;
      r2 = r2ip1;
      if (r2hist && nh < histcap) r2hist[nh++] = r2 / b2;
    }
  }
  for (int k = 0; k < nmass; k++) free(ps[k]);
  free(ps); free(zi); free(zim1); free(r); free(Ap); free(t); free(sg);
  return itn;
}

/* multi-mass solve (stagSolve.nim:347-446) */
int qo_solve_multi(const qo_layout *lo, const double *fat, const double *lng,
                   double **xs, const double *b, const double *masses, int nmass,
                   double r2req, int maxits, double *r2_final) {
  size_t n6 = (size_t)lo->vol * 6;
  const double mass = masses[0];
  double *shifts = (double *)calloc((size_t)nmass, sizeof(double));
  double **ys = (double **)calloc((size_t)nmass, sizeof(double *));
  double *r = (double *)malloc(sizeof(double) * n6);
  double *xt = (double *)calloc(n6, sizeof(double));
  memcpy(r, b, sizeof(double) * n6);
  double b2 = qo_norm2(lo, b, 2), b2e = qo_norm2(lo, b, 0), b2o = qo_norm2(lo, b, 1);
  double r2 = b2e + b2o;
  const double r2stop = r2req * b2;
  for (int k = 0; k < nmass; k++) {
    shifts[k] = (k == 0) ? masses[0] : 4.0 * (masses[k] * masses[k] - mass * mass);
    ys[k] = (double *)calloc(n6, sizeof(double));
    memset(xs[k], 0, sizeof(double) * n6);
  }
  int its = 0;
  while (r2 > r2stop) {
    int mx = maxits - its;
    if (mx <= 0) break;
    double rq = r2stop;
    const double r2stop2 = 0.5 * rq;
    const double r2stope = (b2o <= r2stop2) ? rq - b2o : r2stop2;
    const double r2stopo = (b2e <= r2stop2) ? rq - b2e : r2stop2;
    int even = 1;
    if (b2e > r2stope) { rq = r2stope / b2e; even = 1; }
    else if (b2o > r2stopo) { rq = r2stopo / b2o; even = 0; }
    its += qo_solveXX_multi(lo, fat, lng, ys, r, shifts, nmass, rq, mx, even, NULL, 0);
    int s0 = even ? 0 : lo->volh, s1 = even ? lo->volh : lo->vol;
    for (int k = 0; k < nmass; k++)
      for (int s = s0; s < s1; s++) for (int c = 0; c < 6; c++) xs[k][6 * (size_t)s + c] += 4.0 * ys[k][6 * (size_t)s + c];
    qo_Ddag(lo, fat, lng, xt, xs[0], mass);
    qo_D(lo, fat, lng, r, xt, mass);
    for (size_t i = 0; i < n6; i++) r[i] = b[i] - r[i];
    b2e = qo_norm2(lo, r, 0); b2o = qo_norm2(lo, r, 1);
    r2 = b2e + b2o;
  }
  /* full solutions (:431-438): for m != 0 recompute xt = Ddag(m_k) xs[k]; xs[k] := xt
   * (for m == 0, xt still holds Ddag(mass) xs[0] from the loop above). */
  for (int k = 0; k < nmass; k++) {
    if (k != 0) qo_Ddag(lo, fat, lng, xt, xs[k], masses[k]);
    memcpy(xs[k], xt, sizeof(double) * n6);
  }
  if (r2_final) *r2_final = (b2 != 0 ? r2 / b2 : 0.0);
  for (int k = 0; k < nmass; k++) free(ys[k]);
  free(ys); free(shifts); free(r); free(xt);
  return its;
}

void qo_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ------------------------------------------------------------------ */
/* flow observables: F_munu from clover-type loops, E, Q, Wilson lines */
/* (src/gauge/gaugeUtils.nim:1079-1270)           */
/* ------------------------------------------------------------------ */
/* ordered product of links along `path` (entries +-(d+1)) starting and ending at site x
 * ("origin=true" of gaugeProd, gaugeUtils.nim:951-1077; convention fixed by
 * tests/base/tgaugeprod.nim:13-19: [1,2,-1,-2] is the plaquette U_0(x)U_1(x+0)U_0(x+1)^+U_1(x)^+) */
static void path_prod(const qo_layout *lo, const double *g, int x, const int *path, int n, double *out) {
  double m[18], t[18];
  m_unit(m);
  int cur = x;
  for (int i = 0; i < n; i++) {
    int s = path[i];
    if (s > 0) {
      int d = s - 1;
      m_mul(t, m, GLINK(g, cur, d));
      cur = lo->nb[d][0][cur];
    } else {
      int d = -s - 1;
      cur = lo->nb[d][1][cur];
      m_mul_na(t, m, GLINK(g, cur, d));
    }
    m_copy(m, t);
  }
  m_copy(out, m);
}

/* allCorners (gaugeUtils.nim:1114-1126): the rotations of a closed path that start at a corner */
static int all_corners(const int *path, int np, int out[][16]) {
  int n = 0, old = 0;
  for (int i = 0; i < np; i++) {
    if (path[i] != old) {
      for (int j = 0; j < np; j++) out[n][j] = path[(j + i) % np];
      n++;
      old = path[i];
    }
  }
  return n;
}

/* fmunuCoeffs (gaugeUtils.nim:1128-1146) */
static void fmunu_coeffs(int loop, double k[5]) {
  for (int i = 0; i < 5; i++) k[i] = 0;
  if (loop == 1) { k[0] = 1.0; return; }
  k[4] = (loop == 3) ? 1.0 / 90.0 : (loop == 5 ? 1.0 / 180.0 : 0.0);
  k[0] = 19.0 / 9.0 - 55.0 * k[4];
  k[1] = 1.0 / 36.0 - 16.0 * k[4];
  k[2] = 64.0 * k[4] - 32.0 / 45.0;
  k[3] = 1.0 / 15.0 - 6.0 * k[4];
}

/* fmunu(g, mu, nu, loop) (gaugeUtils.nim:1162-1218): traceless anti-Hermitian F_munu from the
 * 1x1 (, 2x2, 1x2/2x1, 1x3/3x1, 3x3) clover leaves.  f: [vol][18] */
static void fmunu_plane(const qo_layout *lo, const double *g, int mu, int nu, int loop, double *f) {
  int paths[32][16], lens[32], group[32], np = 0;
  const int a = mu + 1, b = nu + 1;
  int tmp[8][16];
#define ADDLOOP(grp, ...) do { const int p_[] = {__VA_ARGS__}; int l_ = (int)(sizeof(p_) / sizeof(int)); \
    int c_ = all_corners(p_, l_, tmp); for (int q_ = 0; q_ < c_; q_++) { for (int j_ = 0; j_ < l_; j_++) paths[np][j_] = tmp[q_][j_]; lens[np] = l_; group[np] = grp; np++; } } while (0)
  ADDLOOP(0, -a, -b, a, b);                                             /* 1x1 */
  if (loop >= 3) ADDLOOP(1, -a, -a, -b, -b, a, a, b, b);                /* 2x2 */
  if (loop >= 4) {
    ADDLOOP(2, -a, -a, -b, a, a, b);                                    /* 2x1 */
    ADDLOOP(2, -a, -b, -b, a, b, b);                                    /* 1x2 */
    ADDLOOP(3, -a, -a, -a, -b, a, a, a, b);                             /* 3x1 */
    ADDLOOP(3, -a, -b, -b, -b, a, b, b, b);                             /* 1x3 */
  }
  if (loop == 3 || loop == 5) ADDLOOP(4, -a, -a, -a, -b, -b, -b, a, a, a, b, b, b);  /* 3x3 */
#undef ADDLOOP
  static const int lpc[5] = {4, 4, 8, 8, 4};
  double cs[5];
  fmunu_coeffs(loop, cs);
#pragma omp parallel for schedule(static)
  for (int x = 0; x < lo->vol; x++) {
    double acc[18], grp[5][18], m[18];
    m_zero(acc);
    for (int j = 0; j < 5; j++) m_zero(grp[j]);
    for (int p = 0; p < np; p++) {
      path_prod(lo, g, x, paths[p], lens[p], m);
      for (int k = 0; k < 18; k++) grp[group[p]][k] += m[k];
    }
    for (int j = 0; j < 5; j++) {
      const double ni = cs[j] / (double)lpc[j];
      for (int k = 0; k < 18; k++) acc[k] += ni * grp[j][k];
    }
    qo_projectTAH(&f[(size_t)x * 18], acc);
  }
}

/* reTrMul (gaugeUtils.nim:1234-1239): sum_x Re tr( x y ) */
static double retr_mul(const qo_layout *lo, const double *x, const double *y) {
  double s = 0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int i = 0; i < lo->vol; i++) {
    const double *a = &x[(size_t)i * 18], *b = &y[(size_t)i * 18];
    double t = 0;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) t += RE(a, r, c) * RE(b, c, r) - IM(a, r, c) * IM(b, c, r);
    s += t;
  }
  return s;
}

/* EQ of tests/base/twflow_topo.nim:4-10: [E_s, E_t, Q] from fmunu(loop), densityE
 * (gaugeUtils.nim:1241-1257) and topoQ (:1259-1271) */
void qo_flow_EQ(const qo_layout *lo, const double *g, int loop, double out[3]) {
  double *F[6];
  for (int i = 0; i < 6; i++) F[i] = (double *)malloc(sizeof(double) * 18 * (size_t)lo->vol);
  for (int mu = 1; mu < 4; mu++)
    for (int nu = 0; nu < mu; nu++) fmunu_plane(lo, g, mu, nu, loop, F[(mu * (mu - 1)) / 2 + nu]);
  double es = 0, et = 0;
  for (int mu = 1; mu < 4; mu++)
    for (int nu = 0; nu < mu; nu++) {
      const double *f = F[(mu * (mu - 1)) / 2 + nu];
      double t = retr_mul(lo, f, f);
      if (mu < 3) es += t; else et += t;
    }
  const double vi = -1.0 / (double)lo->vol;
  /* f[1][0] f[3][2] - f[2][0] f[3][1] + f[2][1] f[3][0] */
  double a = retr_mul(lo, F[0], F[5]), b = retr_mul(lo, F[1], F[4]), c = retr_mul(lo, F[2], F[3]);
  out[0] = vi * es;
  out[1] = vi * et;
  out[2] = -1.0 / (4.0 * 3.14159265358979323846 * 3.14159265358979323846) * (a - b + c);
  for (int i = 0; i < 6; i++) free(F[i]);
}

/* wline (gaugeUtils.nim:1079-1112): volume- and colour-averaged trace of the path product */
void qo_wline(const qo_layout *lo, const double *g, const int *path, int n, double out[2]) {
  double sr = 0, si = 0;
#pragma omp parallel for reduction(+ : sr, si) schedule(static)
  for (int x = 0; x < lo->vol; x++) {
    double m[18];
    path_prod(lo, g, x, path, n, m);
    sr += RE(m,0,0) + RE(m,1,1) + RE(m,2,2);
    si += IM(m,0,0) + IM(m,1,1) + IM(m,2,2);
  }
  const double fac = 1.0 / ((double)lo->vol * 3.0);
  out[0] = sr * fac; out[1] = si * fac;
}

/* ------------------------------------------------------------------ */
/* gauge actions with rectangle / adjoint-plaquette terms and their     */
/* derivatives (completes SURVEY 8 row a14)                             */
/* ------------------------------------------------------------------ */
/* the 18 rectangle "staples" of link (x,mu): for every nu != mu and both signs of nu the three
 * 5-link paths from x to x+mu that close a 1x2 or 2x1 rectangle with U_mu(x)^+.  This is what
 * the rect part of gaugeActionDeriv accumulates through its stf/stu/ru fields
 * (gaugeAction.nim:205-241,275-331): f[mu] += cr * (every rectangle through the link, opened). */
static void rect_staples(const qo_layout *lo, const double *g, int x, int mu, double *acc, double cr) {
  const int a = mu + 1;
  double m[18];
  for (int nu = 0; nu < 4; nu++) {
    if (nu == mu) continue;
    for (int sgn = -1; sgn <= 1; sgn += 2) {
      const int b = sgn * (nu + 1);
      const int p1[5] = {b, b, a, -b, -b};      /* 1 (mu) x 2 (nu): the link is the short side */
      const int p2[5] = {b, a, a, -b, -a};      /* 2 (mu) x 1 (nu): the link is the first long-side link */
      const int p3[5] = {-a, b, a, a, -b};      /*                  the link is the second one */
      path_prod(lo, g, x, p1, 5, m); m_axpy(acc, cr, m);
      path_prod(lo, g, x, p2, 5, m); m_axpy(acc, cr, m);
      path_prod(lo, g, x, p3, 5, m); m_axpy(acc, cr, m);
    }
  }
}

/* gaugeActionDeriv with c.plaq and c.rect (gaugeAction.nim:148-332) */
void qo_gauge_deriv_rect(const qo_layout *lo, const double *g, double *f, double cplaq, double crect) {
  qo_gauge_deriv(lo, g, f, cplaq);
  if (crect == 0.0) return;
  const double cr = crect / 3.0;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < lo->vol * 4; i++) rect_staples(lo, g, i / 4, i % 4, &f[(size_t)i * 18], cr);
}

/* gaugeADeriv (gaugeAction.nim:683-740): every plaquette staple S of link U weighted by
 * cp + ca*tr(S^+ U), cp = c.plaq/nc, ca = 2 c.adjplaq/nc^2 */
void qo_gauge_deriv_adj(const qo_layout *lo, const double *g, double *f, double cplaq, double cadj) {
  const double cp = cplaq / 3.0, ca = 2.0 * cadj / 9.0;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < lo->vol * 4; i++) {
    const int x = i / 4, mu = i % 4, a = mu + 1;
    double acc[18], s[18];
    m_zero(acc);
    const double *U = GLINK(g, x, mu);
    for (int nu = 0; nu < 4; nu++) {
      if (nu == mu) continue;
      for (int sgn = -1; sgn <= 1; sgn += 2) {
        const int b = sgn * (nu + 1);
        const int p[3] = {b, a, -b};
        path_prod(lo, g, x, p, 3, s);
        /* t = dot(S, U) = tr(S^+ U) */
        double tr = 0, ti = 0;
        for (int k = 0; k < 9; k++) { tr += s[2 * k] * U[2 * k] + s[2 * k + 1] * U[2 * k + 1]; ti += s[2 * k] * U[2 * k + 1] - s[2 * k + 1] * U[2 * k]; }
        const double wr = cp + ca * tr, wi = ca * ti;
        for (int k = 0; k < 9; k++) { acc[2 * k] += wr * s[2 * k] - wi * s[2 * k + 1]; acc[2 * k + 1] += wr * s[2 * k + 1] + wi * s[2 * k]; }
      }
    }
    m_copy(&f[(size_t)i * 18], acc);
  }
}

/* force = TAH(U f^+) for the general actions: kind 0: plaq+rect (gaugeForce, gaugeAction.nim:334-338),
 * kind 1: plaq+adjplaq (forceA, :742-747) */
void qo_gauge_force_general(const qo_layout *lo, const double *g, double *f, double cplaq, double c2, int kind) {
  if (kind == 0) qo_gauge_deriv_rect(lo, g, f, cplaq, c2);
  else qo_gauge_deriv_adj(lo, g, f, cplaq, c2);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < lo->vol * 4; i++) {
    double s[18];
    m_mul_na(s, &g[(size_t)i * 18], &f[(size_t)i * 18]);
    qo_projectTAH(&f[(size_t)i * 18], s);
  }
}

/* gaugeAction1 (gaugeAction.nim:61-142): -(1/nc) (c.plaq sum ReTr P + c.rect sum ReTr R), 6 plaquette
 * and 12 rectangle types per site; actionA (:614-681): c.plaq (a0 - sum ReTr P/nc) + c.adjplaq (a0 - sum |tr P|^2/nc^2) */
double qo_gauge_action(const qo_layout *lo, const double *g, double cplaq, double c2, int kind) {
  double sp = 0, sr = 0, sa = 0;
#pragma omp parallel for reduction(+ : sp, sr, sa) schedule(static)
  for (int x = 0; x < lo->vol; x++) {
    double m[18];
    for (int mu = 1; mu < 4; mu++)
      for (int nu = 0; nu < mu; nu++) {
        const int a = mu + 1, b = nu + 1;
        const int pl[4] = {a, b, -a, -b};
        path_prod(lo, g, x, pl, 4, m);
        const double tr = RE(m,0,0) + RE(m,1,1) + RE(m,2,2), ti = IM(m,0,0) + IM(m,1,1) + IM(m,2,2);
        sp += tr;
        sa += tr * tr + ti * ti;
        if (kind == 0 && c2 != 0.0) {
          const int r1[6] = {a, a, b, -a, -a, -b}, r2[6] = {a, b, b, -a, -b, -b};
          path_prod(lo, g, x, r1, 6, m); sr += RE(m,0,0) + RE(m,1,1) + RE(m,2,2);
          path_prod(lo, g, x, r2, 6, m); sr += RE(m,0,0) + RE(m,1,1) + RE(m,2,2);
        }
      }
  }
  if (kind == 0) return (-1.0 / 3.0) * (cplaq * sp + c2 * sr);
  const double a0 = 0.5 * 12.0 * (double)lo->vol;
  return cplaq * (a0 - sp / 3.0) + c2 * (a0 - sa / 9.0);
}

/* gaugeFlow with an action choice (src/flow/flow.nim:22-90): kind 0 "Wilson"/"rect", kind 1 "adj" */
void qo_wflow_general(const qo_layout *lo, double *g, int nsteps, double eps, double cplaq, double c2, int kind) {
  size_t n = (size_t)lo->vol * 4 * 18;
  double *p = (double *)malloc(sizeof(double) * n);
  double *f = (double *)malloc(sizeof(double) * n);
  const double epsnc = eps * 3.0;
  for (int step = 0; step < nsteps; step++)
    for (int stage = 0; stage < 3; stage++) {
      qo_gauge_force_general(lo, g, f, cplaq, c2, kind);
#pragma omp parallel for schedule(static)
      for (int i = 0; i < lo->vol * 4; i++) {
        double v[18], e[18], t[18];
        double *gi = &g[(size_t)i * 18], *fi = &f[(size_t)i * 18], *pi = &p[(size_t)i * 18];
        if (stage == 0) for (int k = 0; k < 18; k++) v[k] = (-1.0 / 4.0) * epsnc * fi[k];
        else if (stage == 1) for (int k = 0; k < 18; k++) v[k] = (-8.0 / 9.0) * epsnc * fi[k] + (-17.0 / 9.0) * pi[k];
        else for (int k = 0; k < 18; k++) v[k] = (-3.0 / 4.0) * epsnc * fi[k] - pi[k];
        qo_exp(e, v);
        m_mul(t, e, gi);
        if (stage < 2) m_copy(pi, v);
        m_copy(gi, t);
      }
    }
  free(p); free(f);
}

/* ------------------------------------------------------------------ */
/* fermion-force outer product                       */
/* ------------------------------------------------------------------ */
/* f[mu](s) (+)= scale(parity of s) * x(s) (x) x(s+mu)^+ :
 *   stagDeriv   (stagD.nim:634-664): +1 on even sites, -1 on odd sites, accumulate (then s.rephase f)
 *   fforce loop (stagg_pv_hmc/staghmc_spv.nim:831-854): the same scale on both parities,
 *                f := for the first field, f += afterwards */
void qo_stag_outer(const qo_layout *lo, double *f, const double *x, double scale_even, double scale_odd, int accumulate) {
  qo_stag_outer_hop(lo, f, x, scale_even, scale_odd, accumulate, 1);
}
/* hop = 3: the Naik part of the HISQ fermion force, p(x) (x) p(x+3mu)^+ (src/examples/hisqhmc.nim:496-516) */
void qo_stag_outer_hop(const qo_layout *lo, double *f, const double *x, double scale_even, double scale_odd, int accumulate, int hop) {
#pragma omp parallel for schedule(static)
  for (int s = 0; s < lo->vol; s++) {
    const double sc = s < lo->volh ? scale_even : scale_odd;
    for (int mu = 0; mu < 4; mu++) {
      int nb = s;
      for (int h = 0; h < hop; h++) nb = lo->nb[mu][0][nb];
      const double *a = &x[6 * (size_t)s], *b = &x[6 * (size_t)nb];
      double *m = &f[((size_t)s * 4 + mu) * 18];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
          /* x_i * conj(y_j) */
          const double re = a[2 * i] * b[2 * j] + a[2 * i + 1] * b[2 * j + 1];
          const double im = a[2 * i + 1] * b[2 * j] - a[2 * i] * b[2 * j + 1];
          if (accumulate) { RE(m,i,j) += sc * re; IM(m,i,j) += sc * im; }
          else { RE(m,i,j) = sc * re; IM(m,i,j) = sc * im; }
        }
    }
  }
}

/* ------------------------------------------------------------------ */
/* link smearing: HISQ (fat7 + Naik) and nHYP     */
/* ------------------------------------------------------------------ */
/* single-matrix fields: double m[vol][18]; gauge fields [vol][4][18] are addressed through
 * (base pointer, stride in doubles between sites). */
typedef struct { const double *p; size_t stride; } mview;
static inline const double *MV(mview v, int s) { return v.p + (size_t)s * v.stride; }
static inline mview gauge_view(const double *g, int mu) { mview v = {g + (size_t)mu * 18, 72}; return v; }
static inline mview field_view(const double *f) { mview v = {f, 18}; return v; }

/* the generic staple of computeGenStaple (fat7l.nim:24-75) = symStaple (smearutil.nim:3-20):
 *   st(x) = A(x) B(x+nu) A(x+mu)^+  +  A(x-nu)^+ B(x-nu) A(x-nu+mu)
 * A: side links (direction nu), B: middle "link" field (direction mu) */
static void gen_staple_site(const qo_layout *lo, mview A, mview B, int mu, int nu, int x, double *st) {
  double t[18], u[18];
  const int xpn = lo->nb[nu][0][x], xpm = lo->nb[mu][0][x], xmn = lo->nb[nu][1][x];
  m_mul_na(t, MV(B, xpn), MV(A, xpm));
  m_mul(st, MV(A, x), t);
  m_mul_an(t, MV(A, xmn), MV(B, xmn));
  m_mul(u, t, MV(A, lo->nb[mu][0][xmn]));
  for (int k = 0; k < 18; k++) st[k] += u[k];
}
/* staple field (optional) and acc += coef*staple */
static void gen_staple(const qo_layout *lo, double *staple, double *acc, size_t acc_stride, double coef,
                       mview A, mview B, int mu, int nu) {
#pragma omp parallel for schedule(static)
  for (int x = 0; x < lo->vol; x++) {
    double st[18];
    gen_staple_site(lo, A, B, mu, nu, x, st);
    if (staple) m_copy(&staple[(size_t)x * 18], st);
    if (acc) m_axpy(&acc[(size_t)x * acc_stride], coef, st);
  }
}

/* makeImpLinks (fat7l.nim:77-161).  fl, ll: gauge-format outputs; gf, gfLong inputs */
void qo_fat7(const qo_layout *lo, double *fl, const double *gf, const double coef[5], double *ll,
             const double *gfLong, double naik) {
  const double c3 = coef[1], c5 = coef[2], c7 = coef[3], cL = coef[4];
  const double c1 = coef[0] - 6.0 * cL;          /* Lepage fix-up, fat7l.nim:104-105 */
  const int have5 = (c5 != 0.0) || (c7 != 0.0) || (cL != 0.0);
  const int have3 = (c3 != 0.0) || have5;
  size_t n = (size_t)lo->vol * 18;
  double *staple = (double *)malloc(sizeof(double) * n), *temp = (double *)malloc(sizeof(double) * n);
  for (int dir = 0; dir < 4; dir++) {
    for (int x = 0; x < lo->vol; x++) m_scale(&fl[((size_t)x * 4 + dir) * 18], c1, &gf[((size_t)x * 4 + dir) * 18]);
    if (!have3) continue;
    double *acc = fl + (size_t)dir * 18;
    for (int nu = 0; nu < 4; nu++) {
      if (nu == dir) continue;
      gen_staple(lo, staple, acc, 72, c3, gauge_view(gf, nu), gauge_view(gf, dir), dir, nu);
      if (cL != 0.0) gen_staple(lo, NULL, acc, 72, cL, gauge_view(gf, nu), field_view(staple), dir, nu);
      if (c5 != 0.0 || c7 != 0.0)
        for (int rho = 0; rho < 4; rho++) {
          if (rho == dir || rho == nu) continue;
          gen_staple(lo, temp, acc, 72, c5, gauge_view(gf, rho), field_view(staple), dir, rho);
          if (c7 != 0.0)
            for (int sig = 0; sig < 4; sig++) {
              if (sig == dir || sig == nu || sig == rho) continue;
              gen_staple(lo, NULL, acc, 72, c7, gauge_view(gf, sig), field_view(temp), dir, sig);
            }
        }
    }
  }
  if (naik != 0.0 && ll) {
    /* ll[dir](x) = naik * U(x) U(x+dir) U(x+2dir)   (fat7l.nim:146-156) */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < lo->vol * 4; i++) {
      const int x = i / 4, dir = i % 4;
      const int x1 = lo->nb[dir][0][x], x2 = lo->nb[dir][0][x1];
      double t[18], u[18];
      m_mul(t, GLINK(gfLong, x1, dir), GLINK(gfLong, x2, dir));
      m_mul(u, GLINK(gfLong, x, dir), t);
      m_scale(&ll[(size_t)i * 18], naik, u);
    }
  }
  free(staple); free(temp);
}

/* HisqCoefs.init + smear (physics/hisqLinks.nim:9-43) */
void qo_hisq_smear(const qo_layout *lo, const double *g, double *fl, double *ll) {
  const double f7lf = 0.0, naik = 1.0;
  /* setHisqFat7 (hisqLinks.nim:9-14) */
  const double c_first[5] = {(1.0 + 3.0 * f7lf + 0.0) / 8.0, -1.0 / 16.0, 1.0 / 64.0, -1.0 / 384.0, -f7lf / 16.0};
  const double f2 = 2.0 - f7lf;
  const double c_second[5] = {(1.0 + 3.0 * f2 + naik) / 8.0, -1.0 / 16.0, 1.0 / 64.0, -1.0 / 384.0, -f2 / 16.0};
  size_t n = (size_t)lo->vol * 72;
  double *t1 = (double *)malloc(sizeof(double) * n), *t2 = (double *)malloc(sizeof(double) * n);
  qo_fat7(lo, t1, g, c_first, NULL, g, 0.0);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < lo->vol * 4; i++) qo_projectU(&t2[(size_t)i * 18], &t1[(size_t)i * 18]);
  qo_fat7(lo, fl, t2, c_second, ll, t2, -naik / 24.0);
  free(t1); free(t2);
}

/* nHYP smearing, forward part of smearGetForce (gauge/hypsmear.nim:49-144):
 *   l1[mu,nu] = P( (1-a1) U_mu + a1/2 staple_nu(U_nu; U_mu) )
 *   l2[mu,nu] = P( (1-a2) U_mu + a2/4 sum_{a != mu,nu} staple_a(l1[a,b]; l1[mu,b]) ),  b = 6-mu-nu-a
 *   fl[mu]    = P( (1-a3) U_mu + a3/6 sum_{nu != mu} staple_nu(l2[nu,mu]; l2[mu,nu]) ) */
void qo_nhyp_smear(const qo_layout *lo, const double *g, double *fl, double a1, double a2, double a3) {
  size_t n = (size_t)lo->vol * 18;
  double *l1[4][4], *l2[4][4], *tmp = (double *)malloc(sizeof(double) * n);
  for (int mu = 0; mu < 4; mu++) for (int nu = 0; nu < 4; nu++) { l1[mu][nu] = l2[mu][nu] = NULL; if (mu != nu) { l1[mu][nu] = (double *)malloc(sizeof(double) * n); l2[mu][nu] = (double *)malloc(sizeof(double) * n); } }
  const double alp1 = a1 / 2.0, alp2 = a2 / 4.0, alp3 = a3 / 6.0;
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) {
      if (nu == mu) continue;
      for (int x = 0; x < lo->vol; x++) m_scale(&tmp[(size_t)x * 18], 1 - a1, GLINK(g, x, mu));
      gen_staple(lo, NULL, tmp, 18, alp1, gauge_view(g, nu), gauge_view(g, mu), mu, nu);
#pragma omp parallel for schedule(static)
      for (int x = 0; x < lo->vol; x++) qo_projectU(&l1[mu][nu][(size_t)x * 18], &tmp[(size_t)x * 18]);
    }
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) {
      if (nu == mu) continue;
      for (int x = 0; x < lo->vol; x++) m_scale(&tmp[(size_t)x * 18], 1 - a2, GLINK(g, x, mu));
      for (int a = 0; a < 4; a++) {
        if (a == mu || a == nu) continue;
        const int b = 1 + 2 + 3 - mu - nu - a;
        gen_staple(lo, NULL, tmp, 18, alp2, field_view(l1[a][b]), field_view(l1[mu][b]), mu, a);
      }
#pragma omp parallel for schedule(static)
      for (int x = 0; x < lo->vol; x++) qo_projectU(&l2[mu][nu][(size_t)x * 18], &tmp[(size_t)x * 18]);
    }
  for (int mu = 0; mu < 4; mu++) {
    for (int x = 0; x < lo->vol; x++) m_scale(&tmp[(size_t)x * 18], 1 - a3, GLINK(g, x, mu));
    for (int nu = 0; nu < 4; nu++) {
      if (nu == mu) continue;
      gen_staple(lo, NULL, tmp, 18, alp3, field_view(l2[nu][mu]), field_view(l2[mu][nu]), mu, nu);
    }
#pragma omp parallel for schedule(static)
    for (int x = 0; x < lo->vol; x++) qo_projectU(&fl[((size_t)x * 4 + mu) * 18], &tmp[(size_t)x * 18]);
  }
  for (int mu = 0; mu < 4; mu++) for (int nu = 0; nu < 4; nu++) if (mu != nu) { free(l1[mu][nu]); free(l2[mu][nu]); }
  free(tmp);
}

/* ------------------------------------------------------------------ */
/* nHYP smeared-force chain (backward direction)     */
/* ------------------------------------------------------------------ */
#include <complex.h>
typedef double complex cx_t;
static inline void to_cx(cx_t *c, const double *m) { for (int i = 0; i < 9; i++) c[i] = m[2 * i] + I * m[2 * i + 1]; }
static inline void from_cx(double *m, const cx_t *c) { for (int i = 0; i < 9; i++) { m[2 * i] = creal(c[i]); m[2 * i + 1] = cimag(c[i]); } }
static inline void cx_mul(cx_t *r, const cx_t *a, const cx_t *b) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    cx_t s = 0; for (int k = 0; k < 3; k++) s += a[3 * i + k] * b[3 * k + j];
    r[3 * i + j] = s;
  }
}
/* adjugate, nc = 3 (maths/projUderiv.nim:8-38) */
static void cx_adjugate(cx_t *r, const cx_t *x) {
  r[0] = x[4] * x[8] - x[5] * x[7]; r[1] = x[7] * x[2] - x[8] * x[1]; r[2] = x[1] * x[5] - x[2] * x[4];
  r[3] = x[5] * x[6] - x[3] * x[8]; r[4] = x[8] * x[0] - x[6] * x[2]; r[5] = x[2] * x[3] - x[0] * x[5];
  r[6] = x[3] * x[7] - x[4] * x[6]; r[7] = x[6] * x[1] - x[7] * x[0]; r[8] = x[0] * x[4] - x[1] * x[3];
}
/* inverse, nc = 3, c = 1 (maths/matinv.nim:90-115) */
static void cx_inverse(cx_t *r, const cx_t *x) {
  const cx_t det0 = x[0] * x[4] - x[1] * x[3], det1 = x[2] * x[3] - x[0] * x[5], det2 = x[1] * x[5] - x[2] * x[4];
  const cx_t det = det0 * x[8] + det1 * x[7] + det2 * x[6];
  const cx_t idet = 1.0 / det;
  r[0] = idet * (x[4] * x[8] - x[5] * x[7]); r[1] = idet * (x[7] * x[2] - x[8] * x[1]); r[2] = idet * det2;
  r[3] = idet * (x[5] * x[6] - x[3] * x[8]); r[4] = idet * (x[8] * x[0] - x[6] * x[2]); r[5] = idet * det1;
  r[6] = idet * (x[3] * x[7] - x[4] * x[6]); r[7] = idet * (x[6] * x[1] - x[7] * x[0]); r[8] = idet * det0;
}
/* sylsolve, nc = 3: solves A X + X A = C (maths/projUderiv.nim:96-147) */
static void cx_sylsolve(cx_t *x, const cx_t *a, const cx_t *c) {
  cx_t ad[9], ac[9], ca[9], aca[9], adc[9], cad[9], adcad[9];
  cx_adjugate(ad, a);
  const cx_t t = a[0] + a[4] + a[8], s = ad[0] + ad[4] + ad[8];
  const cx_t r = a[0] * ad[0] + a[1] * ad[3] + a[2] * ad[6];
  cx_mul(ac, a, c); cx_mul(ca, c, a); cx_mul(aca, ac, a);
  cx_mul(adc, ad, c); cx_mul(cad, c, ad); cx_mul(adcad, adc, ad);
  const cx_t c2 = 1.0 / (2.0 * (s * t - r)), c0 = c2 * (s + t * t), c1 = c2 * (t / r), c4 = c2 * t;
  for (int i = 0; i < 9; i++) x[i] = c0 * c[i] + c1 * adcad[i] + c2 * (aca[i] - adc[i] - cad[i]) - c4 * (ac[i] + ca[i]);
}
/* projectUderiv(r, u, x, chain) (maths/matrixFunctions.nim:329-351):
 * F with  d Re tr(C^+ U(X)) = Re tr(dX^+ F),  U = X (X^+X)^{-1/2}.  r may alias chain. */
void qo_projectUderiv(double *r, const double *u, const double *x, const double *chain) {
  double t[18], zr[18];
  m_mul_an(t, x, x);
  m_add_diag(t, 1e-20);
  rsqrtPHM3(zr, t);
  cx_t z[9], y[9], rr[9], cc[9], uu[9], xx[9], t1[9], t2[9];
  to_cx(z, zr); to_cx(cc, chain); to_cx(uu, u); to_cx(xx, x);
  cx_inverse(y, z);
  cx_mul(rr, cc, z);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {       /* t1 = u^+ r */
    cx_t s = 0; for (int k = 0; k < 3; k++) s += conj(uu[3 * k + i]) * rr[3 * k + j];
    t1[3 * i + j] = s;
  }
  cx_sylsolve(t2, y, t1);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t1[3 * i + j] = t2[3 * i + j] + conj(t2[3 * j + i]);
  cx_mul(t2, xx, t1);
  for (int i = 0; i < 9; i++) rr[i] -= t2[i];
  from_cx(r, rr);
}

/* symStapleDeriv (gauge/smearutil.nim:22-50), gathered per site:
 *   f1(x) += g2(x) g1(x+mu) c(x+nu)^+ + c(x) g1(x+mu) g2(x+nu)^+
 *          + [g2^+ g1 c(+nu) + c^+ g1 g2(+nu)](x-mu)
 *   f2(x) += g1(x) c(x+nu) g1(x+mu)^+ + [g1^+ c g1(+mu)](x-nu)
 * g1: side links (direction nu), g2: middle links (direction mu), c: chain of the staple sum */
static void staple_deriv_c(const qo_layout *lo, double *f1, size_t f1s, double *f2, size_t f2s,
                           mview g1, mview g2, const double *c, size_t cs, int mu, int nu, double coef) {
#pragma omp parallel for schedule(static)
  for (int x = 0; x < lo->vol; x++) {
    const int xpm = lo->nb[mu][0][x], xpn = lo->nb[nu][0][x], xmm = lo->nb[mu][1][x], xmn = lo->nb[nu][1][x];
    const int xmmpn = lo->nb[nu][0][xmm], xmnpm = lo->nb[mu][0][xmn];
    const double *C = c;
    double t[18], u[18], a1[18], a2[18];
#define CC(site) (&C[(size_t)(site) * cs])
    m_zero(a1); m_zero(a2);
    m_mul_na(t, MV(g1, xpm), CC(xpn)); m_mul(u, MV(g2, x), t); m_axpy(a1, 1.0, u);
    m_mul_na(t, MV(g1, xpm), MV(g2, xpn)); m_mul(u, CC(x), t); m_axpy(a1, 1.0, u);
    m_mul(t, MV(g1, xmm), CC(xmmpn)); m_mul_an(u, MV(g2, xmm), t); m_axpy(a1, 1.0, u);
    m_mul(t, MV(g1, xmm), MV(g2, xmmpn)); m_mul_an(u, CC(xmm), t); m_axpy(a1, 1.0, u);
    m_mul_na(t, CC(xpn), MV(g1, xpm)); m_mul(u, MV(g1, x), t); m_axpy(a2, 1.0, u);
    m_mul(t, CC(xmn), MV(g1, xmnpm)); m_mul_an(u, MV(g1, xmn), t); m_axpy(a2, 1.0, u);
    /* f1 and f2 may be the same gauge field (different mu) but never the same matrix */
    m_axpy(&f1[(size_t)x * f1s], coef, a1);
    m_axpy(&f2[(size_t)x * f2s], coef, a2);
  }
#undef CC
}
static void staple_deriv(const qo_layout *lo, double *f1, size_t f1s, double *f2, size_t f2s,
                         mview g1, mview g2, const double *c, int mu, int nu) {
  staple_deriv_c(lo, f1, f1s, f2, f2s, g1, g2, c, 18, mu, nu, 1.0);
}

/* smearGetForce + smearedForce(f, chain) with keepProj (gauge/hypsmear.nim:49-247).
 * g: thin links; chain: d/dV^+ of the action w.r.t. the smeared links; f (out, may alias chain):
 * d/dU^+ w.r.t. the thin links.  fl (optional): the smeared links. */
void qo_nhyp_force(const qo_layout *lo, const double *g, double *fl, double *f, const double *chain,
                   double a1, double a2, double a3) {
  const size_t n = (size_t)lo->vol * 18;
  const int V = lo->vol;
  double *l1x[4][4], *l1[4][4], *l2x[4][4], *l2[4][4], *fl1[4][4], *fl2[4][4], *flx[4], *fc[4];
#define NEWF() ((double *)calloc(n, sizeof(double)))
  for (int mu = 0; mu < 4; mu++) {
    flx[mu] = NEWF(); fc[mu] = NEWF();
    for (int nu = 0; nu < 4; nu++) {
      l1x[mu][nu] = l1[mu][nu] = l2x[mu][nu] = l2[mu][nu] = fl1[mu][nu] = fl2[mu][nu] = NULL;
      if (mu != nu) { l1x[mu][nu] = NEWF(); l1[mu][nu] = NEWF(); l2x[mu][nu] = NEWF(); l2[mu][nu] = NEWF(); fl1[mu][nu] = NEWF(); fl2[mu][nu] = NEWF(); }
    }
  }
  const double alp1 = a1 / 2.0, alp2 = a2 / 4.0, alp3 = a3 / 6.0, ma1 = 1 - a1, ma2 = 1 - a2, ma3 = 1 - a3;
  /* forward (hypsmear.nim:98-143) */
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) {
      if (nu == mu) continue;
      for (int x = 0; x < V; x++) m_scale(&l1x[mu][nu][(size_t)x * 18], ma1, GLINK(g, x, mu));
      gen_staple(lo, NULL, l1x[mu][nu], 18, alp1, gauge_view(g, nu), gauge_view(g, mu), mu, nu);
#pragma omp parallel for schedule(static)
      for (int x = 0; x < V; x++) qo_projectU(&l1[mu][nu][(size_t)x * 18], &l1x[mu][nu][(size_t)x * 18]);
    }
  for (int mu = 0; mu < 4; mu++)
    for (int nu = 0; nu < 4; nu++) {
      if (nu == mu) continue;
      for (int x = 0; x < V; x++) m_scale(&l2x[mu][nu][(size_t)x * 18], ma2, GLINK(g, x, mu));
      for (int a = 0; a < 4; a++) {
        if (a == mu || a == nu) continue;
        const int b = 1 + 2 + 3 - mu - nu - a;
        gen_staple(lo, NULL, l2x[mu][nu], 18, alp2, field_view(l1[a][b]), field_view(l1[mu][b]), mu, a);
      }
#pragma omp parallel for schedule(static)
      for (int x = 0; x < V; x++) qo_projectU(&l2[mu][nu][(size_t)x * 18], &l2x[mu][nu][(size_t)x * 18]);
    }
  for (int mu = 0; mu < 4; mu++) {
    for (int x = 0; x < V; x++) m_scale(&flx[mu][(size_t)x * 18], ma3, GLINK(g, x, mu));
    for (int nu = 0; nu < 4; nu++) {
      if (nu == mu) continue;
      gen_staple(lo, NULL, flx[mu], 18, alp3, field_view(l2[nu][mu]), field_view(l2[mu][nu]), mu, nu);
    }
    if (fl) {
#pragma omp parallel for schedule(static)
      for (int x = 0; x < V; x++) qo_projectU(&fl[((size_t)x * 4 + mu) * 18], &flx[mu][(size_t)x * 18]);
    }
  }
  if (f && chain) {
    /* backward (hypsmear.nim:146-245) */
    /* proj flx -> fl: fc <- chain (u recomputed from x, matrixFunctions.nim:353-357) */
    for (int mu = 0; mu < 4; mu++) {
#pragma omp parallel for schedule(static)
      for (int x = 0; x < V; x++) {
        double u[18];
        qo_projectU(u, &flx[mu][(size_t)x * 18]);
        qo_projectUderiv(&fc[mu][(size_t)x * 18], u, &flx[mu][(size_t)x * 18], &chain[((size_t)x * 4 + mu) * 18]);
      }
    }
    for (int mu = 0; mu < 4; mu++)
      for (int x = 0; x < V; x++) {
        m_scale(&f[((size_t)x * 4 + mu) * 18], ma3, &fc[mu][(size_t)x * 18]);
        for (int k = 0; k < 18; k++) fc[mu][(size_t)x * 18 + k] *= alp3;
      }
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) {
        if (nu == mu) continue;
        staple_deriv(lo, fl2[nu][mu], 18, fl2[mu][nu], 18, field_view(l2[nu][mu]), field_view(l2[mu][nu]), fc[mu], mu, nu);
      }
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) {
        if (nu == mu) continue;
#pragma omp parallel for schedule(static)
        for (int x = 0; x < V; x++) {
          double *p = &fl2[mu][nu][(size_t)x * 18];
          qo_projectUderiv(p, &l2[mu][nu][(size_t)x * 18], &l2x[mu][nu][(size_t)x * 18], p);
        }
      }
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) {
        if (nu == mu) continue;
        for (int x = 0; x < V; x++) {
          double *p = &fl2[mu][nu][(size_t)x * 18];
          m_axpy(&f[((size_t)x * 4 + mu) * 18], ma2, p);
          for (int k = 0; k < 18; k++) p[k] *= alp2;
        }
      }
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) {
        if (nu == mu) continue;
        for (int a = 0; a < 4; a++) {
          if (a == mu || a == nu) continue;
          const int b = 1 + 2 + 3 - mu - nu - a;
          staple_deriv(lo, fl1[a][b], 18, fl1[mu][b], 18, field_view(l1[a][b]), field_view(l1[mu][b]), fl2[mu][nu], mu, a);
        }
      }
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) {
        if (nu == mu) continue;
#pragma omp parallel for schedule(static)
        for (int x = 0; x < V; x++) {
          double *p = &fl1[mu][nu][(size_t)x * 18];
          qo_projectUderiv(p, &l1[mu][nu][(size_t)x * 18], &l1x[mu][nu][(size_t)x * 18], p);
        }
      }
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) {
        if (nu == mu) continue;
        for (int x = 0; x < V; x++) {
          double *p = &fl1[mu][nu][(size_t)x * 18];
          m_axpy(&f[((size_t)x * 4 + mu) * 18], ma1, p);
          for (int k = 0; k < 18; k++) p[k] *= alp1;
        }
      }
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) {
        if (nu == mu) continue;
        staple_deriv(lo, f + (size_t)nu * 18, 72, f + (size_t)mu * 18, 72, gauge_view(g, nu), gauge_view(g, mu), fl1[mu][nu], mu, nu);
      }
  }
  for (int mu = 0; mu < 4; mu++) {
    free(flx[mu]); free(fc[mu]);
    for (int nu = 0; nu < 4; nu++) if (mu != nu) { free(l1x[mu][nu]); free(l1[mu][nu]); free(l2x[mu][nu]); free(l2[mu][nu]); free(fl1[mu][nu]); free(fl2[mu][nu]); }
  }
#undef NEWF
}

/* projTAH(f, g) of the fork (stagg_pv_hmc/staghmc_spv_gforce.nim:256-291):
 * adj = 0 ("no_adj", matter):  f <- TAH(f g^+);  adj = 1 ("adj", gauge):  f <- TAH(g f^+) */
void qo_force_projTAH(const qo_layout *lo, double *f, const double *g, int adj) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < lo->vol * 4; i++) {
    double s[18];
    if (adj) m_mul_na(s, &g[(size_t)i * 18], &f[(size_t)i * 18]);
    else m_mul_na(s, &f[(size_t)i * 18], &g[(size_t)i * 18]);
    qo_projectTAH(&f[(size_t)i * 18], s);
  }
}

/* MD gauge update mdt (src/examples/staghmc_sh.nim:429-435): g[mu][s] := exp(t p[mu][s]) g[mu][s] */
void qo_gauge_exp_update(const qo_layout *lo, double *g, const double *p, double t) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < lo->vol * 4; i++) {
    double tp[18], e[18], r[18];
    m_scale(tp, t, &p[(size_t)i * 18]);
    qo_exp(e, tp);
    m_mul(r, e, &g[(size_t)i * 18]);
    m_copy(&g[(size_t)i * 18], r);
  }
}
/* g.projectSU on a gauge field (gaugeUtils.nim:1333-1334), the "reunit" of the HMC examples */
void qo_gauge_projectSU(const qo_layout *lo, double *g) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < lo->vol * 4; i++) qo_projectSU(&g[(size_t)i * 18], &g[(size_t)i * 18]);
}

/* ------------------------------------------------------------------ */
/* HISQ force: reverse of makeImpLinks and of HisqCoefs.smear           */
/* (gauge/fat7lderiv.nim, gauge/hisqsmear.nim:16-90)                    */
/* ------------------------------------------------------------------ */
/* d(gauge-format, accumulated): derivative w.r.t. gf of  sum Re tr(cfl^+ fl(gf)) + sum Re tr(cll^+ ll(gf)),
 * fl, ll = makeImpLinks(gf, coef, naik) as in qo_fat7.  Reverse accumulation over the same staple graph:
 * every generic staple S(A; B) is differentiated by symStapleDeriv (staple_deriv_c above). */
static void fat7_deriv(const qo_layout *lo, double *d, const double *gf, const double *cfl, const double coef[5],
                       const double *cll, double naik) {
  const double c3 = coef[1], c5 = coef[2], c7 = coef[3], cL = coef[4];
  const double c1 = coef[0] - 6.0 * cL;
  const int have5 = (c5 != 0.0) || (c7 != 0.0) || (cL != 0.0);
  const int have3 = (c3 != 0.0) || have5;
  const size_t n = (size_t)lo->vol * 18;
  double *st1 = (double *)malloc(sizeof(double) * n), *tmp = (double *)malloc(sizeof(double) * n);
  double *ast1 = (double *)malloc(sizeof(double) * n), *atmp = (double *)malloc(sizeof(double) * n);
  for (int dir = 0; dir < 4; dir++) {
    const double *ch = cfl + (size_t)dir * 18;                      /* chain of fl[dir], stride 72 */
    for (int x = 0; x < lo->vol; x++) m_axpy(&d[((size_t)x * 4 + dir) * 18], c1, &ch[(size_t)x * 72]);
    if (!have3) continue;
    for (int nu = 0; nu < 4; nu++) {
      if (nu == dir) continue;
      /* forward: st1 = S(gf_nu; gf_dir) */
      gen_staple(lo, st1, NULL, 0, 0.0, gauge_view(gf, nu), gauge_view(gf, dir), dir, nu);
      /* adjoint of st1 starts with the direct term c3 * chain */
      for (int x = 0; x < lo->vol; x++) m_scale(&ast1[(size_t)x * 18], c3, &ch[(size_t)x * 72]);
      if (cL != 0.0)   /* fl += cL S(gf_nu; st1) */
        staple_deriv_c(lo, d + (size_t)nu * 18, 72, ast1, 18, gauge_view(gf, nu), field_view(st1), ch, 72, dir, nu, cL);
      if (c5 != 0.0 || c7 != 0.0)
        for (int rho = 0; rho < 4; rho++) {
          if (rho == dir || rho == nu) continue;
          gen_staple(lo, tmp, NULL, 0, 0.0, gauge_view(gf, rho), field_view(st1), dir, rho);     /* tmp = S(gf_rho; st1) */
          for (int x = 0; x < lo->vol; x++) m_scale(&atmp[(size_t)x * 18], c5, &ch[(size_t)x * 72]);
          if (c7 != 0.0)
            for (int sig = 0; sig < 4; sig++) {
              if (sig == dir || sig == nu || sig == rho) continue;
              staple_deriv_c(lo, d + (size_t)sig * 18, 72, atmp, 18, gauge_view(gf, sig), field_view(tmp), ch, 72, dir, sig, c7);
            }
          staple_deriv_c(lo, d + (size_t)rho * 18, 72, ast1, 18, gauge_view(gf, rho), field_view(st1), atmp, 18, dir, rho, 1.0);
        }
      staple_deriv_c(lo, d + (size_t)nu * 18, 72, d + (size_t)dir * 18, 72, gauge_view(gf, nu), gauge_view(gf, dir), ast1, 18, dir, nu, 1.0);
    }
  }
  if (naik != 0.0 && cll) {
    /* ll[dir](x) = naik U(x) U(x+d) U(x+2d): the three places a link occupies */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < lo->vol * 4; i++) {
      const int x = i / 4, dir = i % 4;
      const int xp = lo->nb[dir][0][x], xpp = lo->nb[dir][0][xp], xm = lo->nb[dir][1][x], xmm = lo->nb[dir][1][xm];
      double t[18], u[18], acc[18];
      /* first:  C(x) U(x+2d)^+ U(x+d)^+ */
      m_mul_na(t, GLINK(cll, x, dir), GLINK(gf, xpp, dir)); m_mul_na(acc, t, GLINK(gf, xp, dir));
      /* middle: U(x-d)^+ C(x-d) U(x+d)^+ */
      m_mul_an(t, GLINK(gf, xm, dir), GLINK(cll, xm, dir)); m_mul_na(u, t, GLINK(gf, xp, dir)); m_axpy(acc, 1.0, u);
      /* last:   U(x-d)^+ U(x-2d)^+ C(x-2d) */
      m_mul_an(t, GLINK(gf, xmm, dir), GLINK(cll, xmm, dir)); m_mul_an(u, GLINK(gf, xm, dir), t); m_axpy(acc, 1.0, u);
      m_axpy(&d[(size_t)i * 18], naik, acc);
    }
  }
  free(st1); free(tmp); free(ast1); free(atmp);
}

/* derivative of makeImpLinks alone (test hook) */
void qo_fat7_deriv(const qo_layout *lo, double *d, const double *gf, const double *cfl, const double coef[5],
                   const double *cll, double naik) {
  memset(d, 0, sizeof(double) * (size_t)lo->vol * 72);
  fat7_deriv(lo, d, gf, cfl, coef, cll, naik);
}

/* HisqCoefs.smearGetForce -> smearedForce(dsdu, dsdsu, dsdsul) (gauge/hisqsmear.nim:55-90):
 * f = d/dU^+ of  sum Re tr(dsdsu^+ fl(U)) + sum Re tr(dsdsul^+ ll(U)),  fl, ll = HISQ links of U */
void qo_hisq_force(const qo_layout *lo, const double *g, const double *dsdsu, const double *dsdsul, double *f) {
  const double f7lf = 0.0, naik = 1.0, f2 = 2.0 - f7lf;
  const double c_first[5] = {(1.0 + 3.0 * f7lf + 0.0) / 8.0, -1.0 / 16.0, 1.0 / 64.0, -1.0 / 384.0, -f7lf / 16.0};
  const double c_second[5] = {(1.0 + 3.0 * f2 + naik) / 8.0, -1.0 / 16.0, 1.0 / 64.0, -1.0 / 384.0, -f2 / 16.0};
  const size_t n = (size_t)lo->vol * 72;
  double *v = (double *)malloc(sizeof(double) * n), *w = (double *)malloc(sizeof(double) * n), *t = (double *)calloc(n, sizeof(double));
  qo_fat7(lo, v, g, c_first, NULL, g, 0.0);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < lo->vol * 4; i++) qo_projectU(&w[(size_t)i * 18], &v[(size_t)i * 18]);
  fat7_deriv(lo, t, w, dsdsu, c_second, dsdsul, -naik / 24.0);          /* second fat7 + Naik */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < lo->vol * 4; i++) qo_projectUderiv(&t[(size_t)i * 18], &w[(size_t)i * 18], &v[(size_t)i * 18], &t[(size_t)i * 18]);
  memset(f, 0, sizeof(double) * n);
  fat7_deriv(lo, f, g, t, c_first, NULL, 0.0);                          /* first fat7 */
  free(v); free(w); free(t);
}

/* extended-precision twins of the two CGs: the yardstick of tests/parity_log.py (a file of its own: it restates no fp64 path of QEX) */
#include "qex_oracle_ext.inc"
