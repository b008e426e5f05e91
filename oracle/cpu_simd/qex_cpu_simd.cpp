// qex_cpu_simd.cpp -- CPU baseline of bench.py (test infrastructure, NOT part of the product; nothing under qex_amd/
// may use it).  It restates QEX's CPU hot path in the form SURVEY.md 8d / BASELINE.md 2 promise for the
// "reference CPU path timed beside the GPU": QEX's AoSoA data layout with V = 8 sites per SIMD vector and even-odd
// blocks, compiled -O3 -march=native -fopenmp.  The reference itself (Nim + QMP) cannot be built in this image.
//
//   layout        src/layout/qlayout.nim:110-185 with innerGeom = {1,2,2,2}: the lattice is cut into 8 sub-lattices
//                 (halves in y, z, t); lane j of outer site o is the site with o's coordinates inside sub-lattice j.
//                 Outer sites: x fastest, even block first (layoutIndexQ).  A hop that leaves the sub-lattice lands
//                 on the outer site at its opposite face with the lanes exchanged pairwise (the "perm" of
//                 src/layout/shifts.nim:114-129).
//   fields        colour vector [outer][colour][re|im][lane], links [outer][mu][row][col][re|im][lane]
//                 (src/field/fieldET.nim:18-22 over SimdD8)
//   stagD2        src/physics/stagD.nim:349-395     stagD2ee  :434-456     CG  src/solvers/cg.nim:174-214
//
// Checked against oracle/qex_oracle.c (the pinned restatement) in tests/test_cpu_simd.py: same operator to 1e-14,
// same CG residual history.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {
constexpr int V = 8;
typedef double vd __attribute__((vector_size(64), aligned(64)));   // 8 doubles = one AVX-512 register where available

struct Lay {
  int X[4], O[4];        // global and outer extents; inner geometry {1,2,2,2}
  int nOuter, nEven;
  std::vector<int> nbr;  // [dir 0..7 (2 mu + backward)][outer site in storage order] -> outer site
  std::vector<unsigned char> perm;  // same shape: lane exchange mask (bit k: exchange across inner dimension k)
  std::vector<int> outer_of_lex;    // outer lex (x fastest over O) -> storage index
};

inline int inner_of(int mu) { return mu == 0 ? 1 : 2; }

void lay_init(Lay &L, const int X[4]) {
  for (int i = 0; i < 4; i++) { L.X[i] = X[i]; L.O[i] = X[i] / inner_of(i); }
  L.nOuter = L.O[0] * L.O[1] * L.O[2] * L.O[3];
  L.nEven = L.nOuter / 2;
  L.outer_of_lex.assign(L.nOuter, 0);
  std::vector<int> lex_of(L.nOuter);
  int ne = 0, no = 0;
  for (int lex = 0; lex < L.nOuter; lex++) {
    int r = lex, par = 0;
    for (int i = 0; i < 4; i++) { par += r % L.O[i]; r /= L.O[i]; }
    const int idx = (par & 1) ? L.nEven + no++ : ne++;
    L.outer_of_lex[lex] = idx;
    lex_of[idx] = lex;
  }
  L.nbr.assign((size_t)8 * L.nOuter, 0);
  L.perm.assign((size_t)8 * L.nOuter, 0);
  for (int s = 0; s < L.nOuter; s++) {
    int c[4], r = lex_of[s];
    for (int i = 0; i < 4; i++) { c[i] = r % L.O[i]; r /= L.O[i]; }
    for (int d = 0; d < 8; d++) {
      const int mu = d >> 1, hop = (d & 1) ? -1 : 1;
      int y[4] = {c[0], c[1], c[2], c[3]};
      unsigned char pm = 0;
      y[mu] += hop;
      if (y[mu] < 0 || y[mu] >= L.O[mu]) {
        y[mu] = (y[mu] + L.O[mu]) % L.O[mu];
        if (inner_of(mu) == 2) pm = (unsigned char)(1 << (mu - 1));   // lanes: bit 0 = y half, 1 = z half, 2 = t half
      }
      const int lex = y[0] + L.O[0] * (y[1] + L.O[1] * (y[2] + L.O[2] * y[3]));
      L.nbr[(size_t)d * L.nOuter + s] = L.outer_of_lex[lex];
      L.perm[(size_t)d * L.nOuter + s] = pm;
    }
  }
}

inline vd lane_perm(vd a, unsigned m) {
  if (!m) return a;
  vd r;
  for (int j = 0; j < V; j++) r[j] = a[j ^ (int)m];
  return r;
}

struct CVec { vd c[3][2]; };            // colour vector of one outer site
struct Link { vd u[3][3][2]; };         // 3x3 complex matrix of one outer site and direction

// acc += U x   /   acc -= U^+ x
inline void mulacc(CVec &a, const Link &U, const CVec &x) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      a.c[i][0] += U.u[i][j][0] * x.c[j][0] - U.u[i][j][1] * x.c[j][1];
      a.c[i][1] += U.u[i][j][0] * x.c[j][1] + U.u[i][j][1] * x.c[j][0];
    }
}
inline void mulacc_adj_sub(CVec &a, const Link &U, const CVec &x) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      a.c[i][0] -= U.u[j][i][0] * x.c[j][0] + U.u[j][i][1] * x.c[j][1];
      a.c[i][1] -= U.u[j][i][0] * x.c[j][1] - U.u[j][i][1] * x.c[j][0];
    }
}
inline CVec perm_vec(const CVec &x, unsigned m) {
  if (!m) return x;
  CVec r;
  for (int i = 0; i < 3; i++) for (int k = 0; k < 2; k++) r.c[i][k] = lane_perm(x.c[i][k], m);
  return r;
}
inline Link perm_link(const Link &U, unsigned m) {
  if (!m) return U;
  Link r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 2; k++) r.u[i][j][k] = lane_perm(U.u[i][j][k], m);
  return r;
}

struct Ctx {
  Lay L;
  std::vector<Link> g;     // [outer][mu]
};

// r[par] = cb * xs + sum_mu [ U_mu(s) x(s+mu) - U_mu(s-mu)^+ x(s-mu) ]     (stagD2 with a = 0)
void dslash(const Ctx &C, CVec *r, const CVec *x, int par, double cb, const CVec *xs) {
  const Lay &L = C.L;
  const int s0 = par ? L.nEven : 0, s1 = par ? L.nOuter : L.nEven;
#pragma omp parallel for schedule(static)
  for (int s = s0; s < s1; s++) {
    CVec a;
    for (int i = 0; i < 3; i++) for (int k = 0; k < 2; k++) a.c[i][k] = cb != 0.0 ? cb * xs[s].c[i][k] : vd{0, 0, 0, 0, 0, 0, 0, 0};
    for (int mu = 0; mu < 4; mu++) {
      const size_t f = (size_t)(2 * mu) * L.nOuter + s, b = f + L.nOuter;
      const int sf = L.nbr[f], sb = L.nbr[b];
      mulacc(a, C.g[(size_t)s * 4 + mu], perm_vec(x[sf], L.perm[f]));
      const unsigned pb = L.perm[b];
      if (pb) mulacc_adj_sub(a, perm_link(C.g[(size_t)sb * 4 + mu], pb), perm_vec(x[sb], pb));
      else mulacc_adj_sub(a, C.g[(size_t)sb * 4 + mu], x[sb]);
    }
    r[s] = a;
  }
}

double norm2(const CVec *x, int s0, int s1) {
  double t = 0;
#pragma omp parallel for schedule(static) reduction(+ : t)
  for (int s = s0; s < s1; s++) {
    vd a = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; i++) for (int k = 0; k < 2; k++) a += x[s].c[i][k] * x[s].c[i][k];
    for (int j = 0; j < V; j++) t += a[j];
  }
  return t;
}
double redot(const CVec *x, const CVec *y, int s0, int s1) {
  double t = 0;
#pragma omp parallel for schedule(static) reduction(+ : t)
  for (int s = s0; s < s1; s++) {
    vd a = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; i++) for (int k = 0; k < 2; k++) a += x[s].c[i][k] * y[s].c[i][k];
    for (int j = 0; j < V; j++) t += a[j];
  }
  return t;
}

// host V=1 MILC even-odd arrays <-> AoSoA
struct Map { std::vector<int> outer, lane; };   // per V=1 even-odd index
void build_map(const Lay &L, Map &M) {
  const int vol = L.X[0] * L.X[1] * L.X[2] * L.X[3];
  M.outer.resize(vol); M.lane.resize(vol);
  for (int lex = 0; lex < vol; lex++) {
    int c[4], r = lex, par = 0;
    for (int i = 0; i < 4; i++) { c[i] = r % L.X[i]; r /= L.X[i]; par += c[i]; }
    const int idx = lex / 2 + ((par & 1) ? vol / 2 : 0);
    int o[4], lane = 0;
    for (int i = 0; i < 4; i++) { o[i] = c[i] % L.O[i]; if (i > 0 && c[i] >= L.O[i]) lane |= 1 << (i - 1); }
    const int olex = o[0] + L.O[0] * (o[1] + L.O[1] * (o[2] + L.O[2] * o[3]));
    M.outer[idx] = L.outer_of_lex[olex];
    M.lane[idx] = lane;
  }
}
}  // namespace

extern "C" {
struct qcs_handle { Ctx C; Map M; int vol; };

int qcs_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void qcs_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
// links: double[vol][4][3][3][2] in the V=1 even-odd host format, already carrying BC and staggered phases
qcs_handle *qcs_new(const int lat[4], const double *g) {
  for (int i = 1; i < 4; i++) if (lat[i] % 4) return nullptr;        // every sub-lattice keeps even extents
  if (lat[0] % 2) return nullptr;
  qcs_handle *h = new qcs_handle();
  lay_init(h->C.L, lat);
  build_map(h->C.L, h->M);
  h->vol = lat[0] * lat[1] * lat[2] * lat[3];
  h->C.g.resize((size_t)h->C.L.nOuter * 4);
  for (int idx = 0; idx < h->vol; idx++) {
    const int o = h->M.outer[idx], l = h->M.lane[idx];
    for (int mu = 0; mu < 4; mu++)
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 2; k++)
        h->C.g[(size_t)o * 4 + mu].u[i][j][k][l] = g[((((size_t)idx * 4 + mu) * 3 + i) * 3 + j) * 2 + k];
  }
  return h;
}
void qcs_free(qcs_handle *h) { delete h; }
// the site map this baseline lays its fields out by, per V=1 even-odd index: SIMD site = outer * 8 + lane
// (tests/test_simd_layout.py holds the product's restatement of LayoutQ against it: two independent implementations)
void qcs_site_map(const qcs_handle *h, int *simd_of_v1) {
  for (int idx = 0; idx < h->vol; idx++) simd_of_v1[idx] = h->M.outer[idx] * V + h->M.lane[idx];
}

static void to_simd(const qcs_handle *h, const double *v, std::vector<CVec> &x) {
  x.assign(h->C.L.nOuter, CVec());
  for (int idx = 0; idx < h->vol; idx++)
    for (int i = 0; i < 3; i++) for (int k = 0; k < 2; k++) x[h->M.outer[idx]].c[i][k][h->M.lane[idx]] = v[((size_t)idx * 3 + i) * 2 + k];
}
static void from_simd(const qcs_handle *h, const std::vector<CVec> &x, double *v) {
  for (int idx = 0; idx < h->vol; idx++)
    for (int i = 0; i < 3; i++) for (int k = 0; k < 2; k++) v[((size_t)idx * 3 + i) * 2 + k] = x[h->M.outer[idx]].c[i][k][h->M.lane[idx]];
}

// r[parity] = b*x + (2D) x, other parity of r untouched (r in/out, host format): stagD2 with a = 0
void qcs_stagD2(qcs_handle *h, double *r, const double *x, int parity, double b) {
  std::vector<CVec> X, R;
  to_simd(h, x, X);
  to_simd(h, r, R);
  for (int p = (parity == 2 ? 0 : parity); p <= (parity == 2 ? 1 : parity); p++) dslash(h->C, R.data(), X.data(), p, b, X.data());
  from_simd(h, R, r);
}

// solveXX on the even (par_even) or odd sites: 4(m^2 - D_eo D_oe) x = b from x = 0 (cg.nim:132-214);
// returns iterations; hist[k] = r2/b2 after iteration k.  Only the CG loop is timed (seconds_out).
int qcs_solveXX(qcs_handle *h, double *x_host, const double *b_host, double mass, double r2req, int maxits, int par_even,
                double *hist, int histcap, double *seconds_out) {
  const Lay &L = h->C.L;
  const int par = par_even ? 0 : 1;
  const int s0 = par ? L.nEven : 0, s1 = par ? L.nOuter : L.nEven;
  std::vector<CVec> b, x(L.nOuter, CVec()), r, p(L.nOuter, CVec()), Ap(L.nOuter, CVec()), t(L.nOuter, CVec());
  to_simd(h, b_host, b);
  r = b;
  const double m2 = mass * mass;
  const double b2 = norm2(b.data(), s0, s1);
  double r2 = b2, rzo = 1.0;
  const double r2stop = r2req * b2;
  int itn = 0;
  if (hist && histcap > 0) hist[0] = b2 != 0.0 ? 1.0 : 0.0;
#ifdef _OPENMP
  const double t0 = omp_get_wtime();
#endif
  while (itn < maxits && r2 > r2stop) {
    const double beta = r2 / rzo;
    const bool first = itn == 0;
#pragma omp parallel for schedule(static)
    for (int s = s0; s < s1; s++)
      for (int i = 0; i < 3; i++) for (int k = 0; k < 2; k++) p[s].c[i][k] = first ? r[s].c[i][k] : r[s].c[i][k] + beta * p[s].c[i][k];
    itn++;
    dslash(h->C, t.data(), p.data(), 1 - par, 0.0, nullptr);                 // stagDP onto the other parity
    // Ap = 4 m2 p - (2D)(2D) p : second sweep with the minus sign folded in
    dslash(h->C, Ap.data(), t.data(), par, 0.0, nullptr);
#pragma omp parallel for schedule(static)
    for (int s = s0; s < s1; s++)
      for (int i = 0; i < 3; i++) for (int k = 0; k < 2; k++) Ap[s].c[i][k] = 4.0 * m2 * p[s].c[i][k] - Ap[s].c[i][k];
    const double pAp = redot(p.data(), Ap.data(), s0, s1);
    const double alpha = r2 / pAp;
#pragma omp parallel for schedule(static)
    for (int s = s0; s < s1; s++)
      for (int i = 0; i < 3; i++) for (int k = 0; k < 2; k++) {
        x[s].c[i][k] += alpha * p[s].c[i][k];
        r[s].c[i][k] -= alpha * Ap[s].c[i][k];
      }
    rzo = r2;
    r2 = norm2(r.data(), s0, s1);
    if (hist && itn < histcap) hist[itn] = r2 / b2;
  }
#ifdef _OPENMP
  if (seconds_out) *seconds_out = omp_get_wtime() - t0;
#else
  if (seconds_out) *seconds_out = 0;
#endif
  if (x_host) from_simd(h, x, x_host);
  return itn;
}
}
