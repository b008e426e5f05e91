// Host check of the closed-form exp of csrc/su3.h (m3_exp_tah, the flow's opt-in) against the reference-algorithm exp
// (m3_exp: order-4 Taylor at v/2^20 + 20 squarings, matexp.nim) and against a long-double Taylor series:
//   hipcc -O2 -std=c++17 -I qex_amd/csrc tests/cpp/test_exp_ch.cpp -o /tmp/test_exp_ch   (host code only: su3.h is __host__ __device__)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include "su3.h"

typedef long double ld;
struct LM { ld re[9], im[9]; };
static LM lmul(const LM &a, const LM &b) {
  LM r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    ld x = 0, y = 0;
    for (int k = 0; k < 3; k++) { x += a.re[3*i+k]*b.re[3*k+j] - a.im[3*i+k]*b.im[3*k+j]; y += a.re[3*i+k]*b.im[3*k+j] + a.im[3*i+k]*b.re[3*k+j]; }
    r.re[3*i+j] = x; r.im[3*i+j] = y;
  }
  return r;
}
static LM lexp(const M3 &v) {       // scaling and squaring in long double: scale 2^-12, Taylor to order 18
  LM a, r, t;
  for (int k = 0; k < 9; k++) { a.re[k] = (ld)v.e[k].x / 4096.0L; a.im[k] = (ld)v.e[k].y / 4096.0L; }
  for (int k = 0; k < 9; k++) { r.re[k] = (k % 4 == 0) ? 1.0L : 0.0L; r.im[k] = 0; t = r; }
  t = r;
  for (int n = 1; n <= 18; n++) {
    t = lmul(t, a);
    for (int k = 0; k < 9; k++) { t.re[k] /= n; t.im[k] /= n; r.re[k] += t.re[k]; r.im[k] += t.im[k]; }
  }
  for (int s = 0; s < 12; s++) r = lmul(r, r);
  return r;
}
int main() {
  std::mt19937_64 rng(12345);
  std::normal_distribution<double> gs(0.0, 1.0);
  const double scales[] = {1e-9, 1e-5, 1e-4, 3e-4, 1e-3, 0.01, 0.1, 0.5, 1.0, 3.0, 8.0};
  int n = 0, bad = 0;
  for (double sc : scales) {
    double worst_ch = 0, worst_ref = 0, worst_uni_ch = 0, worst_uni_ref = 0, worst_diff = 0;
    for (int rep = 0; rep < 400; rep++) {
      M3 g;
      for (int k = 0; k < 9; k++) g.e[k] = make_double2(gs(rng), gs(rng));
      if (rep % 7 == 0) { g.e[1] = g.e[2] = g.e[3] = g.e[5] = g.e[6] = g.e[7] = make_double2(0, 0); }       // diagonal: degenerate-prone
      if (rep % 11 == 0) { g.e[4] = g.e[0]; }                                                               // two nearly equal eigenvalues
      M3 v = m3_tah(g);
      for (int k = 0; k < 9; k++) { v.e[k].x *= sc; v.e[k].y *= sc; }
      const M3 a = m3_exp_tah(v), b = m3_exp(v);
      const LM ex = lexp(v);
      double ea = 0, eb = 0, dab = 0;
      for (int k = 0; k < 9; k++) {
        ea = fmax(ea, fmax(fabs((double)((ld)a.e[k].x - ex.re[k])), fabs((double)((ld)a.e[k].y - ex.im[k]))));
        eb = fmax(eb, fmax(fabs((double)((ld)b.e[k].x - ex.re[k])), fabs((double)((ld)b.e[k].y - ex.im[k]))));
        dab = fmax(dab, fmax(fabs(a.e[k].x - b.e[k].x), fabs(a.e[k].y - b.e[k].y)));
      }
      auto uni = [](const M3 &m) { M3 p = m3_mul_na(m, m); double d = 0; for (int k = 0; k < 9; k++) d = fmax(d, fmax(fabs(p.e[k].x - (k % 4 == 0)), fabs(p.e[k].y))); return d; };
      worst_ch = fmax(worst_ch, ea); worst_ref = fmax(worst_ref, eb); worst_diff = fmax(worst_diff, dab);
      worst_uni_ch = fmax(worst_uni_ch, uni(a)); worst_uni_ref = fmax(worst_uni_ref, uni(b));
      n++;
    }
    printf("scale %-6g |closed form - exact| %.2e  |reference algorithm - exact| %.2e  |closed - reference| %.2e  unitarity %.2e vs %.2e\n",
           sc, worst_ch, worst_ref, worst_diff, worst_uni_ch, worst_uni_ref);
    // the closed form must be at least as accurate as the algorithm it stands in for (both grow with the norm of v)
    const double lim = 1e-15 * fmax(1.0, 2.0 * sc);
    if (!(worst_ch < 2.0 * lim && worst_ch < fmax(2e-15, 1.5 * worst_ref) && worst_uni_ch < 4.0 * lim)) bad++;
  }
  printf("%d matrices, %d scale(s) out of bounds\n", n, bad);
  return bad ? 1 : 0;
}
