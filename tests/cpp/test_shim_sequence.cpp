// test_shim_sequence.cpp -- the C-call sequence of every proc of qex_amd/nim/qexhip.nim, executed from C++.
//
// The image has no Nim compiler, so the Nim shim a QEX maintainer would build cannot be compiled here.  What can be
// executed is what the shim DOES: each proc below performs, in the same order and with the same arguments, the calls of the
// Nim proc of the same name on V=1 site-major arrays (what toHost / toHostG produce), through the plain C ABI
// (include/qexhip.h), and the results are held to the CPU oracle (oracle/qex_oracle.h: test infrastructure).
// A change of an entry point's signature or meaning breaks this program the way it would break the shim.
//
//   shim proc                      reference call site it stands in for
//   hipSetup                       qudaSetup, src/quda/qudaWrapperImpl.nim:88-123
//   hipSetLinks                    the link copy of qudaSolveXX, :216-240 (plain and Naik: s.g.len 4 | 8)
//   hipStagD2 / hipD / hipDdag     stagD2, s.D, s.Ddag: src/physics/stagD.nim:349-395,566-571
//   hipSolveEE / hipSolveOO        solveXX backend arm, src/physics/stagSolve.nim:65-128
//   hipSolve                       Staggered.solve, :224-294
//   hipSolveXX(xs) / hipSolve(xs)  multi-shift, :296-345,347-446
//   hipGaugeFlow (both forms)      src/gauge/wflow.nim:21-67, src/flow/flow.nim:22-90 (+ hipPlaq, hipFlowMeasure in `measure`)
//   hipSmearGetForce + closure     src/gauge/hypsmear.nim:49-247; gforce / fforce: src/stagg_pv_hmc/staghmc_spv.nim:217-228,716-865
// Build + run: tests/test_cpp_host.py.
#include "qexhip.h"
extern "C" {
#include "qex_oracle.h"
}
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef std::vector<double> Buf;
static int fails = 0;
#define CHECK(cond, ...)                                                   \
  do {                                                                     \
    if (!(cond)) { fails++; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } \
  } while (0)
#define CHK(e)                                                             \
  do {                                                                     \
    int rc_ = (e);                                                         \
    if (rc_ != 0) { printf("libqexhip: %s (%s)\n", qexhip_last_error(), #e); exit(2); } \
  } while (0)
static double relerr(const Buf &a, const Buf &b, size_t lo = 0, size_t hi = 0) {
  if (!hi) hi = a.size();
  double n = 0, d = 0;
  for (size_t i = lo; i < hi; i++) { d += (a[i] - b[i]) * (a[i] - b[i]); n += b[i] * b[i]; }
  return std::sqrt(d / (n > 0 ? n : 1));
}

// ---- the shim, call for call (h = hipParam.h) ----
static qexhip_handle h;
static void hipSetup(const int lat[4]) {
  int ndev = 0;
  CHK(qexhip_device_count(&ndev));
  const int rg[4] = {1, 1, 1, 1}, rc[4] = {0, 0, 0, 0};
  CHK(qexhip_init(&h, 0 % (ndev > 0 ? ndev : 1), lat, rg, rc));      // one rank: no unique id / comm_init
  char tr[16];
  CHK(qexhip_comm_transport(h, tr, 16, nullptr));                    // the line hipSetup prints: "none" without a communicator
  CHECK(std::string(tr) == "none", "transport before comm_init: %s", tr);
}
static void hipSetLinks(const Buf &fat, const Buf *lng) { CHK(qexhip_stag_set_links(h, fat.data(), lng ? lng->data() : nullptr)); }
static void hipStagD2(Buf &r, const Buf &x, double a, double b, int subset) { CHK(qexhip_stag_dslash(h, r.data(), x.data(), subset, a, b)); }
static void hipD(Buf &r, const Buf &x, double m) { CHK(qexhip_stag_D(h, r.data(), x.data(), m, 1.0)); }
static void hipDdag(Buf &r, const Buf &x, double m) { CHK(qexhip_stag_D(h, r.data(), x.data(), m, -1.0)); }
static int hipSolveXX(Buf &r, const Buf &t, double m, double r2req, int maxits, bool parEven, double *r2) {
  int iters = 0;
  CHK(qexhip_stag_solve_xx(h, r.data(), t.data(), m, r2req, maxits, parEven ? 1 : 0, &iters, r2, nullptr, 0));
  return iters;
}
static int hipSolve(Buf &x, const Buf &b, double m, double r2req, int maxits, double *r2) {
  int iters = 0;
  CHK(qexhip_stag_solve(h, x.data(), b.data(), m, r2req, maxits, &iters, r2));
  return iters;
}
static int hipSolveXXmulti(std::vector<Buf> &xs, const Buf &b, const std::vector<double> &ms, double r2req, int maxits, bool parEven) {
  const int n = (int)xs.size();
  std::vector<double *> ptrs(n);
  std::vector<double> shifts(n);
  for (int k = 0; k < n; k++) {
    ptrs[k] = xs[k].data();
    shifts[k] = k == 0 ? ms[0] : 4.0 * (ms[k] * ms[k] - ms[0] * ms[0]);      // stagSolve.nim:391-394
  }
  int iters = 0;
  CHK(qexhip_stag_solve_xx_multi(h, ptrs.data(), b.data(), shifts.data(), n, r2req, maxits, parEven ? 1 : 0, &iters, nullptr, 0));
  return iters;
}
static int hipSolveMulti(std::vector<Buf> &xs, const Buf &b, const std::vector<double> &ms, double r2req, int maxits, double *r2) {
  const int n = (int)xs.size();
  std::vector<double *> ptrs(n);
  for (int k = 0; k < n; k++) ptrs[k] = xs[k].data();
  int iters = 0;
  CHK(qexhip_stag_solve_multi(h, ptrs.data(), b.data(), ms.data(), n, r2req, maxits, &iters, r2));
  return iters;
}
static void hipGaugeSet(const Buf &g) { CHK(qexhip_gauge_set(h, g.data())); }
static void hipGaugeGet(Buf &g) { CHK(qexhip_gauge_get(h, g.data())); }
static void hipPlaq(double o[6]) { CHK(qexhip_plaq(h, o)); }
static void hipFlowMeasure(double pl[6], double eq[3]) { CHK(qexhip_flow_measure(h, pl, eq)); }
static void hipS4Gauge(double o[8]) { CHK(qexhip_plaq_s4(h, o)); }
// hipPloops: (pls, plt) = (mean of the three spatial loops, the temporal one), as meas_ploop (gauge_flow.nim:137-156)
static void hipPloops(double pls[2], double plt[2]) {
  double o[8];
  CHK(qexhip_polyakov_loops(h, o));
  pls[0] = (o[0] + o[2] + o[4]) / 3.0; pls[1] = (o[1] + o[3] + o[5]) / 3.0;
  plt[0] = o[6]; plt[1] = o[7];
}
template <class M> static void hipGaugeFlow(Buf &g, int steps, double eps, M measure) {
  hipGaugeSet(g);
  for (int n = 1;; n++) {
    CHK(qexhip_wflow(h, 1, eps));
    measure(n * eps);
    if (steps > 0 && n + 1 > steps) break;
  }
  hipGaugeGet(g);
}
template <class M> static void hipGaugeFlowGeneral(double plaq, double rect, double adjplaq, const char *flowAct, Buf &g, int steps, double eps, M measure) {
  hipGaugeSet(g);
  const bool adj = flowAct[0] == 'a';
  for (int n = 1;; n++) {
    CHK(qexhip_wflow_general(h, 1, eps, plaq, adj ? adjplaq : rect, adj ? 1 : 0));
    measure(n * eps);
    if (steps > 0 && n + 1 > steps) break;
  }
  hipGaugeGet(g);
}
struct HipSmearedForce { int bc[4]; };
static HipSmearedForce hipSmearGetForce(double a1, double a2, double a3, const Buf &g, Buf &sg, const char *bc) {
  CHK(qexhip_nhyp_prepare(h, g.data(), a1, a2, a3, sg.data()));
  HipSmearedForce sf;
  for (int mu = 0; mu < 4; mu++) sf.bc[mu] = bc[mu] == 'a';
  return sf;
}
static void smearedForce(Buf &f, const Buf &chain) { CHK(qexhip_nhyp_force(h, f.data(), chain.data())); }
static void gforce(Buf &f, double plaq, double rect, double adjplaq) { CHK(qexhip_nhyp_gauge_force(h, f.data(), plaq, rect, adjplaq)); }
static void fforce(const HipSmearedForce &sf, Buf &f, const std::vector<Buf> &psis, const std::vector<double> &scales) {
  std::vector<const double *> ptrs(psis.size());
  for (size_t k = 0; k < psis.size(); k++) ptrs[k] = psis[k].data();
  CHK(qexhip_nhyp_fermion_force(h, f.data(), ptrs.data(), scales.data(), (int)psis.size(), sf.bc, nullptr));
}
static void setLinksFromClosure(const HipSmearedForce &sf) { CHK(qexhip_stag_set_links_nhyp(h, nullptr, 0.0, 0.0, 0.0, sf.bc, nullptr)); }
static void release() { CHK(qexhip_nhyp_release(h)); }

int main() {
  const int lat[4] = {4, 6, 8, 4};
  hipSetup(lat);
  qo_layout *lo = qo_layout_new(lat);
  const size_t vol = (size_t)qo_vol(lo), vh = vol / 2;
  qo_rngfield *rf = qo_rngfield_new(lo, QO_RNG_MILC6, 33);
  Buf g0(vol * 72), g, x(vol * 6), y(vol * 6);
  qo_gauge_warm(lo, rf, 0.5, g0.data());
  qo_vector_gaussian(lo, rf, x.data());
  qo_vector_gaussian(lo, rf, y.data());
  g = g0;
  qo_setBC(lo, g.data());
  const int phases[4] = {8, 9, 11, 0};
  qo_stagPhase(lo, g.data(), phases);

  // ---- hipSetLinks (plain), hipStagD2, hipD, hipDdag ----
  hipSetLinks(g, nullptr);
  for (int sub = 0; sub < 3; sub++) {
    Buf r = y, ref = y;
    hipStagD2(r, x, 0.5, 0.25, sub);
    qo_stagD2(lo, g.data(), nullptr, ref.data(), x.data(), sub, 0.5, 0.25);
    CHECK(relerr(r, ref) < 1e-13, "hipStagD2 subset %d: %g", sub, relerr(r, ref));
  }
  {
    Buf r(vol * 6), ref(vol * 6);
    hipD(r, x, 0.1);
    qo_D(lo, g.data(), nullptr, ref.data(), x.data(), 0.1);
    CHECK(relerr(r, ref) < 1e-13, "hipD: %g", relerr(r, ref));
    hipDdag(r, x, 0.1);
    qo_Ddag(lo, g.data(), nullptr, ref.data(), x.data(), 0.1);
    CHECK(relerr(r, ref) < 1e-13, "hipDdag: %g", relerr(r, ref));
    // hipStagD (stagD.nim:406-409) on one subset with the accumulate term
    Buf sd = y, sdref = y;
    CHK(qexhip_stag_stagD(h, sd.data(), x.data(), QEXHIP_ODD, 0.2, -0.5, 0.7));
    qo_stagD(lo, g.data(), nullptr, sdref.data(), x.data(), 1, 0.2, -0.5, 0.7);
    CHECK(relerr(sd, sdref) < 1e-13, "hipStagD: %g", relerr(sd, sdref));
    // hipEoReduce / hipEoReconstruct (stagD.nim:575-586): r is read and written (the other parity is kept)
    Buf e = y, eref = y;
    CHK(qexhip_stag_eo_reduce(h, e.data(), x.data(), 0.1));
    qo_eoReduce(lo, g.data(), nullptr, eref.data(), x.data(), 0.1);
    CHECK(relerr(e, eref) < 1e-13, "hipEoReduce: %g", relerr(e, eref));
    e = y; eref = y;
    CHK(qexhip_stag_eo_reconstruct(h, e.data(), x.data(), 0.1));
    qo_eoReconstruct(lo, g.data(), nullptr, eref.data(), x.data(), 0.1);
    CHECK(relerr(e, eref) < 1e-13, "hipEoReconstruct: %g", relerr(e, eref));
  }
  // ---- through the seam from QEX's OWN memory: SIMD fields in, SIMD field out (toHostG / toHost / fromHost of qexhip.nim) ----
  // The SIMD arrays are filled here by the closed form of layoutIndexQ for an inner geometry without checkerboard shift
  // (qlayout.nim:110-131: lane = lex of c / outer, outer index = lex(c % outer) / 2 (+ half for odd sites)) -- written out
  // independently of the library's restatement, which then has to undo it.
  {
    int ig[4];
    CHK(qexhip_layout_default_inner(lat, 8, ig));                    // newLayoutX's choice for V = 8 on 4x6x8x4: {2,1,2,2}
    CHECK(ig[0] == 2 && ig[1] == 1 && ig[2] == 2 && ig[3] == 2, "default inner geometry %d %d %d %d", ig[0], ig[1], ig[2], ig[3]);
    const int V = 8, og[4] = {lat[0] / ig[0], lat[1] / ig[1], lat[2] / ig[2], lat[3] / ig[3]};
    const size_t nouter = vol / V;
    std::vector<size_t> simd_of_v1(vol);
    for (size_t lex = 0; lex < vol; lex++) {
      int c[4], par = 0; size_t r = lex;
      for (int i = 0; i < 4; i++) { c[i] = (int)(r % lat[i]); r /= lat[i]; par += c[i]; }
      const size_t j = lex / 2 + ((par & 1) ? vh : 0);
      int lane = 0, lmul = 1; size_t olex = 0, omul = 1;
      for (int i = 0; i < 4; i++) { lane += (c[i] / og[i]) * lmul; lmul *= ig[i]; olex += (size_t)(c[i] % og[i]) * omul; omul *= og[i]; }
      simd_of_v1[j] = (olex / 2 + ((par & 1) ? nouter / 2 : 0)) * V + lane;
    }
    auto vec_to_simd = [&](const Buf &v1, Buf &sv) {
      sv.assign(vol * 6, 0.0);
      for (size_t j = 0; j < vol; j++) for (int k = 0; k < 6; k++) sv[((simd_of_v1[j] / V) * 6 + k) * V + simd_of_v1[j] % V] = v1[j * 6 + k];
    };
    std::vector<Buf> gq(4, Buf(nouter * 18 * V));                    // s.g[mu]: four QEX fields
    for (size_t j = 0; j < vol; j++) for (int mu = 0; mu < 4; mu++) for (int k = 0; k < 18; k++)
      gq[mu][((simd_of_v1[j] / V) * 18 + k) * V + simd_of_v1[j] % V] = g[j * 72 + mu * 18 + k];
    Buf xq, rq(vol * 6, 0.0);
    vec_to_simd(x, xq);
    // hipSetLinks(s): toHostG(s.g, g1) -> qexhip_stag_set_links
    Buf g1(vol * 72), xb(vol * 6), rb(vol * 6);
    const double *gp[4] = {gq[0].data(), gq[1].data(), gq[2].data(), gq[3].data()};
    CHK(qexhip_layout_gauge_simd_to_v1(lat, ig, gp, g1.data()));
    CHECK(relerr(g1, g) == 0.0, "toHostG: %g", relerr(g1, g));
    hipSetLinks(g1, nullptr);
    // hipSolveEE(s, r, t, m, sp): toHost(t) -> solve -> fromHost(r)
    CHK(qexhip_layout_vec_simd_to_v1(lat, ig, xq.data(), xb.data()));
    double r2 = 0, fin = 0;
    const int its = hipSolveXX(rb, xb, 0.1, 1e-12, 5000, true, &r2);
    CHK(qexhip_layout_vec_v1_to_simd(lat, ig, rb.data(), rq.data()));
    Buf ref(vol * 6), refq;
    const int oits = qo_solveXX(lo, g.data(), nullptr, ref.data(), x.data(), 0.1, 1e-12, 5000, 1, nullptr, 0, &fin);
    for (size_t k = vh * 6; k < vol * 6; k++) ref[k] = rb[k];       // the odd half is not part of solveEE's answer
    vec_to_simd(ref, refq);
    CHECK(std::abs(its - oits) <= 1 && relerr(rq, refq) < 1e-6, "hipSolveEE on SIMD-layout fields: its %d / %d, %g", its, oits, relerr(rq, refq));
    // gauge field back into QEX's memory (hipGaugeFlow's last step): fromHostG
    std::vector<Buf> gback(4, Buf(nouter * 18 * V));
    double *gbp[4] = {gback[0].data(), gback[1].data(), gback[2].data(), gback[3].data()};
    CHK(qexhip_layout_gauge_v1_to_simd(lat, ig, g1.data(), gbp));
    for (int mu = 0; mu < 4; mu++) CHECK(relerr(gback[mu], gq[mu]) == 0.0, "fromHostG mu %d", mu);
  }
  // ---- hipSolveEE / hipSolveOO / hipSolve ----
  for (int pe = 1; pe >= 0; pe--) {
    Buf r(vol * 6), ref(vol * 6);
    double r2 = 0, fin = 0;
    const int its = hipSolveXX(r, x, 0.1, 1e-12, 5000, pe, &r2);
    const int oits = qo_solveXX(lo, g.data(), nullptr, ref.data(), x.data(), 0.1, 1e-12, 5000, pe, nullptr, 0, &fin);
    const size_t a = pe ? 0 : vh * 6, b = pe ? vh * 6 : vol * 6;
    CHECK(std::abs(its - oits) <= 1 && relerr(r, ref, a, b) < 1e-6 && r2 <= 1e-12, "hipSolve%s: its %d / %d, %g", pe ? "EE" : "OO", its, oits, relerr(r, ref, a, b));
  }
  {
    Buf r(vol * 6), ref(vol * 6);
    double r2 = 0, fin = 0;
    const int its = hipSolve(r, x, 0.1, 1e-12, 10000, &r2);
    const int oits = qo_solve(lo, g.data(), nullptr, ref.data(), x.data(), 0.1, 1e-12, 10000, &fin);
    CHECK(std::abs(its - oits) <= 2 && relerr(r, ref) < 1e-7 && r2 <= 1e-12, "hipSolve: its %d / %d, %g, r2 %g", its, oits, relerr(r, ref), r2);
  }
  // ---- hipSetLinks (Naik: HISQ fat + long links), multi-shift ----
  {
    Buf fl(vol * 72), ll(vol * 72);
    qo_hisq_smear(lo, g.data(), fl.data(), ll.data());
    hipSetLinks(fl, &ll);
    const std::vector<double> ms = {0.2, 0.4, 0.8};
    std::vector<Buf> xs(3, Buf(vol * 6)), ref(3, Buf(vol * 6));
    const int its = hipSolveXXmulti(xs, x, ms, 1e-14, 5000, true);
    std::vector<double *> rp = {ref[0].data(), ref[1].data(), ref[2].data()};
    const double shifts[3] = {ms[0], 4 * (ms[1] * ms[1] - ms[0] * ms[0]), 4 * (ms[2] * ms[2] - ms[0] * ms[0])};
    const int oits = qo_solveXX_multi(lo, fl.data(), ll.data(), rp.data(), x.data(), shifts, 3, 1e-14, 5000, 1, nullptr, 0);
    CHECK(std::abs(its - oits) <= 1, "hipSolveXX(xs): its %d / %d", its, oits);
    for (int k = 0; k < 3; k++) CHECK(relerr(xs[k], ref[k], 0, vh * 6) < 1e-6, "hipSolveXX(xs) shift %d: %g", k, relerr(xs[k], ref[k], 0, vh * 6));
    double r2 = 0, fin = 0;
    const int its2 = hipSolveMulti(xs, x, ms, 1e-12, 5000, &r2);
    const int oits2 = qo_solve_multi(lo, fl.data(), ll.data(), rp.data(), x.data(), ms.data(), 3, 1e-12, 5000, &fin);
    CHECK(std::abs(its2 - oits2) <= 2, "hipSolve(xs): its %d / %d", its2, oits2);
    for (int k = 0; k < 3; k++) {
      CHECK(relerr(xs[k], ref[k]) < 1e-6, "hipSolve(xs) mass %d: %g", k, relerr(xs[k], ref[k]));
      Buf d(vol * 6);                                               // true residual through the oracle's operator
      qo_D(lo, fl.data(), ll.data(), d.data(), xs[k].data(), ms[k]);
      CHECK(relerr(d, x) < 2e-6, "hipSolve(xs) mass %d: residual %g", k, relerr(d, x));
    }
  }
  // ---- hipGaugeFlow with hipPlaq / hipFlowMeasure in its measure block ----
  {
    Buf gf = g0, gr = g0;
    int nmeas = 0;
    hipGaugeFlow(gf, 3, 0.01, [&](double t) {
      double pl[6], pl2[6], eq[3], opl[6], oeq[3];
      hipPlaq(pl);
      hipFlowMeasure(pl2, eq);
      qo_wflow(lo, gr.data(), 1, 0.01);                             // the oracle, one step behind no more
      qo_plaq(lo, gr.data(), opl);
      qo_flow_EQ(lo, gr.data(), 1, oeq);
      for (int i = 0; i < 6; i++) CHECK(std::fabs(pl[i] - opl[i]) < 1e-14 && std::fabs(pl2[i] - opl[i]) < 1e-14, "flow t=%g plaq[%d] %g %g %g", t, i, pl[i], pl2[i], opl[i]);
      for (int i = 0; i < 3; i++) CHECK(std::fabs(eq[i] - oeq[i]) < 1e-11 * (1 + std::fabs(oeq[i])), "flow t=%g EQ[%d] %g %g", t, i, eq[i], oeq[i]);
      // meas_ploop: g.wline(repeat(i+1, pg[i])) for the four directions
      double pls[2], plt[2], ops[2] = {0, 0}, opt[2] = {0, 0};
      hipPloops(pls, plt);
      for (int d = 0; d < 4; d++) {
        std::vector<int> path((size_t)lat[d], d + 1);
        double w[2];
        qo_wline(lo, gr.data(), path.data(), lat[d], w);
        if (d < 3) { ops[0] += w[0] / 3.0; ops[1] += w[1] / 3.0; } else { opt[0] = w[0]; opt[1] = w[1]; }
      }
      CHECK(std::fabs(pls[0] - ops[0]) < 1e-14 && std::fabs(pls[1] - ops[1]) < 1e-14 && std::fabs(plt[0] - opt[0]) < 1e-14 && std::fabs(plt[1] - opt[1]) < 1e-14,
            "flow t=%g Polyakov loops (%g,%g) (%g,%g) vs (%g,%g) (%g,%g)", t, pls[0], pls[1], plt[0], plt[1], ops[0], ops[1], opt[0], opt[1]);
      // the fork's verbose plaquette measurement (staghmc_spv_meas.nim:25-65)
      double s4[8], os4[8];
      hipS4Gauge(s4);
      qo_s4_gauge(lo, gr.data(), os4);
      for (int i = 0; i < 8; i++) CHECK(std::fabs(s4[i] - os4[i]) < 1e-14, "flow t=%g s4_gauge[%d] %g %g", t, i, s4[i], os4[i]);
      nmeas++;
    });
    CHECK(nmeas == 3 && relerr(gf, gr) < 1e-12, "hipGaugeFlow: %d measurements, links %g", nmeas, relerr(gf, gr));
    // the fork's action-selectable form: "rect" with (plaq, rect), "adj" with (plaq, adjplaq)
    Buf g1 = g0, r1 = g0, g2 = g0, r2 = g0;
    hipGaugeFlowGeneral(5.0 / 3.0, -1.0 / 12.0, 0.0, "rect", g1, 2, 0.01, [](double) {});
    qo_wflow_general(lo, r1.data(), 2, 0.01, 5.0 / 3.0, -1.0 / 12.0, 0);
    hipGaugeFlowGeneral(0.9, 0.0, 0.35, "adj", g2, 2, 0.01, [](double) {});
    qo_wflow_general(lo, r2.data(), 2, 0.01, 0.9, 0.35, 1);
    CHECK(relerr(g1, r1) < 1e-12 && relerr(g2, r2) < 1e-12, "hipGaugeFlow(gc, act): rect %g adj %g", relerr(g1, r1), relerr(g2, r2));
  }
  // ---- hipSmearGetForce and the closure ----
  {
    Buf sg(vol * 72), osg(vol * 72);
    HipSmearedForce sf = hipSmearGetForce(0.4, 0.5, 0.5, g0, sg, "aaaa");
    qo_nhyp_smear(lo, g0.data(), osg.data(), 0.4, 0.5, 0.5);
    CHECK(relerr(sg, osg) < 1e-13, "hipSmearGetForce: smeared links %g", relerr(sg, osg));
    Buf chain(vol * 72), f(vol * 72), ref(vol * 72), ofl(vol * 72);
    qo_gauge_random_tah(lo, rf, chain.data());
    smearedForce(f, chain);
    qo_nhyp_force(lo, g0.data(), ofl.data(), ref.data(), chain.data(), 0.4, 0.5, 0.5);
    CHECK(relerr(f, ref) < 1e-11, "smearedForce: %g", relerr(f, ref));
    // gforce: derivative of the action on the smeared links, chain, TAH(g f^+)        (staghmc_spv.nim:217-228)
    gforce(f, 5.0 / 3.0, -1.0 / 12.0, 0.0);
    qo_gauge_deriv_rect(lo, osg.data(), chain.data(), 5.0 / 3.0, -1.0 / 12.0);
    qo_nhyp_force(lo, g0.data(), ofl.data(), ref.data(), chain.data(), 0.4, 0.5, 0.5);
    qo_force_projTAH(lo, ref.data(), g0.data(), 1);
    CHECK(relerr(f, ref) < 1e-11, "closure.gforce: %g", relerr(f, ref));
    // fforce: sum_k scale_k psi_k (x) psi_k(+mu)^+, rephase (bc aaaa + phases), odd sites re-signed, chain, TAH(f g^+)
    std::vector<Buf> psis = {x, y};
    const std::vector<double> scales = {0.37, -1.9};
    fforce(sf, f, psis, scales);
    std::fill(ref.begin(), ref.end(), 0.0);
    qo_stag_outer(lo, ref.data(), x.data(), scales[0], scales[0], 0);
    qo_stag_outer(lo, ref.data(), y.data(), scales[1], scales[1], 1);
    for (size_t i = 0; i < vol; i++) {
      int c[4];
      qo_coord(lo, (int)i, c);
      for (int mu = 0; mu < 4; mu++)
        if (c[mu] == lat[mu] - 1) for (int k = 0; k < 18; k++) ref[(i * 4 + mu) * 18 + k] = -ref[(i * 4 + mu) * 18 + k];   // 'a' in every direction
    }
    qo_stagPhase(lo, ref.data(), phases);
    for (size_t i = vh * 72; i < vol * 72; i++) ref[i] = -ref[i];
    Buf ref2(vol * 72);
    qo_nhyp_force(lo, g0.data(), ofl.data(), ref2.data(), ref.data(), 0.4, 0.5, 0.5);
    qo_force_projTAH(lo, ref2.data(), g0.data(), 0);
    CHECK(relerr(f, ref2) < 1e-11, "closure.fforce: %g", relerr(f, ref2));
    // sg.rephase(); newStag(sg) from the closure's links: the operator equals the oracle's on rephase(smear(g))
    setLinksFromClosure(sf);
    for (size_t i = 0; i < vol; i++) {
      int c[4];
      qo_coord(lo, (int)i, c);
      for (int mu = 0; mu < 4; mu++)
        if (c[mu] == lat[mu] - 1) for (int k = 0; k < 18; k++) osg[(i * 4 + mu) * 18 + k] = -osg[(i * 4 + mu) * 18 + k];
    }
    qo_stagPhase(lo, osg.data(), phases);
    Buf r(vol * 6), rr(vol * 6);
    hipD(r, x, 0.1);
    qo_D(lo, osg.data(), nullptr, rr.data(), x.data(), 0.1);
    CHECK(relerr(r, rr) < 1e-12, "setLinksFromClosure + hipD: %g", relerr(r, rr));
    release();
  }
  CHK(qexhip_finalize(h));
  if (fails) { printf("%d check(s) FAILED\n", fails); return 1; }
  printf("shim sequence: Passed\n");
  return 0;
}
