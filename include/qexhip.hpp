// qexhip.hpp -- C++ host-side mirror of QEX's staggered operator / solver interface over the C ABI.
//
// QEX's host code is Nim compiled to C; the toolchain is not available in this build image, so
// this header is the compiled-language host layer above include/qexhip.h: same names, argument
// meaning and error behaviour as the reference, so a test written against it reads like the
// reference's own (tests/examples/testStagProp.nim, src/physics/stagSolve.nim:516-680):
//
//   Layout, ColorVector(), newGauge()         src/layout/layoutX.nim:70, src/physics/qcdTypes.nim
//   setBC / stagPhase / rephase               src/gauge/gaugeUtils.nim:124-131, src/physics/stagD.nim:72-80,509-520
//   newStag(g) / newStag3(g, g3)              src/physics/stagD.nim:522-564
//   Staggered::D / Ddag / eoReconstruct       src/physics/stagD.nim:566-586
//   Staggered::solveEE / solveOO / solve      src/physics/stagSolve.nim:134-138,224-294,347-446
//   SolverParams                              src/solvers/solverBase.nim:10-58
//   plaq / gaugeFlow                          src/gauge/gaugeUtils.nim:213-282, src/gauge/wflow.nim:21-67
//
// Header-only; link with -lqexhip.  Fields are std::vector<double> in the V=1 even-odd host format
// (colour vector [vol][3][2], gauge [vol][4][3][3][2]).  Errors throw qex::Error (QEX aborts with
// qexError, src/base/qexInternal.nim:37-45); not converging within maxits is not an error.
#pragma once
#include "qexhip.h"
#include <array>
#include <chrono>
#include <complex>
#include <stdexcept>
#include <string>
#include <vector>

namespace qex {

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};
inline void check(int rc) {
  if (rc != 0) throw Error(std::string("libqexhip error ") + std::to_string(rc) + ": " + qexhip_last_error());
}

using Field = std::vector<double>;

// V=1 even-odd layout of the rank-local lattice (src/layout/qlayout.nim:110-131)
struct Layout {
  std::array<int, 4> physGeom;
  int nSites, nEven;
  std::vector<std::array<int, 4>> coords;  // coords[idx]
  explicit Layout(const std::array<int, 4> &lat) : physGeom(lat) {
    nSites = lat[0] * lat[1] * lat[2] * lat[3];
    nEven = nSites / 2;
    coords.resize(nSites);
    std::array<int, 4> x;
    for (x[3] = 0; x[3] < lat[3]; x[3]++)
      for (x[2] = 0; x[2] < lat[2]; x[2]++)
        for (x[1] = 0; x[1] < lat[1]; x[1]++)
          for (x[0] = 0; x[0] < lat[0]; x[0]++) coords[index(x)] = x;
  }
  int index(const std::array<int, 4> &x) const {
    int lex = 0, p = 0;
    for (int i = 3; i >= 0; i--) lex = lex * physGeom[i] + x[i];
    for (int i = 0; i < 4; i++) p += x[i];
    return (p & 1) ? (lex + nSites) / 2 : lex / 2;
  }
  Field ColorVector() const { return Field((size_t)nSites * 6, 0.0); }
  Field newGauge() const { return Field((size_t)nSites * 72, 0.0); }
};

// U_3 *= -1 on the last t slice (gaugeUtils.nim:124-131)
inline void setBC(const Layout &lo, Field &g) {
  for (int s = 0; s < lo.nSites; s++)
    if (lo.coords[s][3] == lo.physGeom[3] - 1)
      for (int k = 0; k < 18; k++) g[((size_t)s * 4 + 3) * 18 + k] *= -1.0;
}
// eta_mu from the bit masks [8,9,11,0] (stagD.nim:509-520)
inline void stagPhase(const Layout &lo, Field &g, const std::array<int, 4> &phases = {8, 9, 11, 0}) {
  for (int mu = 0; mu < 4; mu++)
    for (int i = 0; i < lo.nSites; i++) {
      int s = 0;
      for (int k = 0; k < 4; k++) s += (phases[mu] >> k) & lo.coords[i][k];
      if (s & 1)
        for (int k = 0; k < 18; k++) g[((size_t)i * 4 + mu) * 18 + k] *= -1.0;
    }
}
inline void rephase(const Layout &lo, Field &g) { setBC(lo, g); stagPhase(lo, g); }

// solverBase.nim:10-58
struct SolverParams {
  double r2req = 1e-6;
  int maxits = 50000;
  int verbosity = 1;
  // outputs
  int calls = 0, iterations = 0, iterationsMax = 0;
  double seconds = 0, flops = 0, r2 = 0;
  std::vector<double> r2hist;  // "CG iteration: k  r2/b2:" values when histcap > 0
  void resetStats() { calls = iterations = iterationsMax = 0; seconds = flops = r2 = 0; r2hist.clear(); }
  int finalIterations() const { return iterations; }
};

class Context {
 public:
  qexhip_handle h = nullptr;
  Layout lo;
  explicit Context(const std::array<int, 4> &latLocal, int device = 0,
                   const std::array<int, 4> &rankGeom = {1, 1, 1, 1}, const std::array<int, 4> &rankCoord = {0, 0, 0, 0})
      : lo(latLocal) {
    check(qexhip_init(&h, device, latLocal.data(), rankGeom.data(), rankCoord.data()));
  }
  ~Context() { if (h) qexhip_finalize(h); }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  std::string info() const { char b[512]; check(qexhip_device_info(h, b, 512)); return b; }
  // multi-rank set-up as hipSetup of qexhip.nim: rank 0 calls uniqueId(), the host broadcasts it, every rank calls commInit
  static std::array<char, QEXHIP_UNIQUE_ID_BYTES> uniqueId() {
    std::array<char, QEXHIP_UNIQUE_ID_BYTES> id{};
    check(qexhip_comm_unique_id(id.data()));
    return id;
  }
  void commInit(const std::array<char, QEXHIP_UNIQUE_ID_BYTES> &id, int nranks, int rank) { check(qexhip_comm_init(h, id.data(), nranks, rank)); }
  // what RCCL reports for the communicator + the PCI bus id of the bound GPU
  struct CommInfo { int nranks, rank, device; std::string busId; };
  CommInfo commInfo() const {
    CommInfo ci{0, -1, 0, ""};
    char bus[64] = "";
    check(qexhip_comm_info(h, &ci.nranks, &ci.rank, &ci.device, bus, 64));
    ci.busId = bus;
    return ci;
  }
  void setOption(const char *name, int value) { check(qexhip_set_option(h, name, value)); }   // e.g. "flow_exp", 0
  static int deviceCount() { int n = 0; check(qexhip_device_count(&n)); return n; }
};

class Staggered {
  Context &c_;
  int nlinks_;
  double flops(int its) const { return double(nlinks_ * 4 * 72 + 60) * c_.lo.nEven * its; }  // stagSolve.nim:92
  template <class F> void timed(SolverParams &sp, int &its, F &&f) {
    auto t0 = std::chrono::steady_clock::now();
    f();
    sp.seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    sp.calls += 1; sp.iterations += its; sp.iterationsMax = std::max(sp.iterationsMax, its); sp.flops += flops(its);
  }

 public:
  Staggered(Context &c, const Field &g) : c_(c), nlinks_(4) { check(qexhip_stag_set_links(c.h, g.data(), nullptr)); }
  Staggered(Context &c, const Field &g, const Field &g3) : c_(c), nlinks_(8) { check(qexhip_stag_set_links(c.h, g.data(), g3.data())); }
  struct FromHisq {};   // links = HisqCoefs.smear(g) built on the device (hisqLinks.nim:32-43)
  struct FromNhyp { double alpha1, alpha2, alpha3; std::array<int, 4> antiperiodic{0, 0, 0, 1}; };   // rephase(nHYP(g))
  Staggered(Context &c, const Field &g, FromHisq) : c_(c), nlinks_(8) { check(qexhip_stag_set_links_hisq(c.h, g.data())); }
  Staggered(Context &c, const Field &g, const FromNhyp &h) : c_(c), nlinks_(4) {
    check(qexhip_stag_set_links_nhyp(c.h, g.data(), h.alpha1, h.alpha2, h.alpha3, h.antiperiodic.data(), nullptr));
  }
  // storage format chosen for the links: 0 = 18 reals, 1 = 2 rows + sign, 2 = 2 rows + determinant
  int linkFormat() const { int n = 0, f = 0; double d = 0; check(qexhip_stag_links_info(c_.h, &n, &f, &d)); return f; }
  void D(Field &r, const Field &x, double m) { check(qexhip_stag_D(c_.h, r.data(), x.data(), m, 1.0)); }
  void Ddag(Field &r, const Field &x, double m) { check(qexhip_stag_D(c_.h, r.data(), x.data(), m, -1.0)); }
  void stagD(Field &r, const Field &x, int subset, double m, double sc = 1.0, double a = 0.0) { check(qexhip_stag_stagD(c_.h, r.data(), x.data(), subset, m, sc, a)); }
  void eoReduce(Field &r, const Field &b, double m) { check(qexhip_stag_eo_reduce(c_.h, r.data(), b.data(), m)); }
  void eoReconstruct(Field &r, const Field &b, double m) { check(qexhip_stag_eo_reconstruct(c_.h, r.data(), b.data(), m)); }
  void stagD2(Field &r, const Field &x, int subset, double a, double b) { check(qexhip_stag_dslash(c_.h, r.data(), x.data(), subset, a, b)); }
  void stagD2ee(Field &r, const Field &x, double m2) { check(qexhip_stag_op_xx(c_.h, r.data(), x.data(), m2, 1)); }
  void stagD2oo(Field &r, const Field &x, double m2) { check(qexhip_stag_op_xx(c_.h, r.data(), x.data(), m2, 0)); }
  // solveXX(s, r, x, m, sp, parEven): r <- solution, x = rhs (stagSolve.nim:57-132)
  void solveXX(Field &r, const Field &x, double m, SolverParams &sp, bool parEven = true, int histcap = 0) {
    int its = 0; double fin = 0;
    std::vector<double> hist(histcap > 0 ? histcap : 1);
    timed(sp, its, [&] {
      check(qexhip_stag_solve_xx(c_.h, r.data(), x.data(), m, sp.r2req, sp.maxits, parEven ? 1 : 0, &its, &fin, hist.data(), histcap));
    });
    sp.r2 = fin;
    sp.r2hist.assign(hist.begin(), hist.begin() + (histcap > 0 ? std::min(histcap, its + 1) : 0));
  }
  void solveEE(Field &r, const Field &x, double m, SolverParams &sp, int histcap = 0) { solveXX(r, x, m, sp, true, histcap); }
  void solveOO(Field &r, const Field &x, double m, SolverParams &sp, int histcap = 0) { solveXX(r, x, m, sp, false, histcap); }
  // Staggered.solve(x, b, m, sp): full lattice, even-odd preconditioned, true-residual restarts
  void solve(Field &x, const Field &b, double m, SolverParams &sp) {
    int its = 0; double fin = 0;
    timed(sp, its, [&] { check(qexhip_stag_solve(c_.h, x.data(), b.data(), m, sp.r2req, sp.maxits, &its, &fin)); });
    sp.r2 = fin;
  }
  // convenience form used by the reference's tests: s.solve(v2, v1, m, 1e-8) (stagSolve.nim:462-472)
  void solve(Field &x, const Field &b, double m, double res) {
    SolverParams sp; sp.r2req = res * res; sp.maxits = 100000;
    solve(x, b, m, sp);
  }
  // n <= 4 independent solves on these links in lock-step (one stream of the links per sweep for all of them);
  // per system the result of solve(x[j], b[j], m[j], sps[j])
  void solveBatch(std::vector<Field> &xs, const std::vector<Field> &bs, const std::vector<double> &ms, std::vector<SolverParams> &sps) {
    const int n = (int)xs.size();
    std::vector<double *> xp; std::vector<const double *> bp; std::vector<double> rq, fin(n); std::vector<int> its(n);
    int maxits = sps.at(0).maxits;
    for (int j = 0; j < n; j++) { xp.push_back(xs[j].data()); bp.push_back(bs.at(j).data()); rq.push_back(sps.at(j).r2req); maxits = std::min(maxits, sps[j].maxits); }
    auto t0 = std::chrono::steady_clock::now();
    check(qexhip_stag_solve_batch(c_.h, n, xp.data(), bp.data(), ms.data(), rq.data(), maxits, its.data(), fin.data()));
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / n;
    for (int j = 0; j < n; j++) {
      SolverParams &sp = sps[j];
      sp.seconds += dt; sp.calls += 1; sp.iterations += its[j]; sp.iterationsMax = std::max(sp.iterationsMax, its[j]);
      sp.flops += flops(its[j]); sp.r2 = fin[j];
    }
  }
  // multi-mass Staggered.solve(xs, b, ms, sp) (stagSolve.nim:347-446)
  void solve(std::vector<Field> &xs, const Field &b, const std::vector<double> &ms, SolverParams &sp) {
    std::vector<double *> p;
    for (auto &x : xs) p.push_back(x.data());
    int its = 0; double fin = 0;
    timed(sp, its, [&] { check(qexhip_stag_solve_multi(c_.h, p.data(), b.data(), ms.data(), (int)ms.size(), sp.r2req, sp.maxits, &its, &fin)); });
    sp.r2 = fin;
  }
};
inline Staggered newStag(Context &c, const Field &g) { return Staggered(c, g); }
inline Staggered newStag3(Context &c, const Field &g, const Field &g3) { return Staggered(c, g, g3); }

// plaq(g) (gaugeUtils.nim:213-282)
inline std::array<double, 6> plaq(Context &c, const Field &g) {
  std::array<double, 6> p;
  check(qexhip_gauge_set(c.h, g.data()));
  check(qexhip_plaq(c.h, p.data()));
  return p;
}
// g.gaugeFlow(steps, eps): measure(wflowT)  (wflow.nim:21-67); g is modified in place
template <class Measure>
inline void gaugeFlow(Context &c, Field &g, int steps, double eps, Measure &&measure) {
  check(qexhip_gauge_set(c.h, g.data()));
  for (int n = 1; n <= steps; n++) {
    check(qexhip_wflow(c.h, 1, eps));
    measure(n * eps);
  }
  check(qexhip_gauge_get(c.h, g.data()));
}
inline void gaugeFlow(Context &c, Field &g, int steps, double eps) {
  check(qexhip_gauge_set(c.h, g.data()));
  check(qexhip_wflow(c.h, steps, eps));
  check(qexhip_gauge_get(c.h, g.data()));
}

// HypCoefs (hypsmear.nim:15-18): smear, and smearGetForce as an object holding the device-resident closure
struct HypCoefs {
  double alpha1 = 0.4, alpha2 = 0.5, alpha3 = 0.5;
  void smear(Context &c, const Field &g, Field &fl) const { check(qexhip_nhyp_smear(c.h, g.data(), fl.data(), alpha1, alpha2, alpha3)); }
  class SmearedForce {
    Context &c_;
   public:
    SmearedForce(Context &c, const HypCoefs &h, const Field &g, Field *fl) : c_(c) {
      check(qexhip_nhyp_prepare(c.h, g.data(), h.alpha1, h.alpha2, h.alpha3, fl ? fl->data() : nullptr));
    }
    // smear the links resident on the device (ResidentMD below); forces are then left there too: gforceResident / fforceResident
    SmearedForce(Context &c, const HypCoefs &h) : c_(c) { check(qexhip_nhyp_prepare(c.h, nullptr, h.alpha1, h.alpha2, h.alpha3, nullptr)); }
    void gforceResident(double plaq, double rect = 0, double adjplaq = 0) { check(qexhip_nhyp_gauge_force(c_.h, nullptr, plaq, rect, adjplaq)); }
    std::vector<int> fforceResident(const std::vector<Field> &phi, const std::vector<double> &mass, const std::vector<double> &scale, double r2req,
                                    int maxits = 1000000, const std::array<int, 4> &antiperiodic = {0, 0, 0, 1}) {
      std::vector<const double *> p;
      for (auto &v : phi) p.push_back(v.data());
      std::vector<double> rq(p.size(), r2req);
      std::vector<int> its(p.size(), 0);
      check(qexhip_nhyp_fforce(c_.h, nullptr, (int)p.size(), p.data(), mass.data(), scale.data(), rq.data(), maxits, antiperiodic.data(), nullptr, its.data()));
      return its;
    }
    ~SmearedForce() { qexhip_nhyp_release(c_.h); }
    SmearedForce(const SmearedForce &) = delete;
    void operator()(Field &f, const Field &chain) { check(qexhip_nhyp_force(c_.h, f.data(), chain.data())); }   // smearedForce(f, chain)
    void gforce(Field &f, double plaq, double rect = 0, double adjplaq = 0) { check(qexhip_nhyp_gauge_force(c_.h, f.data(), plaq, rect, adjplaq)); }
    void fforce(Field &f, const std::vector<Field> &psi, const std::vector<double> &scale, const std::array<int, 4> &antiperiodic = {0, 0, 0, 1}) {
      std::vector<const double *> p;
      for (auto &v : psi) p.push_back(v.data());
      check(qexhip_nhyp_fermion_force(c_.h, f.data(), p.data(), scale.data(), (int)p.size(), antiperiodic.data(), nullptr));
    }
  };
};
// mdt / mdv / force-gradient shifts on device-resident links and momenta (staghmc_sh.nim:429-640; qexhip_md_*)
class ResidentMD {
  Context &c_;
 public:
  enum Source { Gauge = 0, Nhyp = 1 };                      // which device force buffer a kick / shift applies
  ResidentMD(Context &c, const Field &g, const Field &p) : c_(c) { check(qexhip_md_begin(c.h, g.data(), p.data())); }
  void end(Field *g, Field *p) { check(qexhip_md_end(c_.h, g ? g->data() : nullptr, p ? p->data() : nullptr)); }
  double momentumNorm2() { double r = 0; check(qexhip_md_momentum_norm2(c_.h, &r)); return r; }
  void updateLinks(double t) { check(qexhip_md_update_links(c_.h, t)); }                       // mdt
  void gaugeForce(double plaq, double rect = 0, double adjplaq = 0) { check(qexhip_md_gauge_force(c_.h, plaq, rect, adjplaq)); }
  void kick(Source s, double t) { check(qexhip_md_kick(c_.h, (int)s, t)); }                     // mdv: p += t f
  void shiftLinks(Source s, double t) { check(qexhip_md_shift_links(c_.h, (int)s, t)); }        // fgv / fgvf
  void saveLinks() { check(qexhip_md_save_links(c_.h)); }
  void restoreLinks() { check(qexhip_md_restore_links(c_.h)); }
};
// HisqCoefs.smear(g, fl, ll) (hisqLinks.nim:32-43)
inline void hisqSmear(Context &c, const Field &g, Field &fl, Field &ll) { check(qexhip_hisq_smear(c.h, g.data(), fl.data(), ll.data())); }
// loadGauge / saveGauge (gaugeUtils.nim:87-122)
inline void saveGauge(const Layout &lo, const Field &g, const std::string &fn, char prec = 'D') {
  check(qexhip_io_write_gauge(fn.c_str(), lo.physGeom.data(), g.data(), prec, nullptr, nullptr));
}
inline void loadGauge(const Layout &lo, Field &g, const std::string &fn) {
  unsigned a = 0, b = 0;
  check(qexhip_io_read_gauge(fn.c_str(), lo.physGeom.data(), g.data(), &a, &b));
}

// field algebra used by the reference's tests (fieldET.nim:605-625)
inline double norm2(Context &c, const Field &x, int subset = QEXHIP_ALL) {
  double r = 0;
  check(qexhip_norm2(c.h, x.data(), subset, &r));
  return r;
}
inline double redot(Context &c, const Field &x, const Field &y, int subset = QEXHIP_ALL) {      // fieldET.nim:704-724
  double r = 0;
  check(qexhip_redot(c.h, x.data(), y.data(), subset, &r));
  return r;
}
inline std::complex<double> dot(Context &c, const Field &x, const Field &y, int subset = QEXHIP_ALL) {   // fieldET.nim:677-693
  double r[2] = {0, 0};
  check(qexhip_dot(c.h, x.data(), y.data(), subset, r));
  return {r[0], r[1]};
}

}  // namespace qex
