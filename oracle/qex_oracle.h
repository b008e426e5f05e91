/* qex_oracle.h -- CPU oracle for the staggered Dslash / CG / Wilson-flow hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the algorithms of
 * ctpeterson/qex (reference @ 2025-02-23) for the path named in BASELINE.json.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * it; the product (libqexhip.so) never links or calls it.
 *
 * Parity pin: the restatement is checked against the reference's own
 * known-answer vectors (tests/test_oracle_golden.py):
 *   G1 tests/reprod/trandgauge.nim:17      plaquettes of g.random, 8^4
 *   G2 src/gauge/wflow.nim:135-142         plaquettes after gaugeFlow(6,0.01)
 *   G4 tests/base/tmrg32k3a.nim:9-20       MRG32k3a uniforms
 *   G6 tests/base/tstressplaq.nim:29-66    unit-gauge plaquettes
 * Dslash / CG have no asserted KAT in the reference (SURVEY.md 8c "Gap"); for
 * those rows parity is pinned by this restatement anchored through G1/G2.
 *
 * Data format (identical to the C-ABI host format of include/qexhip.h and to
 * the V=1 twin layout of src/quda/qudaWrapperImpl.nim:198-240):
 *   site index  = MILC even-odd order, src/layout/qlayout.nim:110-131 with V=1:
 *                 lex = x0 + L0*(x1 + L1*(x2 + L2*x3)),  idx = lex/2 + parity*vol/2
 *   colour vector  double[vol][3][2]        (re,im)
 *   gauge field    double[vol][4][3][3][2]  ([site][mu][row][col][re,im])
 */
#ifndef QEX_ORACLE_H
#define QEX_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qo_layout qo_layout;

/* ---- layout (qlayout.nim:110-185, layoutX.nim:285-295) ---- */
qo_layout *qo_layout_new(const int L[4]);
void qo_layout_free(qo_layout *lo);
int qo_vol(const qo_layout *lo);
int qo_index(const qo_layout *lo, const int x[4]);
void qo_coord(const qo_layout *lo, int idx, int x[4]);
/* neighbour tables: index of site x + len*mu (len may be negative) */
int qo_neighbor(const qo_layout *lo, int idx, int mu, int len);

/* ---- RNG (rng/milcrng.nim, rng/mrg32k3a.nim, rng/distributionUtils.nim) ---- */
typedef struct qo_rngfield qo_rngfield;
enum { QO_RNG_MILC6 = 0, QO_RNG_MRG32K3A = 1 };
qo_rngfield *qo_rngfield_new(const qo_layout *lo, int kind, uint64_t seed);
void qo_rngfield_free(qo_rngfield *rf);
/* single-generator test hooks */
void qo_milc6_test(uint32_t seed, uint32_t index, int n, float *uniforms, double *gaussians);
void qo_mrg32k3a_test(uint64_t seed, uint64_t index, int n, double *uniforms);

void qo_field_uniform(const qo_layout *lo, qo_rngfield *rf, int ncomp, double *v, int round_f32);
void qo_vector_gaussian(const qo_layout *lo, qo_rngfield *rf, double *v);
void qo_gauge_gaussian(const qo_layout *lo, qo_rngfield *rf, double *g);
void qo_gauge_random(const qo_layout *lo, qo_rngfield *rf, double *g);  /* gaugeUtils.nim:1424-1429 */
void qo_gauge_warm(const qo_layout *lo, qo_rngfield *rf, double s, double *g); /* :1431-1441 */
void qo_gauge_unit(const qo_layout *lo, double *g);
void qo_gauge_random_tah(const qo_layout *lo, qo_rngfield *rf, double *g); /* randomTAH :1377-1382 */

/* ---- SU(3) helpers exposed for unit tests ---- */
void qo_projectU(double *r, const double *x);   /* matrixFunctions.nim:301-313 */
void qo_projectSU(double *r, const double *x);  /* :359-370 */
void qo_projectTAH(double *r, const double *x); /* :375-380 */
void qo_exp(double *r, const double *m);        /* :436-445 + matexp.nim:634-649,707-710 */

/* ---- BC + phases (gaugeUtils.nim:124-131, stagD.nim:509-520) ---- */
void qo_setBC(const qo_layout *lo, double *g);
void qo_stagPhase(const qo_layout *lo, double *g, const int phases[4]);

/* ---- gauge observables / flow ---- */
void qo_plaq(const qo_layout *lo, const double *g, double out[6]);      /* gaugeUtils.nim:213-282 */
void qo_gauge_force(const qo_layout *lo, const double *g, double *f);   /* gaugeAction.nim:334-350, plaq:1.0 */
void qo_gauge_deriv(const qo_layout *lo, const double *g, double *f, double cplaq); /* :148-204 */
void qo_wflow(const qo_layout *lo, double *g, int nsteps, double eps);  /* wflow.nim:21-67 */

/* ---- general gauge actions: kind 0 plaq+rect (gaugeAction.nim:61-142,148-332), kind 1 plaq+adjplaq (:614-747) ---- */
double qo_gauge_action(const qo_layout *lo, const double *g, double cplaq, double c2, int kind);
void qo_gauge_deriv_rect(const qo_layout *lo, const double *g, double *f, double cplaq, double crect);
void qo_gauge_deriv_adj(const qo_layout *lo, const double *g, double *f, double cplaq, double cadj);
void qo_gauge_force_general(const qo_layout *lo, const double *g, double *f, double cplaq, double c2, int kind);
void qo_wflow_general(const qo_layout *lo, double *g, int nsteps, double eps, double cplaq, double c2, int kind); /* flow/flow.nim:22-90 */

/* ---- link smearing: fat7/HISQ (gauge/fat7l.nim:24-161, physics/hisqLinks.nim:9-43),
 * nHYP forward smearing (gauge/hypsmear.nim:49-144).  coef = {oneLink, threeStaple, fiveStaple, sevenStaple, lepage} ---- */
void qo_fat7(const qo_layout *lo, double *fl, const double *gf, const double coef[5], double *ll, const double *gfLong, double naik);
void qo_hisq_smear(const qo_layout *lo, const double *g, double *fl, double *ll);
void qo_nhyp_smear(const qo_layout *lo, const double *g, double *fl, double a1, double a2, double a3);
/* nHYP smeared-force chain: smearGetForce + smearedForce(f, chain) (gauge/hypsmear.nim:49-247, keepProj),
 * symStapleDeriv (gauge/smearutil.nim:22-50), projectUderiv / sylsolve / inverse
 * (maths/matrixFunctions.nim:329-357, maths/projUderiv.nim:8-147, maths/matinv.nim:90-115).
 * fl (nullable) receives the smeared links; f may alias chain. */
void qo_projectUderiv(double *r, const double *u, const double *x, const double *chain);
void qo_nhyp_force(const qo_layout *lo, const double *g, double *fl, double *f, const double *chain, double a1, double a2, double a3);
/* projTAH(f, g) of the fork (stagg_pv_hmc/staghmc_spv_gforce.nim:256-291): adj=0 TAH(f g^+), adj=1 TAH(g f^+) */
void qo_stag_outer_hop(const qo_layout *lo, double *f, const double *x, double scale_even, double scale_odd, int accumulate, int hop);
/* HISQ force: derivative of makeImpLinks (gauge/fat7lderiv.nim) and the HisqCoefs chain (gauge/hisqsmear.nim:55-90) */
void qo_fat7_deriv(const qo_layout *lo, double *d, const double *gf, const double *cfl, const double coef[5], const double *cll, double naik);
void qo_hisq_force(const qo_layout *lo, const double *g, const double *dsdsu, const double *dsdsul, double *f);
/* MD pieces of the HMC examples (src/examples/staghmc_sh.nim:429-435,247-258) */
void qo_gauge_exp_update(const qo_layout *lo, double *g, const double *p, double t);
void qo_gauge_projectSU(const qo_layout *lo, double *g);
void qo_force_projTAH(const qo_layout *lo, double *f, const double *g, int adj);

/* ---- flow observables (gaugeUtils.nim:1079-1270): out = {E_s, E_t, Q} ---- */
void qo_flow_EQ(const qo_layout *lo, const double *g, int loop, double out[3]);
void qo_s4_gauge(const qo_layout *lo, const double *g, double out[8]);   /* staghmc_spv_meas.nim:25-65 */
void qo_wline(const qo_layout *lo, const double *g, const int *path, int n, double out[2]);

/* ---- field algebra (fieldET.nim:605-625,704-724) ; parity: 0 even, 1 odd, 2 all ---- */
double qo_norm2(const qo_layout *lo, const double *x, int parity);
double qo_redot(const qo_layout *lo, const double *x, const double *y, int parity);

/* ---- staggered operator.  fat: 4 links/site; lng: NULL or 4 long links/site ---- */
/* r = a*r + b*x + (2D)x on `parity` of r   (stagD.nim:349-395) */
void qo_stagD2(const qo_layout *lo, const double *fat, const double *lng,
               double *r, const double *x, int parity, double a, double b);
/* r = a*r + m*x + sc*D*x  (stagD.nim:406-409) */
void qo_stagD(const qo_layout *lo, const double *fat, const double *lng,
              double *r, const double *x, int parity, double m, double sc, double a);
void qo_D(const qo_layout *lo, const double *fat, const double *lng, double *r, const double *x, double m);
void qo_Ddag(const qo_layout *lo, const double *fat, const double *lng, double *r, const double *x, double m);
/* r[par] = 4 m2 x - (2D)(2D) x   (stagD.nim:434-469); par_even=1: ee, 0: oo */
void qo_stagD2xx(const qo_layout *lo, const double *fat, const double *lng,
                 double *r, const double *x, double m2, int par_even);
void qo_eoReduce(const qo_layout *lo, const double *fat, const double *lng, double *r, const double *b, double m);
void qo_eoReconstruct(const qo_layout *lo, const double *fat, const double *lng,
                      double *r, const double *b, double m);

/* fermion-force outer product f[mu](s) (+)= scale * x(s) (x) x(s+mu)^+ (stagD.nim:634-664, staghmc_spv.nim:831-854) */
void qo_stag_outer(const qo_layout *lo, double *f, const double *x, double scale_even, double scale_odd, int accumulate);

/* ---- solvers ---- */
/* solveXX (stagSolve.nim:57-132) = CgState.solve (cg.nim:55-272) with op stagD2ee|oo(m^2).
 * r2hist[k] = r2/b2 after iteration k (k=0: initial), up to histcap entries. */
int qo_solveXX(const qo_layout *lo, const double *fat, const double *lng,
               double *r, const double *x, double m, double r2req, int maxits, int par_even,
               double *r2hist, int histcap, double *final_r2_over_b2);
/* full solve D x = b (stagSolve.nim:224-294); returns total CG iterations; r2_final = |b-Dx|^2/|b|^2 */
int qo_solve(const qo_layout *lo, const double *fat, const double *lng,
             double *x, const double *b, double m, double r2req, int maxits, double *r2_final);
int qo_solve_prev(const qo_layout *lo, const double *fat, const double *lng,
                  double *x, const double *b, double m, double r2req, int maxits, int use_prev, double *r2_final);
/* multi-shift solveXX (stagSolve.nim:296-345 + cgm.nim:84-315): shifts[0] is the base MASS,
 * shifts[k>0] = sigma_k.  xs = nmass pointers to full-volume vectors. */
int qo_solveXX_multi(const qo_layout *lo, const double *fat, const double *lng,
                     double **xs, const double *b, const double *shifts, int nmass,
                     double r2req, int maxits, int par_even, double *r2hist, int histcap);
/* multi-mass solve (stagSolve.nim:347-446) */
int qo_solve_multi(const qo_layout *lo, const double *fat, const double *lng,
                   double **xs, const double *b, const double *masses, int nmass,
                   double r2req, int maxits, double *r2_final);

int qo_num_threads(void);
void qo_set_num_threads(int n);

/* extended-precision (IEEE binary128) twins of qo_solveXX / qo_solveXX_multi: the same algorithms with every vector, operator
 * application, accumulation and scalar in binary128 arithmetic; the yardstick tests/parity_log.py measures both the HIP path and
 * the fp64 oracle against along the chaotic tail of hard systems.  x may be NULL. */
int qo_solveXX_ext(const qo_layout *lo, const double *fat, const double *lng, double *x, const double *b, double m, double r2req,
                   int maxits, int par_even, double *r2hist, int histcap, double *final_r2_over_b2);
int qo_solveXX_multi_ext(const qo_layout *lo, const double *fat, const double *lng, const double *b, const double *shifts, int nmass,
                         double r2req, int maxits, int par_even, double *r2hist, int histcap);

#ifdef __cplusplus
}
#endif
#endif
