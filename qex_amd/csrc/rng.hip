// rng.hip -- per-site random number fields and the configuration generators built on them.
// Host code (no GPU) for everything that defines "the same configuration": the deviates must be bit-identical to QEX's,
// whose gaussians go through libm's log / cos and are then rounded to float32, so they are produced with the host's libm,
// one generator per site.  Round 3 adds, at the end of the file, the device form of the RngMilc6 field for the two ENDS of
// an HMC trajectory (momenta, pseudofermion and pbp sources born in HBM): the integer streams advance bit-identically on
// the GPU (one lane per site, each stream sequential), the deviates go through the device's libm.
//
// Restates (file:line in ctpeterson/qex):
//   RngMilc6: seedX / nextI / uniform / gaussian     src/rng/milcrng.nim:92-110,120-133,150-154,158-193
//   MRG32k3a: seedX / skip-ahead / next / uniform / gaussian   src/rng/mrg32k3a.nim:18-30,103-120,158-187,225-232
//   newRNGField: generator j seeded with (seed, lexicographic index of site j, x fastest)   src/rng/distributionUtils.nim:306-331
//   gaussian / uniform / u1 of a field                src/rng/distributionUtils.nim:23-44,64-97,182-211
//   randTah3, randomTAH, warm, random (= gaussian + projectSU)   src/gauge/gaugeUtils.nim:1348-1446
// The per-site stream is consumed field by field (g[mu].gaussian r for mu = 0..3 in turn).
// Pinned through the same golden sets as the oracle (tests/test_rng_product.py): G1 (random -> plaquettes),
// G4 (MRG32k3a uniforms), G5 (randomTAH norm), G7 (momenta, pseudofermions, u1 sources of the HMC log).
#pragma clang fp contract(off)      // every sum / product rounds as written, as in QEX's generated C for these scalars
#include "../../include/qexhip.h"
#include "su3.h"
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

void qexhip_set_error(const char *fmt, ...);

namespace {
struct Milc6 {
  uint32_t r0, r1, r2, r3, r4, r5, r6, ic, mult;
  void seed(uint32_t seed0, uint32_t index) {
    uint32_t s = seed0;
    auto nxt = [&](uint32_t &x) { s = (69607u + 8u * index) * s + 12345u; x = (s >> 8) & 0x00FFFFFFu; };
    nxt(r0); nxt(r1); nxt(r2); nxt(r3); nxt(r4); nxt(r5); nxt(r6);
    s = (69607u + 8u * index) * s + 12345u;
    ic = s;
    mult = 100005u + 8u * index;
  }
  uint32_t next() {
    const uint32_t t = (((r5 >> 7) | (r6 << 17)) ^ ((r4 >> 1) | (r5 << 23))) & 0x00FFFFFFu;
    r6 = r5; r5 = r4; r4 = r3; r3 = r2; r2 = r1; r1 = r0; r0 = t;
    const uint32_t s = ic * mult + 12345u;
    ic = s;
    return t ^ ((s >> 8) & 0x00FFFFFFu);
  }
  float uniform() { return (1.0f / (float)0x01000000) * (float)next(); }
  double gaussian() {                       // all-double arithmetic, deviate rounded to float32 (pinned by G1 and G5)
    const double v = (double)uniform();
    const double p = (double)uniform() * 2.0 * 3.14159265358979323846;
    const double r = std::sqrt(-2.0 * std::log(v + 9.999999999999999e-308));
    return (double)(float)(r * std::cos(p));
  }
};

const uint64_t M1 = 4294967087ull, M2 = 4294944443ull;
uint32_t a1sq[190][3][3], a2sq[190][3][3];
void mrg_tables() {
  static bool done = false;
  if (done) return;
  auto sq = [](uint32_t x[3][3], uint32_t a[3][3], uint64_t m) {
    for (int i = 0; i < 3; i++) {
      uint64_t t[3] = {0, 0, 0};
      for (int k = 0; k < 3; k++)
        for (int j = 0; j < 3; j++) t[j] += ((uint64_t)a[i][k] * a[k][j]) % m;
      for (int j = 0; j < 3; j++) x[i][j] = (uint32_t)(t[j] % m);
    }
  };
  const uint32_t a1[3][3] = {{0, 1, 0}, {0, 0, 1}, {(uint32_t)(M1 - 810728ull), 1403580u, 0}};
  const uint32_t a2[3][3] = {{0, 1, 0}, {0, 0, 1}, {(uint32_t)(M2 - 1370589ull), 0, 527612u}};
  memcpy(a1sq[0], a1, sizeof a1);
  memcpy(a2sq[0], a2, sizeof a2);
  for (int i = 1; i < 190; i++) { sq(a1sq[i], a1sq[i - 1], M1); sq(a2sq[i], a2sq[i - 1], M2); }
  done = true;
}
struct Mrg {
  uint32_t s1[3], s2[3];
  static void mv(uint32_t a[3][3], uint32_t v[3], uint64_t m) {
    const uint64_t v0 = v[0], v1 = v[1], v2 = v[2];
    for (int i = 0; i < 3; i++) v[i] = (uint32_t)((((uint64_t)a[i][0] * v0) % m + ((uint64_t)a[i][1] * v1) % m + ((uint64_t)a[i][2] * v2) % m) % m);
  }
  void seed(uint64_t sd, uint64_t subseq) {
    if (sd != 0) {
      const uint64_t d1 = 12345ull * (uint64_t)((uint32_t)sd ^ 0x55555555u);
      const uint64_t d2 = 12345ull * (uint64_t)((uint32_t)(sd >> 32) ^ 0xAAAAAAAAu);
      s1[0] = (uint32_t)(d1 % M1); s1[1] = (uint32_t)(d2 % M1); s1[2] = (uint32_t)(d1 % M1);
      s2[0] = (uint32_t)(d2 % M2); s2[1] = (uint32_t)(d1 % M2); s2[2] = (uint32_t)(d2 % M2);
    } else {
      for (int i = 0; i < 3; i++) s1[i] = s2[i] = 12345u;
    }
    int i = 0;
    for (uint64_t s = subseq; s > 0; s >>= 1, i++)
      if (s & 1) { mv(a1sq[76 + i], s1, M1); mv(a2sq[76 + i], s2, M2); }
  }
  int64_t next() {
    int64_t p1 = (1403580ll * (int64_t)s1[1] - 810728ll * (int64_t)s1[0]) % (int64_t)M1;
    if (p1 < 0) p1 += (int64_t)M1;
    s1[0] = s1[1]; s1[1] = s1[2]; s1[2] = (uint32_t)p1;
    int64_t p2 = (527612ll * (int64_t)s2[2] - 1370589ll * (int64_t)s2[0]) % (int64_t)M2;
    if (p2 < 0) p2 += (int64_t)M2;
    s2[0] = s2[1]; s2[1] = s2[2]; s2[2] = (uint32_t)p2;
    return (p1 <= p2) ? p1 - p2 + (int64_t)M1 : p1 - p2;
  }
  double uniform() { return 2.328306549295728e-10 * (double)next(); }
  double gaussian() {
    const double v = uniform();
    const double p = uniform() * 2.0 * 3.14159265358979323846;
    return std::sqrt(-2.0 * std::log(v)) * std::cos(p);
  }
};

struct RngField {
  int kind = 0;                       // 0 RngMilc6, 1 MRG32k3a
  int lat[4];                         // local lattice
  size_t vol = 0;
  std::vector<Milc6> m6;
  std::vector<Mrg> mrg;
  double gaussian(size_t j) { return kind == 0 ? m6[j].gaussian() : mrg[j].gaussian(); }
  double uniform(size_t j) { return kind == 0 ? (double)m6[j].uniform() : mrg[j].uniform(); }
};

template <class F> void for_sites(size_t n, F &&f) {
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt ? (nt > 16 ? 16 : nt) : 1;
  if (n < 4096) nt = 1;
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++)
    th.emplace_back([=, &f] { for (size_t j = n * t / nt; j < n * (t + 1) / nt; j++) f(j); });
  for (auto &x : th) x.join();
}
inline M3 load18(const double *p) { M3 m; for (int k = 0; k < 9; k++) m.e[k] = make_double2(p[2 * k], p[2 * k + 1]); return m; }
inline void store18(double *p, const M3 &m) { for (int k = 0; k < 9; k++) { p[2 * k] = m.e[k].x; p[2 * k + 1] = m.e[k].y; } }

// randTah3 (gaugeUtils.nim:1356-1375)
void rand_tah3(double *m, RngField &R, size_t j) {
  const double s2 = 0.70710678118654752440, s3 = 0.57735026918962576450;
  const double r3 = s2 * R.gaussian(j);
  const double r8 = s2 * s3 * R.gaussian(j);
  for (int k = 0; k < 18; k++) m[k] = 0.0;
  m[1] = r8 + r3; m[9] = r8 - r3; m[17] = -2 * r8;
  const double r01 = s2 * R.gaussian(j), r02 = s2 * R.gaussian(j), r12 = s2 * R.gaussian(j);
  const double i01 = s2 * R.gaussian(j), i02 = s2 * R.gaussian(j), i12 = s2 * R.gaussian(j);
  m[2] = r01; m[3] = i01; m[6] = -r01; m[7] = i01;
  m[4] = r02; m[5] = i02; m[12] = -r02; m[13] = i02;
  m[10] = r12; m[11] = i12; m[14] = -r12; m[15] = i12;
}
}  // namespace

struct qexhip_rng : RngField {};

// lat: LOCAL lattice of this rank; glat: global lattice; t_offset: first global t of the local slab (0 on one rank)
extern "C" int qexhip_rng_new(qexhip_rng **out, int kind, unsigned long long seed, const int lat[4], const int glat[4], int t_offset) {
  if (!out || !lat || kind < 0 || kind > 1) return QEXHIP_ERR_ARG;
  const int *G = glat ? glat : lat;
  for (int i = 0; i < 4; i++)
    if (lat[i] < 2 || (lat[i] & 1) || lat[i] > 1024 || G[i] < lat[i] || (i < 3 && G[i] != lat[i])) {
      qexhip_set_error("rng_new: local extents must be even, 2..1024, and equal the global ones except in t (dim %d: local %d, global %d)", i, lat[i], G[i]);
      return QEXHIP_ERR_ARG;
    }
  if (t_offset < 0 || t_offset + lat[3] > G[3]) { qexhip_set_error("rng_new: slab [%d, %d) outside the global t extent %d", t_offset, t_offset + lat[3], G[3]); return QEXHIP_ERR_ARG; }
  auto *R = new (std::nothrow) qexhip_rng();
  if (!R) return QEXHIP_ERR_ARG;
  R->kind = kind;
  for (int i = 0; i < 4; i++) R->lat[i] = lat[i];
  R->vol = (size_t)lat[0] * lat[1] * lat[2] * lat[3];
  if (kind == 0) R->m6.resize(R->vol); else { mrg_tables(); R->mrg.resize(R->vol); }
  const size_t vol = R->vol;
  for_sites(vol, [&](size_t lexl) {
    int x[4];
    size_t r = lexl;
    for (int i = 0; i < 4; i++) { x[i] = (int)(r % lat[i]); r /= lat[i]; }
    const size_t j = lexl / 2 + (((x[0] + x[1] + x[2] + x[3]) & 1) ? vol / 2 : 0);      // V=1 even-odd index of the LOCAL lattice
    const uint64_t lexg = x[0] + (uint64_t)G[0] * (x[1] + (uint64_t)G[1] * (x[2] + (uint64_t)G[2] * (x[3] + t_offset)));
    if (kind == 0) R->m6[j].seed((uint32_t)seed, (uint32_t)lexg);     // seedIndep narrows to uint32 (milcrng.nim:111-112)
    else R->mrg[j].seed(seed, lexg);
  });
  *out = R;
  return 0;
}
extern "C" int qexhip_rng_free(qexhip_rng *R) { delete R; return 0; }
// x.uniform r: ncomp reals per site in storage order
extern "C" int qexhip_rng_uniform(qexhip_rng *R, int ncomp, double *v) {
  if (!R || !v || ncomp < 1) return QEXHIP_ERR_ARG;
  for_sites(R->vol, [&](size_t j) { for (int k = 0; k < ncomp; k++) v[(size_t)ncomp * j + k] = R->uniform(j); });
  return 0;
}
// v.gaussian r for a colour vector [vol][3][2]
extern "C" int qexhip_rng_gaussian_vector(qexhip_rng *R, double *v) {
  if (!R || !v) return QEXHIP_ERR_ARG;
  for_sites(R->vol, [&](size_t j) { for (int k = 0; k < 6; k++) v[6 * j + k] = R->gaussian(j); });
  return 0;
}
// v.u1 r: each colour component exp(2 pi i u)
extern "C" int qexhip_rng_u1_vector(qexhip_rng *R, double *v) {
  if (!R || !v) return QEXHIP_ERR_ARG;
  for_sites(R->vol, [&](size_t j) {
    for (int k = 0; k < 3; k++) {
      const double n = 2.0 * 3.14159265358979323846 * R->uniform(j);
      v[6 * j + 2 * k] = std::cos(n);
      v[6 * j + 2 * k + 1] = std::sin(n);
    }
  });
  return 0;
}
// p.randomTAH r: gauge-shaped field [vol][4][3][3][2], direction by direction
extern "C" int qexhip_rng_random_tah(qexhip_rng *R, double *p) {
  if (!R || !p) return QEXHIP_ERR_ARG;
  for (int mu = 0; mu < 4; mu++) for_sites(R->vol, [&](size_t j) { rand_tah3(p + (j * 4 + mu) * 18, *R, j); });
  return 0;
}
// g.random r: gaussian matrices, then projectSU
extern "C" int qexhip_rng_gauge_random(qexhip_rng *R, double *g) {
  if (!R || !g) return QEXHIP_ERR_ARG;
  for (int mu = 0; mu < 4; mu++)
    for_sites(R->vol, [&](size_t j) {
      double *m = g + (j * 4 + mu) * 18;
      for (int k = 0; k < 18; k++) m[k] = R->gaussian(j);
      store18(m, m3_projectSU(load18(m)));
    });
  return 0;
}
// g.warm s, r: exp(s * randomTAH)
extern "C" int qexhip_rng_gauge_warm(qexhip_rng *R, double s, double *g) {
  if (!R || !g) return QEXHIP_ERR_ARG;
  for (int mu = 0; mu < 4; mu++)
    for_sites(R->vol, [&](size_t j) {
      double *m = g + (j * 4 + mu) * 18;
      rand_tah3(m, *R, j);
      M3 t = load18(m);
      for (int k = 0; k < 9; k++) { t.e[k].x *= s; t.e[k].y *= s; }
      store18(m, m3_exp(t));
    });
  return 0;
}

// generator state of every site, for checkpoints (the fork writes the RngMilc6 field with the generic Writer,
// src/stagg_pv_hmc/staghmc_spv_rng.nim:135-182): RngMilc6 = 9 uint32 per site in the struct's order
// r0..r6, icState, multiplier (src/rng/milcrng.nim:12-14); MRG32k3a = 6 uint32 (s1[3], s2[3])
extern "C" int qexhip_rng_state_words(qexhip_rng *R) { return R ? (R->kind == 0 ? 9 : 6) : 0; }
extern "C" int qexhip_rng_get_state(qexhip_rng *R, unsigned *out) {
  if (!R || !out) return QEXHIP_ERR_ARG;
  for (size_t j = 0; j < R->vol; j++) {
    if (R->kind == 0) { const Milc6 &m = R->m6[j]; const uint32_t w[9] = {m.r0, m.r1, m.r2, m.r3, m.r4, m.r5, m.r6, m.ic, m.mult}; memcpy(out + 9 * j, w, sizeof w); }
    else { memcpy(out + 6 * j, R->mrg[j].s1, 12); memcpy(out + 6 * j + 3, R->mrg[j].s2, 12); }
  }
  return 0;
}
extern "C" int qexhip_rng_set_state(qexhip_rng *R, const unsigned *in) {
  if (!R || !in) return QEXHIP_ERR_ARG;
  for (size_t j = 0; j < R->vol; j++) {
    if (R->kind == 0) { Milc6 &m = R->m6[j]; const unsigned *w = in + 9 * j; m.r0 = w[0]; m.r1 = w[1]; m.r2 = w[2]; m.r3 = w[3]; m.r4 = w[4]; m.r5 = w[5]; m.r6 = w[6]; m.ic = w[7]; m.mult = w[8]; }
    else { memcpy(R->mrg[j].s1, in + 6 * j, 12); memcpy(R->mrg[j].s2, in + 6 * j + 3, 12); }
  }
  return 0;
}


// ================= device-side generation from an RngMilc6 field (the two ends of a trajectory, round 3) =================
// refresh / measure of the HMC drivers (src/examples/staghmc_sh.nim:716-757,774-789; stagg_pv_hmc/staghmc_spv.nim:1180-1250):
//   p.randomTAH r;  psi[k][i].gaussian r;  eta.u1 r      (gaugeUtils.nim:1356-1383, distributionUtils.nim:64-97,182-211)
// with the fields written straight into HBM: the generator states go up (36 B per site), one lane per site advances ITS
// stream sequentially -- integer arithmetic, so the states that come back are bit for bit the host generator's after the
// same draws (tests assert it through qexhip_rng_get_state) -- and the deviates are formed with the device's fp64 log / cos /
// sqrt.  Those agree with glibc's to the last bit or two; after the float32 rounding of RngMilc6.gaussian a deviate therefore
// differs from the host's (by one float32 ulp) only when the double lands within ~1e-16 of a float32 rounding boundary:
// about once per 1e8 deviates (tests/test_rng_product.py counts them).  u1 phases carry no float32 rounding: 1e-16 agreement.
#include "qexhip_internal.h"

__device__ __forceinline__ uint32_t milc6_next(uint32_t *w) {     // w: r0..r6, ic, mult (milcrng.nim:120-133)
  const uint32_t t = (((w[5] >> 7) | (w[6] << 17)) ^ ((w[4] >> 1) | (w[5] << 23))) & 0x00FFFFFFu;
  w[6] = w[5]; w[5] = w[4]; w[4] = w[3]; w[3] = w[2]; w[2] = w[1]; w[1] = w[0]; w[0] = t;
  const uint32_t s = w[7] * w[8] + 12345u;
  w[7] = s;
  return t ^ ((s >> 8) & 0x00FFFFFFu);
}
__device__ __forceinline__ float milc6_uniform(uint32_t *w) { return (1.0f / (float)0x01000000) * (float)milc6_next(w); }
__device__ __forceinline__ double milc6_gaussian(uint32_t *w) {
  const double v = (double)milc6_uniform(w);
  const double p = (double)milc6_uniform(w) * 2.0 * 3.14159265358979323846;
  const double r = sqrt(-2.0 * log(v + 9.999999999999999e-308));
  return (double)(float)(r * cos(p));
}
// what: 0 gaussian colour vector, 1 u1 colour vector (-> field v, one parity half after the other);
//       2 randomTAH (-> natural-layout matrix field P: [parity][tile][mu][9][64])
__global__ void __launch_bounds__(256) k_rng_milc6(Geom g, uint32_t *__restrict__ state, int what, double2 *v, size_t vhalf, double2 *P) {
  const int i = blockIdx.x * 256 + threadIdx.x;          // host site index j = c + parity * Vh
  if (i >= g.V) return;
  uint32_t w[9];
#pragma unroll
  for (int k = 0; k < 9; k++) w[k] = state[(size_t)9 * i + k];
  const int p = i >= g.Vh, c = i - p * g.Vh;
  if (what == 2) {
    const double s2 = 0.70710678118654752440, s3 = 0.57735026918962576450;
#pragma unroll 1
    for (int mu = 0; mu < 4; mu++) {                      // randTah3, draws in the order r3 r8 r01 r02 r12 i01 i02 i12
      const double r3 = s2 * milc6_gaussian(w);
      const double r8 = s2 * s3 * milc6_gaussian(w);
      const double r01 = s2 * milc6_gaussian(w), r02 = s2 * milc6_gaussian(w), r12 = s2 * milc6_gaussian(w);
      const double i01 = s2 * milc6_gaussian(w), i02 = s2 * milc6_gaussian(w), i12 = s2 * milc6_gaussian(w);
      double2 *d = P + (((size_t)p * g.etile + (c >> 6)) * 4 + mu) * 576 + (c & 63);
      d[0 * 64] = make_double2(0.0, r8 + r3); d[1 * 64] = make_double2(r01, i01); d[2 * 64] = make_double2(r02, i02);
      d[3 * 64] = make_double2(-r01, i01);    d[4 * 64] = make_double2(0.0, r8 - r3); d[5 * 64] = make_double2(r12, i12);
      d[6 * 64] = make_double2(-r02, i02);    d[7 * 64] = make_double2(-r12, i12);    d[8 * 64] = make_double2(0.0, -2 * r8);
    }
  } else {
    double2 *d = v + (size_t)p * vhalf + vec_off(c, 0);
#pragma unroll 1
    for (int k = 0; k < 3; k++) {
      if (what == 0) {
        const double re = milc6_gaussian(w);
        const double im = milc6_gaussian(w);
        d[k * 64] = make_double2(re, im);
      } else {
        const double n = 2.0 * 3.14159265358979323846 * (double)milc6_uniform(w);
        d[k * 64] = make_double2(cos(n), sin(n));
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 9; k++) state[(size_t)9 * i + k] = w[k];
}
// states up, one kernel, states down (stream-ordered; the host copy of the field stays the single source of truth)
int rng_dev_generate(qexhip_ctx *c, qexhip_rng *R, int what, DevField *f, double2 *P) {
  if (!c || !R) return QEXHIP_ERR_ARG;
  if (R->kind != 0) { qexhip_set_error("device-side generation is implemented for RngMilc6 fields (MRG32k3a: use the host generators)"); return QEXHIP_ERR_ARG; }
  if (R->vol != (size_t)c->g.V || R->lat[0] != c->g.X[0] || R->lat[1] != c->g.X[1] || R->lat[2] != c->g.X[2] || R->lat[3] != c->g.X[3]) {
    qexhip_set_error("the RNG field's lattice is not the context's local lattice");
    return QEXHIP_ERR_ARG;
  }
  HIPCHK(hipSetDevice(c->device));
  const size_t bytes = R->vol * 9 * sizeof(uint32_t);
  std::vector<uint32_t> h(R->vol * 9);
  CHK(qexhip_rng_get_state(R, h.data()));
  CHK(ensure_stage(c, bytes));
  uint32_t *ds = (uint32_t *)c->stage;
  HIPCHK(hipMemcpyAsync(ds, h.data(), bytes, hipMemcpyHostToDevice, c->stream));
  k_rng_milc6<<<(c->g.V + 255) / 256, 256, 0, c->stream>>>(c->g, ds, what, f ? f->d : nullptr, f ? f->half : 0, P);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(h.data(), ds, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return qexhip_rng_set_state(R, h.data());
}
