// test_host_san.cpp -- the HOST side of the product and the oracle under AddressSanitizer + UBSan (CPU only: GPU ASan and
// XNACK are not available on this pool).  Built by tests/test_sanitizers.py against a host-only (-fsanitize) build of every
// source of libqexhip (hipcc --offload-host-only: no device code, no kernel is ever launched) and a sanitized build of
// oracle/qex_oracle.c.  What runs here:
//   1. the index / table builders the kernels depend on (site_index.h through the debug hooks, tile_order_plane_host) on
//      small, odd-shaped and LARGE lattices (the 64-bit keys of the visiting order: extents up to 1024)
//   2. the host generators of csrc/rng.hip (RngMilc6, MRG32k3a fields; states in / out; sharded construction)
//   3. csrc/scidac_io.cpp: round trips in both precisions, field records, metadata -- and a fuzz loop over truncated and
//      corrupted LIME files (readerQiolite.nim, crc32.nim are what it restates): every malformed input must come back as an
//      error code, never as a crash, an out-of-bounds access or an allocation of attacker-chosen size
//   4. the error paths of the handle entry points without a GPU (qexhip_init must fail, loudly, and leak nothing)
//   5. the oracle: layout, generators, plaquette (G1 value), D, CG, one flow step, nHYP smear + force on 4^4
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "qexhip.h"
extern "C" {
#include "qex_oracle.h"
}

static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { fails++; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

static std::vector<unsigned char> slurp(const std::string &p) {
  std::vector<unsigned char> b;
  FILE *f = std::fopen(p.c_str(), "rb");
  if (!f) return b;
  std::fseek(f, 0, SEEK_END);
  long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  b.resize((size_t)n);
  if (n > 0 && std::fread(b.data(), 1, (size_t)n, f) != (size_t)n) b.clear();
  std::fclose(f);
  return b;
}
static void spit(const std::string &p, const unsigned char *d, size_t n) {
  FILE *f = std::fopen(p.c_str(), "wb");
  if (n) std::fwrite(d, 1, n, f);
  std::fclose(f);
}
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static void test_index_hooks() {
  const int lats[][4] = {{2, 2, 2, 2}, {4, 6, 10, 6}, {8, 8, 8, 8}, {12, 2, 2, 4}, {6, 6, 6, 8}, {16, 4, 2, 6}, {8, 4, 4, 4}, {16, 8, 2, 12}};
  for (auto &L : lats)
    for (int depth = 1; depth <= 3; depth += 2)
      for (int halo = 0; halo < 2; halo++) {
        if (L[3] < depth) continue;
        int geo[8];
        const int rc = qexhip_debug_geom(L, depth, halo, geo);
        const int Fq = L[0] / 2 * L[1] * L[2];
        if (halo && (Fq % 64 != 0 || L[3] < 3)) { CHECK(rc != 0, "t-sharding needs whole tiles per slice: must be refused"); continue; }
        CHECK(rc == 0, "geom");
        const int vh = geo[0], F = geo[1], ext = vh + (halo ? 2 * depth * F : 0);
        for (int p = 0; p < 2; p++)
          for (int c = 0; c < vh; c++) {
            int x[4];
            CHECK(qexhip_debug_site_coord(L, c, p, x) == 0, "coord");
            CHECK(((x[0] + x[1] + x[2] + x[3]) & 1) == p, "parity of site %d", c);
            for (int mu = 0; mu < 4; mu++)
              for (int hop : {1, -1, 3, -3}) {
                if (std::abs(hop) > depth && halo && mu == 3) continue;
                if (std::abs(hop) == 3 && L[mu] < 3) continue;
                const int n = qexhip_debug_nbr_pos(L, depth, halo, c, p, mu, hop);
                CHECK(n >= 0 && n < ext, "nbr_pos %d out of [0,%d)", n, ext);
              }
          }
        CHECK(qexhip_debug_nbr_pos(L, depth, halo, vh, 0, 0, 1) < 0 && qexhip_debug_nbr_pos(L, depth, halo, -1, 0, 0, 1) < 0, "range check");
      }
  int bad[4] = {3, 4, 4, 4}, geo[8];
  CHECK(qexhip_debug_geom(bad, 1, 0, geo) != 0, "odd extent must be refused");
}

static void test_tile_orders() {
  // permutation property on small / ragged shapes, and on shapes whose packed keys need more than 32 bits
  const int lats[][4] = {{4, 6, 10, 6}, {8, 8, 8, 8}, {32, 32, 32, 8}, {64, 64, 64, 16}, {1024, 2, 2, 2}, {2, 2, 2, 1024}, {2, 1024, 2, 16}, {128, 128, 16, 16}};
  for (auto &L : lats) {
    int geo[8];
    CHECK(qexhip_debug_geom(L, 1, 0, geo) == 0, "geom");
    const int ntile = geo[2], n = 8 * ((2 * ntile + 7) / 8);
    std::vector<int> tab((size_t)n);
    std::vector<char> seen((size_t)2 * ntile);
    for (int mu = 0; mu < 4; mu++)
      for (int nu = 0; nu < 4; nu++) {
        if (mu == nu) { CHECK(qexhip_debug_tile_order(L, mu, nu, tab.data(), n) < 0, "mu == nu"); continue; }
        CHECK(qexhip_debug_tile_order(L, mu, nu, tab.data(), n - 1) == -n, "capacity check");
        CHECK(qexhip_debug_tile_order(L, mu, nu, tab.data(), n) == n, "size");
        std::fill(seen.begin(), seen.end(), 0);
        int cnt = 0;
        for (int e : tab) {
          if (e < 0) continue;
          CHECK(e < 2 * ntile && !seen[(size_t)e], "entry %d twice / out of range", e);
          if (e >= 0 && e < 2 * ntile) seen[(size_t)e] = 1;
          cnt++;
        }
        CHECK(cnt == 2 * ntile, "%dx%dx%dx%d plane (%d,%d): %d of %d entries", L[0], L[1], L[2], L[3], mu, nu, cnt, 2 * ntile);
      }
  }
}

static void test_rng() {
  const int lat[4] = {4, 4, 4, 8}, loc[4] = {4, 4, 4, 4};
  const int vol = 4 * 4 * 4 * 8, lvol = vol / 2;
  for (int kind = 0; kind < 2; kind++) {
    qexhip_rng *R = nullptr, *A = nullptr, *B = nullptr;
    CHECK(qexhip_rng_new(&R, kind, 987654321ull, lat, nullptr, 0) == 0, "rng_new");
    CHECK(qexhip_rng_new(&A, kind, 987654321ull, loc, lat, 0) == 0 && qexhip_rng_new(&B, kind, 987654321ull, loc, lat, 4) == 0, "sharded rng");
    std::vector<double> g((size_t)vol * 72), ga((size_t)lvol * 72), gb((size_t)lvol * 72), v((size_t)vol * 6), p((size_t)vol * 72), u((size_t)vol * 3);
    CHECK(qexhip_rng_gauge_random(R, g.data()) == 0 && qexhip_rng_gauge_random(A, ga.data()) == 0 && qexhip_rng_gauge_random(B, gb.data()) == 0, "gauge_random");
    // the two slabs hold the global field's links (site order differs: compare the multiset through a checksum of squares)
    double s = 0, sa = 0;
    for (double t : g) s += t * t;
    for (double t : ga) sa += t * t;
    for (double t : gb) sa += t * t;
    CHECK(std::fabs(s - sa) < 1e-9 * s, "sharded generation: %g vs %g", s, sa);
    CHECK(qexhip_rng_gaussian_vector(R, v.data()) == 0 && qexhip_rng_u1_vector(R, v.data()) == 0 && qexhip_rng_random_tah(R, p.data()) == 0, "vectors");
    CHECK(qexhip_rng_uniform(R, 3, u.data()) == 0 && qexhip_rng_gauge_warm(R, 0.5, g.data()) == 0, "uniform / warm");
    for (double t : u) CHECK(t >= 0.0 && t < 1.0, "uniform out of range");
    const int nw = qexhip_rng_state_words(R);
    CHECK(nw == (kind == 0 ? 9 : 6), "state words %d", nw);
    std::vector<unsigned> st((size_t)vol * nw);
    CHECK(qexhip_rng_get_state(R, st.data()) == 0, "get_state");
    std::vector<double> a((size_t)vol * 6), b((size_t)vol * 6);
    CHECK(qexhip_rng_gaussian_vector(R, a.data()) == 0 && qexhip_rng_set_state(R, st.data()) == 0 && qexhip_rng_gaussian_vector(R, b.data()) == 0, "replay");
    CHECK(std::memcmp(a.data(), b.data(), a.size() * sizeof(double)) == 0, "state restore replays the stream");
    CHECK(qexhip_rng_free(R) == 0 && qexhip_rng_free(A) == 0 && qexhip_rng_free(B) == 0, "free");
  }
  qexhip_rng *R = nullptr;
  const int bad[4] = {4, 4, 4, 0};
  CHECK(qexhip_rng_new(&R, 0, 1, bad, nullptr, 0) != 0 && qexhip_rng_new(&R, 7, 1, lat, nullptr, 0) != 0 && qexhip_rng_new(nullptr, 0, 1, lat, nullptr, 0) != 0, "bad arguments refused");
}

static void test_io_and_fuzz(const std::string &dir) {
  const int lat[4] = {4, 4, 2, 6};
  const int vol = 4 * 4 * 2 * 6;
  qexhip_rng *R = nullptr;
  CHECK(qexhip_rng_new(&R, 0, 12345, lat, nullptr, 0) == 0, "rng");
  std::vector<double> g((size_t)vol * 72), h((size_t)vol * 72);
  qexhip_rng_gauge_random(R, g.data());
  const std::string fd = dir + "/g_d.lime", ff = dir + "/g_f.lime", fz = dir + "/fuzz.lime", fr = dir + "/rng.lime";
  unsigned sa = 0, sb = 0;
  CHECK(qexhip_io_write_gauge(fd.c_str(), lat, g.data(), 'D', "<file/>", "<record/>") == 0, "write D: %s", qexhip_last_error());
  CHECK(qexhip_io_write_gauge(ff.c_str(), lat, g.data(), 'F', nullptr, nullptr) == 0, "write F");
  int l2[4]; char prec = 0; int cks = 0;
  CHECK(qexhip_io_gauge_info(fd.c_str(), l2, &prec, &cks) == 0 && prec == 'D' && cks == 1 && std::memcmp(l2, lat, sizeof(lat)) == 0, "info");
  CHECK(qexhip_io_read_gauge(fd.c_str(), lat, h.data(), &sa, &sb) == 0 && std::memcmp(g.data(), h.data(), g.size() * 8) == 0, "D round trip");
  CHECK(qexhip_io_read_gauge(ff.c_str(), lat, h.data(), &sa, &sb) == 0, "F read");
  double e = 0;
  for (size_t i = 0; i < g.size(); i++) e = std::fmax(e, std::fabs(g[i] - h[i]));
  CHECK(e < 1e-6, "F round trip %g", e);
  std::vector<double> slab((size_t)vol / 3 * 72);
  CHECK(qexhip_io_read_gauge_slab(fd.c_str(), lat, 2, 2, slab.data()) == 0, "slab");
  CHECK(qexhip_io_read_gauge_slab(fd.c_str(), lat, 5, 2, slab.data()) != 0 && qexhip_io_read_gauge_slab(fd.c_str(), lat, 1, 2, slab.data()) != 0, "bad slabs refused");
  char fm[64], rm[64]; int fl = 0, rl = 0;
  CHECK(qexhip_io_metadata(fd.c_str(), fm, 64, rm, 64, &fl, &rl) == 0 && std::string(fm) == "<file/>" && std::string(rm) == "<record/>", "metadata");
  CHECK(qexhip_io_metadata(fd.c_str(), fm, 3, rm, 1, &fl, &rl) == 0 && std::strlen(fm) <= 2, "metadata truncation");
  const int nw = qexhip_rng_state_words(R);
  std::vector<unsigned> st((size_t)vol * nw), st2((size_t)vol * nw);
  qexhip_rng_get_state(R, st.data());
  CHECK(qexhip_io_write_field(fr.c_str(), lat, st.data(), 4 * nw, 4, "QDP_RngMilc6", 'F', 0, 1, nullptr, nullptr) == 0, "field write");
  char dt[64];
  CHECK(qexhip_io_read_field(fr.c_str(), lat, st2.data(), 4 * nw, 4, dt) == 0 && st == st2 && std::string(dt) == "QDP_RngMilc6", "field round trip");
  CHECK(qexhip_io_read_field(fr.c_str(), lat, st2.data(), 4 * nw + 4, 4, dt) != 0, "wrong site size refused");
  int wrong[4] = {4, 4, 2, 8};
  CHECK(qexhip_io_read_gauge(fd.c_str(), wrong, h.data(), &sa, &sb) != 0, "wrong lattice refused");
  CHECK(qexhip_io_read_gauge((dir + "/absent").c_str(), lat, h.data(), &sa, &sb) != 0, "missing file refused");
  qexhip_rng_free(R);

  // ---- fuzz: truncations at every record boundary +-, random truncations, bit flips, length-field attacks ----
  const std::vector<unsigned char> good = slurp(fd);
  CHECK(good.size() > 1000, "file read back");
  auto probe = [&](const std::vector<unsigned char> &b) {
    spit(fz, b.data(), b.size());
    int li[4]; char p = 0; int c = 0;
    unsigned a2 = 0, b2 = 0;
    (void)qexhip_io_gauge_info(fz.c_str(), li, &p, &c);
    const int rc = qexhip_io_read_gauge(fz.c_str(), lat, h.data(), &a2, &b2);
    (void)qexhip_io_read_gauge_slab(fz.c_str(), lat, 0, 2, slab.data());
    char fm2[32], rm2[32]; int x = 0, y = 0;
    (void)qexhip_io_metadata(fz.c_str(), fm2, 32, rm2, 32, &x, &y);
    char d2[64];
    (void)qexhip_io_read_field(fz.c_str(), lat, h.data(), 576, 8, d2);
    return rc;
  };
  int rejected = 0, total = 0;
  for (size_t n = 0; n < good.size(); n += (n < 2048 ? 1 : 997)) {            // every prefix of the headers, then strides
    std::vector<unsigned char> b(good.begin(), good.begin() + (long)n);
    total++;
    if (probe(b) != 0) rejected++;
  }
  CHECK(rejected == total, "every truncated file must be refused: %d of %d", rejected, total);
  int flips_ok = 0;
  for (int k = 0; k < 600; k++) {
    std::vector<unsigned char> b = good;
    const size_t where = k < 300 ? (size_t)(rnd() % 2048) : (size_t)(rnd() % b.size());     // headers get half of the flips
    b[where] ^= (unsigned char)(1u << (rnd() & 7));
    if (probe(b) == 0) flips_ok++;            // a flip in padding / ignored header bytes may legitimately pass
  }
  std::printf("fuzz: %d truncations refused, %d of 600 single-bit flips still read back (padding / unused header bytes)\n", total, flips_ok);
  CHECK(flips_ok < 200, "most flips must be caught (checksums, magic, lengths): %d", flips_ok);
  for (int k = 0; k < 64; k++) {               // 8-byte big-endian record length of each LIME header <- huge / negative values
    std::vector<unsigned char> b = good;
    size_t off = 0, rec = 0;
    while (off + 144 <= b.size() && rec < (size_t)(k % 8)) {                  // walk to record k % 8
      uint64_t len = 0;
      for (int i = 0; i < 8; i++) len = (len << 8) | b[off + 8 + i];
      off += 144 + ((len + 7) / 8) * 8;
      rec++;
    }
    if (off + 144 > b.size()) continue;
    const uint64_t evil[] = {~0ull, 1ull << 62, 1ull << 40, (uint64_t)b.size() * 2, 0ull, 0x7fffffffffffffffull, 1ull << 32, (1ull << 31) - 1};
    const uint64_t v = evil[k / 8];
    for (int i = 0; i < 8; i++) b[off + 8 + i] = (unsigned char)(v >> (56 - 8 * i));
    (void)probe(b);                              // must return (any code), not crash / over-allocate
  }
  std::vector<unsigned char> junk(4096);
  for (auto &c : junk) c = (unsigned char)rnd();
  CHECK(probe(junk) != 0, "random bytes refused");
  CHECK(probe(std::vector<unsigned char>()) != 0, "empty file refused");
}

static void test_no_gpu_error_paths() {
  int n = -1;
  (void)qexhip_device_count(&n);
  if (n > 0) { std::printf("a GPU is visible: skipping the no-GPU error paths\n"); return; }
  qexhip_handle h = nullptr;
  const int lat[4] = {8, 8, 8, 8}, geom[4] = {1, 1, 1, 1}, coord[4] = {0, 0, 0, 0};
  CHECK(qexhip_init(&h, 0, lat, geom, coord) != 0 && h == nullptr, "init without a GPU must fail");
  CHECK(std::strlen(qexhip_last_error()) > 0, "error string");
  double out[6];
  CHECK(qexhip_plaq(nullptr, out) != 0 && qexhip_stag_D(nullptr, out, out, 0.1, 1.0) != 0 && qexhip_finalize(nullptr) != 0, "NULL handle refused");
}

static void test_oracle() {
  const int L[4] = {4, 4, 4, 4};
  qo_layout *lo = qo_layout_new(L);
  const int vol = qo_vol(lo);
  CHECK(vol == 256, "vol");
  qo_rngfield *rf = qo_rngfield_new(lo, 0, 987654321ull);
  std::vector<double> g((size_t)vol * 72), g3((size_t)vol * 72), x((size_t)vol * 6), r((size_t)vol * 6), f((size_t)vol * 72), fl((size_t)vol * 72), ll((size_t)vol * 72);
  qo_gauge_random(lo, rf, g.data());
  double pl[6];
  qo_plaq(lo, g.data(), pl);
  for (double p : pl) CHECK(std::isfinite(p) && std::fabs(p) < 0.05, "plaq %g", p);
  qo_gauge_force(lo, g.data(), f.data());
  std::vector<double> gw = g;
  qo_wflow(lo, gw.data(), 1, 0.01);
  double pw[6];
  qo_plaq(lo, gw.data(), pw);
  double s0 = 0, s1 = 0;
  for (int k = 0; k < 6; k++) { s0 += pl[k]; s1 += pw[k]; }
  CHECK(s1 > s0, "the flow raises the plaquette");
  qo_nhyp_smear(lo, g.data(), fl.data(), 0.4, 0.5, 0.5);
  qo_nhyp_force(lo, g.data(), fl.data(), f.data(), gw.data(), 0.4, 0.5, 0.5);
  qo_hisq_smear(lo, g.data(), fl.data(), ll.data());
  double eq[3];
  qo_flow_EQ(lo, g.data(), 1, eq);
  std::vector<double> gp = g;
  const int ph[4] = {8, 9, 11, 0};
  qo_setBC(lo, gp.data());
  qo_stagPhase(lo, gp.data(), ph);
  qo_vector_gaussian(lo, rf, x.data());
  qo_D(lo, gp.data(), nullptr, r.data(), x.data(), 0.1);
  qo_D(lo, fl.data(), ll.data(), r.data(), x.data(), 0.1);                 // Naik
  std::vector<double> sol((size_t)vol * 6), hist(64);
  double fin = 0;
  const int its = qo_solveXX(lo, gp.data(), nullptr, sol.data(), x.data(), 0.1, 1e-10, 500, 1, hist.data(), 64, &fin);
  CHECK(its > 5 && its < 500 && fin <= 1e-10, "CG: %d its, %g", its, fin);
  double fin2 = 0;
  const int its2 = qo_solve(lo, gp.data(), nullptr, sol.data(), x.data(), 0.1, 1e-10, 2000, &fin2);
  CHECK(its2 > 5 && fin2 <= 1e-10, "solve: %d its, %g", its2, fin2);
  const double sh[3] = {0.1, 4 * (0.04 - 0.01), 4 * (0.16 - 0.01)};
  std::vector<std::vector<double>> xs(3, std::vector<double>((size_t)vol * 6));
  double *xp[3] = {xs[0].data(), xs[1].data(), xs[2].data()};
  const int itm = qo_solveXX_multi(lo, gp.data(), nullptr, xp, x.data(), sh, 3, 1e-10, 500, 1, hist.data(), 64);
  CHECK(itm > 5 && itm < 500, "multi-shift: %d its", itm);
  qo_rngfield_free(rf);
  qo_layout_free(lo);
}

int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  test_index_hooks();
  test_tile_orders();
  test_rng();
  test_io_and_fuzz(dir);
  test_no_gpu_error_paths();
  test_oracle();
  std::printf(fails ? "host sanitizer run: %d check(s) FAILED\n" : "host sanitizer run: Passed\n", fails);
  return fails ? 1 : 0;
}
