// simd_layout.cpp -- QEX's native field memory <-> the library's host format.  Pure host code.
//
// A QEX Field[V,T] is an array over OUTER sites of T over SIMD vectors of V lanes (src/field/fieldET.nim:18-22,123-128):
//   colour vector  double[outer][3][re|im][V]           colour matrix  double[outer][3][3][re|im][V]
// and site i = outer * V + lane.  Which lattice site that is, is LayoutQ's business (src/layout/qlayout.nim):
//   layoutSetupQ   :10-66     outerGeom = localGeom / innerGeom, the inner checkerboard shift (innerCb, innerCbDir)
//   layoutIndexQ   :110-131   coordinates -> index        layoutCoordQ  :133-185   index -> coordinates
// and the default inner geometry comes from layoutX.nim:19-42 (partitionGeom, dist = 1) + :98-111 (fix-up).
// The library's host format is the same function at V = 1 (innerGeom {1,1,1,1}): the MILC even-odd order the QUDA bridge
// uses (src/quda/qudaWrapperImpl.nim:198-260).  The shim's copy loops (v.l.coord -> lo1.rankIndex, per site, in Nim) become
// one call here; tests/test_simd_layout.py checks the restatement against its own inverse on every site, against the
// independent map of oracle/cpu_simd and against the closed form of the default V = 8 geometry.
#include "../../include/qexhip.h"
#include <cstddef>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

void qexhip_set_error(const char *fmt, ...);

namespace {
struct SimdLay {
  int local[4], inner[4], outer[4];
  int nSites = 0, nSitesOuter = 0, nEvenOuter = 0, nSitesInner = 0, innerCb = 0, innerCbDir = 0;
};

// layoutSetupQ (qlayout.nim:10-66), rank-local part
int simd_setup(SimdLay &l, const int local[4], const int inner[4]) {
  int lvol = 1, ovol = 1, icb = 0, icbd = -1;
  for (int i = 0; i < 4; i++) {
    if (local[i] < 1 || inner[i] < 1 || local[i] % inner[i]) {
      qexhip_set_error("simd layout: localGeom[%d] = %d is not a multiple of innerGeom[%d] = %d", i, local[i], i, inner[i]);
      return QEXHIP_ERR_ARG;
    }
    if (local[i] & 1) {
      // QEX takes the parity that splits a field into its even and odd halves from GLOBAL coordinates (qlayout.nim:133-185 adds
      // localGeom * rankCoord before counting): with an odd local extent a rank at an odd coordinate keeps locally-odd sites in its even
      // half, and a map built from local coordinates alone -- all these entry points are given -- would permute fields to the wrong
      // sites without any sign of it.  Every handle libqexhip shards has even local extents (qexhip_init), so nothing is lost.
      qexhip_set_error("simd layout: localGeom[%d] = %d is odd: the even/odd split would depend on the rank's origin, which this map is not given", i, local[i]);
      return QEXHIP_ERR_ARG;
    }
    l.local[i] = local[i]; l.inner[i] = inner[i]; l.outer[i] = local[i] / inner[i];
    lvol *= l.local[i]; ovol *= l.outer[i];
    if (l.inner[i] > 1 && (l.outer[i] & 1) == 1) icb++;
    if (l.inner[i] == 1 && (l.outer[i] & 1) == 0) icbd = i;
  }
  if (icb == 0) icbd = 0;
  else {
    if (icbd < 0) { qexhip_set_error("simd layout: not enough 2's in localGeom (qlayout.nim:31)"); return QEXHIP_ERR_ARG; }
    icb = l.outer[icbd] / 2;
    if ((icb & 1) == 0) { qexhip_set_error("simd layout: error in cb choice (qlayout.nim:42)"); return QEXHIP_ERR_ARG; }
  }
  if (lvol & 1) { qexhip_set_error("simd layout: odd local volume"); return QEXHIP_ERR_ARG; }
  l.nSites = lvol; l.nSitesOuter = ovol; l.nEvenOuter = ovol - ovol / 2; l.nSitesInner = lvol / ovol;
  l.innerCb = icb; l.innerCbDir = icbd;
  return 0;
}

// lex_i (qlayout.nim:89-104): x[0] fastest, optional divisor d
int lex_i(const int *x, const int *s, const int *d) {
  int l = 0;
  for (int i = 3; i >= 0; i--) {
    int xx = x[i];
    if (d) xx /= d[i];
    l = l * s[i] + (xx % s[i]);
  }
  return l;
}

// layoutIndexQ (qlayout.nim:110-131), local coordinates
int simd_index(const SimdLay &l, int c[4]) {
  const int ii = lex_i(c, l.inner, l.outer);
  int ib = 0;
  for (int i = 0; i < 4; i++) ib += ((c[i] / l.outer[i]) % l.inner[i]) * l.outer[i];
  ib &= 1;
  c[l.innerCbDir] += l.innerCb * ib;
  const int oi = lex_i(c, l.outer, nullptr);
  c[l.innerCbDir] -= l.innerCb * ib;
  int p = 0;
  for (int i = 0; i < 4; i++) p += c[i];
  int oi2 = oi / 2;
  if (p & 1) oi2 = (oi + l.nSitesOuter) / 2;
  return oi2 * l.nSitesInner + ii;
}

// layoutCoordQ (qlayout.nim:133-185), local coordinates
void simd_coord(const SimdLay &l, int index, int c[4]) {
  int cr[4], p = 0, ib = 0;
  int ll = index % l.nSitesInner;
  for (int i = 0; i < 4; i++) {
    const int w = l.inner[i], wl = l.outer[i], k = ll % w;
    cr[i] = k * wl;
    p += cr[i];
    ll /= w;
    ib += k * wl;
  }
  ib &= 1;
  int ii = index / l.nSitesInner;
  if (ii >= l.nEvenOuter) { ii -= l.nEvenOuter; p++; }
  ii *= 2;
  for (int i = 0; i < 4; i++) {
    const int wl = l.outer[i];
    int k = ii % wl;
    if (i == l.innerCbDir) k = (k + l.innerCb * ib) % wl;
    c[i] = k;
    p += k;
    ii /= wl;
  }
  if (p & 1) {
    for (int i = 0; i < 4; i++) {
      const int wl = l.outer[i];
      if (i == l.innerCbDir) c[i] = (c[i] + l.innerCb * ib) % wl;
      c[i]++;
      if (c[i] >= wl) {
        c[i] = 0;
        if (i == l.innerCbDir) c[i] = (c[i] + l.innerCb * ib) % wl;
      } else {
        if (i == l.innerCbDir) c[i] = (c[i] + l.innerCb * ib) % wl;
        break;
      }
    }
  }
  for (int i = 0; i < 4; i++) c[i] += cr[i];
}

int build_map_uncached(const int localGeom[4], const int innerGeom[4], std::vector<int> &v1_of_simd, int *V);

// the map of the geometry used last is kept (a solve converts source and solution, a trajectory hundreds of fields, all on one layout)
struct MapCache { int local[4], inner[4], V; std::shared_ptr<const std::vector<int>> map; };
std::mutex g_cache_mu;
MapCache g_cache{{0, 0, 0, 0}, {0, 0, 0, 0}, 0, nullptr};

int build_map(const int localGeom[4], const int innerGeom[4], std::shared_ptr<const std::vector<int>> &out, int *V) {
  if (!localGeom || !innerGeom) return QEXHIP_ERR_ARG;
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if (g_cache.map && !memcmp(g_cache.local, localGeom, sizeof g_cache.local) && !memcmp(g_cache.inner, innerGeom, sizeof g_cache.inner)) {
      out = g_cache.map; *V = g_cache.V;
      return 0;
    }
  }
  auto m = std::make_shared<std::vector<int>>();
  if (int e = build_map_uncached(localGeom, innerGeom, *m, V)) return e;
  std::lock_guard<std::mutex> lk(g_cache_mu);
  memcpy(g_cache.local, localGeom, sizeof g_cache.local); memcpy(g_cache.inner, innerGeom, sizeof g_cache.inner);
  g_cache.V = *V; g_cache.map = m;
  out = m;
  return 0;
}

int build_map_uncached(const int localGeom[4], const int innerGeom[4], std::vector<int> &v1_of_simd, int *V) {
  if (!localGeom || !innerGeom) return QEXHIP_ERR_ARG;
  SimdLay ls, l1;
  const int one[4] = {1, 1, 1, 1};
  if (int e = simd_setup(ls, localGeom, innerGeom)) return e;
  if (int e = simd_setup(l1, localGeom, one)) return e;
  v1_of_simd.resize(ls.nSites);
  for (int i = 0; i < ls.nSites; i++) {
    int c[4];
    simd_coord(ls, i, c);
    if (simd_index(ls, c) != i) {            // the check layoutCoordQ itself ends with (qlayout.nim:177-185)
      qexhip_set_error("simd layout: bad coord: index %d -> %d %d %d %d -> %d", i, c[0], c[1], c[2], c[3], simd_index(ls, c));
      return QEXHIP_ERR_STATE;
    }
    v1_of_simd[i] = simd_index(l1, c);
  }
  *V = ls.nSitesInner;
  return 0;
}

// n reals per site: simd[(o * n + k) * V + lane]  <->  v1[j * stride + off + k]
template <bool TO_V1>
void permute(const std::vector<int> &map, int V, int n, double *v1, size_t stride, size_t off, double *simd) {
  const size_t ns = map.size();
  for (size_t i = 0; i < ns; i++) {
    const size_t o = i / V, lane = i % V, j = (size_t)map[i];
    for (int k = 0; k < n; k++) {
      double &a = v1[j * stride + off + k], &b = simd[(o * n + k) * V + lane];
      if (TO_V1) a = b; else b = a;
    }
  }
}
}  // namespace

// default inner geometry of newLayoutX: partitionGeom(og, ig, lg, V, dist = 1) (layoutX.nim:19-42) + the fix-up of :98-111
extern "C" int qexhip_layout_default_inner(const int localGeom[4], int V, int innerGeom[4]) {
  if (!localGeom || !innerGeom || V < 1 || (V & (V - 1))) return QEXHIP_ERR_ARG;
  int lx[4], nx[4];
  for (int i = 0; i < 4; i++) { nx[i] = 1; lx[i] = localGeom[i]; }
  for (int ww = V; ww > 1; ww /= 2) {
    int k = 3;
    while (lx[k] & 1) {
      if (--k < 0) { qexhip_set_error("simd layout: not enough 2's in partitioned geom (layoutX.nim:27)"); return QEXHIP_ERR_ARG; }
    }
    for (int i = k - 1; i >= 0; i--)
      if ((lx[i] & 1) == 0 && (nx[i] < nx[k] || (nx[i] == nx[k] && lx[i] > lx[k]))) k = i;
    nx[k] *= 2;
    lx[k] /= 2;
  }
  for (int i = 0; i < 4; i++) {
    if (nx[i] > 1 && (lx[i] % 2) == 1) {
      for (int j = 1; j < 4; j++) {
        const int k = (i + j) % 4;
        if (nx[k] == 1 && (lx[k] % 4) == 0) { nx[k] *= 2; lx[k] /= 2; nx[i] /= 2; lx[i] *= 2; break; }
      }
    }
    if ((nx[i] > 1 && (lx[i] % 2) == 1) || nx[i] > 2) { qexhip_set_error("simd layout: can't lay out inner geom (layoutX.nim:109)"); return QEXHIP_ERR_ARG; }
  }
  for (int i = 0; i < 4; i++) innerGeom[i] = nx[i];
  return 0;
}

// v1_of_simd[outer * V + lane] = index of that site in the library's host format (Layout[1].rankIndex(Layout[V].coord(i)))
extern "C" int qexhip_layout_simd_map(const int localGeom[4], const int innerGeom[4], int *v1_of_simd) {
  if (!v1_of_simd) return QEXHIP_ERR_ARG;
  std::shared_ptr<const std::vector<int>> m;
  int V = 0;
  if (int e = build_map(localGeom, innerGeom, m, &V)) return e;
  for (size_t i = 0; i < m->size(); i++) v1_of_simd[i] = (*m)[i];
  return 0;
}

// colour vector field: simd = double[outer][3][2][V] (QEX's memory as it stands)  <->  v1 = double[site][3][2]
extern "C" int qexhip_layout_vec_simd_to_v1(const int localGeom[4], const int innerGeom[4], const double *simd, double *v1) {
  if (!simd || !v1) return QEXHIP_ERR_ARG;
  std::shared_ptr<const std::vector<int>> mp;
  int V = 0;
  if (int e = build_map(localGeom, innerGeom, mp, &V)) return e;
  const std::vector<int> &m = *mp;
  permute<true>(m, V, 6, v1, 6, 0, const_cast<double *>(simd));
  return 0;
}
extern "C" int qexhip_layout_vec_v1_to_simd(const int localGeom[4], const int innerGeom[4], const double *v1, double *simd) {
  if (!simd || !v1) return QEXHIP_ERR_ARG;
  std::shared_ptr<const std::vector<int>> mp;
  int V = 0;
  if (int e = build_map(localGeom, innerGeom, mp, &V)) return e;
  const std::vector<int> &m = *mp;
  permute<false>(m, V, 6, const_cast<double *>(v1), 6, 0, simd);
  return 0;
}
// gauge field: g[mu] = double[outer][3][3][2][V], one QEX field per direction (s.g[mu], or s.g[2 mu] / s.g[2 mu + 1] of a Naik
// operator: the caller picks the four pointers)  <->  v1 = double[site][4][3][3][2]
extern "C" int qexhip_layout_gauge_simd_to_v1(const int localGeom[4], const int innerGeom[4], const double *const g[4], double *v1) {
  if (!g || !v1) return QEXHIP_ERR_ARG;
  std::shared_ptr<const std::vector<int>> mp;
  int V = 0;
  if (int e = build_map(localGeom, innerGeom, mp, &V)) return e;
  const std::vector<int> &m = *mp;
  for (int mu = 0; mu < 4; mu++) {
    if (!g[mu]) return QEXHIP_ERR_ARG;
    permute<true>(m, V, 18, v1, 72, 18 * (size_t)mu, const_cast<double *>(g[mu]));
  }
  return 0;
}
extern "C" int qexhip_layout_gauge_v1_to_simd(const int localGeom[4], const int innerGeom[4], const double *v1, double *const g[4]) {
  if (!g || !v1) return QEXHIP_ERR_ARG;
  std::shared_ptr<const std::vector<int>> mp;
  int V = 0;
  if (int e = build_map(localGeom, innerGeom, mp, &V)) return e;
  const std::vector<int> &m = *mp;
  for (int mu = 0; mu < 4; mu++) {
    if (!g[mu]) return QEXHIP_ERR_ARG;
    permute<false>(m, V, 18, const_cast<double *>(v1), 72, 18 * (size_t)mu, g[mu]);
  }
  return 0;
}
