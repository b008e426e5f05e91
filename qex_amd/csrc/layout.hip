// layout.hip -- geometry, and reorder kernels between the host format (V=1 MILC even-odd order,
// src/quda/qudaWrapperImpl.nim:198-240 is the reference-side producer of that format) and the
// tiled device layout described in qexhip_internal.h  (kernel K11 of SURVEY.md 2.3).
#include <algorithm>
#include <vector>
#include "qexhip_internal.h"
#include "site_index.h"
#include <cstring>

int geom_init(Geom &g, const int X[4], int depth, int halo) {
  for (int i = 0; i < 4; i++) {
    if (X[i] < 2 || (X[i] & 1)) { qexhip_set_error("local lattice extents must be even and >= 2 (got %d in dim %d)", X[i], i); return -1; }
    if (X[i] > 1024) { qexhip_set_error("local lattice extents above 1024 are not supported (got %d in dim %d)", X[i], i); return -1; }
    g.X[i] = X[i];
  }
  g.Xh = X[0] / 2;
  g.V = X[0] * X[1] * X[2] * X[3];
  g.Vh = g.V / 2;
  g.F = g.Xh * X[1] * X[2];
  g.ntile = (g.Vh + QEXHIP_TILE - 1) / QEXHIP_TILE;
  g.halo = halo;
  g.depth = halo ? depth : 0;
  if (halo) {
    if (g.F % QEXHIP_TILE) { qexhip_set_error("sharding in t needs X*Y*Z/2 to be a multiple of 64 (got %d)", g.F); return -1; }
    if (X[3] < 3) { qexhip_set_error("sharding in t needs a local t extent >= 3"); return -1; }
    g.gtile = 3 * g.F / QEXHIP_TILE;  // room for the 3-hop (Naik) ghost depth; g.depth is the depth in use
  } else {
    g.gtile = 0;
  }
  g.etile = g.ntile + 2 * g.gtile;
  return 0;
}

int ensure_stage(qexhip_ctx *c, size_t bytes) {
  if (c->stage_bytes >= bytes) return 0;
  if (c->stage) HIPCHK(hipFree(c->stage));
  c->stage = nullptr; c->stage_bytes = 0;
  HIPCHK(hipMalloc((void **)&c->stage, bytes));
  c->stage_bytes = bytes;
  return 0;
}

int field_alloc(qexhip_ctx *c, DevField &f) {
  f.half = (size_t)c->g.etile * 192;
  size_t bytes = 2 * f.half * sizeof(double2);
  HIPCHK(hipMalloc((void **)&f.d, bytes));
  HIPCHK(hipMemsetAsync(f.d, 0, bytes, c->stream));  // alignedMem.nim:30 zero-initialises fields
  return 0;
}

// one thread per (parity, c): host[(p*Vh+c)][3] double2  <->  tile layout
__global__ void __launch_bounds__(256) k_vec_to_tiles(Geom g, const double2 *__restrict__ host, double2 *dev, size_t half) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.V) return;
  int p = i >= g.Vh, c = i - p * g.Vh;
  double2 *d = dev + (size_t)p * half;
  for (int k = 0; k < 3; k++) d[vec_off(c, k)] = host[(size_t)i * 3 + k];
}
__global__ void __launch_bounds__(256) k_vec_from_tiles(Geom g, double2 *__restrict__ host, const double2 *dev, size_t half) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.V) return;
  int p = i >= g.Vh, c = i - p * g.Vh;
  const double2 *d = dev + (size_t)p * half;
  for (int k = 0; k < 3; k++) host[(size_t)i * 3 + k] = d[vec_off(c, k)];
}

int field_upload(qexhip_ctx *c, DevField &f, const double *host) {
  size_t bytes = (size_t)c->g.V * 6 * sizeof(double);
  CHK(ensure_stage(c, bytes));
  HIPCHK(hipMemcpyAsync(c->stage, host, bytes, hipMemcpyHostToDevice, c->stream));
  k_vec_to_tiles<<<(c->g.V + 255) / 256, 256, 0, c->stream>>>(c->g, (const double2 *)c->stage, f.d, f.half);
  HIPCHK(hipGetLastError());
  return 0;
}

int field_download(qexhip_ctx *c, const DevField &f, double *host) {
  size_t bytes = (size_t)c->g.V * 6 * sizeof(double);
  CHK(ensure_stage(c, bytes));
  k_vec_from_tiles<<<(c->g.V + 255) / 256, 256, 0, c->stream>>>(c->g, (double2 *)c->stage, f.d, f.half);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(host, c->stage, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// ---- links ----
// pack U_3 of the top `depth` t-slices (both parities) for the upper neighbour:
// buf[p][j][9], j = c - (Vh - depth*F)
__global__ void __launch_bounds__(256) k_pack_top_links(Geom g, const double2 *__restrict__ hostfmt, double2 *buf) {
  int n = g.depth * g.F;
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * n) return;
  int p = i / n, j = i - p * n;
  int c = g.Vh - n + j;
  const double2 *U = hostfmt + ((size_t)(p * g.Vh + c) * 4 + 3) * 9;
  for (int k = 0; k < 9; k++) buf[(size_t)i * 9 + k] = U[k];
}

// W[p][tile][d][k][lane]; hostfmt = [idx][mu][9] double2; ghost = lower neighbour's packed top links
template <bool HALO>
__global__ void __launch_bounds__(256) k_links_to_tiles(Geom g, const double2 *__restrict__ hostfmt,
                                                        const double2 *__restrict__ ghost, double2 *W,
                                                        int ndir, int dbase, int hop) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.V) return;
  int p = i >= g.Vh, c = i - p * g.Vh;
  SiteXYZT s = site_coord(g, c, p);
  double2 *w = W + ((size_t)p * g.ntile + (c >> 6)) * ndir * 576 + (c & 63);
  for (int mu = 0; mu < 4; mu++) {
    const double2 *U = hostfmt + ((size_t)i * 4 + mu) * 9;
    double2 *wf = w + (size_t)(dbase + 2 * mu) * 576;
    for (int k = 0; k < 9; k++) wf[k * 64] = U[k];
    // backward: U_mu(s - hop*mu)^+ ; the source site has the opposite parity
    const double2 *B;
    if (HALO && mu == 3 && s.t - hop < 0) {
      int cF = c - s.t * g.F;
      int j = (s.t - hop + g.depth) * g.F + cF;
      B = ghost + ((size_t)(1 - p) * g.depth * g.F + j) * 9;
    } else {
      int cb = nbr_pos<false>(g, c, s, mu, -hop);
      B = hostfmt + ((size_t)((1 - p) * g.Vh + cb) * 4 + mu) * 9;
    }
    double2 *wb = w + (size_t)(dbase + 2 * mu + 1) * 576;
    for (int r = 0; r < 3; r++)
      for (int q = 0; q < 3; q++) {
        double2 v = B[q * 3 + r];
        wb[(r * 3 + q) * 64] = make_double2(v.x, -v.y);
      }
  }
}

int links_upload(qexhip_ctx *c, const double *fat, const double *lng) {
  c->cg_resume.valid = 0;            // another operator: a kept CG state no longer belongs to it
  const Geom &g = c->g;
  int ndir = lng ? 16 : 8;
  if (lng) for (int i = 0; i < 4; i++) if (g.X[i] < 4) { qexhip_set_error("Naik links need local extents >= 4"); return -1; }
  if (lng && g.halo && g.depth < 3) { qexhip_set_error("internal: ghost depth %d < 3 for Naik links", g.depth); return -3; }
  size_t wbytes = (size_t)2 * g.ntile * ndir * 576 * sizeof(double2);
  if (c->W && c->ndir != ndir) { HIPCHK(hipFree(c->W)); c->W = nullptr; }
  if (!c->W) { HIPCHK(hipMalloc((void **)&c->W, wbytes)); }
  HIPCHK(hipMemsetAsync(c->W, 0, wbytes, c->stream));
  c->ndir = ndir;
  size_t gbytes = (size_t)g.V * 72 * sizeof(double);
  size_t ghost_elems = g.halo ? (size_t)2 * g.depth * g.F * 9 : 0;
  size_t ghost_bytes = ghost_elems * sizeof(double2);
  CHK(ensure_stage(c, gbytes + 2 * ghost_bytes + 256));
  double2 *hostfmt = (double2 *)c->stage;
  double2 *sendbuf = (double2 *)((char *)c->stage + gbytes);
  double2 *ghost = (double2 *)((char *)c->stage + gbytes + ghost_bytes);
  for (int pass = 0; pass < (lng ? 2 : 1); pass++) {
    const double *src = pass ? lng : fat;
    HIPCHK(hipMemcpyAsync(hostfmt, src, gbytes, hipMemcpyHostToDevice, c->stream));
    if (g.halo) {
      int n = 2 * g.depth * g.F;
      k_pack_top_links<<<(n + 255) / 256, 256, 0, c->stream>>>(g, hostfmt, sendbuf);
      HIPCHK(hipGetLastError());
      CHK(comm_exchange_raw(c, sendbuf, ghost, ghost_bytes, c->stream));
      k_links_to_tiles<true><<<(g.V + 255) / 256, 256, 0, c->stream>>>(g, hostfmt, ghost, c->W, ndir, pass * 8, pass ? 3 : 1);
    } else {
      k_links_to_tiles<false><<<(g.V + 255) / 256, 256, 0, c->stream>>>(g, hostfmt, nullptr, c->W, ndir, pass * 8, pass ? 3 : 1);
    }
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  CHK(links_compress(c));
  return sweep_autotune(c);
}

// ---- link compression --------------------------------------------------------------------------
// The sweep is HBM-bound with >10x VALU slack, so bytes are what count.  A unitary link is fixed by
// its first two rows and its determinant: row2 = det * conj(row0 x row1).
//   format 1 (96 B/link): thin staggered links = SU(3) times the +-1 of boundary condition and
//             staggered phase, det = +-1: rows 0,1 + one sign bit per link (64-bit mask per tile row);
//   format 2 (112 B/link): U(3) links (nHYP-smeared: projectU output), rows 0,1 + det as a 7th double2;
//   format 0 (144 B/link): everything else (HISQ fat links are not unitary), all 18 reals.
// The format is chosen per set_links: the most compact one that EVERY link of the operator
// satisfies to 5e-14; row 2 is rebuilt in registers by the Dslash kernel.
template <int FMT>
__global__ void __launch_bounds__(256) k_links_compress(size_t nrows, const double2 *__restrict__ W, double2 *Wc,
                                                        unsigned long long *Ws, unsigned int *maxdev) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;     // (row = (parity, tile, dir), lane)
  const size_t row = j >> 6;
  if (row >= nrows) return;
  const int l = j & 63;
  const double2 *w = W + row * 576 + l;
  constexpr int NR = FMT == 1 ? 6 : 7;
  double2 *o = Wc + row * (NR * 64) + l;
  double2 u[9], r[3];
#pragma unroll
  for (int k = 0; k < 9; k++) u[k] = w[k * 64];
#pragma unroll
  for (int k = 0; k < 6; k++) o[k * 64] = u[k];
  double n2 = 0, px = 0, py = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int a = (k + 1) % 3, b = (k + 2) % 3;
    // conj(u0[a] u1[b] - u0[b] u1[a])
    r[k].x = (u[a].x * u[3 + b].x - u[a].y * u[3 + b].y) - (u[b].x * u[3 + a].x - u[b].y * u[3 + a].y);
    r[k].y = -((u[a].x * u[3 + b].y + u[a].y * u[3 + b].x) - (u[b].x * u[3 + a].y + u[b].y * u[3 + a].x));
    n2 += r[k].x * r[k].x + r[k].y * r[k].y;
    px += u[6 + k].x * r[k].x + u[6 + k].y * r[k].y;     // sum row2[k] * conj(r[k])
    py += u[6 + k].y * r[k].x - u[6 + k].x * r[k].y;
  }
  double2 ph;                                            // det estimate: <r, row2> / <r, r>
  if (n2 > 0) ph = make_double2(px / n2, py / n2); else ph = make_double2(1.0, 0.0);
  if (FMT == 1) ph = make_double2(ph.x < 0 ? -1.0 : 1.0, 0.0);
  double dev = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double rx = ph.x * r[k].x - ph.y * r[k].y, ry = ph.x * r[k].y + ph.y * r[k].x;
    dev = fmax(dev, fmax(fabs(u[6 + k].x - rx), fabs(u[6 + k].y - ry)));
  }
  if (FMT == 1) {
    const unsigned long long mask = __ballot(ph.x < 0);
    if (l == 0) Ws[row] = mask;
  } else {
    o[6 * 64] = ph;
  }
  // one value per wavefront, and an atomic only from a wavefront that raises the maximum seen so far (a plain read first:
  // 131072 read-modify-writes on one address cost 1.1 of the kernel's 1.5 ms at 32^4)
  unsigned int bits = __float_as_uint((float)dev);       // dev >= 0: the uint order is the float order
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned int o2 = __shfl_xor(bits, off, 64);
    bits = o2 > bits ? o2 : bits;
  }
  if (l == 0 && bits > __hip_atomic_load(maxdev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxdev, bits);
}
int links_compress(qexhip_ctx *c) {
  const Geom &g = c->g;
  c->recon = 0;
  if (!c->opt_recon || !c->W) return 0;
  const size_t nrows = (size_t)2 * g.ntile * c->ndir;
  if (c->Wc_rows < nrows) {
    if (c->Wc) HIPCHK(hipFree(c->Wc));
    if (c->Ws) HIPCHK(hipFree(c->Ws));
    HIPCHK(hipMalloc((void **)&c->Wc, nrows * 448 * sizeof(double2)));
    HIPCHK(hipMalloc((void **)&c->Ws, nrows * sizeof(unsigned long long)));
    c->Wc_rows = nrows;
  }
  unsigned int *flag = (unsigned int *)&c->dscal[62];    // a slot of its own (qexhip_internal.h)
  const unsigned nblk = (unsigned)((nrows * 64 + 255) / 256);
  for (int fmt = 1; fmt <= 2; fmt++) {
    if (fmt == 2 && c->opt_recon == 1) break;          // QEXHIP_RECON=1: sign format only
    HIPCHK(hipMemsetAsync(flag, 0, sizeof(unsigned int), c->stream));
    if (fmt == 1) k_links_compress<1><<<nblk, 256, 0, c->stream>>>(nrows, c->W, c->Wc, c->Ws, flag);
    else k_links_compress<2><<<nblk, 256, 0, c->stream>>>(nrows, c->W, c->Wc, c->Ws, flag);
    HIPCHK(hipGetLastError());
    unsigned int bits = 0;
    HIPCHK(hipMemcpyAsync(&bits, flag, sizeof(bits), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    float dev;
    memcpy(&dev, &bits, sizeof(dev));
    if (fmt == 1 || dev <= 5e-14f) c->recon_dev = dev;
    if (dev <= 5e-14f) { c->recon = fmt; break; }
  }
  return 0;
}

// Dslash-ready links from a DEVICE gauge field in the natural tile layout [parity][tile][mu][9][64]
// (the output of the smearing kernels): same W as k_links_to_tiles builds from the host format.
__global__ void __launch_bounds__(256) k_links_from_nat(Geom g, const double2 *__restrict__ N, double2 *W, int ndir, int dbase, int hop) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.V) return;
  int p = i >= g.Vh, c = i - p * g.Vh;
  SiteXYZT s = site_coord(g, c, p);
  double2 *w = W + ((size_t)p * g.ntile + (c >> 6)) * ndir * 576 + (c & 63);
  for (int mu = 0; mu < 4; mu++) {
    const double2 *U = N + (((size_t)p * g.etile + (c >> 6)) * 4 + mu) * 576 + (c & 63);
    double2 *wf = w + (size_t)(dbase + 2 * mu) * 576;
    for (int k = 0; k < 9; k++) wf[k * 64] = U[k * 64];
    // the backward neighbour x - hop*mu; t-sharded: below the slab it lives in the ghost_lo slices of N
    // (virtual slice t < 0 -> Xt + 6 + t, the ghost layout of gauge.hip / smear.hip)
    int cb;
    if (g.halo && mu == 3) {
      const int t = c / g.F, tn = t - hop;
      cb = (tn < 0 ? tn + g.X[3] + 6 : tn) * g.F + (c - t * g.F);
    } else {
      cb = nbr_pos<false>(g, c, s, mu, -hop);
    }
    const double2 *B = N + (((size_t)(1 - p) * g.etile + (cb >> 6)) * 4 + mu) * 576 + (cb & 63);
    double2 *wb = w + (size_t)(dbase + 2 * mu + 1) * 576;
    for (int r = 0; r < 3; r++)
      for (int q = 0; q < 3; q++) {
        double2 v = B[(q * 3 + r) * 64];
        wb[(r * 3 + q) * 64] = make_double2(v.x, -v.y);
      }
  }
}
int links_from_natural(qexhip_ctx *c, const double2 *fat, const double2 *lng) {
  c->cg_resume.valid = 0;
  const Geom &g = c->g;
  int ndir = lng ? 16 : 8;
  if (lng) for (int i = 0; i < 4; i++) if (g.X[i] < 4) { qexhip_set_error("Naik links need local extents >= 4"); return -1; }
  if (g.halo) c->g.depth = lng ? 3 : 1;      // vector ghost depth in use (as qexhip_stag_set_links does)
  size_t wbytes = (size_t)2 * g.ntile * ndir * 576 * sizeof(double2);
  if (c->W && c->ndir != ndir) { HIPCHK(hipFree(c->W)); c->W = nullptr; }
  if (!c->W) { HIPCHK(hipMalloc((void **)&c->W, wbytes)); }
  HIPCHK(hipMemsetAsync(c->W, 0, wbytes, c->stream));
  c->ndir = ndir;
  k_links_from_nat<<<(g.V + 255) / 256, 256, 0, c->stream>>>(g, fat, c->W, ndir, 0, 1);
  if (lng) k_links_from_nat<<<(g.V + 255) / 256, 256, 0, c->stream>>>(g, lng, c->W, ndir, 8, 3);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  CHK(links_compress(c));
  return sweep_autotune(c);
}

// ---- host-callable test hooks for the index arithmetic (no GPU needed) ----
// The same inline functions the kernels use, evaluated on the host, so that the CPU test suite
// (tests/test_host_logic.py, incl. the 2-rank gloo test) can check site order, neighbour sense
// and ghost-zone positions against the oracle's tables.
extern "C" int qexhip_debug_geom(const int latLocal[4], int depth, int halo, int out[8]) {
  Geom g;
  if (geom_init(g, latLocal, depth, halo)) return -1;
  out[0] = g.Vh; out[1] = g.F; out[2] = g.ntile; out[3] = g.gtile; out[4] = g.etile;
  out[5] = g.depth; out[6] = g.halo; out[7] = g.Xh;
  return 0;
}
extern "C" int qexhip_debug_nbr_pos(const int latLocal[4], int depth, int halo, int c, int parity, int mu, int hop) {
  Geom g;
  if (geom_init(g, latLocal, depth, halo)) return -1;
  if (c < 0 || c >= g.Vh || mu < 0 || mu > 3) return -1;
  SiteXYZT s = site_coord(g, c, parity);
  return halo ? nbr_pos<true>(g, c, s, mu, hop) : nbr_pos<false>(g, c, s, mu, hop);
}
extern "C" int qexhip_debug_site_coord(const int latLocal[4], int c, int parity, int x[4]) {
  Geom g;
  if (geom_init(g, latLocal, 1, 0)) return -1;
  if (c < 0 || c >= g.Vh) return -1;
  SiteXYZT s = site_coord(g, c, parity);
  x[0] = 2 * s.xh + s.o; x[1] = s.y; x[2] = s.z; x[3] = s.t;
  return 0;
}

// ---- visiting order of the gather kernels ----
// Workgroups are dealt to the 8 XCDs round-robin by block id and start in id order, so block b runs on XCD b&7 as
// that XCD's (b>>3)-th workgroup.  The table gives every XCD one contiguous (t,z) region of the lattice and walks
// it in compact blocks of both parities, so that the ~64 workgroups an XCD has in flight share most of their
// neighbour links through its 4 MB L2 instead of each fetching them from beyond it.
// entry = 2*tile + parity, -1 = padding.  Layout: [8][chunk].
int tile_order_table(qexhip_ctx *c, const int **tab, int *chunk_out) {
  const Geom &g = c->g;
  const int n = 2 * g.ntile, chunk = (n + 7) / 8;
  if (c->tile_order && c->tile_order_n == 8 * chunk) { *tab = c->tile_order; *chunk_out = chunk; return 0; }
  // block extents in (y, z, t) of the walk inside an XCD's region: the winner of the sweep in
  // profiles/r03_flow_stage_order_sweep.log (XCD regions split in z as well as t lost there and are gone)
  constexpr int by = 8, bz = 4, bt = 4;
  struct Ent { unsigned long long key0, key1; int e; };
  std::vector<Ent> v(n);
  for (int p = 0; p < 2; p++)
    for (int tile = 0; tile < g.ntile; tile++) {
      unsigned r = (unsigned)tile * 64u / (unsigned)g.Xh;             // first site of the tile
      const int y = r % g.X[1]; r /= g.X[1];
      const int z = r % g.X[2], t = r / g.X[2];
      Ent &a = v[(size_t)p * g.ntile + tile];
      a.e = 2 * tile + p;
      a.key0 = (((long)t * g.X[2] + z) * g.X[1] + y) * 2 + p;         // plain order: splits the lattice into 8 (t,z) regions
      // six 10-bit fields + parity = 61 bits (extents <= 1024, geom_init)
      a.key1 = ((((((unsigned long long)(t / bt) * 1024u + z / bz) * 1024u + y / by) * 1024u + t % bt) * 1024u + z % bz) * 1024u + (y % by)) * 2 + p;
    }
  std::sort(v.begin(), v.end(), [](const Ent &a, const Ent &b) { return a.key0 < b.key0; });
  std::vector<int> h((size_t)8 * chunk, -1);
  for (int k = 0; k < 8; k++) {
    const int lo = std::min(n, k * chunk), hi = std::min(n, (k + 1) * chunk);
    std::sort(v.begin() + lo, v.begin() + hi, [](const Ent &a, const Ent &b) { return a.key1 < b.key1; });
    for (int j = lo; j < hi; j++) h[(size_t)k * chunk + (j - lo)] = v[j].e;
  }
  // k_force_lds2 takes the two parities of a tile position from adjacent slots (2j, 2j+1) of one XCD's list
  int pairs = (chunk % 2 == 0);
  for (size_t j = 0; pairs && j + 1 < h.size(); j += 2)
    if (!((h[j] < 0 && h[j + 1] < 0) || (h[j] >= 0 && h[j + 1] == h[j] + 1 && (h[j] & 1) == 0))) pairs = 0;
  c->tile_pairs_ok = pairs;
  if (c->tile_order) { (void)hipFree(c->tile_order); c->tile_order = nullptr; }
  HIPCHK(hipMalloc(&c->tile_order, h.size() * sizeof(int)));
  HIPCHK(hipMemcpy(c->tile_order, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
  c->tile_order_n = 8 * chunk;
  *tab = c->tile_order; *chunk_out = chunk;
  return 0;
}


// Visiting order for the kernels whose gathers stay in ONE plane (the generic staple and its derivative: sites
// x, x+-mu, x+-nu, x-mu+nu, x-nu+mu only).  What is in flight on an XCD at any time is a window of ~256 table entries
// (32 CUs x 2 workgroups x 4 wavefronts): the window should be compact IN THE PLANE, so that the shifted reads of its
// wavefronts land on lines their neighbours in the window fetch anyway, and thin in the other directions, where nothing
// is shared.  The XCDs split the lattice along a direction OUTSIDE the plane (no halo between their regions at all).
// A tile is 64 consecutive sites of one parity (x fastest): x never needs blocking, y comes in units of 64/Xh rows.
// Like tile_order_table this is a permutation of the (tile, parity) pairs -- any order gives the same results.
static void tile_order_plane_host(const Geom &g, int mu, int nu, std::vector<int> &h) {
  const int n = 2 * g.ntile, chunk = (n + 7) / 8;
  const bool inpl[4] = {mu == 0 || nu == 0, mu == 1 || nu == 1, mu == 2 || nu == 2, mu == 3 || nu == 3};
  // direction the XCDs split along: the slowest one outside the plane
  int dpart = 3;
  while (dpart > 0 && inpl[dpart]) dpart--;
  // block extents (sites) in y, z, t: a budget of 128 tile positions, spent on the plane's directions first
  const int rows_per_tile = std::max(1, 64 / std::max(1, g.Xh));
  int ext[4] = {0, rows_per_tile, 1, 1};
  long budget = 128;
  auto spend = [&](int d, long cap) {
    long full = d == 1 ? std::max(1, g.X[1] / rows_per_tile) : g.X[d];      // in tile positions
    long e = std::max(1L, std::min(std::min(full, cap), budget));
    ext[d] = (int)(d == 1 ? e * rows_per_tile : e);
    budget = std::max(1L, budget / e);
  };
  int pd[2], npd = 0;
  for (int d = 3; d >= 1; d--) if (inpl[d]) pd[npd++] = d;
  if (npd == 2) { spend(pd[1], 8); spend(pd[0], budget); }           // two blocked directions: 8 x 16 positions
  else if (npd == 1) spend(pd[0], budget);
  for (int d = 1; d <= 3; d++) if (!inpl[d] && d != dpart) spend(d, budget);   // leftover budget: a direction outside the plane
  struct Ent { unsigned long long key0, key1; int e; };
  std::vector<Ent> v(n);
  for (int p = 0; p < 2; p++)
    for (int tile = 0; tile < g.ntile; tile++) {
      unsigned r = (unsigned)tile * 64u / (unsigned)g.Xh;             // first site of the tile
      int xc[4];
      xc[0] = 0;
      xc[1] = r % g.X[1]; r /= g.X[1];
      xc[2] = r % g.X[2]; xc[3] = r / g.X[2];
      Ent &a = v[(size_t)p * g.ntile + tile];
      a.e = 2 * tile + p;
      // key0: the split direction slowest, then the others (storage order)
      // (coordinates < 1024: 3 x 10 bits per index, so the packed keys below stay under 2^61 -- 12-bit fields overflowed 2^63
      //  for t >= 8 in the split direction and wrapped the order around)
      unsigned long long k0 = (unsigned)xc[dpart];
      for (int d = 3; d >= 1; d--) if (d != dpart) k0 = k0 * 1024u + (unsigned)xc[d];
      a.key0 = k0 * 2 + p;
      // key1: block index (slow: directions outside the plane, then inside), then position inside the block, parity last
      unsigned long long kb = 0, ki = 0;
      for (int pass = 0; pass < 2; pass++)
        for (int d = 3; d >= 1; d--) {
          if ((pass == 0) == inpl[d]) continue;                        // pass 0: outside the plane, pass 1: inside
          kb = kb * 1024u + (unsigned)(xc[d] / ext[d]);
          ki = ki * 1024u + (unsigned)(xc[d] % ext[d]);
        }
      a.key1 = ((kb << 30) + ki) * 2 + p;
    }
  std::sort(v.begin(), v.end(), [](const Ent &a, const Ent &b) { return a.key0 < b.key0; });
  h.assign((size_t)8 * chunk, -1);
  for (int k = 0; k < 8; k++) {
    const int lo = std::min(n, k * chunk), hi = std::min(n, (k + 1) * chunk);
    std::sort(v.begin() + lo, v.begin() + hi, [](const Ent &a, const Ent &b) { return a.key1 < b.key1; });
    for (int j = lo; j < hi; j++) h[(size_t)k * chunk + (j - lo)] = v[j].e;
  }
}
int tile_order_plane(qexhip_ctx *c, int mu, int nu, const int **tab, int *chunk_out) {
  const Geom &g = c->g;
  const int n = 2 * g.ntile, chunk = (n + 7) / 8;
  if (mu == nu || mu < 0 || nu < 0 || mu > 3 || nu > 3) return tile_order_table(c, tab, chunk_out);
  int *&slot = c->tile_order_pl[mu * 4 + nu];
  if (slot) { *tab = slot; *chunk_out = chunk; return 0; }
  std::vector<int> h;
  tile_order_plane_host(g, mu, nu, h);
  HIPCHK(hipMalloc(&slot, h.size() * sizeof(int)));
  HIPCHK(hipMemcpy(slot, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
  *tab = slot; *chunk_out = chunk;
  return 0;
}
// host-callable (no GPU): the table tile_order_plane would upload; out holds 8 * ((2 ntile + 7) / 8) entries, -1 = empty slot
extern "C" int qexhip_debug_tile_order(const int latLocal[4], int mu, int nu, int *out, int cap) {
  Geom g;
  if (geom_init(g, latLocal, 1, 0) || mu == nu || mu < 0 || nu < 0 || mu > 3 || nu > 3 || !out) return -1;
  std::vector<int> h;
  tile_order_plane_host(g, mu, nu, h);
  if ((int)h.size() > cap) return -(int)h.size();
  for (size_t i = 0; i < h.size(); i++) out[i] = h[i];
  return (int)h.size();
}
