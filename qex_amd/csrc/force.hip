// force.hip -- fermion-force outer product.
//
// Restates the shifted outer products of the staggered force:
//   stagDeriv  (src/physics/stagD.nim:634-664):  f[mu](s) += x(s) (x) x(s+mu)^+ on even s,
//                                                f[mu](s) -= x(s) (x) x(s+mu)^+ on odd s
//   fforce     (src/stagg_pv_hmc/staghmc_spv.nim:831-854): f[mu](s) (:= | +=) scale psi(s) (x) psi(s+mu)^+
// One lane per site, same tile layout and neighbour arithmetic as the Dslash; the four 3x3 outer
// products of a site are written as contiguous 576-double2 rows of the natural-layout force field
// F[parity][tile][mu][9][64].  Pure streaming: 96 B read, 576 B written per site.
#include "qexhip_internal.h"
#include "site_index.h"

template <bool HALO>
__global__ void __launch_bounds__(256) k_outer(Geom g, const double2 *__restrict__ x0, const double2 *__restrict__ x1,
                                               double2 *F, double se, double so, int accumulate, int hop) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.V) return;
  const int p = i >= g.Vh, c = i - p * g.Vh;
  const double2 *xs = p ? x1 : x0;       // this parity
  const double2 *xn = p ? x0 : x1;       // the neighbours' parity
  const double sc = p ? so : se;
  SiteXYZT s = site_coord(g, c, p);
  double2 a[3];
#pragma unroll
  for (int k = 0; k < 3; k++) a[k] = xs[vec_off(c, k)];
#pragma unroll
  for (int mu = 0; mu < 4; mu++) {
    const int pos = nbr_pos<HALO>(g, c, s, mu, hop);
    double2 b[3];
#pragma unroll
    for (int k = 0; k < 3; k++) b[k] = xn[vec_off(pos, k)];
    double2 *w = F + (((size_t)p * g.etile + (c >> 6)) * 4 + mu) * 576 + (c & 63);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int q = 0; q < 3; q++) {
        double2 v = make_double2(sc * (a[r].x * b[q].x + a[r].y * b[q].y), sc * (a[r].y * b[q].x - a[r].x * b[q].y));
        if (accumulate) { double2 o = w[(r * 3 + q) * 64]; v.x += o.x; v.y += o.y; }
        w[(r * 3 + q) * 64] = v;
      }
  }
}

__global__ void __launch_bounds__(256) k_force_to_tiles(Geom g, const double2 *__restrict__ host, double2 *F) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.V) return;
  int p = i >= g.Vh, c = i - p * g.Vh;
  for (int mu = 0; mu < 4; mu++) {
    double2 *w = F + (((size_t)p * g.etile + (c >> 6)) * 4 + mu) * 576 + (c & 63);
    for (int k = 0; k < 9; k++) w[k * 64] = host[((size_t)i * 4 + mu) * 9 + k];
  }
}
__global__ void __launch_bounds__(256) k_force_from_tiles(Geom g, double2 *__restrict__ host, const double2 *F) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.V) return;
  int p = i >= g.Vh, c = i - p * g.Vh;
  for (int mu = 0; mu < 4; mu++) {
    const double2 *w = F + (((size_t)p * g.etile + (c >> 6)) * 4 + mu) * 576 + (c & 63);
    for (int k = 0; k < 9; k++) host[((size_t)i * 4 + mu) * 9 + k] = w[k * 64];
  }
}

// f (:= | +=) scale * x (x) x(+mu)^+ on a DEVICE force field in the natural layout (single GPU)
int stag_outer_dev(qexhip_ctx *c, DevField &fx, double2 *F, double se, double so, int accumulate, int hop) {
  const Geom &g = c->g;
  if (hop != 1 && hop != 3) { qexhip_set_error("outer product: hop must be 1 or 3"); return -1; }
  if (g.halo) {
    if (g.depth < hop) { qexhip_set_error("outer product with hop 3 on a sharded field needs the Naik operator's ghost depth (set the links first)"); return -3; }
    for (int par = 0; par < 2; par++) CHK(comm_halo_exchange(c, fx, par, 0));      // x(s + hop t) across the slab boundary
    ScopedTimer tm(c, "outer", c->stream);
    k_outer<true><<<(g.V + 255) / 256, 256, 0, c->stream>>>(g, fx.par(0), fx.par(1), F, se, so, accumulate, hop);
    HIPCHK(hipGetLastError());
    return 0;
  }
  ScopedTimer tm(c, "outer", c->stream);
  k_outer<false><<<(g.V + 255) / 256, 256, 0, c->stream>>>(g, fx.par(0), fx.par(1), F, se, so, accumulate, hop);
  HIPCHK(hipGetLastError());
  return 0;
}

// host f (in/out, [vol][4][3][3][2]) and host x; the force field lives in a scratch buffer
int stag_outer_host(qexhip_ctx *c, double *f_host, const double *x_host, double se, double so, int accumulate) {
  const Geom &g = c->g;
  DevField *fx;
  CHK(get_work(c, WK_IN, &fx));
  CHK(field_upload(c, *fx, x_host));
  const size_t n2 = (size_t)2 * g.etile * 4 * 576;
  if (c->outer_Fn < n2) {
    if (c->outer_F) HIPCHK(hipFree(c->outer_F));
    c->outer_F = nullptr; c->outer_Fn = 0;
    HIPCHK(hipMalloc((void **)&c->outer_F, n2 * sizeof(double2)));
    c->outer_Fn = n2;
  }
  double2 *Fd = c->outer_F;
  const size_t gbytes = (size_t)g.V * 72 * sizeof(double);
  if (accumulate) {
    CHK(ensure_stage(c, gbytes));
    HIPCHK(hipMemcpyAsync(c->stage, f_host, gbytes, hipMemcpyHostToDevice, c->stream));
    k_force_to_tiles<<<(g.V + 255) / 256, 256, 0, c->stream>>>(g, (const double2 *)c->stage, Fd);
    HIPCHK(hipGetLastError());
  }
  if (g.halo) {
    // neighbours across the t-faces: refresh the ghost zones of both parity halves
    for (int par = 0; par < 2; par++) CHK(comm_halo_exchange(c, *fx, par, 0));
    ScopedTimer tm(c, "outer", c->stream);
    k_outer<true><<<(g.V + 255) / 256, 256, 0, c->stream>>>(g, fx->par(0), fx->par(1), Fd, se, so, accumulate, 1);
  } else {
    ScopedTimer tm(c, "outer", c->stream);
    k_outer<false><<<(g.V + 255) / 256, 256, 0, c->stream>>>(g, fx->par(0), fx->par(1), Fd, se, so, accumulate, 1);
  }
  HIPCHK(hipGetLastError());
  CHK(ensure_stage(c, gbytes));
  k_force_from_tiles<<<(g.V + 255) / 256, 256, 0, c->stream>>>(g, (double2 *)c->stage, Fd);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(f_host, c->stage, gbytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}
