// gauge_index.h -- site / link addressing of the natural-layout gauge field G[parity][tile][mu][9][64] (double2), shared by the
// gauge kernels (gauge.hip; the FsSite helpers serve the gather calibration of dslash_tune.hip).  Site numbering: checkerboard index c = lex/2 of the rank-local lattice
// (src/layout/qlayout.nim:110-131 with V = 1); t-sharded fields carry ghost tiles addressed as virtual slices.
#pragma once
#include "qexhip_internal.h"

__device__ __forceinline__ void coords_of(const Geom &g, int c, int p, int x[4]) {
  unsigned r = (unsigned)c;
  int xh = r % (unsigned)g.Xh; r /= (unsigned)g.Xh;
  x[1] = r % (unsigned)g.X[1]; r /= (unsigned)g.X[1];
  x[2] = r % (unsigned)g.X[2];
  x[3] = r / (unsigned)g.X[2];
  x[0] = 2 * xh + ((x[1] + x[2] + x[3] + p) & 1);
}
// HALO as a template parameter for the kernels whose register allocation is tight (k_plaq spills with a runtime
// flag); the runtime-flag forms below serve everything else
template <bool HALO>
__device__ __forceinline__ size_t link_off_t(const Geom &g, const int x[4], int mu) {
  int t = x[3];
  if (HALO) t = t < 0 ? t + g.X[3] + 6 : t;          // virtual slices: Xt..Xt+2 -> ghost_hi (in place), -3..-1 -> ghost_lo
  int lex = x[0] + g.X[0] * (x[1] + g.X[1] * (x[2] + g.X[2] * t));
  int p = (x[0] + x[1] + x[2] + x[3]) & 1;
  int c = lex >> 1;
  return (((size_t)p * g.etile + (c >> 6)) * 4 + mu) * 576 + (c & 63);
}
template <bool HALO>
__device__ __forceinline__ void shifted_t(const Geom &g, const int x[4], int mu, int d, int y[4]) {
  y[0] = x[0]; y[1] = x[1]; y[2] = x[2]; y[3] = x[3];
  int v = y[mu] + d;
  if (HALO && mu == 3) { y[3] = v; return; }         // t sharded: no wrap, ghosts
  y[mu] = v >= g.X[mu] ? v - g.X[mu] : (v < 0 ? v + g.X[mu] : v);
}
// shifted_t with a direction that is only known at run time (wavefront-uniform): every coordinate is visited with a
// static index, so x[] and y[] stay in registers (indexing them with mu sends them to scratch)
template <bool HALO>
__device__ __forceinline__ void shifted_dyn(const Geom &g, const int x[4], int mu, int d, int y[4]) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int v = x[k] + (mu == k ? d : 0);
    if (HALO && k == 3) y[k] = v;
    else y[k] = v >= g.X[k] ? v - g.X[k] : (v < 0 ? v + g.X[k] : v);
  }
}

// A wavefront's view of its lane's site: coordinates, lexicographic index, parity.  The operand addresses of a tile
// are this index plus a wavefront-uniform hop (with the wrap of the lane's coordinate): ~15 integer instructions per operand
// instead of the full coordinate -> offset chain of link_off_t (the loaders' instruction stream, not the memory system, set
// the pace of the first form of this kernel: 0.9 us per phase with nothing else running).
struct FsSite {
  int x[4];
  int lex, par;
};
__device__ __forceinline__ void fs_site(const Geom &g, int c, int p, FsSite &s) {
  coords_of(g, c, p, s.x);
  s.lex = s.x[0] + g.X[0] * (s.x[1] + g.X[1] * (s.x[2] + g.X[2] * s.x[3]));
  s.par = p;
}
// one hop of s (+-1) in direction d (wavefront-uniform) applied to (lex, par)
template <bool HALO>
__device__ __forceinline__ void fs_hop(const Geom &g, const FsSite &s, int d, int sgn, int &lex, int &par) {
  const int xd = d == 0 ? s.x[0] : (d == 1 ? s.x[1] : (d == 2 ? s.x[2] : s.x[3]));
  const int Xd = d == 0 ? g.X[0] : (d == 1 ? g.X[1] : (d == 2 ? g.X[2] : g.X[3]));
  const int st = d == 0 ? 1 : (d == 1 ? g.X[0] : (d == 2 ? g.X[0] * g.X[1] : g.X[0] * g.X[1] * g.X[2]));
  const int v = xd + sgn;
  lex += sgn * st;
  if (HALO && d == 3) {
    if (v < 0) lex += (Xd + 6) * st;                     // virtual slice -1 -> ghost_lo (link_off_t: t + Xt + 6)
  } else {
    if (v >= Xd) lex -= Xd * st;
    if (v < 0) lex += Xd * st;
  }
  par ^= 1;
}
__device__ __forceinline__ unsigned fs_link_off(const Geom &g, int lex, int par, int mu) {
  const int c = lex >> 1;
  return ((unsigned)(par * g.etile + (c >> 6)) * 4u + (unsigned)mu) * 576u + (unsigned)(c & 63);
}
