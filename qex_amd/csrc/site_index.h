// site_index.h -- checkerboard site <-> coordinate arithmetic shared by all kernels.
//
// Index convention = QEX V=1 layout (src/layout/qlayout.nim:110-131): lex index with x fastest,
// c = lex/2 within a parity, parity = (x+y+z+t)&1.  Neighbour sense follows the shifts:
// the forward hop of direction mu reads the site s + hop*mu (src/layout/shiftX.nim:76-81).
#pragma once
#include "qexhip_internal.h"

struct SiteXYZT {
  int xh, y, z, t, o;  // x = 2*xh + o
};

__host__ __device__ __forceinline__ SiteXYZT site_coord(const Geom &g, int c, int parity) {
  SiteXYZT s;
  unsigned r = (unsigned)c;
  s.xh = r % (unsigned)g.Xh; r /= (unsigned)g.Xh;
  s.y = r % (unsigned)g.X[1]; r /= (unsigned)g.X[1];
  s.z = r % (unsigned)g.X[2];
  s.t = r / (unsigned)g.X[2];
  s.o = (s.y + s.z + s.t + parity) & 1;
  return s;
}

__host__ __device__ __forceinline__ int wrap(int v, int n) { return v >= n ? v - n : (v < 0 ? v + n : v); }

// position, in the field of the OPPOSITE parity, of the site s + hop*mu (hop = +-1, +-3).
// With g.halo, t-hops that leave the local lattice land in the ghost zones.
template <bool HALO>
__host__ __device__ __forceinline__ int nbr_pos(const Geom &g, int c, const SiteXYZT &s, int mu, int hop) {
  if (mu == 0) {
    int x = 2 * s.xh + s.o;
    int xn = wrap(x + hop, g.X[0]);
    return c - s.xh + (xn >> 1);
  } else if (mu == 1) {
    return c + (wrap(s.y + hop, g.X[1]) - s.y) * g.Xh;
  } else if (mu == 2) {
    return c + (wrap(s.z + hop, g.X[2]) - s.z) * g.Xh * g.X[1];
  } else {
    int tn = s.t + hop;
    if (HALO) {
      // The ghost zones continue the slice numbering: ghost_hi holds the virtual slices
      // t = Xt .. Xt+depth-1 (position t*F + cF, no special case), ghost_lo the virtual slices
      // t = -depth .. -1 stored at t + Xt + 2*depth.  One select, no control flow.  (A three-way
      // select / branch formulation made hipcc hoist the loads of all directions and spill:
      // 256 VGPRs + scratch, 1 wave/SIMD.)
      return c + hop * g.F + (tn < 0 ? (g.X[3] + 2 * g.depth) * g.F : 0);
    } else {
      return c + (wrap(tn, g.X[3]) - s.t) * g.F;
    }
  }
}
