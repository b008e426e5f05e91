// dslash_tune.hip -- experimental variants of the one-parity sweep, for on-GPU A/B timing only.
// Not part of the solver path: qexhip_tune_dslash() runs a variant on scratch fields and returns
// the average launch time measured with hipEvents (interleave variants in ONE process:
// cdna_hip_programming.md 5.4 rule 24).  The winner is folded back into dslash.hip by hand.
#include "qexhip_internal.h"
#include <algorithm>
#include <vector>
#include "site_index.h"
#include "../../include/qexhip.h"

typedef double d2v __attribute__((ext_vector_type(2)));

struct TuneArgs {
  Geom g;
  const double2 *W, *in;
  double2 *out;
  int parity, swz;
};

__device__ __forceinline__ void mv3t(double2 acc[3], const double2 U[9], const double2 v[3]) {
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      acc[i].x += U[3 * i + j].x * v[j].x;
      acc[i].x -= U[3 * i + j].y * v[j].y;
      acc[i].y += U[3 * i + j].x * v[j].y;
      acc[i].y += U[3 * i + j].y * v[j].x;
    }
  }
}

// VAR bit0: non-temporal link loads; bit1: scheduling fence per direction pair;
//     bit2: non-temporal output stores; bit3: fence per single direction
template <int VAR, int BS, int MINW, int NDIR = 8>
__global__ void __launch_bounds__(BS, MINW) k_tune(TuneArgs A) {
  int bid = blockIdx.x;
  if (A.swz) {
    int per = A.swz >> 3;
    bid = (bid & 7) * per + (bid >> 3);
  }
  int c = bid * BS + threadIdx.x;
  if (c >= A.g.Vh) return;
  const Geom &g = A.g;
  SiteXYZT s = site_coord(g, c, A.parity);
  double2 acc[3];
#pragma unroll
  for (int k = 0; k < 3; k++) acc[k] = make_double2(0.0, 0.0);
  const double2 *w = A.W + (size_t)(c >> 6) * (NDIR * 576) + (c & 63);
#pragma unroll
  for (int d = 0; d < NDIR; d++) {
    const int mu = (d >> 1) & 3;
    const int hop = (d >= 8 ? 3 : 1) * ((d & 1) ? -1 : 1);
    int pos = nbr_pos<false>(g, c, s, mu, hop);
    double2 U[9], v[3];
#pragma unroll
    for (int k = 0; k < 9; k++) {
      if (VAR & 1) {
        d2v t = __builtin_nontemporal_load((const d2v *)&w[(size_t)d * 576 + k * 64]);
        U[k] = make_double2(t.x, t.y);
      } else {
        U[k] = w[(size_t)d * 576 + k * 64];
      }
    }
    const double sg = (d & 1) ? -1.0 : 1.0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      double2 t = A.in[vec_off(pos, k)];
      v[k] = make_double2(sg * t.x, sg * t.y);
    }
    mv3t(acc, U, v);
    if ((VAR & 2) && (d & 1)) __builtin_amdgcn_sched_barrier(0);
    if (VAR & 8) __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (VAR & 4) {
      d2v t; t.x = acc[k].x; t.y = acc[k].y;
      __builtin_nontemporal_store(t, (d2v *)&A.out[vec_off(c, k)]);
    } else {
      A.out[vec_off(c, k)] = acc[k];
    }
  }
}

// ---- LDS-staged variant (BASELINE north_star: "gauge links staged in LDS") ----
// Every operand of a hop -- the link (9 x 1 KiB per wavefront) and the neighbour vector (3 x 1 KiB, per-lane gather
// addresses) -- travels global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR destination), double-buffered per
// wavefront: the 12 DMAs of direction d+1 are in flight while direction d is read back from LDS (ds_read_b128) and
// multiplied.  Counted waits: after issuing d+1's 12 DMAs, vmcnt(12) retires exactly direction d's.  24 KiB of LDS
// per wavefront, 128-thread workgroups: 3 workgroups = 6 wavefronts per CU.
template <int NDIR>
__global__ void __launch_bounds__(128, 2) k_tune_lds(TuneArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bid = blockIdx.x;
  if (A.swz) {
    int per = A.swz >> 3;
    bid = (bid & 7) * per + (bid >> 3);
  }
  const int c = bid * 128 + threadIdx.x;
  const Geom &g = A.g;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool live = c < g.Vh;
  const int cc = live ? c : g.Vh - 1;        // dead lanes of the last workgroup still issue (in-range) DMAs
  SiteXYZT s = site_coord(g, cc, A.parity);
  char *wbase = smem + wv * (2 * 12 * 1024);
  const double2 *w = A.W + (size_t)(cc >> 6) * (NDIR * 576) + (cc & 63);
  auto stage = [&](int d) {
    char *buf = wbase + (d & 1) * (12 * 1024);
    const int mu = (d >> 1) & 3;
    const int hop = (d >= 8 ? 3 : 1) * ((d & 1) ? -1 : 1);
    const int pos = nbr_pos<false>(g, cc, s, mu, hop);
#pragma unroll
    for (int k = 0; k < 9; k++)
      __builtin_amdgcn_global_load_lds((const void *)&w[(size_t)d * 576 + k * 64], (__attribute__((address_space(3))) void *)(buf + k * 1024), 16, 0, 0);
#pragma unroll
    for (int k = 0; k < 3; k++)
      __builtin_amdgcn_global_load_lds((const void *)&A.in[vec_off(pos, k)], (__attribute__((address_space(3))) void *)(buf + (9 + k) * 1024), 16, 0, 0);
  };
  double2 acc[3];
#pragma unroll
  for (int k = 0; k < 3; k++) acc[k] = make_double2(0.0, 0.0);
  stage(0);
#pragma unroll
  for (int d = 0; d < NDIR; d++) {
    if (d + 1 < NDIR) {
      stage(d + 1);
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const double2 *buf = (const double2 *)(wbase + (d & 1) * (12 * 1024));
    double2 U[9], v[3];
#pragma unroll
    for (int k = 0; k < 9; k++) U[k] = buf[k * 64 + lane];
    const double sg = (d & 1) ? -1.0 : 1.0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      double2 t = buf[(9 + k) * 64 + lane];
      v[k] = make_double2(sg * t.x, sg * t.y);
    }
    mv3t(acc, U, v);
    // the buffer read here is overwritten by the DMAs issued at the top of the NEXT iteration: LDS reads complete
    // in order with respect to this wavefront's later LDS-DMA writes only after lgkmcnt(0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (live) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      d2v t; t.x = acc[k].x; t.y = acc[k].y;
      __builtin_nontemporal_store(t, (d2v *)&A.out[vec_off(c, k)]);
    }
  }
}
template <int NDIR>
static void launch_tune_lds(TuneArgs &A, int swz_on, hipStream_t st) {
  int nb = (A.g.Vh + 127) / 128;
  A.swz = (swz_on && nb >= 64 && (nb & 7) == 0) ? nb : 0;
  k_tune_lds<NDIR><<<nb, 128, 2 * 2 * 12 * 1024, st>>>(A);
}

template <int VAR, int BS, int MINW, int NDIR = 8>
static void launch_tune(TuneArgs &A, int swz_on, hipStream_t st) {
  int nb = (A.g.Vh + BS - 1) / BS;
  A.swz = (swz_on && nb >= 64 && (nb & 7) == 0) ? nb : 0;
  k_tune<VAR, BS, MINW, NDIR><<<nb, BS, 0, st>>>(A);
}

// variant ids: see table in tests/../scratch/tune_dslash.py
extern "C" int qexhip_tune_dslash(qexhip_handle c, int variant, int swz, int nrep, double *avg_us) {
  if (!c || !c->W || c->g.halo) { qexhip_set_error("tune: needs links, no halo"); return -1; }
  if ((variant >= 100) != (c->ndir == 16)) { qexhip_set_error("tune: variants >= 100 are the Naik ones"); return -1; }
  DevField *fin, *fout;
  CHK(get_work(c, WK_IN, &fin));
  CHK(get_work(c, WK_OUT, &fout));
  TuneArgs A;
  A.g = c->g;
  A.W = c->W;  // parity 0
  A.in = fin->par(1);
  A.out = fout->par(0);
  A.parity = 0;
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  auto run = [&]() {
    switch (variant) {
      case 0: launch_tune<0, 256, 1>(A, swz, c->stream); break;
      case 1: launch_tune<1, 256, 1>(A, swz, c->stream); break;
      case 2: launch_tune<2, 256, 1>(A, swz, c->stream); break;
      case 3: launch_tune<3, 256, 1>(A, swz, c->stream); break;
      case 4: launch_tune<4, 256, 1>(A, swz, c->stream); break;
      case 5: launch_tune<5, 256, 1>(A, swz, c->stream); break;
      case 6: launch_tune<0, 128, 1>(A, swz, c->stream); break;
      case 7: launch_tune<0, 512, 1>(A, swz, c->stream); break;
      case 8: launch_tune<0, 64, 1>(A, swz, c->stream); break;
      case 9: launch_tune<8, 256, 1>(A, swz, c->stream); break;
      case 10: launch_tune<9, 256, 1>(A, swz, c->stream); break;
      case 11: launch_tune<0, 256, 4>(A, swz, c->stream); break;
      case 12: launch_tune<2, 256, 3>(A, swz, c->stream); break;
      case 13: launch_tune<7, 256, 1>(A, swz, c->stream); break;
      case 20: launch_tune_lds<8>(A, swz, c->stream); break;
      case 100: launch_tune<5, 256, 1, 16>(A, swz, c->stream); break;
      case 101: launch_tune<7, 256, 1, 16>(A, swz, c->stream); break;
      case 102: launch_tune<13, 256, 1, 16>(A, swz, c->stream); break;
      case 103: launch_tune<5, 128, 1, 16>(A, swz, c->stream); break;
      case 104: launch_tune<5, 512, 1, 16>(A, swz, c->stream); break;
      case 105: launch_tune<0, 256, 1, 16>(A, swz, c->stream); break;
      case 106: launch_tune<5, 256, 3, 16>(A, swz, c->stream); break;
      default: break;
    }
  };
  for (int i = 0; i < 3; i++) run();
  HIPCHK(hipEventRecord(e0, c->stream));
  for (int i = 0; i < nrep; i++) run();
  HIPCHK(hipEventRecord(e1, c->stream));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  *avg_us = 1e3 * ms / nrep;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return 0;
}

// |out|^2 of the field the variants write (parity 0 of the WK_OUT work field): a variant is only compared on time
// with another when both leave the same result
extern "C" int qexhip_tune_dslash_norm2(qexhip_handle c, double *n2) {
  if (!c || !n2) return -1;
  DevField *fout;
  CHK(get_work(c, WK_OUT, &fout));
  CHK(blas_norm2(c, *fout, 0, &c->dscal[8]));
  return read_scalars(c, &c->dscal[8], 1, n2);
}

// plain device-to-device streaming copy of n bytes with 16-byte accesses: the measured HBM ceiling
__global__ void __launch_bounds__(256) k_copy16(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ void __launch_bounds__(256) k_read16(const double2 *__restrict__ a, double *out, size_t n) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { double2 v = a[i]; s += v.x + v.y; }
  if (s == 1.2345e300) out[0] = s;
}
__global__ void __launch_bounds__(256) k_read16_nt(const double2 *__restrict__ a, double *out, size_t n) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    d2v v = __builtin_nontemporal_load((const d2v *)&a[i]);
    s += v.x + v.y;
  }
  if (s == 1.2345e300) out[0] = s;
}
__global__ void __launch_bounds__(256) k_write16_nt(double2 *__restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    d2v t; t.x = (double)i; t.y = 1.0;
    __builtin_nontemporal_store(t, (d2v *)&b[i]);
  }
}
// mode 0: copy (bytes read + written = 2n*16), mode 1: read only, 2: non-temporal read, 3: nt write
extern "C" int qexhip_tune_stream(qexhip_handle c, int mode, size_t mbytes, int nblocks, int nrep, double *gbs) {
  if (!c) return -1;
  size_t n = mbytes * 1048576 / 16;
  double2 *a, *b;
  HIPCHK(hipMalloc((void **)&a, n * 16));
  HIPCHK(hipMalloc((void **)&b, n * 16));
  HIPCHK(hipMemsetAsync(a, 1, n * 16, c->stream));
  HIPCHK(hipMemsetAsync(b, 0, n * 16, c->stream));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  auto run = [&]() {
    if (mode == 0) k_copy16<<<nblocks, 256, 0, c->stream>>>(a, b, n);
    else if (mode == 1) k_read16<<<nblocks, 256, 0, c->stream>>>(a, (double *)b, n);
    else if (mode == 2) k_read16_nt<<<nblocks, 256, 0, c->stream>>>(a, (double *)b, n);
    else k_write16_nt<<<nblocks, 256, 0, c->stream>>>(b, n);
  };
  for (int i = 0; i < 2; i++) run();
  HIPCHK(hipEventRecord(e0, c->stream));
  for (int i = 0; i < nrep; i++) run();
  HIPCHK(hipEventRecord(e1, c->stream));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  double bytes = (mode == 0 ? 2.0 : 1.0) * n * 16.0 * nrep;
  *gbs = bytes / (ms * 1e-3) / 1e9;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(a); (void)hipFree(b);
  return 0;
}

// ---- fp64 vector ceiling actually held by this chip (the number k_force / k_exp_update are priced against) ----
// `chains` independent v_fma_f64 chains per lane on random-ish data, every SIMD of the chip busy with `wps` wavefronts,
// long enough (>= 2 ms) for the clock to settle.  kind 1: the 3x3 complex products of su3.h in the same dependency
// pattern as m3_exp's squaring loop (what the flow's exp actually issues).
template <int CH>
__global__ void __launch_bounds__(256) k_fma64(double *out, int iters, double a, double b) {
  double v[CH];
#pragma unroll
  for (int k = 0; k < CH; k++) v[k] = 1e-3 * (threadIdx.x + 1) + k;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < CH; k++) v[k] = fma(v[k], a, b);
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < CH; k++) s += v[k];
  if (s == 1.2345e300) out[0] = s;
}
#include "su3.h"
__global__ void __launch_bounds__(256) k_expchain(double *out, int iters, double a) {
  M3 e;
#pragma unroll
  for (int k = 0; k < 9; k++) e.e[k] = make_double2(1e-7 * (threadIdx.x + k), -1e-7 * k * a);
#pragma unroll 1
  for (int i = 0; i < iters; i++) {
    M3 t = e;
    m3_add_diag(t, 2.0);
    e = m3_mul(e, t);
#pragma unroll
    for (int k = 0; k < 9; k++) { e.e[k].x *= 1e-3; e.e[k].y *= 1e-3; }   // keep the values bounded (18 extra multiplies per 216 flop)
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) s += e.e[k].x + e.e[k].y;
  if (s == 1.2345e300) out[0] = s;
}
// kind 2: m3_exp itself (prologue + 20 squarings + the closing product), `iters` times per wavefront, no memory traffic;
// kind 3: ONE m3_exp per wavefront and `iters` x as many wavefronts: the launch shape of k_exp_update without its loads
__global__ void __launch_bounds__(256) k_exponly(double *out, int iters, double a) {
  M3 v;
#pragma unroll
  for (int k = 0; k < 9; k++) v.e[k] = make_double2(1e-3 * (threadIdx.x + k) * a, -1e-3 * k * a);
  v = m3_tah(v);
  M3 u = v;
  m3_add_diag(u, 1.0);
#pragma unroll 1
  for (int i = 0; i < iters; i++) {
    u = m3_mul(m3_exp(v), u);
#pragma unroll
    for (int k = 0; k < 9; k++) { v.e[k].x = 0.5 * v.e[k].x + 1e-9 * u.e[k].y; v.e[k].y = 0.5 * v.e[k].y; }
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) s += u.e[k].x + u.e[k].y;
  if (s == 1.2345e300) out[0] = s;
}
extern "C" int qexhip_tune_fma64(qexhip_handle c, int kind, int chains, int wps, int iters, double *tflops) {
  if (!c || !tflops || wps < 1 || wps > 8) return -1;
  hipDeviceProp_t p;
  HIPCHK(hipGetDeviceProperties(&p, c->device));
  int nblocks = p.multiProcessorCount * wps;            // 256 threads = 4 wavefronts = one per SIMD of a CU
  if (kind == 3) nblocks *= iters;
  double *out;
  HIPCHK(hipMalloc((void **)&out, 64));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  auto run = [&]() {
    if (kind == 1) k_expchain<<<nblocks, 256, 0, c->stream>>>(out, iters, 0.5);
    else if (kind == 2) k_exponly<<<nblocks, 256, 0, c->stream>>>(out, iters, 0.5);
    else if (kind == 3) k_exponly<<<nblocks, 256, 0, c->stream>>>(out, 1, 0.5);
    else if (chains <= 2) k_fma64<2><<<nblocks, 256, 0, c->stream>>>(out, iters, 0.999, 1e-3);
    else if (chains <= 4) k_fma64<4><<<nblocks, 256, 0, c->stream>>>(out, iters, 0.999, 1e-3);
    else if (chains <= 8) k_fma64<8><<<nblocks, 256, 0, c->stream>>>(out, iters, 0.999, 1e-3);
    else k_fma64<16><<<nblocks, 256, 0, c->stream>>>(out, iters, 0.999, 1e-3);
  };
  run();
  HIPCHK(hipEventRecord(e0, c->stream));
  run();
  HIPCHK(hipEventRecord(e1, c->stream));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  const int ch = kind == 1 ? 0 : (chains <= 2 ? 2 : chains <= 4 ? 4 : chains <= 8 ? 8 : 16);
  const double flop = kind == 1 ? (216.0 + 18.0) * iters : kind == 2 ? 23.0 * 216.0 * iters : kind == 3 ? 23.0 * 216.0 : 2.0 * ch * iters;
  *tflops = flop * 256.0 * nblocks / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(out);
  return 0;
}


// ---- what the CU's L2 -> L1 path delivers for the Wilson-flow stage's operand stream (round 3) ----
// The 48 operand matrices of a 64-site tile (the loader / consumer flow stage of round 3, removed: 8 own/back links, 36 plane operands, 4 momenta ~ here all taken
// from U), gathered by NW wavefronts per workgroup into REGISTERS, D matrices in flight per wavefront, nothing else: no LDS,
// no barrier, 18 integer ops per matrix to keep the loads alive.  Persistent workgroups walk tile_order_table.
#include "gauge_index.h"
// R = 9: whole matrices; R = 6: rows 0,1 only (round 6: would gathering 2/3 of every link -- and rebuilding row 2 of an SU(3) link in
// registers -- shorten the stream?  profiles/r06_notes.md section 4)
template <int D, int R = 9>
__global__ void __launch_bounds__(512) k_gather_test(Geom g, const double2 *__restrict__ U, const int *__restrict__ order, int chunk,
                                                       unsigned long long *out) {
  const int nw = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3, jstride = gridDim.x >> 3;
  const int *ord = order + (size_t)xcd * chunk;
  M3 st[D];
  unsigned long long acc = 0;
  // unit u of this wavefront: tile u / npw of the workgroup's list, kind wave + nw * (u % npw); npw kinds per wavefront and tile
  const int npw = 48 / nw;
  int ntile = 0;
  for (int j = j0; j < chunk; j += jstride) { if (ord[j] < 0) break; ntile++; }
  const int nunit = ntile * npw;
  FsSite s;
  int cur = -1;
  auto src = [&](int u) -> const double2 * {
    const int uc = u < nunit ? u : nunit - 1;
    const int ti = uc / npw, k = wave + nw * (uc - ti * npw);
    if (ti != cur) {
      const int e = ord[j0 + ti * jstride];
      int c = (e >> 1) * 64 + lane;
      if (c >= g.Vh) c = g.Vh - 1;
      fs_site(g, c, e & 1, s);
      cur = ti;
    }
    // kind k of 48: group k / 4 (0 own, 1 back, 2..10 the rounds, 11 momenta ~ own), slot w = k % 4
    const int gi = k >> 2, w = k & 3;
    int lex = s.lex, par = s.par, mu = w;
    if (gi == 1) fs_hop<false>(g, s, w, -1, lex, par);
    else if (gi >= 2 && gi <= 10) {
      const int r = (gi - 2) / 3 + 1, kk = (gi - 2) % 3, q = w ^ r;
      if (kk == 0) fs_hop<false>(g, s, q, 1, lex, par);
      else if (kk == 1) { fs_hop<false>(g, s, w, -1, lex, par); mu = q; }
      else { fs_hop<false>(g, s, w, -1, lex, par); fs_hop<false>(g, s, q, 1, lex, par); }
    }
    return U + fs_link_off(g, lex, par, mu);
  };
  if (nunit == 0) return;
  auto gload = [&](M3 &m, const double2 *p) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < R; k++) m.e[k] = p[(size_t)k * 64];
  };
#pragma unroll
  for (int d = 0; d < D; d++) { gload(st[d], src(d)); __builtin_amdgcn_sched_barrier(0); }
  for (int u0 = 0; u0 < nunit; u0 += D) {
#pragma unroll
    for (int d = 0; d < D; d++) {
#pragma unroll
      for (int k = 0; k < R; k++) acc ^= (unsigned long long)__double_as_longlong(st[d].e[k].x) + (unsigned long long)__double_as_longlong(st[d].e[k].y);
      gload(st[d], src(u0 + d + D));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (acc == 0x123456789abcdefull) out[0] = acc;
}
// nw wavefronts per workgroup (divides 48), wgpc workgroups per CU, depth matrices in flight per wavefront (2, 4, 6, 8);
// returns the average launch time; the lattice's resident links (qexhip_gauge_set) are the table
extern "C" int qexhip_tune_gather_rows(qexhip_handle c, int nw, int wgpc, int depth, int rows, int nrep, double *avg_us);
extern "C" int qexhip_tune_gather(qexhip_handle c, int nw, int wgpc, int depth, int nrep, double *avg_us) {
  return qexhip_tune_gather_rows(c, nw, wgpc, depth, 3, nrep, avg_us);
}
// rows = 3: whole matrices, 2: rows 0,1 only
extern "C" int qexhip_tune_gather_rows(qexhip_handle c, int nw, int wgpc, int depth, int rows, int nrep, double *avg_us) {
  if (!c || !avg_us || nw < 1 || nw > 8 || 48 % nw || wgpc < 1 || rows < 2 || rows > 3) return -1;
  const double2 *U = gauge_links_dev(c);
  if (!U) { qexhip_set_error("tune_gather: qexhip_gauge_set first"); return -3; }
  const int *order = nullptr; int chunk = 0;
  CHK(tile_order_table(c, &order, &chunk));
  unsigned long long *out;
  HIPCHK(hipMalloc((void **)&out, 8));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  const int nb = 256 * wgpc;
  auto run = [&]() {
    if (rows == 2) {
      if (depth == 2) k_gather_test<2, 6><<<nb, 64 * nw, 0, c->stream>>>(c->g, U, order, chunk, out);
      else if (depth == 4) k_gather_test<4, 6><<<nb, 64 * nw, 0, c->stream>>>(c->g, U, order, chunk, out);
      else if (depth == 6) k_gather_test<6, 6><<<nb, 64 * nw, 0, c->stream>>>(c->g, U, order, chunk, out);
      else k_gather_test<1, 6><<<nb, 64 * nw, 0, c->stream>>>(c->g, U, order, chunk, out);
    } else if (depth == 2) k_gather_test<2><<<nb, 64 * nw, 0, c->stream>>>(c->g, U, order, chunk, out);
    else if (depth == 4) k_gather_test<4><<<nb, 64 * nw, 0, c->stream>>>(c->g, U, order, chunk, out);
    else if (depth == 6) k_gather_test<6><<<nb, 64 * nw, 0, c->stream>>>(c->g, U, order, chunk, out);
    else k_gather_test<1><<<nb, 64 * nw, 0, c->stream>>>(c->g, U, order, chunk, out);
  };
  run();
  HIPCHK(hipEventRecord(e0, c->stream));
  for (int i = 0; i < nrep; i++) run();
  HIPCHK(hipEventRecord(e1, c->stream));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  *avg_us = 1e3 * ms / nrep;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(out);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- round 5: the same 48-operand gather stream with the links interpreted in a BRICK tile shape (round-4 verdict, Next 3) ----
// A tile position = an 8(x) x 4(y) x 4(z) x 1(t) brick = 128 sites, 64 per parity; lane = (xi >> 1) + 4 * (yi + 4 * zi).  In-tile hop
// fraction 7/8 + 3/4 + 3/4 + 0 = 2.375 of 4 against 1.75 for the product's 128 consecutive sites (4 x-rows).  The buffer is the
// product's link field read as [parity][brick][mu][9][64] (same size when 8 | X, 4 | Y, 4 | Z: only the address pattern matters to a
// gather-rate measurement).  Same operand kinds, same persistent-workgroup walk, visiting order built the same way (one contiguous
// (t, z) region per XCD, walked in 8y x 4z x 4t blocks, parities adjacent).
struct BrickGeom { int X[4], nb[3], nbrick; };
struct BrickSite { int b[3], i[3], t, par; };      // brick coordinates, in-brick coordinates
__device__ __forceinline__ void brick_site(const BrickGeom &g, int brick, int par, int lane, BrickSite &s) {
  unsigned r = (unsigned)brick;
  s.b[0] = r % g.nb[0]; r /= g.nb[0];
  s.b[1] = r % g.nb[1]; r /= g.nb[1];
  s.b[2] = r % g.nb[2]; s.t = r / g.nb[2];
  const int xh = lane & 3, yi = (lane >> 2) & 3, zi = lane >> 4;
  s.i[1] = yi; s.i[2] = zi;
  s.i[0] = 2 * xh + ((yi + zi + s.t + par) & 1);           // brick origins are even in every coordinate
  s.par = par;
}
// one hop of +-1 in direction d (wavefront-uniform) -> (brick, lane, parity) of the neighbour
__device__ __forceinline__ void brick_hop(const BrickGeom &g, BrickSite &s, int d, int sgn) {
  s.par ^= 1;
  if (d == 3) { s.t += sgn; s.t = s.t >= g.X[3] ? 0 : (s.t < 0 ? g.X[3] - 1 : s.t); return; }
  const int ext = d == 0 ? 8 : 4;
  int v = (d == 0 ? s.i[0] : (d == 1 ? s.i[1] : s.i[2])) + sgn;
  int b = d == 0 ? s.b[0] : (d == 1 ? s.b[1] : s.b[2]);
  const int nb = d == 0 ? g.nb[0] : (d == 1 ? g.nb[1] : g.nb[2]);
  if (v >= ext) { v = 0; b = b + 1 >= nb ? 0 : b + 1; }
  if (v < 0) { v = ext - 1; b = b == 0 ? nb - 1 : b - 1; }
  if (d == 0) { s.i[0] = v; s.b[0] = b; } else if (d == 1) { s.i[1] = v; s.b[1] = b; } else { s.i[2] = v; s.b[2] = b; }
}
__device__ __forceinline__ unsigned brick_link_off(const BrickGeom &g, const BrickSite &s, int mu) {
  const unsigned brick = s.b[0] + g.nb[0] * (s.b[1] + g.nb[1] * (s.b[2] + g.nb[2] * (unsigned)s.t));
  const unsigned lane = (s.i[0] >> 1) + 4 * (s.i[1] + 4 * s.i[2]);
  return ((unsigned)(s.par * g.nbrick + brick) * 4u + (unsigned)mu) * 576u + lane;
}
template <int D>
__global__ void __launch_bounds__(512) k_gather_brick(BrickGeom g, const double2 *__restrict__ U, const int *__restrict__ order, int chunk,
                                                        unsigned long long *out) {
  const int nw = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3, jstride = gridDim.x >> 3;
  const int *ord = order + (size_t)xcd * chunk;
  M3 st[D];
  unsigned long long acc = 0;
  const int npw = 48 / nw;
  int ntile = 0;
  for (int j = j0; j < chunk; j += jstride) { if (ord[j] < 0) break; ntile++; }
  const int nunit = ntile * npw;
  BrickSite s0;
  int cur = -1;
  auto src = [&](int u) -> const double2 * {
    const int uc = u < nunit ? u : nunit - 1;
    const int ti = uc / npw, k = wave + nw * (uc - ti * npw);
    if (ti != cur) {
      const int e = ord[j0 + ti * jstride];
      brick_site(g, e >> 1, e & 1, lane, s0);
      cur = ti;
    }
    const int gi = k >> 2, w = k & 3;
    BrickSite s = s0;
    int mu = w;
    if (gi == 1) brick_hop(g, s, w, -1);
    else if (gi >= 2 && gi <= 10) {
      const int r = (gi - 2) / 3 + 1, kk = (gi - 2) % 3, q = w ^ r;
      if (kk == 0) brick_hop(g, s, q, 1);
      else if (kk == 1) { brick_hop(g, s, w, -1); mu = q; }
      else { brick_hop(g, s, w, -1); brick_hop(g, s, q, 1); }
    }
    return U + brick_link_off(g, s, mu);
  };
  if (nunit == 0) return;
#pragma unroll
  for (int d = 0; d < D; d++) { st[d] = m3_load(src(d), 64); __builtin_amdgcn_sched_barrier(0); }
  for (int u0 = 0; u0 < nunit; u0 += D) {
#pragma unroll
    for (int d = 0; d < D; d++) {
#pragma unroll
      for (int k = 0; k < 9; k++) acc ^= (unsigned long long)__double_as_longlong(st[d].e[k].x) + (unsigned long long)__double_as_longlong(st[d].e[k].y);
      st[d] = m3_load(src(u0 + d + D), 64);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (acc == 0x123456789abcdefull) out[0] = acc;
}
// as qexhip_tune_gather, brick tile shape; lattices with 8 | X, 4 | Y, 4 | Z only
extern "C" int qexhip_tune_gather_brick(qexhip_handle c, int nw, int wgpc, int depth, int nrep, double *avg_us) {
  if (!c || !avg_us || nw < 1 || nw > 8 || 48 % nw || wgpc < 1) return -1;
  const double2 *U = gauge_links_dev(c);
  if (!U) { qexhip_set_error("tune_gather_brick: qexhip_gauge_set first"); return -3; }
  const Geom &gg = c->g;
  if (gg.X[0] % 8 || gg.X[1] % 4 || gg.X[2] % 4 || gg.halo) { qexhip_set_error("tune_gather_brick: needs 8 | X, 4 | Y, 4 | Z, no halo"); return -1; }
  BrickGeom g;
  for (int i = 0; i < 4; i++) g.X[i] = gg.X[i];
  g.nb[0] = gg.X[0] / 8; g.nb[1] = gg.X[1] / 4; g.nb[2] = gg.X[2] / 4;
  g.nbrick = g.nb[0] * g.nb[1] * g.nb[2] * gg.X[3];
  if (g.nbrick != gg.ntile) { qexhip_set_error("tune_gather_brick: brick count != tile count"); return -1; }
  // visiting order: 8 XCD regions contiguous in (t, z), each walked in blocks of 8 y x 4 z x 4 t (2 x 1 x 4 bricks, all of x), parities adjacent
  const int n = 2 * g.nbrick, chunk = (n + 7) / 8;
  struct Ent { unsigned long long key0, key1; int e; };
  std::vector<Ent> v(n);
  for (int p = 0; p < 2; p++)
    for (int b = 0; b < g.nbrick; b++) {
      unsigned r = (unsigned)b;
      const int bx = r % g.nb[0]; r /= g.nb[0];
      const int by = r % g.nb[1]; r /= g.nb[1];
      const int bz = r % g.nb[2], t = r / g.nb[2];
      Ent &a = v[(size_t)p * g.nbrick + b];
      a.e = 2 * b + p;
      a.key0 = ((((unsigned long long)t * 1024u + bz) * 1024u + by) * 1024u + bx) * 2 + p;
      a.key1 = (((((((unsigned long long)(t / 4) * 1024u + bz) * 1024u + by / 2) * 1024u + t % 4) * 1024u + by % 2) * 1024u + bx)) * 2 + p;
    }
  std::sort(v.begin(), v.end(), [](const Ent &a, const Ent &b) { return a.key0 < b.key0; });
  std::vector<int> h((size_t)8 * chunk, -1);
  for (int k = 0; k < 8; k++) {
    const int lo = std::min(n, k * chunk), hi = std::min(n, (k + 1) * chunk);
    std::sort(v.begin() + lo, v.begin() + hi, [](const Ent &a, const Ent &b) { return a.key1 < b.key1; });
    for (int j = lo; j < hi; j++) h[(size_t)k * chunk + (j - lo)] = v[j].e;
  }
  int *order;
  HIPCHK(hipMalloc((void **)&order, h.size() * sizeof(int)));
  HIPCHK(hipMemcpy(order, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
  unsigned long long *out;
  HIPCHK(hipMalloc((void **)&out, 8));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  const int nblk = 256 * wgpc;
  auto run = [&]() {
    if (depth == 2) k_gather_brick<2><<<nblk, 64 * nw, 0, c->stream>>>(g, U, order, chunk, out);
    else if (depth == 4) k_gather_brick<4><<<nblk, 64 * nw, 0, c->stream>>>(g, U, order, chunk, out);
    else k_gather_brick<1><<<nblk, 64 * nw, 0, c->stream>>>(g, U, order, chunk, out);
  };
  run();
  HIPCHK(hipEventRecord(e0, c->stream));
  for (int i = 0; i < nrep; i++) run();
  HIPCHK(hipEventRecord(e1, c->stream));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  *avg_us = 1e3 * ms / nrep;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(out); (void)hipFree(order);
  HIPCHK(hipGetLastError());
  return 0;
}
