// api.cpp -- the extern "C" boundary declared in include/qexhip.h.
// Host-pointer entry points upload into work fields, run the device path, download the result:
// the same choreography as qudaSolveXX (src/quda/qudaWrapperImpl.nim:165-261).
#include "qexhip_internal.h"
#include <string>
#include "../../include/qexhip.h"
#include <algorithm>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <vector>

static thread_local char g_err[1024] = "";

void qexhip_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char *qexhip_last_error(void) { return g_err; }

// ---- timers ----
// levels: 1 everything; 2 only the Dslash sweeps (the dominant kernel), so that the event records perturb a timed region as
// little as possible; 3 the sharded iteration's anatomy: Dslash sweeps (interior "dslash", faces "dslash_bnd"), the face
// exchange ("exchange": events on the stream the RCCL group is posted on) and the all-reduces ("allreduce")
static bool timer_class_on(const qexhip_ctx *c, const char *name) {
  if (!c->timers_on) return false;
  if (c->timers_on == 2) return strncmp(name, "dslash", 6) == 0;
  if (c->timers_on == 3) return strncmp(name, "dslash", 6) == 0 || strcmp(name, "exchange") == 0 || strcmp(name, "allreduce") == 0;
  return true;
}
ScopedTimer::ScopedTimer(qexhip_ctx *c_, const char *name, hipStream_t st_) : c(c_), st(st_) {
  if (!c->timers_on || !name) return;
  if (!timer_class_on(c, name)) return;
  s = &c->timers[name];
  if (s->used + 2 > s->ev.size()) {
    size_t old = s->ev.size();
    s->ev.resize(old + 512);
    for (size_t i = old; i < s->ev.size(); i++) (void)hipEventCreate(&s->ev[i]);
  }
  (void)hipEventRecord(s->ev[s->used], st);
}
ScopedTimer::~ScopedTimer() {
  if (!s) return;
  (void)hipEventRecord(s->ev[s->used + 1], st);
  s->used += 2;
}
bool timer_event_pair(qexhip_ctx *c, const char *name, hipEvent_t *e0, hipEvent_t *e1) {
  if (!timer_class_on(c, name)) return false;
  TimerSlot *s = &c->timers[name];
  if (s->used + 2 > s->ev.size()) {
    size_t old = s->ev.size();
    s->ev.resize(old + 512);
    for (size_t i = old; i < s->ev.size(); i++) (void)hipEventCreate(&s->ev[i]);
  }
  *e0 = s->ev[s->used];
  *e1 = s->ev[s->used + 1];
  s->used += 2;
  return true;
}
int timers_collect(qexhip_ctx *c) {
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipStreamSynchronize(c->cstream));   // "exchange" events of an overlapped sweep live there
  for (auto &kv : c->timers) {
    TimerSlot &s = kv.second;
    for (size_t i = 0; i + 1 < s.used; i += 2) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, s.ev[i], s.ev[i + 1]) == hipSuccess) { s.total_ms += ms; s.count++; }
    }
    s.used = 0;
  }
  return 0;
}

extern "C" int qexhip_timers_enable(qexhip_handle c, int on) { if (!c) return QEXHIP_ERR_ARG; c->timers_on = on; return 0; }
extern "C" int qexhip_timers_reset(qexhip_handle c) {
  if (!c) return QEXHIP_ERR_ARG;
  CHK(timers_collect(c));
  for (auto &kv : c->timers) { kv.second.count = 0; kv.second.total_ms = 0; }
  return 0;
}
extern "C" int qexhip_timers_get(qexhip_handle c, const char *name, long *count, double *total_ms) {
  if (!c || !name) return QEXHIP_ERR_ARG;
  CHK(timers_collect(c));
  auto it = c->timers.find(name);
  if (count) *count = (it == c->timers.end()) ? 0 : it->second.count;
  if (total_ms) *total_ms = (it == c->timers.end()) ? 0.0 : it->second.total_ms;
  return 0;
}

// ---- context ----
static int init_body(qexhip_ctx *c, int device, const int latLocal[4], const int rankGeom[4], const int rankCoord[4]) {
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) { qexhip_set_error("no HIP device visible: libqexhip has no CPU fallback"); return QEXHIP_ERR_HIP; }
  if (device < 0 || device >= ndev) { qexhip_set_error("device %d out of range (%d visible)", device, ndev); return QEXHIP_ERR_ARG; }
  c->device = device;
  for (int i = 0; i < 4; i++) {
    c->rankGeom[i] = rankGeom ? rankGeom[i] : 1;
    c->rankCoord[i] = rankCoord ? rankCoord[i] : 0;
  }
  for (int i = 0; i < 3; i++)
    if (c->rankGeom[i] != 1) { qexhip_set_error("only rankGeom = {1,1,1,N} (split along t) is supported"); return QEXHIP_ERR_ARG; }
  const int halo = c->rankGeom[3] > 1;
  if (geom_init(c->g, latLocal, 1, halo)) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  {
    // the comm stream carries the face exchange and the boundary launch that waits for it: at the highest priority, so that the
    // exchange kernel is dispatched at once and not behind the thousands of interior workgroups the compute stream has queued
    // (one-rank rehearsal, where nothing has to be hidden: no difference either way, profiles/r04_prio_ab.log)
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo) HIPCHK(hipStreamCreateWithPriority(&c->cstream, hipStreamNonBlocking, hi));
    else HIPCHK(hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking));
  }
  HIPCHK(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
  c->part2_off = std::max(6144, (c->g.Vh + 255) / 256 + 8);   // >= 6*1024 for the plaquette partials
  c->npartials = c->part2_off + 2048 + 64;
  HIPCHK(hipMalloc((void **)&c->partials, sizeof(double) * c->npartials));
  HIPCHK(hipMalloc((void **)&c->dscal, sizeof(double) * 64));
  HIPCHK(hipMemset(c->dscal, 0, sizeof(double) * 64));
  HIPCHK(hipMalloc((void **)&c->cg, sizeof(CgScal)));
  HIPCHK(hipMemset(c->cg, 0, sizeof(CgScal)));
  HIPCHK(hipHostMalloc(&c->pinned, 4096, hipHostMallocDefault));
  CHK(devjoin_init(c));
  if (getenv("QEXHIP_TEST_FAIL_INIT")) {      // test hook (tests/test_gpu_misc_ops.py): fail the way a late HIP error would, everything above built
    qexhip_set_error("QEXHIP_TEST_FAIL_INIT: injected failure");
    return QEXHIP_ERR_HIP;
  }
  // the environment switches of the library (include/qexhip.h "Environment")
  if (const char *e = getenv("QEXHIP_OVERLAP")) c->opt_overlap = atoi(e);
  if (const char *e = getenv("QEXHIP_HOP_SPLIT")) { const int v = atoi(e); c->opt_hop_split = v < 0 ? -1 : (v ? 2 : 0); }
  if (const char *e = getenv("QEXHIP_RECON")) c->opt_recon = atoi(e);
  if (const char *e = getenv("QEXHIP_FLOW_EXP")) c->opt_flow_exp = atoi(e);
  {
    // the gauge kernels stage links through 144 KiB (k_force_lds2) / 72 KiB (k_force_lds, k_flow_obs_clover) of LDS per
    // workgroup: gfx950 offers 160 KiB.  The library is built for that one target (qex_amd/Makefile pins gfx950); anything
    // smaller is refused here rather than at the first flow step.
    int a = 0, b = 0;
    (void)hipDeviceGetAttribute(&a, hipDeviceAttributeMaxSharedMemoryPerBlock, device);
    (void)hipDeviceGetAttribute(&b, hipDeviceAttributeSharedMemPerBlockOptin, device);
    (void)hipGetLastError();
    const int lds = a > b ? a : b;
    if (lds < 147456 + 1024) {      // + the static LDS (reductions) the kernels add on top of the dynamic 144 KiB
      qexhip_set_error("device %d offers %d bytes of LDS per workgroup; libqexhip is built for gfx950 (160 KiB)", device, lds);
      return QEXHIP_ERR_STATE;
    }
  }
  c->nranks = 1;  // until qexhip_comm_init
  c->rank = 0;
  return 0;
}

extern "C" int qexhip_init(qexhip_handle *h, int device, const int latLocal[4], const int rankGeom[4],
                           const int rankCoord[4]) {
  if (!h || !latLocal) return QEXHIP_ERR_ARG;
  qexhip_ctx *c = new qexhip_ctx();
  const int e = init_body(c, device, latLocal, rankGeom, rankCoord);
  if (e) {                      // every failing path releases what was built so far (streams, events, device and pinned memory)
    (void)qexhip_finalize(c);
    return e;
  }
  *h = c;
  return 0;
}

extern "C" int qexhip_device_count(int *n) {
  if (!n) return QEXHIP_ERR_ARG;
  HIPCHK(hipGetDeviceCount(n));
  return 0;
}

extern "C" int qexhip_finalize(qexhip_handle c) {
  if (!c) return QEXHIP_ERR_ARG;
  if (c->stream) {                 // (a context that failed before its first HIP object has nothing on a device)
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
  }
  for (auto &kv : c->fields) (void)hipFree(kv.second.d);
  c->fields.clear();
  for (auto &kv : c->timers) for (auto e : kv.second.ev) (void)hipEventDestroy(e);
  nhyp_state_free(c);
  hisq_state_free(c);
  batch_state_free(c);
  gauge_free(c);
  comm_destroy(c);
  if (c->W) (void)hipFree(c->W);
  if (c->outer_F) (void)hipFree(c->outer_F);
  if (c->obs_table) (void)hipFree(c->obs_table);
  if (c->tile_order) (void)hipFree(c->tile_order);
  for (int *&t : c->tile_order_pl) if (t) { (void)hipFree(t); t = nullptr; }
  if (c->cgm_scal) (void)hipFree(c->cgm_scal);
  if (c->Wc) (void)hipFree(c->Wc);
  if (c->Ws) (void)hipFree(c->Ws);
  if (c->stage) (void)hipFree(c->stage);
  if (c->partials) (void)hipFree(c->partials);
  if (c->dscal) (void)hipFree(c->dscal);
  if (c->cg) (void)hipFree(c->cg);
  if (c->hist) (void)hipFree(c->hist);
  if (c->pinned) (void)hipHostFree(c->pinned);
  if (c->fz_buf) (void)hipFree(c->fz_buf);
  devjoin_destroy(c);
  if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->cstream) (void)hipStreamDestroy(c->cstream);
  (void)hipGetLastError();
  delete c;
  return 0;
}

extern "C" int qexhip_sync(qexhip_handle c) {
  if (!c) return QEXHIP_ERR_ARG;
  CHK(devjoin_flush(c));
  HIPCHK(hipStreamSynchronize(c->cstream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return devjoin_check(c);
}

extern "C" int qexhip_device_info(qexhip_handle c, char *buf, int buflen) {
  if (!c || !buf) return QEXHIP_ERR_ARG;
  hipDeviceProp_t p;
  HIPCHK(hipGetDeviceProperties(&p, c->device));
  // tile_pairs: -1 until a gather kernel has built the visiting order, then whether its slots pair the two parities of a
  // tile position (k_force_lds2 needs that; otherwise the one-tile kernels run)
  int n = snprintf(buf, buflen, "%s (%s) CUs=%d mem=%.0fGiB local=%dx%dx%dx%d ranks=%d halo=%d tile_pairs=%d", p.name, p.gcnArchName,
                   p.multiProcessorCount, p.totalGlobalMem / 1073741824.0, c->g.X[0], c->g.X[1], c->g.X[2], c->g.X[3],
                   c->rankGeom[3], c->g.halo, c->tile_order ? c->tile_pairs_ok : -1);
  // what set_links measured and decided for the sweeps of a t-sharded slab (qexhip_stag_sweep_tuning has the numbers as numbers):
  // timing-dependent decisions change the grouping of the dot partials, i.e. the last bits of a residual history -- keep them visible
  if (c->g.halo && c->W && n > 0 && n < buflen) {
    double t[8];
    if (qexhip_stag_sweep_tuning(c, t) == 0)
      snprintf(buf + n, buflen - n, "; sweep: overlap=%d form=%s exchange=%.1fus%s boundary_at=%.2f spin=%.0fus tuned[first|sites|fused]=%.0f|%.0f|%.0f us",
               (int)t[7], t[5] == 2 ? "fused" : "by-sites", t[0], c->xchg_us[c->ndir == 16] > 0 ? "(measured)" : "(estimate)", t[1], t[6], t[2], t[3], t[4]);
  }
  return 0;
}

static int drop_fields_for_regeom(qexhip_ctx *c) {
  c->cg_resume.valid = 0;
  // geometry (ghost zones) changed: user fields keep their ids but are re-allocated empty
  HIPCHK(hipStreamSynchronize(c->stream));
  for (auto &kv : c->fields) {
    HIPCHK(hipFree(kv.second.d));
    kv.second.d = nullptr;
    CHK(field_alloc(c, kv.second));
  }
  return 0;
}

extern "C" int qexhip_comm_force_halo(qexhip_handle c, int on) {
  if (!c) return QEXHIP_ERR_ARG;
  if (c->rankGeom[3] > 1) return 0;  // already on
  c->force_halo = on;
  int depth = c->g.depth ? c->g.depth : 1;
  if (geom_init(c->g, c->g.X, depth, on ? 1 : 0)) return QEXHIP_ERR_ARG;
  if (c->W) { HIPCHK(hipFree(c->W)); c->W = nullptr; c->ndir = 0; }
  return drop_fields_for_regeom(c);
}

// ---- staggered operator ----
extern "C" int qexhip_stag_set_links(qexhip_handle c, const double *fat, const double *lng) {
  if (!c || !fat) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  if (c->rankGeom[3] > 1 && !comm_ready(c)) { qexhip_set_error("rankGeom[3] = %d but qexhip_comm_init was not called", c->rankGeom[3]); return QEXHIP_ERR_STATE; }
  if (c->g.halo) {
    int depth = lng ? 3 : 1;
    if (c->g.X[3] < depth) { qexhip_set_error("local t extent %d < hop length %d", c->g.X[3], depth); return QEXHIP_ERR_ARG; }
    c->g.depth = depth;  // ghost tiles were allocated for depth 3
  }
  return links_upload(c, fat, lng);
}

static int host_in(qexhip_ctx *c, int slot, const double *host, DevField **f) {
  CHK(get_work(c, slot, f));
  return field_upload(c, **f, host);
}

extern "C" int qexhip_stag_dslash(qexhip_handle c, double *r, const double *x, int parity, double a, double b) {
  if (!c || !r || !x || parity < 0 || parity > 2) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fx, *fr;
  CHK(host_in(c, WK_IN, x, &fx));
  CHK(host_in(c, WK_OUT, r, &fr));  // r is read when a != 0, and the untouched parity is preserved
  for (int p = (parity == 2 ? 0 : parity); p <= (parity == 2 ? 1 : parity); p++) {
    DslashOpts o;
    o.ca = a; o.cb = b; o.rin = fr; o.xs = fx;
    CHK(dslash_sweep(c, *fr, *fx, p, o));
  }
  return field_download(c, *fr, r);
}

extern "C" int qexhip_stag_D(qexhip_handle c, double *r, const double *x, double m, double sc) {
  if (!c || !r || !x || sc == 0.0) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fx, *fr;
  CHK(host_in(c, WK_IN, x, &fx));
  CHK(get_work(c, WK_OUT, &fr));
  CHK(op_D(c, *fr, *fx, m, sc));
  return field_download(c, *fr, r);
}

extern "C" int qexhip_stag_D_acc(qexhip_handle c, double *r, const double *x, double m, double sc, double a) {
  if (!c || !r || !x || sc == 0.0) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fx, *fr;
  CHK(host_in(c, WK_IN, x, &fx));
  CHK(host_in(c, WK_OUT, r, &fr));
  CHK(op_D(c, *fr, *fx, m, sc, a));
  return field_download(c, *fr, r);
}

extern "C" int qexhip_stag_op_xx(qexhip_handle c, double *r, const double *x, double m2, int par_even) {
  if (!c || !r || !x) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fx, *fr;
  CHK(host_in(c, WK_IN, x, &fx));
  CHK(host_in(c, WK_OUT, r, &fr));
  CHK(op_xx(c, *fr, *fx, m2, par_even, 0, nullptr));
  return field_download(c, *fr, r);
}

int op_eo_reconstruct_pub(qexhip_ctx *c, DevField &r, DevField &b, double m);
extern "C" int qexhip_stag_eo_reconstruct(qexhip_handle c, double *r, const double *b, double m) {
  if (!c || !r || !b || m == 0.0) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fb, *fr;
  CHK(host_in(c, WK_IN, b, &fb));
  CHK(host_in(c, WK_OUT, r, &fr));
  CHK(op_eo_reconstruct_pub(c, *fr, *fb, m));
  return field_download(c, *fr, r);
}

extern "C" int qexhip_stag_sweep_info(qexhip_handle c, int out[8]) {
  if (!c || !out) return QEXHIP_ERR_ARG;
  int lo_end = 0, hi_beg = 0, overlap = 0;
  sweep_plan(c, &lo_end, &hi_beg, &overlap);
  const int slot = c->ndir == 16;
  out[0] = c->g.halo;
  out[1] = overlap;
  out[2] = c->g.halo ? hi_beg - lo_end : c->g.Vh;
  out[3] = c->g.halo ? c->g.depth * c->g.F * 48 : 0;
  out[4] = c->overlap_auto[slot] >= 0;
  out[5] = (int)(c->overlap_tune_us[slot][0] + 0.5);
  out[6] = (int)(std::max(c->overlap_tune_us[slot][1], 0.0) + 0.5);
  out[7] = c->opt_overlap;
  return 0;
}

extern "C" int qexhip_stag_sweep_tuning(qexhip_handle c, double out[8]) {
  if (!c || !out) return QEXHIP_ERR_ARG;
  int lo_end = 0, hi_beg = 0, overlap = 0;
  sweep_plan(c, &lo_end, &hi_beg, &overlap);
  const int slot = c->ndir == 16;
  const double meas = c->xchg_us[slot];
  // (the estimate the placement falls back on where nothing was measured: dslash.hip exchange_estimate_us)
  const double link = (c->emu_link_gbs > 0 ? c->emu_link_gbs : 45.0) * 1e3;
  out[0] = meas > 0 ? meas : (c->g.halo ? 3.0 + (double)c->g.depth * c->g.F * 48.0 / link : 0.0);
  out[1] = c->g.halo && c->ndir ? sweep_push_fraction(c, hi_beg - lo_end) : 0.0;
  for (int k = 0; k < 3; k++) out[2 + k] = c->overlap_tune_us[slot][k];
  out[5] = sweep_form(c, overlap);
  out[6] = c->opt_fused_spin_us == -2 ? -1.0 : (c->opt_fused_spin_us >= 0 ? (double)c->opt_fused_spin_us : std::max(25.0, out[0]));
  out[7] = overlap;
  return 0;
}

extern "C" int qexhip_stag_stagD(qexhip_handle c, double *r, const double *x, int parity, double m, double sc, double a) {
  if (!c || !r || !x || parity < 0 || parity > 2 || sc == 0.0) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fx, *fr;
  CHK(host_in(c, WK_IN, x, &fx));
  CHK(host_in(c, WK_OUT, r, &fr));       // r is read when a != 0; the other subset is kept
  for (int p = (parity == 2 ? 0 : parity); p <= (parity == 2 ? 1 : parity); p++) CHK(op_stagD_pub(c, *fr, *fx, p, m, sc, a));
  return field_download(c, *fr, r);
}

extern "C" int qexhip_stag_eo_reduce(qexhip_handle c, double *r, const double *b, double m) {
  if (!c || !r || !b) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fb, *fr;
  CHK(host_in(c, WK_IN, b, &fb));
  CHK(host_in(c, WK_OUT, r, &fr));       // the odd half of r is kept
  CHK(op_eo_reduce_pub(c, *fr, *fb, m));
  return field_download(c, *fr, r);
}

extern "C" int qexhip_stag_outer(qexhip_handle c, double *f, const double *x, double scale_even, double scale_odd,
                                 int accumulate) {
  if (!c || !f || !x) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  return stag_outer_host(c, f, x, scale_even, scale_odd, accumulate);
}

// ---- solvers ----
extern "C" int qexhip_stag_solve_xx(qexhip_handle c, double *x, const double *b, double mass, double r2req,
                                    int maxits, int par_even, int *iters, double *r2_over_b2, double *hist,
                                    int histcap) {
  if (!c || !x || !b) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fb, *fx;
  CHK(host_in(c, WK_IN, b, &fb));
  CHK(get_work(c, WK_OUT, &fx));
  CHK(solve_xx_dev(c, *fx, *fb, mass, r2req, maxits, par_even, iters, r2_over_b2, hist, histcap));
  return field_download(c, *fx, x);
}

extern "C" int qexhip_stag_solve(qexhip_handle c, double *x, const double *b, double mass, double r2req,
                                 int maxits, int *iters, double *r2_final) {
  if (!c || !x || !b) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fb, *fx;
  CHK(host_in(c, WK_IN, b, &fb));
  CHK(get_work(c, WK_OUT, &fx));
  CHK(solve_full_dev(c, *fx, *fb, mass, r2req, maxits, iters, r2_final));
  return field_download(c, *fx, x);
}

extern "C" int qexhip_stag_solve_prev(qexhip_handle c, double *x, const double *b, double mass, double r2req,
                                      int maxits, int use_prev, int *iters, double *r2_final) {
  if (!c || !x || !b) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fb, *fx;
  CHK(host_in(c, WK_IN, b, &fb));
  if (use_prev) CHK(host_in(c, WK_OUT, x, &fx));
  else CHK(get_work(c, WK_OUT, &fx));
  CHK(solve_full_dev(c, *fx, *fb, mass, r2req, maxits, iters, r2_final, use_prev));
  return field_download(c, *fx, x);
}

static int multi_common(qexhip_ctx *c, double *const *xs, const double *b, const double *vals, int nmass,
                        double r2req, int maxits, int par_even, int full, int *iters, double *out, double *hist,
                        int histcap) {
  if (!c || !xs || !b || !vals || nmass < 1) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fb;
  CHK(host_in(c, WK_IN, b, &fb));
  if (nmass > 32) { qexhip_set_error("multishift: nmass <= 32"); return QEXHIP_ERR_ARG; }
  std::vector<DevField *> xp(nmass);
  for (int k = 0; k < nmass; k++) CHK(pool_field(c, POOL_XS + k, &xp[k]));   // persistent: nothing to free on any path
  if (full) CHK(solve_multi_dev(c, xp, *fb, vals, nmass, r2req, maxits, iters, out));
  else CHK(solve_xx_multi_dev(c, xp, *fb, vals, nmass, r2req, maxits, par_even, iters, hist, histcap));
  for (int k = 0; k < nmass; k++) CHK(field_download(c, *xp[k], xs[k]));
  return 0;
}

extern "C" int qexhip_stag_solve_xx_multi(qexhip_handle c, double *const *xs, const double *b, const double *shifts,
                                          int nmass, double r2req, int maxits, int par_even, int *iters,
                                          double *hist, int histcap) {
  return multi_common(c, xs, b, shifts, nmass, r2req, maxits, par_even, 0, iters, nullptr, hist, histcap);
}
extern "C" int qexhip_stag_solve_multi(qexhip_handle c, double *const *xs, const double *b, const double *masses,
                                       int nmass, double r2req, int maxits, int *iters, double *r2_final) {
  return multi_common(c, xs, b, masses, nmass, r2req, maxits, 1, 1, iters, r2_final, nullptr, 0);
}

// ---- field algebra hooks ----
extern "C" int qexhip_norm2(qexhip_handle c, const double *x, int parity, double *out) {
  if (!c || !x || !out) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fx;
  CHK(host_in(c, WK_IN, x, &fx));
  CHK(blas_norm2(c, *fx, parity, &c->dscal[8]));
  return read_scalars(c, &c->dscal[8], 1, out);
}
extern "C" int qexhip_redot(qexhip_handle c, const double *x, const double *y, int parity, double *out) {
  if (!c || !x || !y || !out) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fx, *fy;
  CHK(host_in(c, WK_IN, x, &fx));
  CHK(host_in(c, WK_IN2, y, &fy));
  CHK(blas_redot(c, *fx, *fy, parity, &c->dscal[8]));
  return read_scalars(c, &c->dscal[8], 1, out);
}
extern "C" int qexhip_dot(qexhip_handle c, const double *x, const double *y, int parity, double out[2]) {
  if (!c || !x || !y || !out || parity < 0 || parity > 2) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fx, *fy;
  CHK(host_in(c, WK_IN, x, &fx));
  CHK(host_in(c, WK_IN2, y, &fy));
  CHK(blas_cdot(c, *fx, *fy, parity, &c->dscal[8]));
  return read_scalars(c, &c->dscal[8], 2, out);
}
extern "C" int qexhip_axpy(qexhip_handle c, double a, const double *x, double *y, int parity) {
  if (!c || !x || !y) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fx, *fy;
  CHK(host_in(c, WK_IN, x, &fx));
  CHK(host_in(c, WK_OUT, y, &fy));
  CHK(blas_axpy(c, a, *fx, *fy, parity));
  return field_download(c, *fy, y);
}
extern "C" int qexhip_xpay(qexhip_handle c, const double *x, double a, double *y, int parity) {
  if (!c || !x || !y) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fx, *fy;
  CHK(host_in(c, WK_IN, x, &fx));
  CHK(host_in(c, WK_OUT, y, &fy));
  CHK(blas_xpay(c, *fx, a, *fy, parity));
  return field_download(c, *fy, y);
}

// ---- device-resident fields ----
static int find_field(qexhip_ctx *c, int id, DevField **f) {
  auto it = c->fields.find(id);
  if (id <= 0 || it == c->fields.end()) { qexhip_set_error("unknown field id %d", id); return QEXHIP_ERR_ARG; }
  *f = &it->second;
  return 0;
}
extern "C" int qexhip_field_new(qexhip_handle c, int *id) {
  if (!c || !id) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField f;
  CHK(field_alloc(c, f));
  *id = c->next_field++;
  c->fields[*id] = f;
  return 0;
}
extern "C" int qexhip_field_free(qexhip_handle c, int id) {
  if (c) c->cg_resume.valid = 0;
  if (!c) return QEXHIP_ERR_ARG;
  DevField *f;
  CHK(find_field(c, id, &f));
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipFree(f->d));
  c->fields.erase(id);
  return 0;
}
extern "C" int qexhip_field_upload(qexhip_handle c, int id, const double *host) {
  if (!c || !host) return QEXHIP_ERR_ARG;
  DevField *f;
  CHK(find_field(c, id, &f));
  CHK(field_upload(c, *f, host));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}
extern "C" int qexhip_field_download(qexhip_handle c, int id, double *host) {
  if (!c || !host) return QEXHIP_ERR_ARG;
  DevField *f;
  CHK(find_field(c, id, &f));
  return field_download(c, *f, host);
}
extern "C" int qexhip_field_zero(qexhip_handle c, int id) {
  if (!c) return QEXHIP_ERR_ARG;
  DevField *f;
  CHK(find_field(c, id, &f));
  return blas_zero(c, *f, 2);
}
extern "C" int qexhip_dev_dslash(qexhip_handle c, int r_id, int x_id, int parity, double a, double b) {
  if (!c || parity < 0 || parity > 2) return QEXHIP_ERR_ARG;
  DevField *fr, *fx;
  CHK(find_field(c, r_id, &fr));
  CHK(find_field(c, x_id, &fx));
  for (int p = (parity == 2 ? 0 : parity); p <= (parity == 2 ? 1 : parity); p++) {
    DslashOpts o;
    o.ca = a; o.cb = b; o.rin = fr; o.xs = fx;
    CHK(dslash_sweep(c, *fr, *fx, p, o));
  }
  return 0;
}
extern "C" int qexhip_dev_op_xx(qexhip_handle c, int r_id, int x_id, double m2, int par_even) {
  if (!c) return QEXHIP_ERR_ARG;
  DevField *fr, *fx;
  CHK(find_field(c, r_id, &fr));
  CHK(find_field(c, x_id, &fx));
  return op_xx(c, *fr, *fx, m2, par_even, 0, nullptr);
}
extern "C" int qexhip_dev_solve_xx(qexhip_handle c, int x_id, int b_id, double mass, double r2req, int maxits,
                                   int par_even, int *iters, double *r2_over_b2, double *hist, int histcap) {
  if (!c) return QEXHIP_ERR_ARG;
  DevField *fx, *fb;
  CHK(find_field(c, x_id, &fx));
  CHK(find_field(c, b_id, &fb));
  return solve_xx_dev(c, *fx, *fb, mass, r2req, maxits, par_even, iters, r2_over_b2, hist, histcap);
}

extern "C" int qexhip_dev_solve_xx_continue(qexhip_handle c, int x_id, double r2req, int maxits, int *iters, double *r2_over_b2,
                                            double *hist, int histcap) {
  if (!c) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fx;
  CHK(find_field(c, x_id, &fx));
  return solve_xx_continue_dev(c, *fx, r2req, maxits, iters, r2_over_b2, hist, histcap);
}

extern "C" int qexhip_dev_solve_xx_multi(qexhip_handle c, const int *x_ids, int b_id, const double *shifts, int nmass,
                                         double r2req, int maxits, int par_even, int *iters, double *hist, int histcap) {
  if (!c || !x_ids || !shifts || nmass < 1 || nmass > 32) return QEXHIP_ERR_ARG;
  DevField *fb;
  CHK(find_field(c, b_id, &fb));
  std::vector<DevField *> xp(nmass);
  for (int k = 0; k < nmass; k++) {
    CHK(find_field(c, x_ids[k], &xp[k]));
    // the solver zeroes every xs[k] before it forms |b|^2: a solution field that is the source, or another solution, would
    // silently give b2 = 0 / one field holding the last shift only
    if (x_ids[k] == b_id) { qexhip_set_error("dev_solve_xx_multi: x_ids[%d] is the source field", k); return QEXHIP_ERR_ARG; }
    for (int j = 0; j < k; j++)
      if (x_ids[j] == x_ids[k]) { qexhip_set_error("dev_solve_xx_multi: x_ids[%d] == x_ids[%d]", j, k); return QEXHIP_ERR_ARG; }
  }
  return solve_xx_multi_dev(c, xp, *fb, shifts, nmass, r2req, maxits, par_even, iters, hist, histcap);
}

// The multi-shift solvers keep up to 3 x nmass full fields (search directions, per-parity and host-entry solutions) between
// calls so that a trajectory's repeated solves allocate nothing; a host that is about to need the memory for something
// else (the 52-field nHYP closure, a larger batch) hands it back here.  The next multi-shift solve re-allocates.
extern "C" int qexhip_release_workspace(qexhip_handle c) {
  if (!c) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (auto it = c->fields.begin(); it != c->fields.end();) {
    if (it->first <= -1000) {
      HIPCHK(hipFree(it->second.d));
      it = c->fields.erase(it);
    } else {
      ++it;
    }
  }
  gauge_release_scratch(c);
  return 0;
}

// norm2 / redot / Staggered.D on resident fields (the host-pointer forms above upload their arguments first)
extern "C" int qexhip_dev_norm2(qexhip_handle c, int x_id, int parity, double *out) {
  if (!c || !out || parity < 0 || parity > 2) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fx;
  CHK(find_field(c, x_id, &fx));
  CHK(blas_norm2(c, *fx, parity, &c->dscal[8]));
  return read_scalars(c, &c->dscal[8], 1, out);
}
extern "C" int qexhip_dev_redot(qexhip_handle c, int x_id, int y_id, int parity, double *out) {
  if (!c || !out || parity < 0 || parity > 2) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fx, *fy;
  CHK(find_field(c, x_id, &fx));
  CHK(find_field(c, y_id, &fy));
  CHK(blas_redot(c, *fx, *fy, parity, &c->dscal[8]));
  return read_scalars(c, &c->dscal[8], 1, out);
}
extern "C" int qexhip_dev_dot(qexhip_handle c, int x_id, int y_id, int parity, double out[2]) {
  if (!c || !out || parity < 0 || parity > 2) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fx, *fy;
  CHK(find_field(c, x_id, &fx));
  CHK(find_field(c, y_id, &fy));
  CHK(blas_cdot(c, *fx, *fy, parity, &c->dscal[8]));
  return read_scalars(c, &c->dscal[8], 2, out);
}
extern "C" int qexhip_dev_D(qexhip_handle c, int r_id, int x_id, double m, double sc) {
  if (!c || r_id == x_id) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *fr, *fx;
  CHK(find_field(c, r_id, &fr));
  CHK(find_field(c, x_id, &fx));
  return op_D(c, *fr, *fx, m, sc);
}

// ---- link smearing ----
extern "C" int qexhip_fat7(qexhip_handle c, const double *g, const double coef[5], double *fl, double *ll, double naik) {
  if (!c || !g || !coef || !fl) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  return smear_fat7_host(c, g, coef, fl, ll, naik);
}
extern "C" int qexhip_hisq_smear(qexhip_handle c, const double *g, double *fl, double *ll) {
  if (!c || !g || !fl || !ll) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  return smear_hisq_host(c, g, fl, ll);
}
extern "C" int qexhip_nhyp_smear(qexhip_handle c, const double *g, double *fl, double a1, double a2, double a3) {
  if (!c || !g || !fl) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  return smear_nhyp_host(c, g, fl, a1, a2, a3);
}

extern "C" int qexhip_set_option(qexhip_handle c, const char *name, int value) {
  if (!c || !name) return QEXHIP_ERR_ARG;
  const std::string n(name);
  if (n == "recon") c->opt_recon = value;            // takes effect at the next set_links
  else if (n == "overlap") c->opt_overlap = value;
  else if (n == "transport") {
    if (comm_ready(c)) { qexhip_set_error("option transport must be set before qexhip_comm_init"); return QEXHIP_ERR_STATE; }
    if (value < 0 || value > 3) { qexhip_set_error("option transport: 0 auto, 1 rccl, 2 peer, 3 rccl + mailbox sums"); return QEXHIP_ERR_ARG; }
    c->opt_transport = value;
  }
  else if (n == "batch_multi") c->opt_batch_multi = value;
  else if (n == "multi_reduce") c->opt_multi_reduce = value;
  else if (n == "flow_exp") c->opt_flow_exp = value;
  else if (n == "smear_ca") c->opt_smear_ca = value;
  else if (n == "chain_overlap") c->opt_chain_overlap = value;
  else if (n == "hop_split") {
    if (value != -1 && value != 0 && value != 2) { qexhip_set_error("option hop_split: -1 measured, 0 split by sites, 2 fused (the two-launch form 1 left the library in round 6)"); return QEXHIP_ERR_ARG; }
    c->opt_hop_split = value;
  }
  else if (n == "fused_spin_us") c->opt_fused_spin_us = value;
  else if (n == "emu_exchange_us") c->emu_exchange_us = value;
  else if (n == "emu_allreduce_us") c->emu_allreduce_us = value;
  else if (n == "emu_link_gbs") c->emu_link_gbs = value;
  else if (n == "obs_clover") c->opt_obs_clover = value;
  else if (n == "force_pair") c->opt_force_pair = value;
  else { qexhip_set_error("unknown option"); return QEXHIP_ERR_ARG; }
  return 0;
}
extern "C" int qexhip_stag_solve_xx_batch(qexhip_handle c, int n, double *const *x, const double *const *b, const double *mass,
                                          const double *r2req, int maxits, int par_even, int *iters, double *r2_over_b2) {
  if (!c || !x || !b || !mass || !r2req) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  return solve_batch_host(c, n, x, b, mass, r2req, maxits, par_even ? 1 : 0, iters, r2_over_b2);
}
extern "C" int qexhip_stag_solve_batch(qexhip_handle c, int n, double *const *x, const double *const *b, const double *mass,
                                       const double *r2req, int maxits, int *iters, double *r2_over_b2) {
  if (!c || !x || !b || !mass || !r2req) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  return solve_batch_host(c, n, x, b, mass, r2req, maxits, -1, iters, r2_over_b2);
}
extern "C" int qexhip_stag_links_info(qexhip_handle c, int *nlinks, int *compressed, double *max_dev) {
  if (!c) return QEXHIP_ERR_ARG;
  if (nlinks) *nlinks = c->W ? c->ndir : 0;
  if (compressed) *compressed = c->recon;
  if (max_dev) *max_dev = c->recon_dev;
  return 0;
}
extern "C" int qexhip_hisq_prepare(qexhip_handle c, const double *g, double *fl, double *ll) {
  if (!c || !g) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  return hisq_prepare(c, g, fl, ll);
}
extern "C" int qexhip_hisq_closure_force(qexhip_handle c, const double *dsdsu, const double *dsdsul, double *f) {
  if (!c || !dsdsu || !dsdsul || !f) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  return hisq_closure_force(c, dsdsu, dsdsul, f);
}
extern "C" int qexhip_hisq_fermion_force(qexhip_handle c, double *f, const double *const *psi, const double *scale, int n) {
  if (!c || !f || !psi || !scale) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  return hisq_fermion_force(c, f, psi, scale, n);
}
extern "C" int qexhip_hisq_release(qexhip_handle c) {
  if (!c) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  hisq_state_free(c);
  return 0;
}
extern "C" int qexhip_hisq_force(qexhip_handle c, const double *g, const double *dsdsu, const double *dsdsul, double *f) {
  if (!c || !g || !dsdsu || !dsdsul || !f) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  return smear_hisq_force_host(c, g, dsdsu, dsdsul, f);
}
extern "C" int qexhip_fat7_deriv(qexhip_handle c, const double *g, const double *dfl, const double coef[5], const double *dll,
                                 double naik, double *d) {
  if (!c || !g || !dfl || !coef || !d) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  return smear_fat7_deriv_host(c, g, dfl, coef, dll, naik, d);
}
extern "C" int qexhip_stag_set_links_hisq(qexhip_handle c, const double *g) {
  if (!c) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  return smear_set_links_hisq(c, g);
}
extern "C" int qexhip_stag_set_links_nhyp(qexhip_handle c, const double *g, double a1, double a2, double a3,
                                          const int antiperiodic[4], const int phases[4]) {
  if (!c) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  static const int defph[4] = {8, 9, 11, 0};
  int mask = 0;
  for (int mu = 0; mu < 4; mu++) if (antiperiodic ? antiperiodic[mu] : (mu == 3)) mask |= 1 << mu;
  return smear_set_links_nhyp(c, g, a1, a2, a3, mask, phases ? phases : defph);
}

extern "C" int qexhip_nhyp_prepare(qexhip_handle c, const double *g, double a1, double a2, double a3, double *fl) {
  if (!c) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  return nhyp_prepare(c, g, a1, a2, a3, fl);
}
extern "C" int qexhip_nhyp_force(qexhip_handle c, double *f, const double *chain) {
  if (!c || !f || !chain) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  return nhyp_force_host(c, f, chain);
}
extern "C" int qexhip_nhyp_gauge_force(qexhip_handle c, double *f, double cplaq, double crect, double cadj) {
  if (!c) return QEXHIP_ERR_ARG;
  if (crect != 0.0 && cadj != 0.0) { qexhip_set_error("rect and adjplaq together are not a QEX action"); return QEXHIP_ERR_ARG; }
  HIPCHK(hipSetDevice(c->device));
  return nhyp_gauge_force(c, f, cplaq, cadj != 0.0 ? cadj : crect, cadj != 0.0 ? 1 : 0);
}
extern "C" int qexhip_nhyp_fermion_force(qexhip_handle c, double *f, const double *const *psi, const double *scale, int n,
                                         const int antiperiodic[4], const int phases[4]) {
  if (!c || !psi || !scale) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  static const int defph[4] = {8, 9, 11, 0};
  int mask = 0;
  for (int mu = 0; mu < 4; mu++) if (antiperiodic ? antiperiodic[mu] : (mu == 3)) mask |= 1 << mu;
  return nhyp_fermion_force(c, f, psi, scale, n, mask, phases ? phases : defph);
}
extern "C" int qexhip_nhyp_fforce(qexhip_handle c, double *f, int n, const double *const *phi, const double *mass,
                                  const double *scale, const double *r2req, int maxits, const int antiperiodic[4],
                                  const int phases[4], int *iters) {
  if (!c || !phi || !mass || !scale || !r2req) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  static const int defph[4] = {8, 9, 11, 0};
  int mask = 0;
  for (int mu = 0; mu < 4; mu++) if (antiperiodic ? antiperiodic[mu] : (mu == 3)) mask |= 1 << mu;
  return nhyp_fforce(c, f, n, phi, mass, scale, r2req, maxits, mask, phases ? phases : defph, iters);
}
extern "C" int qexhip_nhyp_fforce_dev(qexhip_handle c, double *f, int n, const int *phi_ids, const double *mass,
                                      const double *scale, const double *r2req, int maxits, const int antiperiodic[4],
                                      const int phases[4], int *iters) {
  if (!c || !phi_ids || !mass || !scale || !r2req || n < 1 || n > 64) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  static const int defph[4] = {8, 9, 11, 0};
  int mask = 0;
  for (int mu = 0; mu < 4; mu++) if (antiperiodic ? antiperiodic[mu] : (mu == 3)) mask |= 1 << mu;
  std::vector<DevField *> pf(n);
  for (int k = 0; k < n; k++) CHK(find_field(c, phi_ids[k], &pf[k]));
  return nhyp_fforce(c, f, n, nullptr, mass, scale, r2req, maxits, mask, phases ? phases : defph, iters, pf.data());
}
// ---- the two ends of a trajectory on resident fields (round 3) ----
struct qexhip_rng;
extern "C" int qexhip_rng_dev_gaussian_vector(qexhip_handle c, qexhip_rng *rng, int field_id) {
  if (!c || !rng) return QEXHIP_ERR_ARG;
  DevField *f;
  CHK(find_field(c, field_id, &f));
  return rng_dev_generate(c, rng, 0, f, nullptr);
}
extern "C" int qexhip_rng_dev_u1_vector(qexhip_handle c, qexhip_rng *rng, int field_id) {
  if (!c || !rng) return QEXHIP_ERR_ARG;
  DevField *f;
  CHK(find_field(c, field_id, &f));
  return rng_dev_generate(c, rng, 1, f, nullptr);
}
extern "C" int qexhip_md_refresh_momenta(qexhip_handle c, qexhip_rng *rng) {
  if (!c || !rng) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  double2 *M = nullptr;
  CHK(md_momenta_dev(c, &M));
  return rng_dev_generate(c, rng, 2, nullptr, M);
}
extern "C" int qexhip_dev_zero(qexhip_handle c, int id, int parity) {
  if (!c || parity < 0 || parity > 2) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevField *f;
  CHK(find_field(c, id, &f));
  return blas_zero(c, *f, parity);
}
extern "C" int qexhip_dev_solve_batch(qexhip_handle c, int n, const int *x_ids, const int *b_ids, const double *mass,
                                      const double *r2req, int maxits, int *iters, double *r2) {
  if (!c || !x_ids || !b_ids || !mass || !r2req || n < 1 || n > 64) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  // every solution is zeroed while its system is set up, in order: a solution field that is also a source (of ANY system of
  // the call) or another system's solution would be destroyed silently -- refuse such aliases for the whole call
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++)
      if (x_ids[i] == b_ids[j] || (i != j && x_ids[i] == x_ids[j])) {
        qexhip_set_error("dev_solve_batch: solution field %d (system %d) aliases %s of system %d", x_ids[i], i,
                         x_ids[i] == b_ids[j] ? "the source" : "the solution", j);
        return QEXHIP_ERR_ARG;
      }
  for (int k0 = 0; k0 < n; k0 += 4) {          // lock-step batches of four, as qexhip_stag_solve_batch
    const int k = std::min(4, n - k0);
    DevField *xs[4], *bs[4];
    for (int j = 0; j < k; j++) {
      CHK(find_field(c, x_ids[k0 + j], &xs[j]));
      CHK(find_field(c, b_ids[k0 + j], &bs[j]));
    }
    CHK(solve_full_batch_dev(c, k, xs, bs, mass + k0, r2req + k0, maxits, iters ? iters + k0 : nullptr, r2 ? r2 + k0 : nullptr));
  }
  return 0;
}
extern "C" int qexhip_nhyp_release(qexhip_handle c) {
  if (!c) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  nhyp_state_free(c);
  return 0;
}

// ---- gauge / flow ----
extern "C" int qexhip_gauge_set(qexhip_handle c, const double *g) { if (!c || !g) return QEXHIP_ERR_ARG; return gauge_set(c, g); }
extern "C" int qexhip_gauge_get(qexhip_handle c, double *g) { if (!c || !g) return QEXHIP_ERR_ARG; return gauge_get(c, g); }
extern "C" int qexhip_plaq(qexhip_handle c, double out[6]) { if (!c || !out) return QEXHIP_ERR_ARG; return gauge_plaq(c, out); }
extern "C" int qexhip_gauge_force(qexhip_handle c, double *f, double cplaq) { if (!c || !f) return QEXHIP_ERR_ARG; return gauge_force(c, f, cplaq); }
extern "C" int qexhip_gauge_force_general(qexhip_handle c, double *f, double cplaq, double c2, int kind) {
  if (!c || !f || kind < 0 || kind > 1) return QEXHIP_ERR_ARG;
  return gauge_force(c, f, cplaq, c2, kind);
}
extern "C" int qexhip_wflow_general(qexhip_handle c, int nsteps, double eps, double cplaq, double c2, int kind) {
  if (!c || nsteps < 0 || kind < 0 || kind > 1) return QEXHIP_ERR_ARG;
  return gauge_wflow(c, nsteps, eps, cplaq, c2, kind);
}
extern "C" int qexhip_flow_EQ(qexhip_handle c, int loop, double out[3]) { if (!c || !out) return QEXHIP_ERR_ARG; return gauge_flow_obs(c, loop, out); }
extern "C" int qexhip_flow_measure(qexhip_handle c, double plaq[6], double EQ[3]) {
  if (!c || !plaq || !EQ) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  // the dedicated kernel when it is available (one pass over the links for all nine numbers), two passes otherwise
  const int rc = gauge_flow_obs(c, 1, EQ, plaq);
  if (rc != -3) return rc;
  CHK(gauge_plaq(c, plaq));
  return gauge_flow_obs(c, 1, EQ);
}
extern "C" int qexhip_gauge_action(qexhip_handle c, double cplaq, double crect, double cadj, double *out) {
  if (!c || !out) return QEXHIP_ERR_ARG;
  if (crect != 0.0 && cadj != 0.0) { qexhip_set_error("rect and adjplaq together are not a QEX action"); return QEXHIP_ERR_ARG; }
  HIPCHK(hipSetDevice(c->device));
  return gauge_action(c, cplaq, cadj != 0.0 ? cadj : crect, cadj != 0.0 ? 1 : 0, out);
}
extern "C" int qexhip_gauge_update(qexhip_handle c, const double *p, double t) {
  if (!c || !p) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  return gauge_md_update(c, p, t);
}
extern "C" int qexhip_gauge_reunit(qexhip_handle c) {
  if (!c) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  return gauge_reunit(c);
}
extern "C" int qexhip_wline(qexhip_handle c, const int *path, int n, double out[2]) {
  if (!c || !path || !out) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  return gauge_wline(c, path, n, out);
}
extern "C" int qexhip_plaq_s4(qexhip_handle c, double out[8]) {
  if (!c || !out) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  return gauge_plaq_s4(c, out);
}
extern "C" int qexhip_polyakov_loops(qexhip_handle c, double out[8]) {
  if (!c || !out) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  return gauge_polyakov(c, out);
}
extern "C" int qexhip_wflow(qexhip_handle c, int nsteps, double eps) { if (!c || nsteps < 0) return QEXHIP_ERR_ARG; return gauge_wflow(c, nsteps, eps); }

// ---- resident molecular dynamics (gauge.hip) ----
extern "C" int qexhip_md_begin(qexhip_handle c, const double *g, const double *p) {
  if (!c) return QEXHIP_ERR_ARG;
  HIPCHK(hipSetDevice(c->device));
  return md_begin(c, g, p);
}
extern "C" int qexhip_md_end(qexhip_handle c, double *g, double *p) { if (!c) return QEXHIP_ERR_ARG; HIPCHK(hipSetDevice(c->device)); return md_end(c, g, p); }
extern "C" int qexhip_md_momentum_norm2(qexhip_handle c, double *p2) { if (!c || !p2) return QEXHIP_ERR_ARG; HIPCHK(hipSetDevice(c->device)); return md_momentum_norm2(c, p2); }
extern "C" int qexhip_md_update_links(qexhip_handle c, double t) { if (!c) return QEXHIP_ERR_ARG; HIPCHK(hipSetDevice(c->device)); return md_update_links(c, t); }
extern "C" int qexhip_md_gauge_force(qexhip_handle c, double cplaq, double crect, double cadj) {
  if (!c) return QEXHIP_ERR_ARG;
  if (crect != 0.0 && cadj != 0.0) { qexhip_set_error("rect and adjplaq together are not a QEX action"); return QEXHIP_ERR_ARG; }
  HIPCHK(hipSetDevice(c->device));
  return md_gauge_force(c, cplaq, cadj != 0.0 ? cadj : crect, cadj != 0.0 ? 1 : 0);
}
extern "C" int qexhip_md_kick(qexhip_handle c, int source, double t) { if (!c) return QEXHIP_ERR_ARG; HIPCHK(hipSetDevice(c->device)); return md_kick(c, source, t); }
extern "C" int qexhip_md_shift_links(qexhip_handle c, int source, double t) { if (!c) return QEXHIP_ERR_ARG; HIPCHK(hipSetDevice(c->device)); return md_shift_links(c, source, t); }
extern "C" int qexhip_md_save_links(qexhip_handle c) { if (!c) return QEXHIP_ERR_ARG; HIPCHK(hipSetDevice(c->device)); return md_save_links(c); }
extern "C" int qexhip_md_restore_links(qexhip_handle c) { if (!c) return QEXHIP_ERR_ARG; HIPCHK(hipSetDevice(c->device)); return md_restore_links(c); }
