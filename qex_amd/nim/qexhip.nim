## qexhip.nim -- Nim binding of libqexhip.so for QEX (ctpeterson/qex).
##
## Drop this file next to src/quda/ (e.g. src/hip/qexhip.nim), build QEX with
##   -d:qexhipDir=/path/to/repo
## and wire the procs below into the call sites named beside each of them (INTEGRATION.md has the
## one-line patches).  It mirrors src/quda/qudaWrapperImpl.nim:165-261 (qudaSolveXX): build a V=1 twin
## layout once, turn the SIMD fields into site-major host arrays (one qexhip_layout_* call per field: the library
## restates QEX's index map in C), make the C call(s), turn the result back.
##
##   call site in QEX                                                    proc here
##   src/physics/stagSolve.nim:65-128  `case sp.backend` (solveEE/OO)    hipSolveEE / hipSolveOO
##   src/physics/stagSolve.nim:224-294 Staggered.solve(x, b, m, sp)      hipSolve
##   src/physics/stagSolve.nim:347-446 Staggered.solve(xs, b, ms, sp)    hipSolve (seq form; the `sbQex`-only slot :340-341)
##   src/physics/stagSolve.nim:296-345 Staggered.solveXX(xs, b, ms, ..)  hipSolveXX (seq form)
##   src/physics/stagD.nim:566-571     s.D / s.Ddag                      hipD / hipDdag
##     (stagg_pv_hmc/staghmc_spv.nim:418,442,468,543,546,660)
##   src/physics/stagD.nim:349-395     stagD2(s.so, r, s.g, x, a, b)     hipStagD2
##   src/gauge/wflow.nim:21-67, src/flow/flow.nim:22-90  gaugeFlow       hipGaugeFlow (both forms)
##   src/gauge/gaugeUtils.nim:213-282  g.plaq                            hipPlaq
##   src/flow/gauge_flow.nim:360-379   EQ                                hipFlowMeasure
##   src/flow/gauge_flow.nim:137-156   meas_ploop (4 x g.wline)          hipPloops
##   stagg_pv_hmc/staghmc_spv_meas.nim:25-65  g.s4_gauge               hipS4Gauge
##   src/gauge/hypsmear.nim:49-247     coef.smearGetForce(g, sg, info)   hipSmearGetForce (returns the closure)
##     (stagg_pv_hmc/staghmc_spv.nim:989-993)
##   stagg_pv_hmc/staghmc_spv.nim:217-228  gforce(act, g, sg, f, sf)     closure.gforce
##   stagg_pv_hmc/staghmc_spv.nim:716-865  fforce + smeared_one_link_force   closure.fforce
##
## NOTE: written against the reference sources but NOT compiled in this repository's pipeline (there is no
## Nim compiler in the build image).  What IS executed here is the call sequence of every proc below, argument
## for argument, from C++ on V=1 arrays: tests/cpp/test_shim_sequence.cpp (checked against the CPU oracle).

import os, strutils
import base, layout, field
import comms/qmp                      # QMP_broadcast (comms/qmp.nim:46)
import physics/qcdTypes
import physics/stagD
import solvers/solverBase
import gauge/hypsmear                 # HypCoefs

const qexhipDir {.strDefine.} = getHomeDir() & "qexhip"
{.passC: "-I" & qexhipDir & "/include".}
{.passL: "-L" & qexhipDir & "/qex_amd -lqexhip -Wl,-rpath," & qexhipDir & "/qex_amd".}
{.pragma: qh, importc, header: "qexhip.h".}

type QexhipHandle* = pointer

proc qexhip_init(h: ptr QexhipHandle; device: cint; latLocal, rankGeom, rankCoord: ptr cint): cint {.qh.}
proc qexhip_finalize(h: QexhipHandle): cint {.qh.}
proc qexhip_device_count(n: ptr cint): cint {.qh.}
proc qexhip_comm_info(h: QexhipHandle; nranks, rank, device: ptr cint; busid: cstring; buslen: cint): cint {.qh.}
proc qexhip_last_error(): cstring {.qh.}
proc qexhip_comm_unique_id(id: ptr char): cint {.qh.}
proc qexhip_comm_init(h: QexhipHandle; id: ptr char; nranks, rank: cint): cint {.qh.}
proc qexhip_comm_transport(h: QexhipHandle; name: cstring; len: cint; stats: ptr clong): cint {.qh.}
proc qexhip_layout_vec_simd_to_v1(localGeom, innerGeom: ptr cint; simd, v1: ptr cdouble): cint {.qh.}
proc qexhip_layout_vec_v1_to_simd(localGeom, innerGeom: ptr cint; v1, simd: ptr cdouble): cint {.qh.}
proc qexhip_layout_gauge_simd_to_v1(localGeom, innerGeom: ptr cint; g: ptr ptr cdouble; v1: ptr cdouble): cint {.qh.}
proc qexhip_layout_gauge_v1_to_simd(localGeom, innerGeom: ptr cint; v1: ptr cdouble; g: ptr ptr cdouble): cint {.qh.}
proc qexhip_stag_set_links(h: QexhipHandle; fat, lng: ptr cdouble): cint {.qh.}
proc qexhip_stag_dslash(h: QexhipHandle; r, x: ptr cdouble; parity: cint; a, b: cdouble): cint {.qh.}
proc qexhip_stag_D(h: QexhipHandle; r, x: ptr cdouble; m, sc: cdouble): cint {.qh.}
proc qexhip_stag_stagD(h: QexhipHandle; r, x: ptr cdouble; parity: cint; m, sc, a: cdouble): cint {.qh.}
proc qexhip_stag_eo_reduce(h: QexhipHandle; r, b: ptr cdouble; m: cdouble): cint {.qh.}
proc qexhip_stag_eo_reconstruct(h: QexhipHandle; r, b: ptr cdouble; m: cdouble): cint {.qh.}
proc qexhip_stag_solve_xx(h: QexhipHandle; x, b: ptr cdouble; mass, r2req: cdouble;
                          maxits, parEven: cint; iters: ptr cint; r2: ptr cdouble;
                          hist: ptr cdouble; histcap: cint): cint {.qh.}
proc qexhip_stag_solve(h: QexhipHandle; x, b: ptr cdouble; mass, r2req: cdouble; maxits: cint;
                       iters: ptr cint; r2: ptr cdouble): cint {.qh.}
proc qexhip_stag_solve_xx_multi(h: QexhipHandle; xs: ptr ptr cdouble; b: ptr cdouble; shifts: ptr cdouble; nmass: cint;
                                r2req: cdouble; maxits, parEven: cint; iters: ptr cint;
                                hist: ptr cdouble; histcap: cint): cint {.qh.}
proc qexhip_stag_solve_multi(h: QexhipHandle; xs: ptr ptr cdouble; b: ptr cdouble; masses: ptr cdouble; nmass: cint;
                             r2req: cdouble; maxits: cint; iters: ptr cint; r2: ptr cdouble): cint {.qh.}
proc qexhip_stag_solve_batch(h: QexhipHandle; n: cint; x, b: ptr ptr cdouble; mass, r2req: ptr cdouble;
                             maxits: cint; iters: ptr cint; r2: ptr cdouble): cint {.qh.}
proc qexhip_gauge_set(h: QexhipHandle; g: ptr cdouble): cint {.qh.}
proc qexhip_gauge_get(h: QexhipHandle; g: ptr cdouble): cint {.qh.}
proc qexhip_plaq(h: QexhipHandle; o: ptr cdouble): cint {.qh.}
proc qexhip_wflow(h: QexhipHandle; nsteps: cint; eps: cdouble): cint {.qh.}
proc qexhip_wflow_general(h: QexhipHandle; nsteps: cint; eps, cplaq, c2: cdouble; kind: cint): cint {.qh.}
proc qexhip_flow_EQ(h: QexhipHandle; loop: cint; o: ptr cdouble): cint {.qh.}
proc qexhip_flow_measure(h: QexhipHandle; plaq, eq: ptr cdouble): cint {.qh.}
# the HMC-side entry points (INTEGRATION.md 5b): smearing closure, MD forces, gauge-sector pieces
proc qexhip_stag_set_links_nhyp(h: QexhipHandle; g: ptr cdouble; a1, a2, a3: cdouble;
                                antiperiodic, phases: ptr cint): cint {.qh.}
proc qexhip_stag_set_links_hisq(h: QexhipHandle; g: ptr cdouble): cint {.qh.}
proc qexhip_nhyp_prepare(h: QexhipHandle; g: ptr cdouble; a1, a2, a3: cdouble; fl: ptr cdouble): cint {.qh.}
proc qexhip_nhyp_force(h: QexhipHandle; f, chain: ptr cdouble): cint {.qh.}
proc qexhip_nhyp_gauge_force(h: QexhipHandle; f: ptr cdouble; cplaq, crect, cadjplaq: cdouble): cint {.qh.}
proc qexhip_nhyp_fermion_force(h: QexhipHandle; f: ptr cdouble; psi: ptr ptr cdouble; scale: ptr cdouble;
                               n: cint; antiperiodic, phases: ptr cint): cint {.qh.}
proc qexhip_nhyp_release(h: QexhipHandle): cint {.qh.}
proc qexhip_gauge_action(h: QexhipHandle; cplaq, crect, cadjplaq: cdouble; o: ptr cdouble): cint {.qh.}
proc qexhip_gauge_update(h: QexhipHandle; p: ptr cdouble; t: cdouble): cint {.qh.}
proc qexhip_gauge_reunit(h: QexhipHandle): cint {.qh.}
proc qexhip_wline(h: QexhipHandle; path: ptr cint; n: cint; o: ptr cdouble): cint {.qh.}
proc qexhip_polyakov_loops(h: QexhipHandle; o: ptr cdouble): cint {.qh.}
proc qexhip_plaq_s4(h: QexhipHandle; o: ptr cdouble): cint {.qh.}
proc qexhip_io_read_gauge(path: cstring; lat: ptr cint; g: ptr cdouble; suma, sumb: ptr cuint): cint {.qh.}
proc qexhip_rng_get_state(r: pointer; o: ptr cuint): cint {.qh.}
proc qexhip_rng_set_state(r: pointer; i: ptr cuint): cint {.qh.}
proc qexhip_io_write_field(path: cstring; lat: ptr cint; data: pointer; siteBytes, wordBytes: cint; datatype: cstring;
                           prec: cchar; colors, datacount: cint; fileMd, recMd: cstring): cint {.qh.}
proc qexhip_io_read_field(path: cstring; lat: ptr cint; data: pointer; siteBytes, wordBytes: cint; datatype: cstring): cint {.qh.}
proc qexhip_md_begin(h: pointer; g, p: ptr cdouble): cint {.qh.}
proc qexhip_md_end(h: pointer; g, p: ptr cdouble): cint {.qh.}
proc qexhip_md_momentum_norm2(h: pointer; p2: ptr cdouble): cint {.qh.}
proc qexhip_md_update_links(h: pointer; t: cdouble): cint {.qh.}
proc qexhip_md_gauge_force(h: pointer; cplaq, crect, cadj: cdouble): cint {.qh.}
proc qexhip_md_kick(h: pointer; source: cint; t: cdouble): cint {.qh.}
proc qexhip_md_shift_links(h: pointer; source: cint; t: cdouble): cint {.qh.}
proc qexhip_md_save_links(h: pointer): cint {.qh.}
proc qexhip_md_restore_links(h: pointer): cint {.qh.}
proc qexhip_io_metadata(path: cstring; fileMd: cstring; fileCap: cint; recMd: cstring; recCap: cint; fileLen, recLen: ptr cint): cint {.qh.}
proc qexhip_io_write_gauge(path: cstring; lat: ptr cint; g: ptr cdouble; prec: cchar; fileMd, recMd: cstring): cint {.qh.}

template chk(e: untyped) =
  ## libqexhip reports errors by return code; QEX aborts (base/qexInternal.nim:37-45)
  let rc = e
  if rc != 0: qexError "libqexhip: " & $qexhip_last_error()

type HipParams = object
  h: QexhipHandle
  layout1: Layout[1]          # V=1 twin, as qudaParam.layout (qudaWrapperImpl.nim:118)
  initialized: bool
var hipParam: HipParams

proc hipHandle*(): QexhipHandle = hipParam.h

proc hipSetup*(l: Layout): Layout[1] =
  ## once per lattice geometry (cf. qudaSetup, qudaWrapperImpl.nim:88-123)
  if not hipParam.initialized:
    doAssert l.rankGeom[0] == 1 and l.rankGeom[1] == 1 and l.rankGeom[2] == 1,
      "libqexhip shards along t only: run with -rankgeom:1,1,1,N"
    var lat, rg, rc: array[4, cint]
    var rcs = newSeq[cint](4)
    l.rankCoordsFromRank(rcs, l.myRank)     # layout/layoutX.nim
    for i in 0..3:
      lat[i] = l.localGeom[i].cint
      rg[i] = l.rankGeom[i].cint
      rc[i] = rcs[i]
    # one GPU per rank, as qudaInit assumes: the rank's index on its node (launcher-provided) modulo the visible devices
    var ndev: cint
    chk qexhip_device_count(ndev.addr)
    var localRank = l.myRank
    for v in ["OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "SLURM_LOCALID", "LOCAL_RANK"]:
      if existsEnv(v):
        localRank = parseInt(getEnv(v))
        break
    let dev = (localRank mod ndev.int).cint
    chk qexhip_init(hipParam.h.addr, dev, lat[0].addr, rg[0].addr, rc[0].addr)
    if l.nRanks > 1:
      var id: array[128, char]
      if l.myRank == 0: chk qexhip_comm_unique_id(id[0].addr)
      QMP_broadcast(id[0].addr, 128.csize_t)   # comms/qmp.nim:46
      chk qexhip_comm_init(hipParam.h, id[0].addr, l.nRanks.cint, l.myRank.cint)
      var nr, rk, dv: cint
      var bus = newString(64)
      chk qexhip_comm_info(hipParam.h, nr.addr, rk.addr, dv.addr, bus.cstring, 64)
      var tr = newString(16)
      chk qexhip_comm_transport(hipParam.h, tr.cstring, 16, nil)      # "rccl" between distinct GPUs, "peer" when ranks share one
      echoAll "libqexhip: rank ", l.myRank, " = rank ", rk, " of ", nr, " on device ", dv, " (", $bus.cstring, "), transport ", $tr.cstring
    hipParam.layout1 = l.physGeom.newLayout 1
    hipParam.initialized = true
  hipParam.layout1

# ---------------------------------------------------------------------------------------------------------------
# host buffers in the library's format: V=1 MILC even-odd site order (Layout[1].rankIndex), site-major,
#   colour vector  [site][3][re,im]            gauge field  [site][mu][row][col][re,im]
# The site map is the one qudaSolveXX uses (qudaWrapperImpl.nim:198-240): l.coord -> lo1.rankIndex.
# ---------------------------------------------------------------------------------------------------------------
type HostBuf* = seq[cdouble]
proc p(b: var HostBuf): ptr cdouble = cast[ptr cdouble](b[0].addr)

# The per-site copy loops of qudaWrapperImpl.nim:198-260 (l.coord -> lo1.rankIndex, element by element) are ONE call into
# the library each: qexhip_layout_* restate layoutIndexQ / layoutCoordQ (qlayout.nim:110-185) in C, take the field's own
# memory ([outer][...][re|im][V lanes], fieldET.nim:18-22) and the layout's local / inner geometry, and are tested on the CPU
# against an independent V = 8 map and the closed form (tests/test_simd_layout.py).  Double-precision fields only.
type Geom4 = array[4, cint]
proc geoms(l: Layout): tuple[lg, ig: Geom4] =
  for i in 0..3:
    result.lg[i] = l.localGeom[i].cint
    result.ig[i] = l.innerGeom[i].cint
template raw(f: Field): ptr cdouble = cast[ptr cdouble](f.dataPtr)     # qudaWrapperImpl.nim:190-191 uses the same accessor

proc toHost*(v: Field; b: var HostBuf) =
  ## colour vector field -> host buffer (replaces the copy loop of qudaWrapperImpl.nim:198-207)
  let lo1 = v.l.hipSetup
  b.setLen(lo1.nSites * 6)
  var (lg, ig) = v.l.geoms
  chk qexhip_layout_vec_simd_to_v1(lg[0].addr, ig[0].addr, v.raw, b.p)

proc fromHost*(v: Field; b: HostBuf) =
  ## host buffer -> colour vector field (qudaWrapperImpl.nim:252-260)
  discard v.l.hipSetup
  var (lg, ig) = v.l.geoms
  chk qexhip_layout_vec_v1_to_simd(lg[0].addr, ig[0].addr, cast[ptr cdouble](b[0].unsafeAddr), v.raw)

proc toHostG*(g: openArray[Field]; b: var HostBuf; first = 0; stride = 1) =
  ## gauge field g[first], g[first+stride], .. (4 directions) -> host buffer (qudaWrapperImpl.nim:216-240;
  ## first = 0 / 1, stride = 2 pick the fat / long links of a Naik operator's interleaved s.g)
  let l = g[first].l
  let lo1 = l.hipSetup
  b.setLen(lo1.nSites * 72)
  var (lg, ig) = l.geoms
  var ptrs: array[4, ptr cdouble]
  for mu in 0..3: ptrs[mu] = g[first + stride*mu].raw
  chk qexhip_layout_gauge_simd_to_v1(lg[0].addr, ig[0].addr, ptrs[0].addr, b.p)

proc fromHostG*(g: openArray[Field]; b: HostBuf) =
  let l = g[0].l
  discard l.hipSetup
  var (lg, ig) = l.geoms
  var ptrs: array[4, ptr cdouble]
  for mu in 0..3: ptrs[mu] = g[mu].raw
  chk qexhip_layout_gauge_v1_to_simd(lg[0].addr, ig[0].addr, cast[ptr cdouble](b[0].unsafeAddr), ptrs[0].addr)

# ---------------------------------------------------------------------------------------------------------------
# the operator: Staggered.g -> device (once per link update; QUDA re-uploads per solve, qudaWrapperImpl.nim:216-240)
# ---------------------------------------------------------------------------------------------------------------
proc hipSetLinks*(s: Staggered) =
  var g1, g3: HostBuf
  let naik = s.g.len == 8
  if naik:
    toHostG(s.g, g1, 0, 2)             # s.g[2 mu]     fat   (stagD.nim:552-564)
    toHostG(s.g, g3, 1, 2)             # s.g[2 mu + 1] long
  else:
    toHostG(s.g, g1)
  chk qexhip_stag_set_links(hipParam.h, g1.p, if naik: g3.p else: nil)

proc subsetCode(name: string): cint =
  case name
  of "even": 0
  of "odd": 1
  else: 2

proc hipStagD2*(s: Staggered; r, x: Field; a, b: SomeNumber; subset = "all") =
  ## stagD2(s.se|so|sa, r, s.g, x, a, b) (stagD.nim:349-395): r[subset] = a r + b x + (2D) x.  Links: hipSetLinks(s) first.
  var rb, xb: HostBuf
  toHost(r, rb)                        # the a-term reads r
  toHost(x, xb)
  chk qexhip_stag_dslash(hipParam.h, rb.p, xb.p, subsetCode(subset), a.cdouble, b.cdouble)
  fromHost(r, rb)

proc hipD*(s: Staggered; r, x: Field; m: SomeNumber) =
  ## s.D(r, x, m) (stagD.nim:566-568): r = m x + D x on both parities
  var rb, xb: HostBuf
  toHost(x, xb)
  rb.setLen(xb.len)
  chk qexhip_stag_D(hipParam.h, rb.p, xb.p, m.cdouble, 1.0)
  fromHost(r, rb)

proc hipDdag*(s: Staggered; r, x: Field; m: SomeNumber) =
  ## s.Ddag(r, x, m) (stagD.nim:569-571): r = m x - D x
  var rb, xb: HostBuf
  toHost(x, xb)
  rb.setLen(xb.len)
  chk qexhip_stag_D(hipParam.h, rb.p, xb.p, m.cdouble, -1.0)
  fromHost(r, rb)

proc hipStagD*(s: Staggered; r, x: Field; m: SomeNumber; sc: SomeNumber = 1.0; a: SomeNumber = 0.0; subset = "all") =
  ## stagD(s.se | s.so, r, s.g, x, m, sc, a) (stagD.nim:406-409): r[subset] = a r + m x + sc D x; the rest of r is kept
  var rb, xb: HostBuf
  toHost(x, xb)
  toHost(r, rb)
  chk qexhip_stag_stagD(hipParam.h, rb.p, xb.p, subsetCode(subset), m.cdouble, sc.cdouble, a.cdouble)
  fromHost(r, rb)

proc hipEoReduce*(s: Staggered; r, b: Field; m: SomeNumber) =
  ## s.eoReduce(r, b, m) (stagD.nim:575-581): r.even = (D^+ b).even; r.odd is kept
  var rb, bb: HostBuf
  toHost(b, bb)
  toHost(r, rb)
  chk qexhip_stag_eo_reduce(hipParam.h, rb.p, bb.p, m.cdouble)
  fromHost(r, rb)

proc hipEoReconstruct*(s: Staggered; r, b: Field; m: SomeNumber) =
  ## s.eoReconstruct(r, b, m) (stagD.nim:582-586): r.odd = (b.odd - D_oe r.even)/m; r.even is kept
  var rb, bb: HostBuf
  toHost(b, bb)
  toHost(r, rb)
  chk qexhip_stag_eo_reconstruct(hipParam.h, rb.p, bb.p, m.cdouble)
  fromHost(r, rb)

# ---------------------------------------------------------------------------------------------------------------
# solvers
# ---------------------------------------------------------------------------------------------------------------
proc hipSolveXX*(s: Staggered; r, t: Field; m: SomeNumber; sp: var SolverParams; parEven = true) =
  ## same contract as qudaSolveXX: r <- solution on the even (odd) subset of
  ## 4(m^2 - D_eo D_oe) r = t, r = 0 start, sp.iterations set.
  var tb, rb: HostBuf
  toHost(t, tb)
  rb.setLen(tb.len)
  var iters: cint
  var r2: cdouble
  chk qexhip_stag_solve_xx(hipParam.h, rb.p, tb.p, m.cdouble, sp.r2req.cdouble, sp.maxits.cint,
                           (if parEven: 1 else: 0).cint, iters.addr, r2.addr, nil, 0)
  sp.iterations = iters.int
  sp.r2.init r2
  fromHost(r, rb)

proc hipSolveEE*(s: Staggered; r, t: Field; m: SomeNumber; sp: var SolverParams) =
  hipSolveXX(s, r, t, m, sp, parEven = true)
proc hipSolveOO*(s: Staggered; r, t: Field; m: SomeNumber; sp: var SolverParams) =
  hipSolveXX(s, r, t, m, sp, parEven = false)

proc hipSolve*(s: Staggered; x, b: Field; m: SomeNumber; sp: var SolverParams) =
  ## Staggered.solve(x, b, m, sp) (stagSolve.nim:224-294): D(m) x = b with the even-odd reconstruction and the outer
  ## true-residual restarts done on the device; x starts from 0 (sp.usePrevSoln: qexhip_stag_solve_prev)
  var xb, bb: HostBuf
  toHost(b, bb)
  xb.setLen(bb.len)
  var iters: cint
  var r2: cdouble
  chk qexhip_stag_solve(hipParam.h, xb.p, bb.p, m.cdouble, sp.r2req.cdouble, sp.maxits.cint, iters.addr, r2.addr)
  sp.iterations = iters.int
  sp.r2.init r2
  fromHost(x, xb)

proc hipSolveXX*(s: Staggered; xs: openArray[Field]; b: Field; ms: openArray[SomeNumber]; sp: var SolverParams; parEven = true) =
  ## Staggered.solveXX(xs, b, ms, sp, subset) (stagSolve.nim:296-345): shifts[0] = m0, shifts[k] = 4 (m_k^2 - m0^2) (:391-394)
  let n = xs.len
  var bb: HostBuf
  toHost(b, bb)
  var xb = newSeq[HostBuf](n)
  var ptrs = newSeq[ptr cdouble](n)
  var shifts = newSeq[cdouble](n)
  for k in 0..<n:
    xb[k].setLen(bb.len)
    ptrs[k] = xb[k].p
    shifts[k] = if k == 0: ms[0].cdouble else: (4.0 * (ms[k]*ms[k] - ms[0]*ms[0])).cdouble
  var iters: cint
  chk qexhip_stag_solve_xx_multi(hipParam.h, ptrs[0].addr, bb.p, shifts[0].addr, n.cint, sp.r2req.cdouble, sp.maxits.cint,
                                 (if parEven: 1 else: 0).cint, iters.addr, nil, 0)
  sp.iterations = iters.int
  for k in 0..<n: fromHost(xs[k], xb[k])

proc hipSolve*(s: Staggered; xs: openArray[Field]; b: Field; ms: openArray[SomeNumber]; sp: var SolverParams) =
  ## Staggered.solve(xs, b, ms, sp) (stagSolve.nim:347-446): D(m_k) xs[k] = b for every mass, one multi-shift CG
  let n = xs.len
  var bb: HostBuf
  toHost(b, bb)
  var xb = newSeq[HostBuf](n)
  var ptrs = newSeq[ptr cdouble](n)
  var masses = newSeq[cdouble](n)
  for k in 0..<n:
    xb[k].setLen(bb.len)
    ptrs[k] = xb[k].p
    masses[k] = ms[k].cdouble
  var iters: cint
  var r2: cdouble
  chk qexhip_stag_solve_multi(hipParam.h, ptrs[0].addr, bb.p, masses[0].addr, n.cint, sp.r2req.cdouble, sp.maxits.cint,
                              iters.addr, r2.addr)
  sp.iterations = iters.int
  sp.r2.init r2
  for k in 0..<n: fromHost(xs[k], xb[k])

# ---------------------------------------------------------------------------------------------------------------
# gauge observables and the Wilson flow (the links stay on the device between the steps of a flow)
# ---------------------------------------------------------------------------------------------------------------
proc hipGaugeSet*(g: openArray[Field]) =
  var gb: HostBuf
  toHostG(g, gb)
  chk qexhip_gauge_set(hipParam.h, gb.p)

proc hipGaugeGet*(g: openArray[Field]) =
  var gb: HostBuf
  gb.setLen(g[0].l.hipSetup.nSites * 72)
  chk qexhip_gauge_get(hipParam.h, gb.p)
  fromHostG(g, gb)

proc hipPlaq*(): seq[float] =
  ## g.plaq (gaugeUtils.nim:213-282) of the resident field: six values, index mu (mu - 1) / 2 + nu
  var o: array[6, cdouble]
  chk qexhip_plaq(hipParam.h, o[0].addr)
  result = newSeq[float](6)
  for i in 0..5: result[i] = o[i]

proc hipFlowMeasure*(): tuple[plaq: seq[float]; es, et, q: float] =
  ## what the measure block of a flow loop prints (flow/gauge_flow.nim:139-156,360-379): plaquettes and the clover
  ## E_s, E_t, Q of the resident field, one pass over the links
  var pl: array[6, cdouble]
  var eq: array[3, cdouble]
  chk qexhip_flow_measure(hipParam.h, pl[0].addr, eq[0].addr)
  result.plaq = newSeq[float](6)
  for i in 0..5: result.plaq[i] = pl[i]
  result.es = eq[0]; result.et = eq[1]; result.q = eq[2]

proc hipS4Gauge*(): seq[array[2, float]] =
  ## s4_gauge (stagg_pv_hmc/staghmc_spv_meas.nim:25-65) of the resident field: peo[dir][even/odd], already normalised
  var o: array[8, cdouble]
  chk qexhip_plaq_s4(hipParam.h, o[0].addr)
  result = newSeq[array[2, float]](4)
  for d in 0..3: result[d] = [o[2*d].float, o[2*d+1].float]

proc hipPloops*(): tuple[pls, plt: tuple[re, im: float]] =
  ## meas_ploop (flow/gauge_flow.nim:137-156): the four Polyakov loops g.wline(repeat(i+1, L_i)) of the resident field in one
  ## call; the spatial ones averaged, the temporal one alone, as there
  var o: array[8, cdouble]
  chk qexhip_polyakov_loops(hipParam.h, o[0].addr)
  result.pls = (re: (o[0] + o[2] + o[4]) / 3.0, im: (o[1] + o[3] + o[5]) / 3.0)
  result.plt = (re: o[6].float, im: o[7].float)

template hipGaugeFlow*(g: array|seq; steps: int; eps: float; measure: untyped) =
  ## g.gaugeFlow(steps, eps): measure (gauge/wflow.nim:21-67).  `wflowT` is injected for `measure`, as there; the flowed
  ## links are on the DEVICE while `measure` runs (use hipPlaq / hipFlowMeasure in it), g itself is updated at the end.
  block:
    hipGaugeSet(g)
    var n = 1
    while true:
      chk qexhip_wflow(hipParam.h, 1, eps.cdouble)
      let wflowT {.inject, used.} = n * eps
      measure
      inc n
      if steps > 0 and n > steps: break
    hipGaugeGet(g)

template hipGaugeFlow*(gc: GaugeActionCoeffs; flowAct: string; g: array|seq; steps: int; eps: float; measure: untyped) =
  ## the fork's gc.gaugeFlow(flow_act, g, steps, eps): measure (src/flow/flow.nim:22-90): "Wilson" | "rect" -> gaugeForce with
  ## (plaq, rect), "adj" -> forceA with (plaq, adjplaq)
  block:
    hipGaugeSet(g)
    let kind = (if flowAct == "adj": 1 else: 0).cint
    let c2 = (if flowAct == "adj": gc.adjplaq else: gc.rect).cdouble
    var n = 1
    while true:
      chk qexhip_wflow_general(hipParam.h, 1, eps.cdouble, gc.plaq.cdouble, c2, kind)
      let wflowT {.inject, used.} = n * eps
      measure
      inc n
      if steps > 0 and n > steps: break
    hipGaugeGet(g)

# ---------------------------------------------------------------------------------------------------------------
# nHYP smearing and the forces that run through its closure (stagg_pv_hmc)
# ---------------------------------------------------------------------------------------------------------------
type HipSmearedForce* = object
  ## what `coef.smearGetForce(g, sg, info)` returns in QEX is a closure `proc(f, chain)`; here the intermediate fields
  ## live on the device until `release` (or the next hipSmearGetForce)
  bc*: array[4, cint]                  ## per-direction antiperiodic flags of the fork's `bc` string (setBC_cust)

proc hipSmearGetForce*(coef: HypCoefs; g, sg: openArray[Field]; bc = "aaaa"): HipSmearedForce =
  ## smeared_force = gsmear.hypcoeffs.smearGetForce(g, sg, gsmear.info)   (staghmc_spv.nim:989-993, hypsmear.nim:49-247)
  var gb, sgb: HostBuf
  toHostG(g, gb)
  sgb.setLen(gb.len)
  chk qexhip_nhyp_prepare(hipParam.h, gb.p, coef.alpha1.cdouble, coef.alpha2.cdouble, coef.alpha3.cdouble, sgb.p)
  fromHostG(sg, sgb)
  for mu in 0..3: result.bc[mu] = (if bc[mu] == 'a': 1 else: 0).cint

proc smearedForce*(sf: HipSmearedForce; f, chain: openArray[Field]) =
  ## smearedForce(f, chain) (hypsmear.nim:145-247): f = dS/dU^+ from chain = dS/dV^+; f may be chain (f.smeared_force(f))
  var fb, cb: HostBuf
  toHostG(chain, cb)
  fb.setLen(cb.len)
  chk qexhip_nhyp_force(hipParam.h, fb.p, cb.p)
  fromHostG(f, fb)

proc gforce*(sf: HipSmearedForce; gc: GaugeActionCoeffs; f: openArray[Field]) =
  ## act.gforce(g, sg, f, smeared_force) (staghmc_spv.nim:217-228): action derivative on the smeared links, chain, projTAH
  var fb: HostBuf
  fb.setLen(f[0].l.hipSetup.nSites * 72)
  chk qexhip_nhyp_gauge_force(hipParam.h, fb.p, gc.plaq.cdouble, gc.rect.cdouble, gc.adjplaq.cdouble)
  fromHostG(f, fb)

proc fforce*(sf: HipSmearedForce; f: openArray[Field]; psis: openArray[Field]; scales: openArray[float]) =
  ## the force part of s.fforce (staghmc_spv.nim:716-865) once the solves are done: f = TAH(chain(sum_k scale_k psi_k x psi_k(+mu)^+) g^+)
  let n = psis.len
  var pb = newSeq[HostBuf](n)
  var ptrs = newSeq[ptr cdouble](n)
  var sc = newSeq[cdouble](n)
  for k in 0..<n:
    toHost(psis[k], pb[k])
    ptrs[k] = pb[k].p
    sc[k] = scales[k].cdouble
  var fb: HostBuf
  fb.setLen(f[0].l.hipSetup.nSites * 72)
  var bcv = sf.bc
  chk qexhip_nhyp_fermion_force(hipParam.h, fb.p, ptrs[0].addr, sc[0].addr, n.cint, bcv[0].addr, nil)
  fromHostG(f, fb)

proc setLinksFromClosure*(sf: HipSmearedForce) =
  ## sg.rephase(); stag = newStag(sg) (staghmc_spv.nim:601-604,1004): the operator's links straight from the closure's
  ## smeared links (they never cross PCIe); the alphas are ignored when g = nil
  var bcv = sf.bc
  chk qexhip_stag_set_links_nhyp(hipParam.h, nil, 0.0, 0.0, 0.0, bcv[0].addr, nil)

proc release*(sf: HipSmearedForce) =
  chk qexhip_nhyp_release(hipParam.h)
