## qexhip.nim -- Nim binding of libqexhip.so for QEX (ctpeterson/qex).
##
## Drop this file next to src/quda/ (e.g. src/hip/qexhip.nim), build QEX with
##   -d:qexhipDir=/path/to/repo
## and wire `hipSolveEE/hipSolveOO` into the backend switch of
## src/physics/stagSolve.nim:65-128 (see INTEGRATION.md).  It mirrors
## src/quda/qudaWrapperImpl.nim:165-261 (qudaSolveXX): build a V=1 twin layout, copy the SIMD
## fields site by site into site-major arrays, make ONE C call, copy the solution back.
##
## NOTE: written against the reference sources but NOT compiled in this repository's pipeline
## (there is no Nim compiler in the build image); the C ABI it binds is exercised by the
## Python/ctypes tests instead.

import os, strutils
import base, layout, field
import physics/qcdTypes
import physics/stagD
import solvers/solverBase

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
proc qexhip_stag_set_links(h: QexhipHandle; fat, lng: ptr cdouble): cint {.qh.}
proc qexhip_stag_D(h: QexhipHandle; r, x: ptr cdouble; m, sc: cdouble): cint {.qh.}
proc qexhip_stag_solve_xx(h: QexhipHandle; x, b: ptr cdouble; mass, r2req: cdouble;
                          maxits, parEven: cint; iters: ptr cint; r2: ptr cdouble;
                          hist: ptr cdouble; histcap: cint): cint {.qh.}
proc qexhip_wflow(h: QexhipHandle; nsteps: cint; eps: cdouble): cint {.qh.}
proc qexhip_gauge_set(h: QexhipHandle; g: ptr cdouble): cint {.qh.}
proc qexhip_gauge_get(h: QexhipHandle; g: ptr cdouble): cint {.qh.}
proc qexhip_plaq(h: QexhipHandle; o: ptr cdouble): cint {.qh.}
# the HMC-side entry points (INTEGRATION.md 5b): smearing closure, MD forces, batched solves, gauge-sector pieces
proc qexhip_stag_solve(h: QexhipHandle; x, b: ptr cdouble; mass, r2req: cdouble; maxits: cint;
                       iters: ptr cint; r2: ptr cdouble): cint {.qh.}
proc qexhip_stag_solve_batch(h: QexhipHandle; n: cint; x, b: ptr ptr cdouble; mass, r2req: ptr cdouble;
                             maxits: cint; iters: ptr cint; r2: ptr cdouble): cint {.qh.}
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
proc qexhip_flow_EQ(h: QexhipHandle; loop: cint; o: ptr cdouble): cint {.qh.}
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
      QMP_broadcast(id[0].addr, 128)         # comms/qmp.nim
      chk qexhip_comm_init(hipParam.h, id[0].addr, l.nRanks.cint, l.myRank.cint)
      var nr, rk, dv: cint
      var bus = newString(64)
      chk qexhip_comm_info(hipParam.h, nr.addr, rk.addr, dv.addr, bus.cstring, 64)
      echoAll "libqexhip: rank ", l.myRank, " = RCCL rank ", rk, " of ", nr, " on device ", dv, " (", $bus.cstring, ")"
    hipParam.layout1 = l.physGeom.newLayout 1
    hipParam.initialized = true
  hipParam.layout1

proc hipSetLinks*(s: Staggered) =
  ## upload s.g once per link update (QUDA re-uploads per solve, qudaWrapperImpl.nim:216-240)
  let lo1 = s.g[0].l.hipSetup
  var g1, g3: D4LatticeColorMatrix
  g1.new lo1
  let naik = s.g.len == 8
  if naik: g3.new lo1
  threads:
    for i in s.g[0].sites:
      var cv: array[4, cint]
      s.g[0].l.coord(cv, (s.g[0].l.myRank, i))
      let j = lo1.rankIndex(cv).index
      forO mu, 0, 3:
        forO a, 0, 2:
          forO b, 0, 2:
            if naik:
              g1[j][mu][a,b] := s.g[2*mu]{i}[a,b]
              g3[j][mu][a,b] := s.g[2*mu+1]{i}[a,b]
            else:
              g1[j][mu][a,b] := s.g[mu]{i}[a,b]
  chk qexhip_stag_set_links(hipParam.h, cast[ptr cdouble](g1.dataPtr),
                            if naik: cast[ptr cdouble](g3.dataPtr) else: nil)

proc hipSolveXX*(s: Staggered; r, t: Field; m: SomeNumber; sp: var SolverParams; parEven = true) =
  ## same contract as qudaSolveXX: r <- solution on the even (odd) subset of
  ## 4(m^2 - D_eo D_oe) r = t, r = 0 start, sp.iterations set.
  let lo1 = r.l.hipSetup
  var t1, r1: DLatticeColorVector
  t1.new lo1
  r1.new lo1
  threads:
    for i in r.sites:
      var cv: array[4, cint]
      r.l.coord(cv, (r.l.myRank, i))
      let j = lo1.rankIndex(cv).index
      forO a, 0, 2:
        t1[j][a] := t{i}[a]
  var iters: cint
  var r2: cdouble
  chk qexhip_stag_solve_xx(hipParam.h, cast[ptr cdouble](r1.dataPtr), cast[ptr cdouble](t1.dataPtr),
                           m.cdouble, sp.r2req.cdouble, sp.maxits.cint, (if parEven: 1 else: 0).cint,
                           iters.addr, r2.addr, nil, 0)
  sp.iterations = iters.int
  threads:
    for i in r.sites:
      var cv: array[4, cint]
      r.l.coord(cv, (r.l.myRank, i))
      let j = lo1.rankIndex(cv).index
      forO a, 0, 2:
        r{i}[a] := r1[j][a]

proc hipSolveEE*(s: Staggered; r, t: Field; m: SomeNumber; sp: var SolverParams) =
  hipSolveXX(s, r, t, m, sp, parEven = true)
proc hipSolveOO*(s: Staggered; r, t: Field; m: SomeNumber; sp: var SolverParams) =
  hipSolveXX(s, r, t, m, sp, parEven = false)
