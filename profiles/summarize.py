"""Reduce the rocprofv3 output directories written by profiles/collect.sh to the small CSV / JSON summaries that are
committed under profiles/:

    python3 profiles/summarize.py OUTDIR TAG

    OUTDIR/trace_*/  (--kernel-trace --stats)   -> OUTDIR/TAG_<name>_kernel_stats.csv  (copied as rocprofv3 wrote it)
    OUTDIR/pmc_<counters>/ (--pmc ...)          -> OUTDIR/TAG_pmc_summary.csv: mean counter value per kernel and counter
                                                -> OUTDIR/TAG_dslash_traffic.json: HBM bytes per Dslash launch (what bench.py's
                                                   roofline.traffic quotes)
                                                -> OUTDIR/TAG_kernel_traffic.json: per priced kernel its algorithmic bytes per
                                                   launch, the HBM bytes the counters saw, their ratio, and the average launch
                                                   time of the kernel-trace pass of the same workload

FETCH_SIZE is scaled by the factor the 1 GiB k_read16 calibration kernel gives in the same pass (gfx950 reports half of a
wide coalesced read: MI355X_MICROARCH.md, HBM section), WRITE_SIZE by k_copy16's.

Kernels are matched by their EXACT name: the function name, plus the template argument list where a row asks for one.
(Round 2 matched substrings: "k_force" averaged k_force_lds, k_force_gen and k_force_projtah, "k_plaq" averaged k_plaq
with k_plaq_final.)
"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

outdir, tag = sys.argv[1], sys.argv[2]
VOL = 32 ** 4          # profiles/pmc_workload.py runs 32^4
for d in sorted(glob.glob(os.path.join(outdir, "trace_*"))):
    name = os.path.basename(d)[len("trace_"):]
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(outdir, "%s_%s_kernel_stats.csv" % (tag, name)))


def split_name(full):
    """'k_dslash<8, false>(Args...)' -> ('k_dslash', '<8, false>'); 'void k_x(...)' -> ('k_x', '')"""
    s = full.strip()
    s = re.sub(r"^void\s+", "", s)
    m = re.match(r"([A-Za-z_][A-Za-z0-9_:]*)", s)
    fn = m.group(1) if m else s
    rest = s[len(fn):]
    targs = ""
    if rest.startswith("<"):
        depth = 0
        for i, ch in enumerate(rest):
            depth += ch == "<"
            depth -= ch == ">"
            if depth == 0:
                targs = rest[: i + 1]
                break
    return fn, targs.replace(" ", "")


acc = defaultdict(lambda: [0.0, 0])
for d in sorted(glob.glob(os.path.join(outdir, "pmc_*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"], row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"])
            acc[k][1] += 1
kernels = sorted({k for k, _ in acc})
counters = sorted({c for _, c in acc})
if kernels:
    with open(os.path.join(outdir, "%s_pmc_summary.csv" % tag), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "dispatches"] + ["%s_mean" % c for c in counters])
        for k in kernels:
            n = max(acc[(k, c)][1] for c in counters if (k, c) in acc)
            w.writerow([k, n] + ["%.1f" % (acc[(k, c)][0] / acc[(k, c)][1]) if (k, c) in acc else "" for c in counters])


def matches(full, fn, targs):
    """exact function name; template arguments equal, or -- a kernel that has GROWN template parameters since the pattern was written --
    equal up to trailing defaults (false / 0): `<8,false,true,true,0>` also names `<8,false,true,true,0,false,0>`"""
    f, t = split_name(full)
    if f != fn:
        return False
    if targs is None:
        return True
    want = targs.replace(" ", "")
    if t == want:
        return True
    if t.startswith(want[:-1] + ","):
        return all(x in ("false", "0") for x in t[len(want):-1].split(","))
    return False


def mean(fn, targs, counter):
    """dispatch-weighted mean of `counter` over the kernels named exactly fn (with exactly these template arguments)"""
    tot, n = 0.0, 0
    for k in kernels:
        if matches(k, fn, targs) and (k, counter) in acc:
            tot += acc[(k, counter)][0]
            n += acc[(k, counter)][1]
    return (tot / n, n) if n else (None, 0)


# average launch time (us) from the kernel-trace pass of the same workload
times = {}
for f in glob.glob(os.path.join(outdir, "trace_workload", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        times[row["Name"]] = (float(row["AverageNs"]) / 1e3, int(row["Calls"]), float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3)


def avg_us(fn, targs):
    tot, n, lo, hi = 0.0, 0, None, None
    for k, (a, c, mn, mx) in times.items():
        if matches(k, fn, targs):
            tot += a * c
            n += c
            lo = mn if lo is None else min(lo, mn)
            hi = mx if hi is None else max(hi, mx)
    return (tot / n, n, lo, hi) if n else (None, 0, None, None)


GiB = 1024.0 ** 3
rd, _ = mean("k_read16", None, "FETCH_SIZE")
cp_w, _ = mean("k_copy16", None, "WRITE_SIZE")
if rd and cp_w:
    fcorr = GiB / (rd * 1024.0)          # FETCH_SIZE / WRITE_SIZE are reported in KiB
    wcorr = GiB / (cp_w * 1024.0)
    note = ("HBM bytes per launch = FETCH_SIZE*1024*fetch_correction + WRITE_SIZE*1024*write_correction, dispatch-weighted means over "
            "the dispatches of profiles/pmc_workload.py; corrections from the 1 GiB k_read16 / k_copy16 kernels of the same passes; "
            "kernels matched by exact function name and template arguments")
    tj = {"source": "profiles/collect.sh " + tag, "fetch_correction": fcorr, "write_correction": wcorr, "note": note}
    for fn, targs, key in (("k_dslash", "<8, false, false, false, 0>", "dslash8_sweep1_18real"), ("k_dslash", "<8, false, true, true, 0>", "dslash8_sweep2_18real"),
                           ("k_dslash", "<8, false, false, false, 1>", "dslash8_sweep1_recon12"), ("k_dslash", "<8, false, true, true, 1>", "dslash8_sweep2_recon12"),
                           ("k_dslash", "<16, false, false, false, 0>", "dslash16_sweep1_18real"), ("k_dslash", "<16, false, true, true, 0>", "dslash16_sweep2_18real")):
        fe, _ = mean(fn, targs, "FETCH_SIZE")
        wr, _ = mean(fn, targs, "WRITE_SIZE")
        if fe is not None and wr is not None:
            tj[key + "_bytes"] = fe * 1024.0 * fcorr + wr * 1024.0 * wcorr
    if "dslash8_sweep1_18real_bytes" in tj and "dslash8_sweep2_18real_bytes" in tj:
        tj["hbm_bytes_per_launch_32x4"] = 0.5 * (tj["dslash8_sweep1_18real_bytes"] + tj["dslash8_sweep2_18real_bytes"])
    if "dslash8_sweep1_recon12_bytes" in tj and "dslash8_sweep2_recon12_bytes" in tj:
        tj["hbm_bytes_per_launch_32x4_recon12"] = 0.5 * (tj["dslash8_sweep1_recon12_bytes"] + tj["dslash8_sweep2_recon12_bytes"])
    json.dump(tj, open(os.path.join(outdir, "%s_dslash_traffic.json" % tag), "w"), indent=1)

    # ---- per priced kernel: algorithmic bytes, HBM bytes, ratio, time (DESIGN.md section 4 quotes these) ----
    Vh = VOL // 2
    M = 144                                        # bytes of one 3x3 complex fp64 matrix
    # (function, template args or None = all instantiations that ran, algorithmic bytes per launch, what they are)
    priced = [
        ("k_dslash", "<8, false, false, false, 0>", (8 * M + 48 + 48) * Vh, "8 links + vector in + vector out per output site (SURVEY 8d: 1248 B)"),
        ("k_dslash", "<8, false, true, true, 0>", (8 * M + 48 + 48 + 48) * Vh, "the same + the 4 m^2 x term (1296 B)"),
        ("k_dslash", "<16, false, false, false, 0>", (16 * M + 48 + 48) * Vh, "16 links + vector in + out (2400 B)"),
        ("k_dslash", "<16, false, true, true, 0>", (16 * M + 48 + 48 + 48) * Vh, "the same + the 4 m^2 x term"),
        ("k_cg_xpay", None, 144 * Vh, "p = r + beta p: 2 reads + 1 write of 48 B"),
        ("k_cg_update", None, 288 * Vh, "x, r updates: 4 reads + 2 writes of 48 B"),
        ("k_cgm_update", None, (48 + 96 + 9 * 192) * Vh, "r in, ps[0] in/out, 9 x (xs, ps in/out), 10 shifts"),
        ("k_force_lds", None, (4 * M * 3 + 4 * M * 2.0 / 3.0) * VOL, "U in, U' out, momentum out, momentum in for stages 2-3 (2112 B/site average)"),
        ("k_force_lds2", None, (4 * M * 3 + 4 * M * 2.0 / 3.0) * VOL, "U in, U' out, momentum out, momentum in for stages 2-3 (2112 B/site average); both parities of a tile position per workgroup"),
        ("k_plaq", None, 4 * M * VOL, "4 links per site read once (576 B)"),
        ("k_flow_obs_clover", None, 4 * M * VOL, "4 links per site read once (576 B)"),
        ("k_flow_obs_clover2", None, 4 * M * VOL, "4 links per site read once (576 B): plaquettes + clover E, Q in one pass, both parities of a tile position per workgroup"),
        ("k_flow_obs_all", None, 4 * M * VOL, "4 links per site read once (576 B): plaquette + clover E, Q in one pass"),
        ("k_gen_staple", None, 4 * M * VOL, "two input matrices read once, accumulator read + written: 576 B/site unique"),
        ("k_staple_deriv_pair", None, 10 * M * VOL, "a (mu,nu)/(nu,mu) pair: 6 matrices read, 2 read and written back = 1440 B/site (rounds 2-3 quoted the 8 reads alone, 1152 B, against read + write traffic)"),
        ("k_projUderiv_batch", None, None, "864 B per link of the batch (sizes differ per level: see dispatches)"),
    ]
    kt = {"source": "profiles/collect.sh " + tag + " (profiles/pmc_workload.py, 32^4)", "fetch_correction": fcorr, "write_correction": wcorr,
          "note": note + "; avg_us from the --kernel-trace --stats pass of the same workload; achieved_alg_gbs = alg_bytes / avg_us", "kernels": {}}
    for fn, targs, alg, what in priced:
        fe, nfe = mean(fn, targs, "FETCH_SIZE")
        wr, _ = mean(fn, targs, "WRITE_SIZE")
        if fe is None or wr is None:
            continue
        hbm_r, hbm_w = fe * 1024.0 * fcorr, wr * 1024.0 * wcorr
        us, ncalls, lo, hi = avg_us(fn, targs)
        e = {"dispatches_pmc": nfe, "hbm_read_bytes": round(hbm_r), "hbm_write_bytes": round(hbm_w), "hbm_bytes": round(hbm_r + hbm_w), "what_alg_counts": what}
        if alg is not None:
            e["alg_bytes"] = int(alg)
            e["ratio_hbm_over_alg"] = round((hbm_r + hbm_w) / alg, 3)
        if us:
            e.update({"avg_us": round(us, 2), "min_us": round(lo, 2), "max_us": round(hi, 2), "calls_trace": ncalls})
            if alg is not None:
                e["achieved_alg_gbs"] = round(alg / us / 1e3, 1)
                e["frac_of_8TBs"] = round(alg / us / 1e3 / 8000.0, 4)
            e["achieved_hbm_gbs"] = round((hbm_r + hbm_w) / us / 1e3, 1)
        for extra in ("TCC_HIT_sum", "TCC_MISS_sum", "TCP_TCC_READ_REQ_sum"):
            v, _ = mean(fn, targs, extra)
            if v is not None:
                e[extra] = round(v)
        kt["kernels"][fn + (targs or "")] = e
    json.dump(kt, open(os.path.join(outdir, "%s_kernel_traffic.json" % tag), "w"), indent=1)
print("summaries written to", outdir)
