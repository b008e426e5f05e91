"""Reduce the rocprofv3 output directories written by profiles/collect.sh to the small CSV / JSON summaries that are
committed under profiles/:

    python3 profiles/summarize.py OUTDIR TAG

    OUTDIR/trace_*/  (--kernel-trace --stats)   -> OUTDIR/TAG_<name>_kernel_stats.csv  (copied as rocprofv3 wrote it)
    OUTDIR/pmc_<counters>/ (--pmc ...)          -> OUTDIR/TAG_pmc_summary.csv: mean counter value per kernel and counter
                                                -> OUTDIR/TAG_dslash_traffic.json: HBM bytes per Dslash launch, FETCH_SIZE
                                                   scaled by the factor the 1 GiB k_read16 calibration kernel gives in the
                                                   same pass, WRITE_SIZE by k_copy16's
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

outdir, tag = sys.argv[1], sys.argv[2]
for d in sorted(glob.glob(os.path.join(outdir, "trace_*"))):
    name = os.path.basename(d)[len("trace_"):]
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(outdir, "%s_%s_kernel_stats.csv" % (tag, name)))

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


def mean(kpat, counter):
    v = [acc[(k, counter)][0] / acc[(k, counter)][1] for k in kernels if kpat in k and (k, counter) in acc]
    return sum(v) / len(v) if v else None


GiB = 1024.0 ** 3
rd, cp_w = mean("k_read16", "FETCH_SIZE"), mean("k_copy16", "WRITE_SIZE")
if rd and cp_w:
    fcorr = GiB / (rd * 1024.0)          # FETCH_SIZE / WRITE_SIZE are reported in KiB
    wcorr = GiB / (cp_w * 1024.0)
    tj = {"source": "profiles/collect.sh " + tag, "fetch_correction": fcorr, "write_correction": wcorr,
          "note": "HBM bytes per launch = FETCH_SIZE*1024*fetch_correction + WRITE_SIZE*1024*write_correction, per-kernel means over "
                  "the dispatches of profiles/pmc_workload.py; corrections from the 1 GiB k_read16 / k_copy16 kernels of the same passes"}
    for kpat, key in (("k_dslash<8, false, false, false, 0>", "dslash8_sweep1_18real"), ("k_dslash<8, false, true, true, 0>", "dslash8_sweep2_18real"),
                      ("k_dslash<8, false, false, false, 1>", "dslash8_sweep1_recon12"), ("k_dslash<8, false, true, true, 1>", "dslash8_sweep2_recon12"),
                      ("k_dslash<16, false, false, false, 0>", "dslash16_sweep1_18real"), ("k_dslash<16, false, true, true, 0>", "dslash16_sweep2_18real"),
                      ("k_force", "k_force"), ("k_plaq", "k_plaq"), ("k_flow_obs_clover", "k_flow_obs_clover"), ("k_projUderiv_batch", "k_projUderiv_batch"), ("k_staple_deriv<", "k_staple_deriv"), ("k_staple_deriv_pair", "k_staple_deriv_pair"), ("k_gen_staple", "k_gen_staple"), ("k_cgm_update", "k_cgm_update")):
        fe, wr = mean(kpat, "FETCH_SIZE"), mean(kpat, "WRITE_SIZE")
        if fe is not None and wr is not None:
            tj[key + "_bytes"] = fe * 1024.0 * fcorr + wr * 1024.0 * wcorr
    if "dslash8_sweep1_18real_bytes" in tj and "dslash8_sweep2_18real_bytes" in tj:
        tj["hbm_bytes_per_launch_32x4"] = 0.5 * (tj["dslash8_sweep1_18real_bytes"] + tj["dslash8_sweep2_18real_bytes"])
    if "dslash8_sweep1_recon12_bytes" in tj and "dslash8_sweep2_recon12_bytes" in tj:
        tj["hbm_bytes_per_launch_32x4_recon12"] = 0.5 * (tj["dslash8_sweep1_recon12_bytes"] + tj["dslash8_sweep2_recon12_bytes"])
    json.dump(tj, open(os.path.join(outdir, "%s_dslash_traffic.json" % tag), "w"), indent=1)
print("summaries written to", outdir)
