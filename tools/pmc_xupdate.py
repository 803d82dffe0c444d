#!/usr/bin/env python3
"""HBM-side traffic of the LSQR ITERATION kernels per LSQR iteration, by configuration, from the rocprofv3 --pmc passes of tools/pmc_xupdate.sh
(FETCH_SIZE and WRITE_SIZE in separate passes over tools/xupdate_times.py <cfg>), written to profiles/xupdate_traffic.json -- what bench.py reports
as xupdate.roofline.traffic.

    python tools/pmc_xupdate.py gpurun_out/pmcx_<tag> [summary.txt]

Iteration kernels = k_ks_persist (all iterations of a solve in one launch), or k_ks_a + k_ks_b (two launches per iteration): the launches
bench.py's `ms_lsqr_kernels` times.  Correction as MI355X_MICROARCH.md section HBM prescribes: counters in KB (x 1024); FETCH_SIZE counts a
16-byte-per-lane coalesced read at half its bytes -> x 2 (the kernels' operand loads are double2 = 16 B per lane; the 8-byte tag polls of the
one-launch form are a width the guide leaves uncalibrated and are doubled with the rest: an upper bound); WRITE_SIZE as counted.
Bytes per LSQR iteration = corrected bytes of all such dispatches of the process / the LSQR iterations it ran per slice (slices of a batch iterate
together, so this is the traffic of one iteration of the whole batch -- the unit of bench.py's us_per_lsqr_iteration)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = {0: ("spiral_T200_B1", "spiral cut3, one slice"), 2: ("spiral_T1000_B1", "spiral cut0 (T = 1000), one slice"),
       3: ("spiral_T200_B15", "spiral cut3, 15 slices"), 4: ("epi_T200_B15", "EPI cut3, 15 slices")}


def total(d, counter):
    tot, n, names = 0.0, 0, set()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            nm = r["Kernel_Name"]
            if r.get("Counter_Name") == counter and ("k_ks_persist" in nm or "k_ks_a<" in nm or "k_ks_b<" in nm):
                tot += float(r["Counter_Value"]) * 1024
                n += 1
                names.add(nm.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:40])
    return tot, n, sorted(names)


def main():
    out_dir = sys.argv[1]
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/xupdate_times.py <cfg>; tools/pmc_xupdate.py: FETCH x 2 + WRITE",
           "configs": {}}
    for cfg, (key, text) in CFG.items():
        log = os.path.join(out_dir, f"c{cfg}_FETCH_SIZE.log")
        iters = None
        if os.path.exists(log):
            for line in open(log):
                if line.startswith("{"):
                    iters = json.loads(line).get("lsqr_iters_per_slice_all_runs")
        f, nf, names = total(os.path.join(out_dir, f"c{cfg}_FETCH_SIZE"), "FETCH_SIZE")
        w, nw, _ = total(os.path.join(out_dir, f"c{cfg}_WRITE_SIZE"), "WRITE_SIZE")
        if not iters or not nf or not nw:
            continue
        res["configs"][key] = {"what": text, "kernels": " + ".join(names), "dispatches": nf, "lsqr_iterations_per_slice": iters,
                               "fetch_counted_bytes": int(f), "write_bytes": int(w), "corrected_bytes": int(2 * f + w),
                               "bytes_per_lsqr_iteration": int((2 * f + w) / iters)}
    with open(os.path.join(ROOT, "profiles", "xupdate_traffic.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    txt = json.dumps(res, indent=1)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as fh:
            fh.write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
