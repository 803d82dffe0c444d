#!/usr/bin/env python3
"""HBM-side traffic of the dominant kernel (k_conv6 at the 224 x 224 x 64 level) from two SEPARATE rocprofv3 --pmc passes
(FETCH_SIZE in one, WRITE_SIZE in the other -- they do not fit one pass, MI355X_MICROARCH.md section rocprofv3 PMC slots), corrected as
that guide's HBM section prescribes, written to profiles/conv_traffic.json (which bench.py reports as roofline.traffic).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/prof_net.py 1 3
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 tools/prof_net.py 1 3
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write [profiles/rNN_pmc_conv_traffic.txt]
    python tools/pmc_traffic.py --batch 15 gpurun_out/pmc_fetch15 gpurun_out/pmc_write15 [txt]     (passes taken on tools/prof_net.py 15 2: the
        persistent k_conv6p<0, ..> of slice batches; writes profiles/conv_traffic_batch15.json, which bench.py --workload slices reports)

Corrections.  FETCH_SIZE / WRITE_SIZE are in KB (x 1024).  On gfx950 FETCH_SIZE counts a 16-byte-per-lane coalesced read at exactly half
its bytes; WRITE_SIZE is exact for 16-byte-per-lane stores.  With BLOCKED interior tensors (k_conv6<.., true, ..>: DESIGN.md section 4)
every request of the kernel -- weights, activations, residual operands -- is 16 B per lane, so the whole of FETCH_SIZE is doubled.
The residual share still checks the factor: the launches of one forward pass alternate between layers without and with a residual
operand, the two populations separate cleanly, and their difference as counted must be half of the 12 845 056 bytes a residual
operand has (it is: 6.43 MB).  (Planar tensors, the layout before: activation requests were 4 B per lane, a width the guide leaves
uncalibrated, taken as counted, and only the residual share was doubled -- profiles/r02_d_pmc_conv_traffic.txt.)"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALG = {"input_with_halo": 196 * 64 * 18 * 18 * 4, "residual": 64 * 224 * 224 * 4, "output": 64 * 224 * 224 * 4, "weights_f16_pairs": 64 * 64 * 9 * 4}


def rows(d, counter):
    out = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                out.append((int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]), float(r["Counter_Value"])))
    return sorted(out)


def family_per_forward(dfetch, dwrite, marker):
    """Round 5: bench.py's roofline covers the whole 3x3 family of a forward pass as timed UNITS (a launch per layer, a split-K layer with its reduce
    kernel, a resident-tile launch whole).  This is the matching traffic: every dispatch of the family (k_conv6 / k_conv6p / k_conv6r / k_conv6_reduce*)
    from the first dispatch whose name contains `marker` on -- the forward passes of tools/prof_net.py; what comes before is qmri_set_denoiser's
    calibration probe -- FETCH_SIZE x 2 (16 B per lane requests; the reduce kernel's 4-byte partial-sum reads are a width the guide does not
    calibrate: counted as they are, they are < 2 % of the total) + WRITE_SIZE, divided by the number of forward passes (= marker dispatches / 2 for
    k_conv6r, two per pass; the head's launches for batches)."""
    fam = lambda n: ("k_conv6<" in n or "k_conv6p<" in n or "k_conv6r<" in n or "k_conv6_reduce" in n)
    rf, rw = rows(dfetch, "FETCH_SIZE"), rows(dwrite, "WRITE_SIZE")
    out = {}
    for key, rs, fac in (("fetch", rf, 2.0), ("write", rw, 1.0)):
        ids = [i for (i, n, g, v) in rs if marker in n]
        if not ids:
            return None
        first = min(ids)
        tot = 0.0
        for (i, n, g, v) in rs:
            if i >= first and fam(n):
                tot += v * 1024 * (1.0 if (key == "fetch" and "k_conv6_reduce" in n) else fac)
        out[key] = tot
        out["n_marker_" + key] = len(ids)
    return out


def main_batch(dfetch, dwrite, batch):
    """The persistent kernel of slice batches at the 224 x 224 x 64 level: k_conv6p<0, NRES>, one launch = `batch` slices = batch * 196 tiles
    on 256 workgroups.  Populations by the template argument NRES (0 plain, 1 one residual operand, 2 residual + skip)."""
    per = {}
    for nres in (0, 1, 2):
        tagname = "k_conv6p<0, %d" % nres
        f = sorted(v for (_, n, g, v) in rows(dfetch, "FETCH_SIZE") if tagname in n)
        w = sorted(v for (_, n, g, v) in rows(dwrite, "WRITE_SIZE") if tagname in n)
        # the 224 x 224 level's launches are the ones that write batch * 12.8 MB (the 112 x 112 level writes half of that)
        w224 = [v for v in w if abs(v * 1024 - batch * ALG["output"]) < 0.05 * batch * ALG["output"]]
        n224 = len(w224)
        if not n224 or not f:
            continue
        f224 = f[-n224:] if nres else f[len(f) - n224:]              # the level with the largest tensors fetches most
        med = lambda a: sorted(a)[len(a) // 2]
        per[nres] = {"launches": n224, "fetch_counted": int(med(f224) * 1024), "write": int(med(w224) * 1024)}
    if 0 not in per or 1 not in per:
        raise SystemExit("no k_conv6p<0, 0 / 1> launches of the 224 x 224 level found: " + str(per))
    alg_plain = batch * (ALG["input_with_halo"] + ALG["output"]) + ALG["weights_f16_pairs"] * 256       # (every workgroup streams the layer's weights once per tile: from L2)
    alg = {0: batch * (ALG["input_with_halo"] + ALG["output"]), 1: batch * (ALG["input_with_halo"] + ALG["output"] + ALG["residual"]),
           2: batch * (ALG["input_with_halo"] + ALG["output"] + 2 * ALG["residual"])}
    res_counted = per[1]["fetch_counted"] - per[0]["fetch_counted"]
    out = {"kernel": "k_conv6p<0, NRES> (224 x 224 x 64 level, %d slices per launch, 256 persistent workgroups)" % batch, "batch": batch,
           "per_nres": {}, "residual_read_as_counted": int(res_counted), "residual_read_true": batch * ALG["residual"],
           "tensor_format": "blocked [c/8][w][h][8]",
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/prof_net.py %d 2; tools/pmc_traffic.py --batch; FETCH x2 "
                     "(16 B per lane requests; the residual share as counted = half its bytes confirms the factor)" % batch}
    tot_b, tot_n = 0.0, 0
    for nres, d in per.items():
        corr = 2 * d["fetch_counted"] + d["write"]
        out["per_nres"][str(nres)] = {**d, "corrected_bytes": int(corr), "algorithmic_bytes": int(alg[nres]), "ratio": round(corr / alg[nres], 3)}
        tot_b += corr * d["launches"]; tot_n += d["launches"]
    out["corrected_bytes_per_launch_per_slice"] = int(tot_b / tot_n / batch)
    fam = family_per_forward(dfetch, dwrite, "k_conv6p<0, 0")
    if fam:
        nfwd = REPS if REPS > 0 else 2                             # (tools/prof_net.py <batch> 2: two forward passes)
        out["family_per_forward"] = {"forward_passes": nfwd, "corrected_bytes": int((fam["fetch"] + fam["write"]) / nfwd),
                                     "what": "every 3x3-family dispatch of a forward pass of %d slices: FETCH_SIZE x 2 + WRITE_SIZE" % batch}
    with open(os.path.join(ROOT, "profiles", "conv_traffic_batch%d.json" % batch), "w") as fh:
        json.dump(out, fh, indent=1)
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as fh:
            fh.write(txt + "\n")
    print(txt)


def main_resident(f, w, family=None):
    """k_conv6r: one launch = the eight ResBlock layers of the 224 x 224 x 64 level with LDS-resident tiles (conv6_kernels.hip).  Two launches per
    forward pass: the down path's (no skip operand) and the up path's (+ the skip tensor at the last layer): the two populations of FETCH_SIZE.
    Every request is 16 B per lane: FETCH_SIZE x 2.  Algorithmic bytes per launch: the input tile with its ring once, the four block inputs read
    back as residual operands, the four ResBlock outputs written, seven ring exchanges (196 tiles x 368 triples x 64 B written and read), the
    weights of eight layers once per XCD -- and for the up path the skip tensor."""
    med = lambda a: sorted(a)[len(a) // 2]
    half = len(f) // 2
    f_down, f_up = med(f[:half]) * 1024, med(f[half:]) * 1024
    wm = med(w) * 1024
    xch = 196 * 368 * 64
    alg_w = 8 * ALG["weights_f16_pairs"] * 8
    alg_read_down = ALG["input_with_halo"] + 4 * ALG["residual"] + 7 * xch + alg_w
    alg_read_up = alg_read_down + ALG["residual"]
    alg_write = 4 * ALG["output"] + 7 * xch
    per_layer = ((2 * f_down + wm) + (2 * f_up + wm)) / 2 / 8
    out = {"kernel": "k_conv6r (224 x 224 x 64 level: the eight ResBlock layers of a path in one launch, 196 workgroups with LDS-resident tiles)", "batch": 1,
           "launches": {"down_path": half, "up_path": len(f) - half}, "layers_per_launch": 8,
           "raw": {"fetch_down_bytes": int(f_down), "fetch_up_bytes": int(f_up), "write_bytes": int(wm)},
           "corrected": {"fetch_down_bytes": int(2 * f_down), "fetch_up_bytes": int(2 * f_up), "write_bytes": int(wm)},
           "algorithmic": {"read_down_bytes": alg_read_down, "read_up_bytes": alg_read_up, "write_bytes": alg_write, "ring_exchange_bytes_per_layer": xch, **ALG},
           "ratio_corrected_over_algorithmic": {"down_path": round((2 * f_down + wm) / (alg_read_down + alg_write), 3), "up_path": round((2 * f_up + wm) / (alg_read_up + alg_write), 3)},
           "corrected_bytes_per_launch": int(((2 * f_down + wm) + (2 * f_up + wm)) / 2),
           "corrected_bytes_per_launch_per_slice": int(per_layer),
           "per_layer_note": "corrected_bytes_per_launch_per_slice = bytes per LAYER (a launch is eight layers): against 29.5 / 42.6 MB of a plain / residual layer "
                             "launched alone (profiles/r04_l_pmc_conv_traffic.txt): the ReLU intermediates never leave the chip",
           "tensor_format": "blocked [c/8][w][h][8]",
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/prof_net.py 1 3; tools/pmc_traffic.py; FETCH x2 (every request of the kernel is 16 B per lane)"}
    if family:
        nfwd = REPS if REPS > 0 else max(1, family["n_marker_fetch"] // 2)   # (two k_conv6r launches per pass)
        out["family_per_forward"] = {"forward_passes": nfwd, "corrected_bytes": int((family["fetch"] + family["write"]) / nfwd),
                                     "what": "every 3x3-family dispatch of a one-slice forward pass (k_conv6r x 2, k_conv6 x 40, k_conv6_reduce_blk x 8): FETCH_SIZE x 2 + WRITE_SIZE"}
    with open(os.path.join(ROOT, "profiles", "conv_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as fh:
            fh.write(txt + "\n")
    print(txt)


REPS = 0      # forward passes of the profiled run (--reps N; default: 2 for batches, counted from the k_conv6r launches for one slice)


def main():
    global REPS
    batch = 1
    if sys.argv[1] == "--batch":
        batch = int(sys.argv[2]); del sys.argv[1:3]
    if sys.argv[1] == "--reps":
        REPS = int(sys.argv[2]); del sys.argv[1:3]
    dfetch, dwrite = sys.argv[1], sys.argv[2]
    if batch > 1:
        return main_batch(dfetch, dwrite, batch)
    rf, rw = [v for (_, n, g, v) in rows(dfetch, "FETCH_SIZE") if "k_conv6r" in n], [v for (_, n, g, v) in rows(dwrite, "WRITE_SIZE") if "k_conv6r" in n]
    if rf and rw:
        return main_resident(sorted(rf), sorted(rw), family_per_forward(dfetch, dwrite, "k_conv6r"))
    sel = lambda rs: [v for (_, n, g, v) in rs if "k_conv6<0, 2" in n and g == 196 * 512]
    f, w = sel(rows(dfetch, "FETCH_SIZE")), sel(rows(dwrite, "WRITE_SIZE"))
    if not f or not w:
        raise SystemExit("no k_conv6<0, 2> launches with 196 workgroups found in the counter files")
    f.sort()
    # populations (separated by > 15 % gaps): the head layer (10 input channels), plain layers (input + weights), residual layers
    # (+ the residual operand), the level's last layer (+ residual + skip).  plain = the most populous, residual = the next larger one
    clusters = [[f[0]]]
    for v in f[1:]:
        if v > 1.15 * clusters[-1][-1]: clusters.append([v])
        else: clusters[-1].append(v)
    ip = max(range(len(clusters)), key=lambda i: len(clusters[i]))
    if ip + 1 >= len(clusters):
        raise SystemExit("could not separate plain and residual launches: " + str([(len(c), c[0]) for c in clusters]))
    plain, res = clusters[ip], clusters[ip + 1]
    med = lambda a: sorted(a)[len(a) // 2]
    fp, fr, wm = med(plain) * 1024, med(res) * 1024, med(w) * 1024
    res_counted = fr - fp
    blocked = any("k_conv6<0, 2, true" in n for (_, n, g, v) in rows(dfetch, "FETCH_SIZE") if g == 196 * 512)
    corr_plain, corr_res = (2 * fp, 2 * fr) if blocked else (fp, fp + 2 * res_counted)
    n_p, n_r = len(plain), len(res)
    mean_corr = (n_p * (corr_plain + wm) + n_r * (corr_res + wm)) / (n_p + n_r)
    mean_raw = (n_p * (fp + wm) + n_r * (fr + wm)) / (n_p + n_r)
    alg_plain = ALG["input_with_halo"] + ALG["output"] + ALG["weights_f16_pairs"]
    alg_res = alg_plain + ALG["residual"]
    out = {"kernel": "k_conv6<0, 2> (224 x 224 x 64 level, 196 workgroups)", "batch": 1, "launches": {"plain": n_p, "residual": n_r},
           "raw": {"fetch_plain_bytes": int(fp), "fetch_residual_bytes": int(fr), "write_bytes": int(wm)},
           "corrected": {"fetch_plain_bytes": int(corr_plain), "fetch_residual_bytes": int(corr_res), "write_bytes": int(wm),
                         "residual_read_as_counted": int(res_counted), "residual_read_true": ALG["residual"]},
           "algorithmic": {"plain_layer_bytes": alg_plain, "residual_layer_bytes": alg_res, **ALG},
           "ratio_corrected_over_algorithmic": {"plain": round((corr_plain + wm) / alg_plain, 3), "residual": round((corr_res + wm) / alg_res, 3)},
           "raw_bytes_per_launch_per_slice": int(mean_raw), "corrected_bytes_per_launch_per_slice": int(mean_corr),
           "tensor_format": "blocked [c/8][w][h][8]" if blocked else "planar [c][w][h]",
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/prof_net.py 1 3; tools/pmc_traffic.py; "
                     + ("FETCH x2 (every request of the kernel is 16 B per lane; the residual share as counted = half its bytes confirms the factor)"
                        if blocked else "FETCH x2 on the 16-B-per-lane residual reads (measured share), 4-B-per-lane activation reads as counted")}
    with open(os.path.join(ROOT, "profiles", "conv_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as fh:
            fh.write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
