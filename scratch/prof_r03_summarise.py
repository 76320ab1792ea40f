#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of scratch/prof_r03.sh into the tracked round-3 evidence:

    python3 scratch/prof_r03_summarise.py <gpurun_out/prof_r03> <dest prefix dir>

writes  <dest>/bench_4a_{overlap,serial}_kernel_stats.csv + _top.txt   per-kernel launches / average / share,
        <dest>/pmc_conv_kernels.json   per conv kernel: HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, the gfx950 correction of
                                       MI355X_MICROARCH.md: rocprofv3 reports KiB, FETCH_SIZE counts 32-byte requests as 64) with the raw
                                       per-dispatch rows of the dominant kernel, the issue-side counters (MFMA busy share, LDS wait, VMEM),
                                       and the sha256 of each kernel's source file -- bench.py reports `roofline.traffic` only while the
                                       source it was measured on is the source in the tree,
        <dest>/issue_counters.md       the same counters as a table.
"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {  # tag in the kernel name -> (display name, source file)
    "conv_band16_k": ("conv_band16_k", "conv_band16.hip"),
    "conv_ring16_k": ("conv_ring16_k<256,256,2,4>", "conv_ring16.hip"),
    "conv3x3_c64n64_duo": ("conv3x3_c64n64_duo_k", "conv_patch3.hip"),
    "conv_halo_k": ("conv_halo_k", "conv_halo.hip"),
    "conv_ring_k<256, 128": ("conv_ring_k<256,128,4,2>", "conv_ring.hip"),
    "conv_s2d_pool_k": ("conv_s2d_pool_k", "conv_s2d.hip"),
    "conv_s2d_k": ("conv_s2d_k", "conv_s2d.hip"),
}


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    serial_avg = {}
    for mode in ("overlap", "serial"):
        fs = glob.glob(f"{src}/{mode}/*kernel_trace.csv")
        if not fs:
            continue
        rows = list(csv.DictReader(open(fs[0])))
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in rows:
            agg[r["Kernel_Name"]][0] += 1
            agg[r["Kernel_Name"]][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot = sum(v[1] for v in agg.values())
        order = sorted(agg.items(), key=lambda kv: -kv[1][1])
        with open(f"{dst}/bench_4a_{mode}_kernel_stats.csv", "w") as fo:
            fo.write('"Name","Calls","TotalDurationUs","AverageUs","Percentage"\n')
            for k, (n, t) in order:
                fo.write('"%s",%d,%.1f,%.2f,%.2f\n' % (k.replace('"', "'"), n, t, t / n, 100 * t / tot))
        with open(f"{dst}/bench_4a_{mode}_top.txt", "w") as fo:
            fo.write("# %s streams: total kernel time %.1f ms over the traced steps, %d distinct kernels\n" % (mode, tot / 1e3, len(agg)))
            for k, (n, t) in order[:36]:
                fo.write("%8.1f us/launch  n=%5d  %5.1f%%  %s\n" % (t / n, n, 100 * t / tot, k[:120]))
        if mode == "serial":
            for k, (n, t) in agg.items():
                tag = next((w for w in KERNELS if w in k), None)
                if tag:
                    d = serial_avg.setdefault(tag, [0, 0.0])
                    d[0] += n
                    d[1] += t
    # ---- PMC passes ----
    per = collections.OrderedDict()
    raw_rows = collections.defaultdict(list)
    for f in sorted(glob.glob(f"{src}/p*/*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            tag = next((w for w in KERNELS if w in k), None)
            if tag is None:
                continue
            per.setdefault(tag, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                raw_rows[tag].append({"counter": r["Counter_Name"], "value_kib": float(r["Counter_Value"]), "grid": r.get("Grid_Size", ""),
                                      "dispatch": r.get("Dispatch_Id", "")})
    out = {"units": "rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB per dispatch; hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction)",
           "command": "scratch/prof_r03.sh: AUTOMOE_PARALLEL_EXPERTS=0 AUTOMOE_OVERLAP_BACKBONE=0 AUTOMOE_PREFETCH_EXPERTS=0 AUTOMOE_HIPGRAPH=0 rocprofv3 "
                      "--kernel-trace --pmc <one group per pass> -- python3 bench.py --steps 2 --warmup 1 --no-extras (per-GPU batch 32)",
           "batch": 32, "kernels": {}}
    for tag, cs in per.items():
        name, srcfile = KERNELS[tag]
        mean = {c: sum(v) / len(v) for c, v in cs.items()}
        e = {"launches_traced": {c: len(v) for c, v in cs.items()}, "mean_per_launch": mean,
             "source": f"self-driving-model_amd/csrc/{srcfile}", "source_sha256": sha(os.path.join(ROOT, "self-driving-model_amd", "csrc", srcfile))}
        if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
            e["hbm_bytes_per_launch"] = (2.0 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024.0
        if mean.get("SQ_VALU_MFMA_BUSY_CYCLES") and mean.get("GRBM_GUI_ACTIVE"):
            # busy cycles are summed over the 1024 SIMDs of the chip, GRBM_GUI_ACTIVE over the 8 XCDs
            e["mfma_busy_frac"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (mean["GRBM_GUI_ACTIVE"] / 8.0)
        if mean.get("SQ_WAVE_CYCLES"):
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU"):
                if c in mean:
                    e.setdefault("share_of_wave_cycles", {})[c] = mean[c] / mean["SQ_WAVE_CYCLES"]
        if tag in serial_avg:
            e["serial_trace_avg_us"] = serial_avg[tag][1] / serial_avg[tag][0]
            e["serial_trace_launches"] = serial_avg[tag][0]
        e["raw_rows"] = raw_rows.get(tag, [])
        out["kernels"][name] = e
    json.dump(out, open(f"{dst}/pmc_conv_kernels.json", "w"), indent=1)
    with open(f"{dst}/issue_counters.md", "w") as fo:
        fo.write("# Round 3: issue-side counters of the 4a step's conv kernels (inside the real step, eager, streams serialised)\n\n"
                 "`SQ_VALU_MFMA_BUSY_CYCLES` counts cycles per SIMD (summed over 1,024 SIMDs), `GRBM_GUI_ACTIVE` is summed over the 8 XCDs, "
                 "`SQ_WAVE_CYCLES` / `SQ_WAIT_*` / `SQ_ACTIVE_INST_*` are quad-cycles summed over waves (MI355X_MICROARCH.md).\n\n")
        for name, e in out["kernels"].items():
            fo.write(f"## {name}\n\n")
            if "mfma_busy_frac" in e:
                fo.write(f"MFMA busy: **{100 * e['mfma_busy_frac']:.1f} %** of the launch (per SIMD, against GRBM_GUI_ACTIVE / 8)\n\n")
            if "hbm_bytes_per_launch" in e:
                fo.write(f"HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE): {e['hbm_bytes_per_launch'] / 1e6:.1f} MB\n\n")
            if "share_of_wave_cycles" in e:
                fo.write("share of wave cycles: " + ", ".join(f"{c} {100 * v:.1f} %" for c, v in e["share_of_wave_cycles"].items()) + "\n\n")
            fo.write("| counter | launches | mean per launch |\n|---|---|---|\n")
            for c, v in e["mean_per_launch"].items():
                fo.write(f"| {c} | {e['launches_traced'][c]} | {v:.4g} |\n")
            fo.write("\n")
    print(open(f"{dst}/issue_counters.md").read()[:2500])


if __name__ == "__main__":
    main()
