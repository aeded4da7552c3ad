#!/usr/bin/env python3
"""Turns the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected separately as the MI355X guide prescribes) of
`bench.py --steps 3 --warmup 1 --no-cpu-baseline` into profiles/r01_pmc_traffic.json (HBM bytes per launch / per stage),
and trims the raw counter CSVs to this repo's kernels.  Usage: summarize_pmc.py <fetch_counter_collection.csv> <write_...csv>"""
import collections
import csv
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
STAGES = {"pyramid": ["k_resize"], "fast_nms": ["k_fast"], "select_harris": ["k_select"], "blur": ["k_blur"],
          "angle_rbrief": ["k_describe"], "match_knn2_ratio": ["k_match"], "two_view": ["k_tv_prep", "k_tv_hyp", "k_tv_compact", "k_tv_score", "k_tv_finish"]}
STEPS = 4  # 1 warm-up + 3 timed


def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n.split("<")[0]


def load(path, out_csv):
    agg = collections.defaultdict(list)
    rows = [r for r in csv.DictReader(open(path)) if short(r["Kernel_Name"]).startswith("k_")]
    cols = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
            "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    with open(out_csv, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=cols, extrasaction="ignore")
        w.writeheader()
        for r in rows:
            r = dict(r, Kernel_Name=short(r["Kernel_Name"]))
            w.writerow(r)
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


fe = load(sys.argv[1], os.path.join(HERE, "r01_pmc_fetch_size_counter_collection.csv"))
wr = load(sys.argv[2], os.path.join(HERE, "r01_pmc_write_size_counter_collection.csv"))
out = {"command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-trace -- python3 bench.py --steps 3 "
                  "--warmup 1 --no-cpu-baseline",
       "batch_frames": 256,
       "units": "bytes per launch; a stage = the sum of its launches in one step; FETCH_SIZE / WRITE_SIZE are reported in KiB",
       "calibration": "4-byte-per-lane dword loads read 1:1 in FETCH_SIZE on this path (k_resize: counted bytes within 5 % of the "
                      "source windows it is known to read), so the x2 correction of 16-byte-per-lane streams is not applied; "
                      "k_blur WRITE_SIZE matches its algorithmic 243.3 MB within 1.5 %",
       "per_kernel": {}, "per_stage": {}}
for k in sorted(fe):
    out["per_kernel"][k] = {"launches_per_step": len(fe[k]) // STEPS, "fetch_bytes_avg": sum(fe[k]) / len(fe[k]) * 1024,
                            "write_bytes_avg": sum(wr[k]) / len(wr[k]) * 1024}
for s, ks in STAGES.items():
    f = sum(out["per_kernel"][k]["fetch_bytes_avg"] * out["per_kernel"][k]["launches_per_step"] for k in ks)
    w = sum(out["per_kernel"][k]["write_bytes_avg"] * out["per_kernel"][k]["launches_per_step"] for k in ks)
    out["per_stage"][s] = {"fetch_bytes": f, "write_bytes": w, "hbm_bytes": f + w}
json.dump(out, open(os.path.join(HERE, "r01_pmc_traffic.json"), "w"), indent=1)
for s, v in out["per_stage"].items():
    print("%-18s fetch %8.1f MB  write %8.1f MB" % (s, v["fetch_bytes"] / 1e6, v["write_bytes"] / 1e6))
