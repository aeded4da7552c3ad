#!/usr/bin/env python3
"""profiles/summarize.py <dir> <tag> <command> -- condenses the rocprofv3 passes of profiles/collect.sh into <tag>_pmc_per_kernel.json
(per launch: SQ instruction counts, LDS cycles, FETCH_SIZE, WRITE_SIZE, HBM bytes, average duration) and trims the raw CSVs to this
repo's kernels.  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE: on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes
(MI355X_MICROARCH.md, HBM section); round 2 checked the factor on this path (profiles/r02_pmc_per_kernel.json: k_blur must read
243.3 MB and the counter reports 124.1 MB)."""
import collections
import csv
import json
import os
import sys

D, TAG, CMD = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""


def short(name):
    return name.split("(")[0].replace("void ", "").split("<")[0]


def load(path, out_csv):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    rows = [r for r in csv.DictReader(open(path)) if short(r["Kernel_Name"]).startswith("k_")]
    cols = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Counter_Name",
            "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    with open(out_csv, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=cols, extrasaction="ignore")
        w.writeheader()
        for r in rows:
            r = dict(r, Kernel_Name=short(r["Kernel_Name"]))
            w.writerow(r)
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


sq = load(os.path.join(D, "sq", "sq_counter_collection.csv"), os.path.join(D, TAG + "_pmc_sq.csv"))
fe = load(os.path.join(D, "fetch", "fetch_counter_collection.csv"), os.path.join(D, TAG + "_pmc_fetch.csv"))
wr = load(os.path.join(D, "write", "write_counter_collection.csv"), os.path.join(D, TAG + "_pmc_write.csv"))
dur = {}
try:
    for r in csv.DictReader(open(os.path.join(D, TAG + "_kernel_stats.csv"))):
        dur[short(r["Name"])] = (float(r["AverageNs"]), int(r["Calls"]))
except Exception as e:
    print("no kernel stats:", e)
avg = lambda xs: sum(xs) / max(len(xs), 1)
scene = "smooth" if "--scene smooth" in CMD else "survey8d"
batch = 256
if "--batch" in CMD.split():
    batch = int(CMD.split()[CMD.split().index("--batch") + 1])
out = {"command": CMD, "batch_frames": batch, "scene": scene,
       "units": "per launch; FETCH_SIZE / WRITE_SIZE arrive in KiB; hbm_bytes = 2 x fetch_bytes + write_bytes; avg_ns from the "
                "--kernel-trace --stats pass", "kernels": {}}
for k in sorted(sq):
    f = avg(fe[k]["FETCH_SIZE"]) * 1024 if k in fe else None
    w = avg(wr[k]["WRITE_SIZE"]) * 1024 if k in wr else None
    e = {"launches": len(sq[k]["SQ_INSTS_VALU"]), "valu_insts": avg(sq[k]["SQ_INSTS_VALU"]), "salu_insts": avg(sq[k]["SQ_INSTS_SALU"]),
         "lds_insts": avg(sq[k]["SQ_INSTS_LDS"]), "lds_bank_conflict_cycles": avg(sq[k]["SQ_LDS_BANK_CONFLICT"]),
         "lds_active_cycles": avg(sq[k]["SQ_LDS_IDX_ACTIVE"]), "wait_inst_lds": avg(sq[k]["SQ_WAIT_INST_LDS"]),
         "busy_cycles_sum_over_32_se": avg(sq[k]["SQ_BUSY_CYCLES"]), "wave_cycles": avg(sq[k]["SQ_WAVE_CYCLES"]),
         "fetch_bytes": f, "write_bytes": w, "hbm_bytes": (2 * f + w) if f is not None and w is not None else None,
         "avg_ns": dur.get(k, (None, 0))[0]}
    out["kernels"][k] = e
json.dump(out, open(os.path.join(D, TAG + "_pmc_per_kernel.json"), "w"), indent=1)
for k, e in out["kernels"].items():
    t = e["avg_ns"]
    print("%-14s %8s us  valu %.3g (2-cyc frac %s) lds %.3g conflict/active %.2f  fetch %s MB write %s MB" % (
        k, "%.1f" % (t / 1e3) if t else "-", e["valu_insts"], "%.2f" % (e["valu_insts"] * 2 / (t * 1e-9 * 2.4e9 * 1024)) if t else "-",
        e["lds_insts"], e["lds_bank_conflict_cycles"] / max(e["lds_active_cycles"], 1),
        "%.1f" % (e["fetch_bytes"] / 1e6) if e["fetch_bytes"] else "-", "%.1f" % (e["write_bytes"] / 1e6) if e["write_bytes"] else "-"))
