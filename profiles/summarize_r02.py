#!/usr/bin/env python3
"""profiles/summarize_r02.py <dir> -- condenses the rocprofv3 passes of profiles/collect_r02.sh into
r02_pmc_per_kernel.json (per launch: SQ instruction counts, FETCH_SIZE, WRITE_SIZE, HBM bytes) and trims the raw CSVs to this
repo's kernels.  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE: on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes
(MI355X_MICROARCH.md, HBM section); the factor is checked below on k_blur, which must read its whole 243 MB input from beyond L2
(working set 500 MB per step) - the JSON carries that calibration ratio."""
import collections
import csv
import json
import os
import sys

D = sys.argv[1]
BLUR_ALG_READ = 256 * 950532  # bytes k_blur must fetch per launch (raw pyramid incl. level 0, 256 frames)


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


sq = load(os.path.join(D, "sq", "sq_counter_collection.csv"), os.path.join(D, "r02_pmc_sq.csv"))
fe = load(os.path.join(D, "fetch", "fetch_counter_collection.csv"), os.path.join(D, "r02_pmc_fetch.csv"))
wr = load(os.path.join(D, "write", "write_counter_collection.csv"), os.path.join(D, "r02_pmc_write.csv"))
avg = lambda xs: sum(xs) / max(len(xs), 1)
out = {"command": "profiles/collect_r02.sh: bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-optin --no-extras, blur serialised "
                  "(VSLAM_AMD_SERIAL_BLUR=1) for the counter passes", "batch_frames": 256,
       "units": "per launch; FETCH_SIZE / WRITE_SIZE arrive in KiB; hbm_bytes = 2 x fetch_bytes + write_bytes", "kernels": {}}
for k in sorted(sq):
    f = avg(fe[k]["FETCH_SIZE"]) * 1024 if k in fe else None
    w = avg(wr[k]["WRITE_SIZE"]) * 1024 if k in wr else None
    e = {"launches": len(sq[k]["SQ_INSTS_VALU"]), "valu_insts": avg(sq[k]["SQ_INSTS_VALU"]), "salu_insts": avg(sq[k]["SQ_INSTS_SALU"]),
         "lds_insts": avg(sq[k]["SQ_INSTS_LDS"]), "lds_bank_conflict_cycles": avg(sq[k]["SQ_LDS_BANK_CONFLICT"]),
         "lds_active_cycles": avg(sq[k]["SQ_LDS_IDX_ACTIVE"]), "busy_cycles_sum_over_32_se": avg(sq[k]["SQ_BUSY_CYCLES"]),
         "fetch_bytes": f, "write_bytes": w, "hbm_bytes": (2 * f + w) if f is not None and w is not None else None}
    out["kernels"][k] = e
cal_path = os.path.join(D, "fetch_full", "fetch_counter_collection.csv")
if os.path.exists(cal_path):  # pass with VSLAM_AMD_BLUR=full: k_blur reads every level completely
    cal = load(cal_path, os.path.join(D, "r02_pmc_fetch_full_blur.csv"))
    fb = avg(cal["k_blur"]["FETCH_SIZE"]) * 1024
    out["fetch_calibration"] = {"kernel": "k_blur (VSLAM_AMD_BLUR=full pass)", "must_read_bytes": BLUR_ALG_READ, "fetch_size_bytes": fb,
                                "ratio_must_read_over_fetch_size": BLUR_ALG_READ / fb}
json.dump(out, open(os.path.join(D, "r02_pmc_per_kernel.json"), "w"), indent=1)
for k, e in out["kernels"].items():
    print("%-14s valu %.3g salu %.3g lds %.3g  fetch %s MB write %s MB" % (
        k, e["valu_insts"], e["salu_insts"], e["lds_insts"], "%.1f" % (e["fetch_bytes"] / 1e6) if e["fetch_bytes"] else "-",
        "%.1f" % (e["write_bytes"] / 1e6) if e["write_bytes"] else "-"))
