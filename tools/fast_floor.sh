#!/bin/bash
# tools/fast_floor.sh <outdir> -- ON the GPU box: SQ_INSTS_VALU / SQ_INSTS_LDS and the duration of k_fast per launch, for the shipped build
# and the ablation builds of tools/fast_ablate.sh (variants/libfast_{s1,s2a,s2,s3}.so) -> <outdir>/fast_floor_pmc.json (tools/fast_floor.py --pmc)
set -e
# (no -e inside the loop: rocprofv3 passes of garbage-producing ablation builds may exit non-zero)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=${1:-$R/gpurun_out/r04g}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --prewarm-ms 50 --no-cpu-baseline --no-optin --no-extras --no-check --frames-cache /tmp/bench_frames"
$B > /dev/null 2>&1   # (frames generated and cached outside the counter passes)
for v in full s1 s2a s2 s3; do
  lib=$R/visual-slam_amd/variants/libfast_$v.so; [ "$v" = full ] && lib=$R/visual-slam_amd/libvslam_amd.so
  export VSLAM_AMD_LIB=$lib
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-include-regex 'k_fast' --kernel-trace -d $O/pmc_$v -o p --output-format csv -- $B > /dev/null 2> $O/pmc_$v.err || true
  rocprofv3 --kernel-trace --stats -d $O/st_$v -o s --output-format csv -- $B > /dev/null 2> $O/st_$v.err || true
done
unset VSLAM_AMD_LIB
python3 - $O <<'PY'
import csv, json, sys, collections, glob, os
O = sys.argv[1]
out = {}
for v in ("full", "s1", "s2a", "s2", "s3"):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(O, "pmc_" + v, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_fast" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    ms = None
    for f in glob.glob(os.path.join(O, "st_" + v, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_fast" in r["Name"]:
                ms = float(r["AverageNs"]) / 1e6
    avg = lambda k: sum(acc[k]) / max(len(acc[k]), 1)
    out[v] = {"valu": avg("SQ_INSTS_VALU"), "lds": avg("SQ_INSTS_LDS"), "salu": avg("SQ_INSTS_SALU"), "ms": ms, "launches": len(acc["SQ_INSTS_VALU"])}
json.dump(out, open(os.path.join(O, "fast_floor_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/pmc_* $O/st_*
