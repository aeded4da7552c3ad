#!/bin/bash
# tools/trace_levels.sh [kernel-name-prefix]: per-launch durations of one kernel from a rocprofv3 kernel trace of a short bench run
# (blur serialised), grouped by grid size - e.g. the seven k_resize2 launches of a step, one per pyramid level.
R=${GRAFT_REPO_ROOT:-/root/repo}
K=${1:-k_resize2}
O=$R/gpurun_out/trace_levels
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o t --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-optin --no-extras > /dev/null 2> $O/err.txt
python3 - "$O" "$K" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Kernel_Name"].startswith(sys.argv[2]) or (" " + sys.argv[2]) in r["Kernel_Name"]:
        g = (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        acc[g].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for g, v in sorted(acc.items(), reverse=True):
    v = sorted(v); m = v[len(v) // 2]; tot += m
    print("grid %s: %d launches, median %.1f us (min %.1f)" % (g, len(v), m, v[0]))
print("sum of medians %.1f us" % tot)
PY
