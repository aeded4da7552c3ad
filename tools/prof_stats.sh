#!/bin/bash
# tools/prof_stats.sh [pattern] -- ON the GPU box: rocprofv3 kernel-trace summary of a short bench run, kernels matching pattern
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_stats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o ps --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-optin --no-extras > $O/bench.json 2> $O/err.log
f=$(find $O -name "ps_kernel_stats.csv" | head -1)
test -n "$f"
python3 - "$f" "${1:-k_}" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print("%-60s calls %4s avg_us %9.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
