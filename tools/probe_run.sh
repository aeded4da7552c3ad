#!/bin/bash
# tools/probe_run.sh <tag> -- ON the GPU box: tools/single_frame_probe.py plain and under rocprofv3 --kernel-trace, timeline summary
set -e
TAG=${1:-r04a}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
python3 $R/tools/single_frame_probe.py --json $O/probe.json > $O/probe.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/trace -o probe --output-format csv -- python3 $R/tools/single_frame_probe.py --iters 20 > $O/probe_prof.log 2>&1
python3 $R/profiles/summarize_timeline.py $O/trace/probe_kernel_trace.csv --first k_ingest --last k_pack_out > $O/timeline_detect.txt 2>&1 || true
python3 $R/profiles/summarize_timeline.py $O/trace/probe_kernel_trace.csv --first k_match_lds --last k_copy_out > $O/timeline_pair.txt 2>&1 || true
cp $O/trace/probe_kernel_stats.csv $O/probe_kernel_stats.csv || true
ls $O/trace
head -c 3000 $O/timeline_detect.txt; cat $O/timeline_pair.txt
