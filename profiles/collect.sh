#!/bin/bash
# profiles/collect.sh <tag> [extra bench flags] -- run ON the GPU box (gpurun): the rocprofv3 evidence behind bench.py's roofline object.
#   1. --kernel-trace --stats of the bench command                      -> <tag>_kernel_stats.csv, <tag>_bench_under_rocprof.json
#   2. one SQ counter pass                                               -> <tag>_pmc_sq.csv
#   3. FETCH_SIZE and WRITE_SIZE in SEPARATE passes (guide: TCC slots)   -> <tag>_pmc_fetch.csv, <tag>_pmc_write.csv
# then profiles/summarize.py writes <tag>_pmc_per_kernel.json.  Copy gpurun_out/<tag>/<tag>_* into profiles/ afterwards.
# (Every kernel of the default build runs in line on one stream, so the --stats averages are stand-alone durations.)
set -e
TAG=${1:-r04}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --prewarm-ms 50 --no-cpu-baseline --no-optin --no-extras --frames-cache /tmp/bench_frames $*"
# Counters are collected for this repo's kernels only (--kernel-include-regex).  That filter is also what keeps `rocprofv3 --pmc` alive:
# with the 8 SQ counters collected on EVERY dispatch, the frame generator's torch kernels (thousands of tiny launches, index kernels with
# multi-KB by-value arguments) brought the profiler down in round 3 (host SIGSEGV in its dispatch interception under
# at::native::index_kernel) and to a crawl in round 4 (tools/pmc_segv_probe.sh: filter alone = fine, cache alone = fine, neither with one
# counter = fine, neither with the 8 SQ counters = no output for 7 minutes; profiles/README.md).  The frame cache below is only a
# speed-up: the frames are generated once for the four passes.
INC="--kernel-include-regex ^(void\s)?k_[a-z0-9_]+"
rocprofv3 --kernel-trace --stats -d $O/stats -o $TAG --output-format csv -- $B > $O/${TAG}_bench_under_rocprof.json 2> $O/stats.err
cp $O/stats/${TAG}_kernel_stats.csv $O/${TAG}_kernel_stats.csv
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
  $INC --kernel-trace -d $O/sq -o sq --output-format csv -- $B > /dev/null 2> $O/sq.err
rocprofv3 --pmc FETCH_SIZE $INC --kernel-trace -d $O/fetch -o fetch --output-format csv -- $B > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE $INC --kernel-trace -d $O/write -o write --output-format csv -- $B > /dev/null 2> $O/write.err
python3 $R/profiles/summarize.py $O $TAG "$B"
rm -rf $O/stats $O/sq $O/fetch $O/write
ls -la $O
