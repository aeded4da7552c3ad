#!/bin/bash
# round-4 evidence in one GPU call: full GPU suite, the un-profiled bench line, rocprofv3 stats + counter passes, single-frame kernel timeline, stream rates
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -3 $O/tests.log
python bench.py --steps 20 --warmup 3 > $O/r04_bench_full.json 2> $O/bench.err; tail -2 $O/bench.err
echo bench done
bash profiles/collect.sh r04 > $O/collect.log 2>&1; tail -3 $O/collect.log
bash tools/probe_run.sh r04_probe > $O/probe_run.log 2>&1; tail -20 $O/probe_run.log
python tools/stream_probe.py 2>&1 | grep chunk | tee $O/r04_stream_probe.txt
python tools/stream_rate.py 2>&1 | tail -3 | tee $O/r04_stream_rate.txt
