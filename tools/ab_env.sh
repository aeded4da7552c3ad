#!/bin/bash
# tools/ab_env.sh VAR val1 val2 ... -- interleaved bench.py runs on one box with VAR set to each value in turn (3 rounds):
# ms_per_step and the per-stage spans of every run.  Usage on the GPU box: bash tools/ab_env.sh VSLAM_AMD_MATCHER lds mfma
VAR=$1; shift
for round in 1 2 3; do
  for v in "$@"; do
    env $VAR=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-optin --no-extras --frames-cache /tmp/bench_frames 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$VAR=$v', d['ms_per_step'], ' '.join('%s %.4f' % (k, x) for k, x in d['stage_ms'].items()))"
  done
done
