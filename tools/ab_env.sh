#!/bin/bash
# tools/ab_env.sh VAR val_a val_b [rounds]: interleaved bench.py runs with an environment knob at two values (same box, same build);
# prints ms_per_step of every run.  Used for run-time switches read at context creation (VSLAM_AMD_*).
var=$1; a=$2; b=$3; rounds=${4:-3}
for r in $(seq $rounds); do
  for v in $a $b; do
    env $var=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-optin --no-extras | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$v', d['ms_per_step'], {k: round(x, 3) for k, x in d['stage_ms'].items()})"
  done
done
