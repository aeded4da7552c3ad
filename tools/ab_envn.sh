#!/bin/bash
# tools/ab_envn.sh VAR rounds val...: like ab_env.sh for any number of values
var=$1; rounds=$2; shift 2
for r in $(seq $rounds); do
  for v in "$@"; do
    env $var=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-optin --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$v', d['ms_per_step'], {k: round(x, 3) for k, x in d['stage_ms'].items()})"
  done
done
