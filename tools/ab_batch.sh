#!/bin/bash
# tools/ab_batch.sh b1 b2 ...: bench.py --batch b for each value (frames/s and ms per frame), e.g. 512 vs 513 (a rank > 0 of a
# sharded run extracts one halo frame more than a multiple of 8)
for b in "$@"; do
  python bench.py --batch $b --steps 20 --warmup 4 --no-cpu-baseline --no-optin --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch $b', d['value'], 'frames/s', d['ms_per_step'], 'ms/step', round(d['ms_per_step'] / $b * 1e3, 3), 'us/frame', {k: round(x, 3) for k, x in d['stage_ms'].items()})"
done
