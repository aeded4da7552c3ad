#!/bin/bash
# tools/ab_lib.sh name1 name2 ... -- interleaved bench.py runs (3 rounds) of visual-slam_amd/variants/lib<name>.so ("cur" = the in-tree build)
for round in 1 2 3; do
  for v in "$@"; do
    lib=visual-slam_amd/variants/lib$v.so; [ "$v" = cur ] && lib=visual-slam_amd/libvslam_amd.so
    VSLAM_AMD_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-optin --no-extras --frames-cache /tmp/bench_frames 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-10s' % '$v', d['ms_per_step'], ' '.join('%s %.4f' % (k, x) for k, x in d['stage_ms'].items()))"
  done
done
