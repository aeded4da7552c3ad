set -e
python -m pytest tests/test_gpu_twoview.py tests/test_gpu_track.py tests/test_gpu_fundamental.py tests/test_gpu_dropin.py tests/test_gpu_multirank.py tests/test_gpu_geometry_api.py -x -q > gpurun_out/tv1.log 2>&1 || { tail -30 gpurun_out/tv1.log; exit 1; }
tail -2 gpurun_out/tv1.log
for n in 1 2 3 4; do
  VSLAM_AMD_TV_FIRST=$n VSLAM_AMD_SERIAL_BLUR=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-optin --no-extras | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('first=$n', d['ms_per_step'], d['stage_ms']['two_view'])"
done
VSLAM_AMD_LIB=visual-slam_amd/variants/libprev.so VSLAM_AMD_SERIAL_BLUR=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-optin --no-extras | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('prev', d['ms_per_step'], d['stage_ms']['two_view'])"
