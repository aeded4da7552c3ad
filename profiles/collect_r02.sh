#!/bin/bash
# profiles/collect_r02.sh -- run ON the GPU box (gpurun): the rocprofv3 evidence behind bench.py's roofline object.
#   1. --kernel-trace --stats of the default bench command            -> r02_kernel_stats.csv, r02_bench_under_rocprof.json
#      and of the same command with the blur serialised               -> r02_kernel_stats_serial_blur.csv
#   2. one SQ counter pass (blur serialised: stand-alone kernels)     -> r02_pmc_sq.csv
#   3. FETCH_SIZE and WRITE_SIZE in SEPARATE passes (guide: TCC slots) -> r02_pmc_fetch.csv, r02_pmc_write.csv
#      (+ one FETCH_SIZE pass with whole-level blur for the counter calibration)
# then profiles/summarize_r02.py writes r02_pmc_per_kernel.json.  Copy gpurun_out/r02/* into profiles/ afterwards.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --prewarm-ms 50 --no-cpu-baseline --no-optin --no-extras"
rocprofv3 --kernel-trace --stats -d $O/stats -o r02 --output-format csv -- $B > $O/r02_bench_under_rocprof.json 2> $O/stats.err
cp $O/stats/r02_kernel_stats.csv $O/r02_kernel_stats.csv
# the same with the blur serialised: every kernel alone on the GPU - these averages are what bench.py's roofline.kernel_ms /
# stage_ms_standalone must agree with (in the default run k_fast and k_blur stretch one another)
VSLAM_AMD_SERIAL_BLUR=1 rocprofv3 --kernel-trace --stats -d $O/stats_serial -o r02s --output-format csv -- $B > /dev/null 2> $O/stats_serial.err
cp $O/stats_serial/r02s_kernel_stats.csv $O/r02_kernel_stats_serial_blur.csv
VSLAM_AMD_SERIAL_BLUR=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
  --kernel-trace -d $O/sq -o sq --output-format csv -- $B > /dev/null 2> $O/sq.err
VSLAM_AMD_SERIAL_BLUR=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o fetch --output-format csv -- $B > /dev/null 2> $O/fetch.err
VSLAM_AMD_SERIAL_BLUR=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o write --output-format csv -- $B > /dev/null 2> $O/write.err
# FETCH_SIZE calibration: with VSLAM_AMD_BLUR=full k_blur reads every level completely (243.3 MB per launch, far beyond L2), a byte
# count the counter can be checked against; the default build skips the level margins nothing reads
VSLAM_AMD_SERIAL_BLUR=1 VSLAM_AMD_BLUR=full rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch_full -o fetch --output-format csv -- $B > /dev/null 2> $O/fetch_full.err
python3 $R/profiles/summarize_r02.py $O
ls -la $O
