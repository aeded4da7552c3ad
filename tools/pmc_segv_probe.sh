#!/bin/bash
# tools/pmc_segv_probe.sh -- which of the two workarounds of profiles/collect.sh keeps `rocprofv3 --pmc` from the host SIGSEGV inside torch's
# index kernels (round 3, gpurun_out/r03a/sq.err)?  A: kernel filter only (frames generated under the profiler), B: frame cache only (no
# filter: counters collected for every kernel of the process, torch's included, but the generator does not run under the profiler).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --prewarm-ms 20 --no-cpu-baseline --no-optin --no-extras"
rocprofv3 --pmc SQ_INSTS_VALU --kernel-include-regex '^(void\s)?k_[a-z0-9_]+' --kernel-trace -d $O/a -o a --output-format csv -- $B > $O/a.out 2> $O/a.err
echo "A (filter only) exit $?" | tee $O/result.txt
$B --frames-cache /tmp/pmc_probe_frames > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace -d $O/b -o b --output-format csv -- $B --frames-cache /tmp/pmc_probe_frames > $O/b.out 2> $O/b.err
echo "B (cache only) exit $?" | tee -a $O/result.txt
grep -l "SIGSEGV" $O/a.err $O/b.err | tee -a $O/result.txt
grep -h -A3 "SIGSEGV" $O/a.err $O/b.err | head -12 | tee -a $O/result.txt
grep -h "index_kernel\|index_elementwise" $O/a.err $O/b.err | head -4 | cut -c1-200 | tee -a $O/result.txt
rm -rf $O/a $O/b
# C: neither workaround (the round-3 failure): counters on every kernel, generator under the profiler
rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace -d $O/c -o c --output-format csv -- $B > $O/c.out 2> $O/c.err
echo "C (neither, 1 counter) exit $?" | tee -a $O/result.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d $O/d -o d --output-format csv -- $B > $O/d.out 2> $O/d.err
echo "D (neither, the 8 SQ counters of collect.sh) exit $?" | tee -a $O/result.txt
grep -h -m1 -A2 "SIGSEGV" $O/c.err $O/d.err | tee -a $O/result.txt
grep -h -m2 "index_kernel\|index_elementwise" $O/c.err $O/d.err | cut -c1-160 | tee -a $O/result.txt
rm -rf $O/c $O/d
