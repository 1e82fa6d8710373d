#!/bin/bash
# Round 3: the profiler evidence behind the bench line.  rocprofv3 --kernel-trace --stats of the SAME command as the
# bench line's headline workload (weak mode, N = 1), then FETCH_SIZE / WRITE_SIZE in separate --pmc passes (never with
# trace domains other than --kernel-trace).  Summaries -> gpurun_out/r3p/ (copied into profiles/ by hand).
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r3p
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof_k1 -o k1 -- python3 $R/bench.py --scaling weak --no-cpu-baseline --steps 50 --warmup 10 > $O/r03_k1_bench_under_rocprof.json 2> $O/prof.err
python3 $R/tools/rocpd_summary.py stats $(find $O/prof_k1 -name '*.db' | head -1) $O/r03_k1_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/prof_fetch -o fetch -- python3 $R/bench.py --scaling weak --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>> $O/prof.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/prof_write -o write -- python3 $R/bench.py --scaling weak --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>> $O/prof.err
python3 $R/tools/rocpd_summary.py pmc $(find $O/prof_fetch -name '*.db' | head -1) $O/r03_k1_pmc_fetch.csv
python3 $R/tools/rocpd_summary.py pmc $(find $O/prof_write -name '*.db' | head -1) $O/r03_k1_pmc_write.csv
grep -h k_seg_reduce $O/r03_k1_bench_kernel_stats.csv $O/r03_k1_pmc_fetch.csv $O/r03_k1_pmc_write.csv | cut -c1-200
rm -rf $O/prof_k1 $O/prof_fetch $O/prof_write
