set -e
cd /root/repo
export TMPDIR=/tmp
O=$PWD/gpurun_out
python tools/bench_k1_widths.py > $O/r2_k1_widths.json 2>$O/r2_k1_widths.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof_k1 -o k1 -- python3 /root/repo/bench.py --no-cpu-baseline --steps 50 --warmup 10 > $O/r2_bench_under_rocprof.json 2> $O/r2_prof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/prof_fetch -o fetch -- python3 /root/repo/bench.py --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>> $O/r2_prof.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/prof_write -o write -- python3 /root/repo/bench.py --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>> $O/r2_prof.err
find $O/prof_k1 $O/prof_fetch $O/prof_write -type f | head -30
