set -e
cd /root/repo
export TMPDIR=/tmp
O=$PWD/gpurun_out
L=${1:-256}
cd /tmp
rm -rf $O/prof_pre
rocprofv3 --kernel-trace --stats -d $O/prof_pre -o pre -- python3 /root/repo/tools/bench_preproject_bf16.py $L > $O/pre_under_rocprof.json 2> $O/pre_prof.err
DB=$(find $O/prof_pre -name "*.db" | head -1)
python3 /root/repo/tools/rocpd_summary.py stats $DB $O/pre_kernel_stats.csv
head -14 $O/pre_kernel_stats.csv | cut -c1-160
