# rocprofv3 per-kernel summary of the EC-IN fp32 training step (default fp32 MLP path): bash tools/prof_train_fp32.sh [L]
set -e
cd /tmp
export TMPDIR=/tmp
O=/root/repo/gpurun_out
L=${1:-256}
rm -rf $O/prof_train_fp32
TRAIN_HIP_ONLY=1 rocprofv3 --kernel-trace --stats -d $O/prof_train_fp32 -o t -- python3 /root/repo/tools/bench_model_train.py $L ckpt > $O/train_fp32_prof.json 2> $O/train_fp32_prof.err
DB=$(find $O/prof_train_fp32 -name "*.db" | head -1)
python3 /root/repo/tools/rocpd_summary.py stats $DB $O/train_fp32_kernel_stats_L$L.csv
head -22 $O/train_fp32_kernel_stats_L$L.csv | cut -c1-170
