set -e
cd /tmp
export TMPDIR=/tmp
O=/root/repo/gpurun_out
TRAIN_HIP_ONLY=1 rocprofv3 --kernel-trace --stats -d $O/prof_train_bf16 -o t -- python3 /root/repo/tools/bench_model_train.py 256 ckpt bf16 > $O/r2_train_prof.json 2> $O/r2_train_prof.err
python3 /root/repo/tools/rocpd_summary.py stats $O/prof_train_bf16/t_results.db $O/r2_train_bf16_kernel_stats.csv
head -25 $O/r2_train_bf16_kernel_stats.csv | cut -c1-150
