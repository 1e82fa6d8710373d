# SQ / LDS counters of the bf16 feature-split edge MLP (one launch shape per run): bash tools/pmc_split.sh L
set -e
L=${1:-256}
cd /tmp
export TMPDIR=/tmp
O=/root/repo/gpurun_out
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace -d $O/pmc_split_$T -o p -- python3 /root/repo/tools/time_split.py $L > /dev/null 2>> $O/pmc_split.err || echo "pmc pass $T failed"
  python3 /root/repo/tools/rocpd_summary.py pmc $O/pmc_split_$T/p_results.db $O/pmc_split_$T.csv || true
  grep k_mlp_bf16_split $O/pmc_split_$T.csv | cut -d'"' -f3 || true
done
