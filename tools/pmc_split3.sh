#!/bin/bash
# Round 3: what the split-bf16 fp32 edge MLP (k_mlp_f32_split3, latent 256, M = 2M) does with its cycles.
# One counter group per rocprofv3 pass, --kernel-trace only (never with other trace domains).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_split3
mkdir -p $OUT
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace -d $OUT/p$i -o p$i -- python3 $R/tools/run_mlp_split3_once.py ${1:-256} > $OUT/log$i.txt 2>&1 || echo "pass $i ($grp) failed" >> $OUT/errors.txt
  db=$(find $OUT/p$i -name '*.db' | head -1)
  [ -n "$db" ] && python3 $R/tools/rocpd_summary.py pmc $db $OUT/p$i.csv
done
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/stats -o st -- python3 $R/tools/run_mlp_split3_once.py ${1:-256} > $OUT/log_stats.txt 2>&1
python3 $R/tools/rocpd_summary.py stats $(find $OUT/stats -name '*.db' | head -1) $OUT/stats.csv
grep -h "k_mlp_f32_split3\|k_linear_f32_split3" $OUT/p*.csv $OUT/stats.csv | cut -d, -f2- | cut -c1-160 > $OUT/summary.txt
cat $OUT/summary.txt
rm -rf $OUT/p[0-9] $OUT/stats
