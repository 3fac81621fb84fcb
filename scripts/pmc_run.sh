#!/bin/bash
# rocprofv3 PMC passes (each counter set in its own run; no trace options combined) on the kernel micro-benchmark.
#   scripts/pmc_run.sh <outdir under gpurun_out> [kbench args...]
set -e
out=gpurun_out/$1; shift
export TMPDIR=/tmp
mkdir -p $out
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $out/p$i --output-format csv -- python3 scripts/kbench.py "$@" > $out/p$i.log 2>&1 || echo "pass $i failed"
done
python3 scripts/pmc_summary.py $out > $out/summary.csv
grep "qd_k_tile\|qd_k_candidates\|qd_k_gs_" $out/summary.csv
