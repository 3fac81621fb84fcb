#!/bin/bash
# one PMC pass (instruction counts + wave cycles) per library variant:  scripts/pmc_quick.sh <tag> <lib.so> [kbench args]
out=gpurun_out/pmcq_$1; lib=$2; shift; shift
export TMPDIR=/tmp; mkdir -p $out
QDSIM_LIB=$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS -d $out/p1 --output-format csv -- python3 scripts/kbench.py "$@" > $out/p1.log 2>&1
python3 scripts/pmc_summary.py $out | grep "qd_k_tile" | sed "s/^/$out /"
