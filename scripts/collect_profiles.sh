#!/bin/bash
# Everything the round's profiles/ entries come from, in one GPU call:
#   scripts/collect_profiles.sh <tag>      (writes gpurun_out/<tag>/...)
# bench lines (headline, stochastic stages on, config 2, config 1, mixed-N config 5, 2-rank rehearsal on one GPU),
# rocprofv3 --kernel-trace --stats of the headline bench command, and the PMC passes on the kernel micro-benchmark.
tag=${1:-r03}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python bench.py --steps 10 --warmup 3 > $out/bench_headline.json 2> $out/bench_headline.err || echo "headline failed"
python bench.py --steps 10 --warmup 3 --noise --no-cpu-baseline > $out/bench_headline_noise_on.json 2> /dev/null || echo "noise failed"
python bench.py --config config2 --steps 30 --warmup 5 --cpu-seconds 5 > $out/bench_config2_4dot_256env.json 2> /dev/null || echo "config2 failed"
python bench.py --config config1 --steps 200 --warmup 20 --cpu-seconds 3 > $out/bench_config1_2dot_1env.json 2> /dev/null || echo "config1 failed"
python bench.py --config mixed --envs 1024 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_config5_mixed_1024env.json 2> /dev/null || echo "mixed failed"
python bench.py --gpus 2 --share-gpu --backend gloo --envs 1024 --steps 6 --warmup 2 > $out/bench_2ranks_rehearsal_one_gpu.json 2> /dev/null || echo "2-rank failed"
rocprofv3 --kernel-trace --stats -d $out/kstats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/kstats.log 2>&1 || echo "kstats failed"
cp $out/kstats/*/*kernel_stats.csv $out/kernel_stats_bench_headline.csv 2>/dev/null
# PMC passes on ONE launch chunk of the bench (CHUNK envs, 8-dot 64x64) in the bench's regime: envs reset, then stepped with
# uniform random actions for e % 12 + 1 steps (kbench mode wild12); the last dispatch of every kernel is the timed launch
CHUNK=${CHUNK:-180}
mkdir -p $out/pmc; i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $out/pmc/p$i --output-format csv -- python3 scripts/kbench.py --envs $CHUNK --modes wild12 --iters 1 > $out/pmc/p$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 scripts/kbench.py --envs $CHUNK --modes wild12,start,near,mid --iters 3 --each > $out/kbench_${CHUNK}env.log 2>&1
python3 scripts/pmc_summary.py $out/pmc --last 1 > $out/pmc_summary.csv
rm -rf $out/kstats $out/pmc/p*/
ls $out
