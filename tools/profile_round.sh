#!/bin/bash
# Runs on the GPU box (via gpurun): bench line + rocprofv3 kernel stats + PMC passes for the round.
# Usage: tools/profile_round.sh rNN
R=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
HYPHY_HIP_ALL_TIMINGS=1 python bench.py --steps 200 --warmup 20 --pipelined --no-cpu-baseline > $OUT/bench_alltimings.json 2>/dev/null
python bench.py --steps 200 --warmup 20 --pipelined --branch-cache --site-fits 4 --fel > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
HYPHY_HIP_ALL_TIMINGS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --branch-cache --site-fits 4 > $OUT/stats.log 2>&1
# counters: own runs, kernel-trace only (guide: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2 -> separate passes)
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --site-fits 2 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
./tools/ubench_mfma_f64 > $OUT/ubench_mfma_f64.txt 2>&1
bash tools/sweep_small_shards.sh > $OUT/kernel_choice_by_shard_size.txt 2>&1
bash tools/bench_all_workloads.sh > $OUT/all_workloads.txt 2>&1
# the real host through the adapter: mode B (device exponentials), then mode A
(HYPHY_HIP_DEVICE_EXPM=always timeout 200 python tests/adapter_rate.py 12000 1,16; HYPHY_HIP_DEVICE_EXPM=0 timeout 200 python tests/adapter_rate.py 4000 1,16) 2>/dev/null | grep '"host"' > $OUT/adapter_rate.jsonl
# randomised stress runs, seeds other than the test-suite's
(HYPHY_HIP_POISON=1 timeout 300 python tests/stress_codon.py 80 5000 | tail -1; HYPHY_HIP_POISON=1 timeout 300 python tests/stress_generic.py 120 7000 | tail -1) > $OUT/stress.txt 2>&1
find $OUT -name "*.csv" | head -30
