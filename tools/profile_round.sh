#!/bin/bash
# Runs on the GPU box (via gpurun): bench line + rocprofv3 kernel stats + PMC passes for the round.
# Usage: tools/profile_round.sh rNN
R=${1:-r01}
OUT=/root/repo/gpurun_out/$R
mkdir -p $OUT
cd /root/repo
HYPHY_HIP_ALL_TIMINGS=1 python bench.py --steps 200 --warmup 20 --pipelined > $OUT/bench_alltimings.json 2>/dev/null
python bench.py --steps 200 --warmup 20 --pipelined > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
HYPHY_HIP_ALL_TIMINGS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python /root/repo/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/stats.log 2>&1
# counters: own runs, kernel-trace only (guide: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2 -> separate passes)
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc -- python /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
done
cd /root/repo
./tools/ubench_mfma_f64 > $OUT/ubench_mfma_f64.txt 2>&1
python tools/cpu_thread_sweep.py 1 8 16 32 64 > $OUT/cpu_thread_sweep.jsonl 2>/dev/null
for A in 63 62 1 16 2 4; do echo -n "ablate=$A "; HYPHY_HIP_ABLATE=$A python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['roofline']['kernel_ms']*1000,1), 'us')"; done > $OUT/ablation.txt
find $OUT -name "*.csv" | head -30
