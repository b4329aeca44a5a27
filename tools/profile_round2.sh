#!/bin/bash
# Runs on the GPU box (via gpurun): the round's evidence — default bench line (with CPU baseline, parity, live traffic),
# rocprofv3 kernel stats of the same command, PMC passes (memory side and SQ side) for the headline workload and the
# 4-state workloads, every workload's line, the real host through the adapter, randomised stress.
# Usage: tools/profile_round2.sh r02
R=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_line.json 2> $OUT/bench.err
timeout 600 python bench.py --steps 200 --warmup 20 --pipelined --branch-cache --site-fits 4 > $OUT/bench.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --branch-cache --site-fits 4 > $OUT/stats.log 2>&1
for wl in mg94_64x10k gtr_32x50k gtr_32x1m busted3_64x10k; do
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 20 --warmup 6 --no-cpu-baseline --no-traffic > /dev/null 2>&1
  done
done
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_mg94_64x10k -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-traffic > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_mg94_128x100k -- python $GRAFT_REPO_ROOT/bench.py --workload mg94_128x100k --steps 10 --warmup 6 --no-cpu-baseline --no-traffic > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
for wl in mg94_64x10k mg94_32x5k busted3_64x10k mg94_128x100k mg94_64x5000 mg94_64x2500 mg94_64x1250 gtr_32x50k gtr_32x1m hky_8x1k; do
  steps=200; [ $wl = mg94_128x100k ] && steps=30
  HYPHY_HIP_VERBOSE=1 timeout 300 python bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline --no-traffic > $OUT/wl_$wl.json 2> $OUT/wl_$wl.err
  grep "schedule tuner" $OUT/wl_$wl.err | tail -1
  python - $wl $OUT/wl_$wl.json <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(path) if l.startswith("{")][-1]); r = j["roofline"]
    print(f"{tag:18s} {j['value']:9.1f} evals/s  step {j['ms_per_step']*1e3:8.1f} us  {r['kernel']} {r['kernel_ms']*1e3:8.1f} us  {r['achieved']:8.2f} {r['unit']}  frac {r['frac']:.3f}  expm {r['expm_ms']}  reduce {r['reduce_ms']}")
except Exception as e:
    print(f"{tag:18s} FAILED ({e})")
PY
done > $OUT/all_workloads.txt 2>&1
# the real host through the adapter: mode B with and without template mode, then mode A
(NO_REFERENCE=1 HYPHY_HIP_DEVICE_EXPM=always timeout 300 python tests/adapter_rate.py 40000 1; NO_REFERENCE=1 HYPHY_HIP_TEMPLATES=0 HYPHY_HIP_DEVICE_EXPM=always timeout 300 python tests/adapter_rate.py 12000 1; NO_REFERENCE=1 HYPHY_HIP_DEVICE_EXPM=0 timeout 300 python tests/adapter_rate.py 6000 16; timeout 300 python tests/adapter_rate.py 16 16 | grep reference) 2>/dev/null | grep '"host"' > $OUT/adapter_rate.jsonl
(HYPHY_HIP_POISON=1 timeout 300 python tests/stress_codon.py 80 15000 | tail -1; HYPHY_HIP_POISON=1 timeout 300 python tests/stress_generic.py 120 17000 | tail -1) > $OUT/stress.txt 2>&1
HYPHY_HIP_CHAIN_M=12 HYPHY_HIP_TIMELINE=$OUT/timeline_m12.txt timeout 200 python bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-traffic > /dev/null 2>&1
find $OUT -name "*.csv" | wc -l
