#!/bin/bash
# Runs on the GPU box (via gpurun): the round's evidence.  Usage: tools/profile_round3.sh [tag]
#   bench_driver_line.json   the driver's exact command (CPU baseline with thread sweep, parity, live PMC traffic, value_cold)
#   stats/                   rocprofv3 --kernel-trace --stats of the same command (no CPU legs)
#   pmc_<workload>/          separate --pmc passes (SQ sets for the MFMA workloads, FETCH_SIZE / WRITE_SIZE for the 4-state ones)
#   all_workloads.txt        one line per workload with the schedule tuner's report
#   stress.txt               randomised stress with poisoned allocations, fresh seeds
R=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_line.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $OUT/stats.log 2>&1
for wl in gtr_32x50k gtr_32x1m; do
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 20 --warmup 6 --no-cpu-baseline --no-traffic > /dev/null 2>&1
  done
done
for wl in mg94_64x10k mg94_64x1250; do
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 20 --warmup 6 --no-cpu-baseline --no-traffic > /dev/null 2>&1
  done
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
    fr = r["frac"] if r.get("frac") is not None else float("nan")
    extra = f"  VALU {r['valu_tflops']:.2f} TFLOP/s" if r.get("valu_tflops") is not None else ""
    print(f"{tag:18s} {j['value']:9.1f} evals/s  step {j['ms_per_step']*1e3:8.1f} us  {r['kernel']} {r['kernel_ms']*1e3:8.1f} us  {r['achieved']:8.2f} {r['unit']}  frac {fr:.3f}{extra}  expm {r['expm_ms']}  reduce {r['reduce_ms']}")
except Exception as e:
    print(f"{tag:18s} FAILED ({e})")
PY
done > $OUT/all_workloads.txt 2>&1
(HYPHY_HIP_POISON=1 timeout 300 python tests/stress_codon.py 60 31000 | tail -1; HYPHY_HIP_POISON=1 timeout 300 python tests/stress_generic.py 90 33000 | tail -1) > $OUT/stress.txt 2>&1
python bench.py --steps 100 --warmup 10 --collective cabi --no-cpu-baseline --no-traffic 2>/dev/null | grep "^{" > $OUT/bench_one_rank_communicator.json
find $OUT -name "*.csv" | wc -l
