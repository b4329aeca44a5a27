#!/bin/bash
# Runs on the GPU box (via gpurun): the round's evidence.  Usage: tools/profile_round.sh <tag> [parts]   (parts: default "all";
# any of: line workloads stats pmc nuc adapter phases ubench)
#   bench_driver_line.json   the driver's exact command (CPU baseline with thread sweep, parity, live PMC traffic, value_cold)
#   all_workloads.txt        one line per workload with the schedule tuner's report
#   stats/                   rocprofv3 --kernel-trace --stats, tuner off: subtree repeats on and the trunk's production cut forced
#                            (steady-state averages)
#   pmc_<workload>/          separate --pmc passes, production cut forced: SQ sets (wave cycles, instruction mix, MFMA pipe),
#                            FETCH_SIZE, WRITE_SIZE
#   adapter_rate.jsonl       evaluations per second through the real host (tools/adapter_rate.py)
#   phases_*.txt             where a wave's cycles go (trace build of the wave kernel, tools/timeline_waves.py)
#   ubench_*.txt             instruction / edge-product / edge+leaf microbenchmarks (tools/ubench)
R=${1:-r06}; PARTS=${2:-all}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
has() { [ "$PARTS" = all ] || echo "$PARTS" | grep -qw "$1"; }
cd $GRAFT_REPO_ROOT
if has line; then timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_line.json 2> $OUT/bench.err; fi
if has workloads; then
for wl in mg94_64x10k mg94_32x5k busted3_64x10k mix3_64x10k mg94_128x100k mg94_64x5000 mg94_64x2500 mg94_64x1250 gtr_32x50k gtr_32x1m hky_8x1k; do
  steps=200; [ $wl = mg94_128x100k ] && steps=30
  HYPHY_HIP_VERBOSE=1 timeout 300 python bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline --no-traffic > $OUT/wl_$wl.json 2> $OUT/wl_$wl.err
  grep "schedule tuner" $OUT/wl_$wl.err | tail -1
  python - $wl $OUT/wl_$wl.json <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(path) if l.startswith("{")][-1]); r = j["roofline"]
    if r.get("bound") == "hbm":
        # 4 states: the algorithmic bytes (every conditional vector through HBM) are not what the memory system moves — the kernel keeps
        # them on chip — so the row reports COUNTER bytes / time against the achievable HBM rate (live PMC passes, else the committed
        # profiles/pmc_traffic.json entry of this workload and kernel), never an algorithmic figure above the peak
        if r.get("traffic_rate_gbs") is not None:
            roof = f"{r['traffic_rate_gbs']:8.1f} GB/s by counters ({r['traffic'] / 1e6:.1f} MB per launch) = {r['traffic_frac_of_achievable']:.3f} of the achievable HBM rate"
        else:
            roof = "counter traffic not available"
        roof += f"  VALU {r['valu_tflops']:.2f} TFLOP/s"
    else:
        roof = f"{r['achieved']:8.2f} {r['unit']}  frac {r['frac']:.3f}"
        if r.get("repeat_ratio") is not None and r["repeat_ratio"] < 1.0:
            roof += f" (executed flops; repeat ratio {r['repeat_ratio']:.3f}, effective {r['effective_tflops']:.1f} TFLOP/s)"
    print(f"{tag:18s} {j['value']:9.1f} evals/s  step {j['ms_per_step']*1e3:8.1f} us  {r['kernel']} {r['kernel_ms']*1e3:8.1f} us  {roof}  expm {r['expm_ms']}  reduce {r['reduce_ms']}")
except Exception as e:
    print(f"{tag:18s} FAILED ({e})")
PY
done > $OUT/all_workloads.txt 2>&1
fi
cd /tmp && export TMPDIR=/tmp
if has stats; then
  HYPHY_HIP_REPEATS=1 HYPHY_HIP_TRUNK_WALK=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-traffic > $OUT/stats.log 2>&1
  # 4 states (r06): the run-time generated kernel (nucgen_kernel) in steady state; bench.py waits for it before anything is timed
  for wl in gtr_32x1m gtr_32x50k; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-traffic > $OUT/stats_$wl.log 2>&1
  done
fi
pmc() { wl=$1; shift
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    env "$@" timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 20 --warmup 6 --no-cpu-baseline --no-traffic > /dev/null 2>&1
  done
}
if has pmc; then
  # (the forms the production runs settle on, forced so that the tuner's choice cannot differ between the counter passes)
  pmc mg94_64x10k HYPHY_HIP_REPEATS=1 HYPHY_HIP_TRUNK_WALK=1
  pmc mg94_128x100k HYPHY_HIP_REPEATS=1 HYPHY_HIP_TRUNK_WALK=1
fi
if has nuc; then
  pmc gtr_32x1m X=1
  pmc gtr_32x50k X=1
fi
cd $GRAFT_REPO_ROOT
if has adapter; then timeout 1500 python tools/adapter_rate.py headline,class2,cat3,mix3,manylf 10000 400 2>/dev/null > $OUT/adapter_rate.jsonl; fi
if has phases; then
  # r06: the kernels the headline RUNS — the trunk of the class-compressed form as a row-split walk per tile (trunk_walk_kernel, trace
  # build: one record per workgroup), the lower phase (class_table_team_kernel, the same record) — then the wave-per-tile trunk it
  # replaced (prune_wave_kernel<.., REP>) and the plain form for reference
  for spec in "mg94_64x10k 8 624" "mg94_128x100k 16 6250"; do
    set -- $spec
    HYPHY_HIP_REPEATS=1 HYPHY_HIP_TRUNK_WALK=1 HYPHY_HIP_WALK_TIMELINE=$OUT/tlw_$1.txt timeout 300 python bench.py --workload $1 --steps 3 --warmup 3 --no-cpu-baseline --no-traffic > /dev/null 2>&1
    (echo "# $1, TRUNK of the class-compressed form: trunk_walk_kernel, trace build (HYPHY_HIP_WALK_TIMELINE; one record per workgroup = (tile, chain); 'index maps' = the tile's leaf table into LDS, 'publish' = the hand-over at the root and the root epilogue)"; python tools/rep_team_timeline.py $OUT/tlw_$1.txt) > $OUT/phases_a_trunk_walk_$1.txt
    rm -f $OUT/tlw_$1.txt
    HYPHY_HIP_REPEATS=1 HYPHY_HIP_REP_TIMELINE=$OUT/tlr_$1.txt timeout 300 python bench.py --workload $1 --steps 3 --warmup 3 --no-cpu-baseline --no-traffic > /dev/null 2>&1
    (echo "# $1, LOWER PHASE of the class-compressed form: class_table_team_kernel, trace build (HYPHY_HIP_REP_TIMELINE; the last level's launch when the pass has several)"; python tools/rep_team_timeline.py $OUT/tlr_$1.txt) > $OUT/phases_b_lower_$1.txt
    rm -f $OUT/tlr_$1.txt
    HYPHY_HIP_REPEATS=1 HYPHY_HIP_TRUNK_WALK=0 HYPHY_HIP_CHAIN_M=$2 HYPHY_HIP_TIMELINE=$OUT/tl_$1.txt timeout 300 python bench.py --workload $1 --steps 3 --warmup 3 --no-cpu-baseline --no-traffic > /dev/null 2>&1
    (echo "# $1, the trunk under the wave-per-tile kernel it replaced (HYPHY_HIP_TRUNK_WALK=0), chain cut m = $2, trace build of prune_wave_kernel<.., REP> (HYPHY_HIP_TIMELINE; every stamp is an s_memtime + lgkmcnt(0): ~10-25 % slower than production)"; python tools/timeline_waves.py $OUT/tl_$1.txt $3) > $OUT/phases_c_trunk_wave_$1.txt
    rm -f $OUT/tl_$1.txt
  done
  HYPHY_HIP_CHAIN_M=12 HYPHY_HIP_TIMELINE=$OUT/tl_plain.txt timeout 300 python bench.py --workload mg94_64x10k --steps 3 --warmup 3 --no-cpu-baseline --no-traffic > /dev/null 2>&1
  (echo "# mg94_64x10k, PLAIN form (every node at every pattern; what r04 ran), chain cut m = 12, trace build of prune_wave_kernel"; python tools/timeline_waves.py $OUT/tl_plain.txt 624) > $OUT/phases_d_plain_mg94_64x10k.txt
  rm -f $OUT/tl_plain.txt
fi
if has ubench; then
  (cd tools/ubench && OLD_ONLY=1 ./mfma4_skew > $OUT/ubench_agpr_vs_vgpr.txt 2>&1; ./mfma4_skew > $OUT/ubench_edge_product.txt 2>&1; ./overlap_probe > $OUT/ubench_edge_plus_leaf.txt 2>&1)
fi
find $OUT -name "*.csv" | wc -l
# the raw counter / trace files are tens of MiB: summarise HERE and keep the summaries only (gpurun_out/ travels back up to 64 MiB)
python tools/summarize_profiles.py $OUT $R $OUT/summary > $OUT/summary.log 2>&1
rm -rf $OUT/pmc_* $OUT/stats $OUT/stats_gtr_32x1m $OUT/stats_gtr_32x50k
cat $OUT/summary.log | tail -12
