#!/bin/bash
# GPU box: experimental instantiations of prune_wave_kernel (HYPHY_HIP_WAVE_VARIANT) against the production one
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/abv; mkdir -p $OUT
for v in ${VARIANTS:-1 2}; do
  sl=3; [ $v -ge 2 ] && sl=2
  echo "== stress, variant $v"
  HYPHY_HIP_WAVE_VARIANT=$v STRESS_SLOTS=$sl HYPHY_HIP_POISON=1 STRESS_KERNEL=1 timeout 300 python tests/stress_codon.py 12 9800 2>&1 | tail -1
done
one() { tag=$1; wl=$2; m=$3; steps=$4; shift 4
  env "$@" HYPHY_HIP_CHAIN_M=$m timeout 300 python bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline > $OUT/$tag.json 2> $OUT/$tag.err
  python - $tag $OUT/$tag.json <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(path) if l.startswith("{")][-1]); r = j["roofline"]
    print(f"{tag:40s} step {j['ms_per_step']*1e3:8.1f} us  prune {r['kernel_ms']*1e3:8.1f} us  frac {r['frac']:.3f}")
except Exception as e:
    print(f"{tag:40s} FAILED ({e})")
PY
}
for rep in 1 2; do
  one v0_head_m12_r$rep mg94_64x10k 12 200 HYPHY_HIP_WAVE_VARIANT=0
  one v0np_head_m12_r$rep mg94_64x10k 12 200 HYPHY_HIP_WAVE_VARIANT=0 HYPHY_HIP_SLOTS=2
  for v in ${VARIANTS:-1 2}; do
    sl=3; [ $v -ge 2 ] && sl=2
    for m in ${MS:-12 8}; do one v${v}_head_m${m}_r$rep mg94_64x10k $m 200 HYPHY_HIP_WAVE_VARIANT=$v HYPHY_HIP_SLOTS=$sl; done
  done
done
one v0_big_m40 mg94_128x100k 40 30 HYPHY_HIP_WAVE_VARIANT=0
for v in ${VARIANTS:-1 2}; do sl=3; [ $v -ge 2 ] && sl=2; one v${v}_big_m40 mg94_128x100k 40 30 HYPHY_HIP_WAVE_VARIANT=$v HYPHY_HIP_SLOTS=$sl; one v${v}_1250_m5 mg94_64x1250 5 200 HYPHY_HIP_WAVE_VARIANT=$v HYPHY_HIP_SLOTS=$sl; done
one v0_1250_m5 mg94_64x1250 5 200 HYPHY_HIP_WAVE_VARIANT=0
