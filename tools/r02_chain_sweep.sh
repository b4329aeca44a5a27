#!/bin/bash
# GPU box: chain schedules of the wave-per-tile kernel vs the level-peeled fragments, by shard size and source size m
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_chain
mkdir -p $OUT
line() { python - "$1" "$2" <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(path) if l.startswith("{")][-1])
    r = j["roofline"]
    print(f"{tag:44s} {j['value']:9.1f} evals/s  step {j['ms_per_step']*1e3:8.1f} us  prune {r['kernel_ms']*1e3:8.1f} us  {r['achieved']:6.2f} {r['unit']}  frac {r['frac']:.3f}  {r['kernel']} x{r['launches_per_step']}  logL {j['logl_last']!r}")
except Exception as e:
    print(f"{tag:44s} FAILED ({e})")
PY
}
run() { # tag workload env...
  tag=$1; wl=$2; shift 2
  env "$@" timeout 300 python bench.py --workload $wl --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline > $OUT/$tag.json 2> $OUT/$tag.err
  line "$tag" $OUT/$tag.json
}
echo "== correctness first: randomised stress against the oracle, wave kernel, chain schedules"
HYPHY_HIP_POISON=1 STRESS_KERNEL=1 timeout 600 python tests/stress_codon.py 16 9300 2>&1 | tail -4
HYPHY_HIP_POISON=1 STRESS_KERNEL=1 timeout 600 python tests/stress_generic.py 16 9400 2>&1 | tail -3
echo "== headline mg94_64x10k"
run head_levels mg94_64x10k HYPHY_HIP_CUT=levels
for m in 4 6 8 10 12 16; do run head_chain_m$m mg94_64x10k HYPHY_HIP_CHAIN_M=$m; done
run head_chain_auto mg94_64x10k HYPHY_HIP_VERBOSE=1
echo "== shard sizes (one rank's share at 2 / 4 / 8 GPUs)"
for wl in mg94_64x5000 mg94_64x2500 mg94_64x1250 mg94_32x5k; do
  run ${wl}_wgkernel $wl HYPHY_HIP_KERNEL=0
  for m in 3 5 8 12 16; do run ${wl}_wave_chain_m$m $wl HYPHY_HIP_KERNEL=1 HYPHY_HIP_CHAIN_M=$m; done
done
echo "== busted3 (3 classes batched)"
run busted3_levels busted3_64x10k HYPHY_HIP_CUT=levels
for m in 8 12 16 24 62; do run busted3_chain_m$m busted3_64x10k HYPHY_HIP_CHAIN_M=$m; done
echo "== 128 taxa x 100k codons"
STEPS=20 run big_levels mg94_128x100k HYPHY_HIP_CUT=levels
for m in 8 16 24 40; do STEPS=20 run big_chain_m$m mg94_128x100k HYPHY_HIP_CHAIN_M=$m; done
