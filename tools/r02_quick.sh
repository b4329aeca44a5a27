#!/bin/bash
# GPU box: quick correctness + timing check of the default configuration (schedule tuner on)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_quick
mkdir -p $OUT
HYPHY_HIP_POISON=1 STRESS_KERNEL=1 timeout 600 python tests/stress_codon.py ${NSTRESS:-16} ${SEED:-9500} 2>&1 | tail -2
HYPHY_HIP_POISON=1 STRESS_KERNEL=1 timeout 600 python tests/stress_generic.py ${NSTRESS:-16} ${SEED:-9600} 2>&1 | tail -2
for wl in ${WLS:-mg94_64x10k mg94_64x5000 mg94_64x1250 mg94_32x5k busted3_64x10k mg94_128x100k}; do
  steps=100; [ $wl = mg94_128x100k ] && steps=20
  HYPHY_HIP_VERBOSE=1 timeout 300 python bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline > $OUT/$wl.json 2> $OUT/$wl.err
  grep "schedule tuner" $OUT/$wl.err | tail -1
  python - $wl $OUT/$wl.json <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(path) if l.startswith("{")][-1])
    r = j["roofline"]
    print(f"{tag:24s} {j['value']:9.1f} evals/s  step {j['ms_per_step']*1e3:8.1f} us  prune {r['kernel_ms']*1e3:8.1f} us  {r['achieved']:7.2f} {r['unit']}  frac {r['frac']:.3f}  {r['kernel']} x{r['launches_per_step']}  expm {r['expm_ms']} reduce {r['reduce_ms']}")
except Exception as e:
    print(f"{tag:24s} FAILED ({e})")
PY
done
