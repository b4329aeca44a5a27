#!/bin/bash
# GPU box: headline workload at a list of chain cuts. usage: MS="9 10 11" [WL=..] tools/ab_m.sh
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/abm; mkdir -p $OUT
for m in $MS; do
  HYPHY_HIP_CHAIN_M=$m HYPHY_HIP_VERBOSE=1 timeout 300 python bench.py --workload ${WL:-mg94_64x10k} --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline --no-traffic > $OUT/m$m.json 2> $OUT/m$m.err
  grep "chain schedule" $OUT/m$m.err | head -1 | cut -c1-200
  python - m$m $OUT/m$m.json <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(path) if l.startswith("{")][-1]); r = j["roofline"]
    print(f"{tag:40s} step {j['ms_per_step']*1e3:8.1f} us  prune {r['kernel_ms']*1e3:8.1f} us  frac {r['frac']:.3f}")
except Exception as e:
    print(f"{tag:40s} FAILED ({e})")
PY
done
