#!/bin/bash
# GPU box: A/B of environment settings on the same box. usage: CASES="tag:ENV=v,ENV2=v tag2:..." WL=... M=... tools/ab_env.sh
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/abe; mkdir -p $OUT
for rep in 1 2; do for c in $CASES; do
  tag=${c%%:*}; envs=$(echo ${c#*:} | tr ',' ' ')
  env $envs HYPHY_HIP_CHAIN_M=${M:-12} timeout 300 python bench.py --workload ${WL:-mg94_64x10k} --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline > $OUT/$tag.json 2> $OUT/$tag.err
  python - ${tag}_r$rep $OUT/$tag.json <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(path) if l.startswith("{")][-1]); r = j["roofline"]
    print(f"{tag:40s} step {j['ms_per_step']*1e3:8.1f} us  prune {r['kernel_ms']*1e3:8.1f} us  frac {r['frac']:.3f}")
except Exception as e:
    print(f"{tag:40s} FAILED ({e})")
PY
done; done
