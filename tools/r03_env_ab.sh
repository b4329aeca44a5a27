#!/bin/bash
# GPU box: A/B of environment settings of ONE library build, interleaved.
# usage: CASES="tag[@ENV=val[,ENV=val]] ..." WLS="mg94_64x10k:200 mg94_64x1250:200" [REPS=2] tools/r03_env_ab.sh
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03ab; mkdir -p $OUT
REPS=${REPS:-2}
one() { tag=$1; wl=$2; steps=$3
  timeout 300 python bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline --no-traffic > $OUT/$tag.json 2> $OUT/$tag.err
  python - $tag $OUT/$tag.json <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(path) if l.startswith("{")][-1]); r = j["roofline"]
    ex = r.get("expm_ms"); rd = r.get("reduce_ms")
    print(f"{tag:44s} {j['value']:9.1f} evals/s  step {j['ms_per_step']*1e3:8.1f} us  prune {r['kernel_ms']*1e3:8.1f} us  frac {(r['frac'] if r.get('frac') is not None else float('nan')):.3f}"
          f"  expm {ex*1e3 if ex else float('nan'):6.1f}  reduce {rd*1e3 if rd else float('nan'):5.1f}  logL {j['logl_last']!r}")
except Exception as e:
    print(f"{tag:44s} FAILED ({e})")
PY
}
for rep in $(seq 1 $REPS); do
  for cs in $CASES; do
    name=${cs%%@*}; envs=""; [ "$cs" != "$name" ] && envs=${cs#*@}
    for kv in ${envs//,/ }; do export "$kv"; done
    for w in $WLS; do one ${name}_${w%%:*}_r$rep ${w%%:*} ${w##*:}; done
    for kv in ${envs//,/ }; do unset ${kv%%=*}; done
  done
done
