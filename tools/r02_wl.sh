#!/bin/bash
# GPU box: one bench line per workload with the tuner's report. usage: tools/r02_wl.sh "wl1 wl2 ..." [extra bench args]
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/wl; mkdir -p $OUT
for wl in $1; do
  steps=200; [ $wl = mg94_128x100k ] && steps=30
  HYPHY_HIP_VERBOSE=1 timeout 300 python bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline --no-traffic $2 > $OUT/wl_$wl.json 2> $OUT/wl_$wl.err
  grep "schedule tuner" $OUT/wl_$wl.err | tail -1
  python - $wl $OUT/wl_$wl.json <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(path) if l.startswith("{")][-1]); r = j["roofline"]
    print(f"{tag:18s} {j['value']:9.1f} evals/s  step {j['ms_per_step']*1e3:8.1f} us  {r['kernel']} {r['kernel_ms']*1e3:8.1f} us  {r['achieved']:8.2f} {r['unit']}  frac {(r['frac'] if r['frac'] is not None else float('nan')):.3f}  expm {r['expm_ms']}  reduce {r['reduce_ms']}")
except Exception as e:
    print(f"{tag:18s} FAILED ({e})")
PY
done
