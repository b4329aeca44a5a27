#!/bin/bash
# GPU box: what does the schedule tuner pick per workload, and what does the default bench line look like
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_tuner
mkdir -p $OUT
for wl in mg94_64x10k mg94_64x5000 mg94_64x2500 mg94_64x1250 mg94_32x5k busted3_64x10k mg94_128x100k gtr_32x50k hky_8x1k; do
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
echo "== driver-style default line (with CPU baseline + parity)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/default_line.json 2> $OUT/default_line.err
tail -c 3000 $OUT/default_line.json
tail -3 $OUT/default_line.err
