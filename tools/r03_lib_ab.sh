#!/bin/bash
# GPU box: A/B of library builds on the tuner's own choice (no forced cut), interleaved.
# usage: LIBS="new=hyphy_amd/lib/libhyphy_hip.so old=hyphy_amd/lib_ab/libhyphy_hip_<commit>.so" WLS="mg94_64x10k:300" [REPS=3] tools/r03_lib_ab.sh
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03lib; mkdir -p $OUT
REPS=${REPS:-3}
for rep in $(seq 1 $REPS); do
  for kv in $LIBS; do
    name=${kv%%=*}; lib=$GRAFT_REPO_ROOT/${kv#*=}
    for w in $WLS; do
      tag=${name}_${w%%:*}_r$rep
      HYPHY_HIP_LIB=$lib HYPHY_HIP_VERBOSE=1 timeout 300 python bench.py --workload ${w%%:*} --steps ${w##*:} --warmup 10 --no-cpu-baseline --no-traffic > $OUT/$tag.json 2> $OUT/$tag.err
      python - $tag $OUT/$tag.json $OUT/$tag.err <<'PY'
import json, sys, re
tag, path, err = sys.argv[1:4]
try:
    j = json.loads([l for l in open(path) if l.startswith("{")][-1]); r = j["roofline"]
    pick = [ln.split("->")[-1].strip() for ln in open(err) if "schedule tuner" in ln]
    print(f"{tag:36s} {j['value']:9.1f} evals/s  step {j['ms_per_step']*1e3:7.1f} us  prune {r['kernel_ms']*1e3:7.1f} us  expm {(r.get('expm_ms') or 0)*1e3:5.1f}  reduce {(r.get('reduce_ms') or 0)*1e3:4.1f}  tuner -> {pick[-1] if pick else '?'}")
except Exception as e:
    print(f"{tag:36s} FAILED ({e})")
PY
    done
  done
done
