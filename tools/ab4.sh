#!/bin/bash
# GPU box: A/B builds of libhyphy_hip.so on the same box, interleaved, production settings (tuner on) unless ENV is given.
# usage: LIBS="base=hyphy_amd/lib_base/libhyphy_hip.so new=hyphy_amd/lib/libhyphy_hip.so" [WLS="mg94_64x10k ..."] [REPS=2] tools/ab4.sh
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab; mkdir -p $OUT
WLS=${WLS:-"mg94_64x10k mg94_128x100k mg94_32x5k mg94_64x2500"}
REPS=${REPS:-2}
one() { tag=$1; lib=$2; wl=$3; steps=$4
  HYPHY_HIP_VERBOSE=1 HYPHY_HIP_LIB=$lib timeout 300 python bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline --no-traffic > $OUT/$tag.json 2> $OUT/$tag.err
  python - $tag $OUT/$tag.json $OUT/$tag.err <<'PY'
import json, sys, re
tag, path, err = sys.argv[1:4]
try:
    j = json.loads([l for l in open(path) if l.startswith("{")][-1]); r = j["roofline"]
    sched = ""
    for l in open(err):
        m = re.search(r"-> (\S+)", l)
        if "schedule tuner" in l and m: sched = m.group(1)
    print(f"{tag:44s} step {j['ms_per_step']*1e3:8.1f} us  {r['kernel']:18s} {r['kernel_ms']*1e3:8.1f} us  frac {r['frac']:.3f}  expm {1e3*(r.get('expm_ms') or 0):5.1f} reduce {1e3*(r.get('reduce_ms') or 0):4.1f}  {sched}")
except Exception as e:
    print(f"{tag:44s} FAILED ({e})")
PY
}
for rep in $(seq 1 $REPS); do
  for wl in $WLS; do
    steps=200; [ $wl = mg94_128x100k ] && steps=30
    for kv in $LIBS; do
      name=${kv%%=*}; lib=$GRAFT_REPO_ROOT/${kv#*=}
      one ${name}_${wl}_r$rep $lib $wl $steps
    done
  done
done
