#!/bin/bash
# GPU box: headline + two other sizes for a list of library builds (LIBS="name=path[@ENV=val] ..."), interleaved, two repetitions
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/abq; mkdir -p $OUT
one() { tag=$1; lib=$2; wl=$3; m=$4; steps=$5
  HYPHY_HIP_LIB=$lib HYPHY_HIP_CHAIN_M=$m timeout 300 python bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline --no-traffic > $OUT/$tag.json 2> $OUT/$tag.err
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
  for kv in $LIBS; do
    name=${kv%%=*}; rest=${kv#*=}; lib=$GRAFT_REPO_ROOT/${rest%%@*}; extra=""; [ "$rest" != "${rest%%@*}" ] && extra=${rest#*@}
    [ -n "$extra" ] && export $extra
    one ${name}_head_m12_r$rep $lib mg94_64x10k 12 200
    [ $rep = 1 ] && one ${name}_1250_m5_r$rep $lib mg94_64x1250 5 200
    [ $rep = 1 ] && one ${name}_big_m40_r$rep $lib mg94_128x100k 40 30
    [ -n "$extra" ] && unset ${extra%%=*}
  done
done
