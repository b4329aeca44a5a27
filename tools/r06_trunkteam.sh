#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; wl=$1; shift
  env "$@" HYPHY_HIP_VERBOSE=1 timeout 120 python bench.py --workload $wl --steps 200 --warmup 10 --no-cpu-baseline --no-traffic > $OUT/tt_${tag}_$wl.json 2> $OUT/tt_${tag}_$wl.err
  echo "$tag $wl $(python -c "
import json,sys
j=json.loads([l for l in open('$OUT/tt_${tag}_$wl.json') if l.startswith('{')][-1]); r=j['roofline']
print(round(j['value'],1),'evals/s step',round(j['ms_per_step']*1e3,1),'us kernel',r['kernel'],round(r['kernel_ms']*1e3,1),'us', 'logl', j['logl_last'])
" 2>&1)"
  grep "schedule tuner\|repeats:" $OUT/tt_${tag}_$wl.err | tail -3 | cut -c1-400
}
run team mg94_64x10k X=1
run noteam mg94_64x10k HYPHY_HIP_TRUNK_TEAM=0
run team mg94_32x5k X=1
run team mg94_64x2500 X=1
run team busted3_64x10k X=1
run team mg94_128x100k X=1
