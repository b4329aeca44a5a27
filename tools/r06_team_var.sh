#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for var in 0 1 2 3 4 5; do
  for wl in mg94_64x10k mg94_32x5k mg94_128x100k; do
    steps=200; [ $wl = mg94_128x100k ] && steps=30
    HYPHY_HIP_REP_TEAM_VAR=$var HYPHY_HIP_REPEATS=1 HYPHY_HIP_VERBOSE=1 timeout 300 python bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline --no-traffic > $OUT/var${var}_$wl.json 2> $OUT/var${var}_$wl.err
    echo "var=$var $wl $(python -c "
import json,sys
j=json.loads([l for l in open('$OUT/var${var}_$wl.json') if l.startswith('{')][-1]); r=j['roofline']
print(round(j['value'],1),'evals/s step',round(j['ms_per_step']*1e3,1),'us kernel',round(r['kernel_ms']*1e3,1),'us')
" 2>&1)"
  done
done 2>&1 | tee $OUT/team_var.txt
