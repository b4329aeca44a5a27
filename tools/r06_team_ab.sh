#!/bin/bash
# r06: lower phase, row-split team walk (default) against the one-wave walk (HYPHY_HIP_REP_TEAM=0), on the GPU box
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_repeats.py -x -q -m gpu > $OUT/t_repeats.log 2>&1; tail -3 $OUT/t_repeats.log
for team in 1 0; do
  for wl in mg94_64x10k mg94_32x5k busted3_64x10k mg94_128x100k mg94_64x2500; do
    steps=200; [ $wl = mg94_128x100k ] && steps=30
    HYPHY_HIP_REP_TEAM=$team HYPHY_HIP_VERBOSE=1 timeout 300 python bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline --no-traffic > $OUT/ab_team${team}_$wl.json 2> $OUT/ab_team${team}_$wl.err
    echo "team=$team $wl $(python -c "
import json,sys
j=json.loads([l for l in open('$OUT/ab_team${team}_$wl.json') if l.startswith('{')][-1]); r=j['roofline']
print(round(j['value'],1),'evals/s step',round(j['ms_per_step']*1e3,1),'us kernel',round(r['kernel_ms']*1e3,1),'us', r.get('lower_ms'), r.get('trunk_ms'))
" 2>&1)"
    grep "repeats:" $OUT/ab_team${team}_$wl.err | tail -1
  done
done 2>&1 | tee $OUT/team_ab.txt
