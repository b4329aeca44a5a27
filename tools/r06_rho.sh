#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; wl=$1; shift
  env "$@" HYPHY_HIP_VERBOSE=1 timeout 300 python bench.py --workload $wl --steps 200 --warmup 10 --no-cpu-baseline --no-traffic > $OUT/rho_${tag}_$wl.json 2> $OUT/rho_${tag}_$wl.err
  echo "$tag $wl $(python -c "
import json,sys
j=json.loads([l for l in open('$OUT/rho_${tag}_$wl.json') if l.startswith('{')][-1]); r=j['roofline']
print(round(j['value'],1),'evals/s step',round(j['ms_per_step']*1e3,1),'us kernel',round(r['kernel_ms']*1e3,1),'us rr',r.get('repeat_ratio'))
" 2>&1)  $(grep 'repeats:' $OUT/rho_${tag}_$wl.err | tail -1 | cut -c1-110)"
}
for wl in mg94_64x10k mg94_128x100k; do
run base $wl HYPHY_HIP_REPEATS=1
run rho03 $wl HYPHY_HIP_REPEATS=1 HYPHY_HIP_REP_RHO=0.3
run rho06 $wl HYPHY_HIP_REPEATS=1 HYPHY_HIP_REP_RHO=0.6
run th05 $wl HYPHY_HIP_REPEATS=1 HYPHY_HIP_REP_THETA=0.5
run th07 $wl HYPHY_HIP_REPEATS=1 HYPHY_HIP_REP_THETA=0.7
run th05rho03 $wl HYPHY_HIP_REPEATS=1 HYPHY_HIP_REP_THETA=0.5 HYPHY_HIP_REP_RHO=0.3
run th025 $wl HYPHY_HIP_REPEATS=1 HYPHY_HIP_REP_THETA=0.25
done
