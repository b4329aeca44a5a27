#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; wl=$1; shift
  env "$@" HYPHY_HIP_VERBOSE=1 timeout 300 python bench.py --workload $wl --steps 200 --warmup 10 --no-cpu-baseline --no-traffic > $OUT/q_${tag}_$wl.json 2> $OUT/q_${tag}_$wl.err
  echo "$tag $wl $(python -c "
import json,sys
j=json.loads([l for l in open('$OUT/q_${tag}_$wl.json') if l.startswith('{')][-1]); r=j['roofline']
print(round(j['value'],1),'evals/s step',round(j['ms_per_step']*1e3,1),'us kernel',round(r['kernel_ms']*1e3,1),'us expm',r.get('expm_ms'),'reduce',r.get('reduce_ms'))
" 2>&1)"
}
run base mg94_64x10k X=1
run fused mg94_64x10k HYPHY_HIP_FUSED_REDUCE=1
run base mg94_32x5k X=1
run fused mg94_32x5k HYPHY_HIP_FUSED_REDUCE=1
run base mg94_64x2500 X=1
run base gtr_32x50k X=1
run base gtr_32x1m X=1
run base hky_8x1k X=1
cd /tmp && export TMPDIR=/tmp
HYPHY_HIP_REPEATS=1 HYPHY_HIP_CHAIN_M=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_q -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-traffic > $OUT/stats_q.log 2>&1
f=$(find $OUT/stats_q -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
