#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_nucgen.py -x -q -m gpu > $OUT/t_nucgen.log 2>&1; tail -3 $OUT/t_nucgen.log
run() { tag=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py --workload $wl --steps 200 --warmup 10 --no-cpu-baseline --no-traffic > $OUT/n2_${tag}_$wl.json 2> $OUT/n2_${tag}_$wl.err
  echo "$tag $wl $(python -c "
import json,sys
j=json.loads([l for l in open('$OUT/n2_${tag}_$wl.json') if l.startswith('{')][-1]); r=j['roofline']
print(round(j['value'],1),'evals/s step',round(j['ms_per_step']*1e3,1),'us kernel',r['kernel'],round(r['kernel_ms']*1e3,1),'us expm',r.get('expm_ms'),'reduce',r.get('reduce_ms'))
" 2>&1)"
}
for wl in gtr_32x50k hky_8x1k; do
run small_fold_fuse $wl X=1
run small_nofold_fuse $wl HYPHY_HIP_NUC_FOLD=0
run small_fold_nofuse $wl HYPHY_HIP_FUSED_REDUCE=0
run small_nofold_nofuse $wl HYPHY_HIP_NUC_FOLD=0 HYPHY_HIP_FUSED_REDUCE=0
run scalar $wl HYPHY_HIP_NUCGEN_SMALL=0
done
run small gtr_32x1m HYPHY_HIP_NUCGEN_SMALL=1
