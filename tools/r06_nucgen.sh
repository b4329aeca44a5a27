#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_nucgen.py -x -q -m gpu > $OUT/t_nucgen.log 2>&1; tail -15 $OUT/t_nucgen.log
for ng in 0 1; do
  for wl in gtr_32x50k gtr_32x1m hky_8x1k; do
    HYPHY_HIP_NUCGEN=$ng HYPHY_HIP_VERBOSE=1 timeout 300 python bench.py --workload $wl --steps 200 --warmup 10 --no-cpu-baseline --no-traffic > $OUT/ng${ng}_$wl.json 2> $OUT/ng${ng}_$wl.err
    echo "nucgen=$ng $wl $(python -c "
import json,sys
j=json.loads([l for l in open('$OUT/ng${ng}_$wl.json') if l.startswith('{')][-1]); r=j['roofline']
print(round(j['value'],1),'evals/s step',round(j['ms_per_step']*1e3,1),'us kernel',r['kernel'],round(r['kernel_ms']*1e3,1),'us expm',r.get('expm_ms'),'reduce',r.get('reduce_ms'))
" 2>&1)"
    grep nucgen $OUT/ng${ng}_$wl.err | tail -2
  done
done 2>&1 | tee $OUT/nucgen_ab.txt
