#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in hyphy_amd/lib_a/libhyphy_hip.so hyphy_amd/lib/libhyphy_hip.so; do
  HYPHY_HIP_LIB=$GRAFT_REPO_ROOT/$lib HYPHY_HIP_CHAIN_M=12 timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); r=j['roofline']; print('$lib', 'step', round(j['ms_per_step']*1e3,1), 'expm', round(r['expm_ms']*1e3,2), 'prune', round(r['kernel_ms']*1e3,1))"
done; done
