#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_line.json 2> $OUT/bench.err
python - <<'PY'
import json
j=json.loads([l for l in open('/root/repo/gpurun_out/r06/bench_driver_line.json') if l.startswith('{')][-1])
r=j['roofline']
print({k:j[k] for k in ('value','value_cold','ms_per_step','shader_clock_ghz','shader_clock_ghz_cold')})
print({k:r.get(k) for k in ('kernel','kernel_ms','frac','achieved','traffic','expm_ms','reduce_ms','repeat_ratio')})
print(j.get('cpu_baseline'), j.get('parity'))
PY
