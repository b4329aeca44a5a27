#!/bin/bash
# GPU box: A/B runs, one bench line per "label|ENV=.. ENV=..|workload|steps" spec ($SPECS, separated by ';'), condensed; the whole
# list is repeated $REPS times (interleaved repetitions).  Library builds compete through HYPHY_HIP_LIB=<path> in a spec's ENV.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab; mkdir -p $OUT
for rep in $(seq 1 ${REPS:-1}); do
echo "$SPECS" | tr ';' '\n' | while IFS='|' read -r tag envs wl steps; do
  [ -z "$tag" ] && continue
  env HYPHY_HIP_VERBOSE=1 $envs timeout 300 python bench.py --workload $wl --steps ${steps:-200} --warmup 10 --no-cpu-baseline --no-traffic > $OUT/$tag.json 2> $OUT/$tag.err
  python - "$tag" $OUT/$tag.json $OUT/$tag.err <<'PY'
import json, sys, re
tag, path, err = sys.argv[1:4]
try:
    j = json.loads([l for l in open(path) if l.startswith("{")][-1]); r = j["roofline"]
    sched = ""
    for l in open(err):
        m = re.search(r"-> (\S+)", l)
        if "schedule tuner" in l and m: sched = m.group(1)
    print(f"{tag:44s} step {j['ms_per_step']*1e3:8.1f} us  {r['kernel']:18s} {r['kernel_ms']*1e3:8.1f} us  frac {(r['frac'] or 0):.3f}  expm {1e3*(r.get('expm_ms') or 0):5.1f} reduce {1e3*(r.get('reduce_ms') or 0):4.1f}  {sched}")
except Exception as e:
    print(f"{tag:44s} FAILED ({e})")
PY
done
done
