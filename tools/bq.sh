#!/bin/bash
# One short line per bench run: tools/bq.sh <workload> <steps> [ENV=VALUE ...]
wl=$1; steps=$2; shift 2
env "$@" python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps $steps --warmup 20 --no-cpu-baseline --no-traffic 2>/dev/null | python -c '
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = j["roofline"]
print(sys.argv[1:], "%.1f evals/s  step %.1f us  kernel %.1f us  expm %s  reduce %s" % (j["value"], 1e3 * j["ms_per_step"], 1e3 * r["kernel_ms"], r["expm_ms"], r["reduce_ms"]))
' $wl "$@"
