#!/bin/bash
# One bench line per workload (no CPU baseline), condensed.  Usage: tools/bench_all_workloads.sh > out.txt
for W in mg94_64x10k mg94_32x5k busted3_64x10k mg94_64x5000 mg94_64x2500 mg94_64x1250 mg94_128x100k gtr_32x50k gtr_32x1m hky_8x1k; do
  EXTRA=""; [ $W = mg94_64x10k ] && EXTRA="--branch-cache --pipelined"
  STEPS=200; [ $W = mg94_128x100k ] && STEPS=40
  python bench.py --workload $W --steps $STEPS --warmup 20 --no-cpu-baseline $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('%-16s %8.1f evals/s  step %6.1f us  %s %6.1f us  %.2f %s (frac %.3f)%s%s' % (d['config']['workload'], d['value'], d['ms_per_step']*1e3, r['kernel'], r['kernel_ms']*1e3, r['achieved'], r['unit'], r['frac'],
  ('  pipelined %.1f evals/s' % d['value_pipelined_no_host_sync']) if d.get('value_pipelined_no_host_sync') else '',
  ('  branch-cache %.0f evals/s (build %.0f us)' % (d['branch_cache']['evals_per_s'], d['branch_cache']['build_ms']*1e3)) if d.get('branch_cache') else ''))"
done
