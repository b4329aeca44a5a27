run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --workload $WLN 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', round(d['value'],1), 'evals/s; prune', round(r['kernel_ms']*1e3,1), 'us launches', r.get('launches_per_step'))"; }
for WLN in ${WLS:-mg94_64x1250 mg94_64x2500 mg94_64x5000}; do for K in 0 1; do for F in ${FS:-auto}; do
  [ $K = 0 ] && [ $F != auto ] && continue
  export HYPHY_HIP_KERNEL=$K; if [ $F = auto ]; then unset HYPHY_HIP_FRAGMENT; else export HYPHY_HIP_FRAGMENT=$F; fi
  run "$WLN K=$K F=$F"
done; done; done
