run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', round(d['value'],1), 'evals/s; prune', round(r['kernel_ms']*1e3,1), 'us launches', r.get('launches_per_step'))"; }
for T in 1 2; do for F in auto 2 4 8 12 16 31 62; do
  export HYPHY_HIP_TILES=$T; if [ $F = auto ]; then unset HYPHY_HIP_FRAGMENT; else export HYPHY_HIP_FRAGMENT=$F; fi
  run "T=$T F=$F"
done; done
