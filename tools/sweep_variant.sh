# compare pruning-kernel variants on the headline workload (run on the GPU box)
# usage: tools/sweep_variant.sh "<kernels>" "<slots>" "<fragment sizes>"
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline $WL 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', round(d['value'],1), 'evals/s; prune', round(r['kernel_ms']*1e3,1), 'us launches', r.get('launches_per_step'))"; }
for K in ${1:-0 1}; do for S in ${2:-3 2}; do for F in ${3:-auto}; do
  export HYPHY_HIP_KERNEL=$K HYPHY_HIP_SLOTS=$S; if [ $F = auto ]; then unset HYPHY_HIP_FRAGMENT; else export HYPHY_HIP_FRAGMENT=$F; fi
  run "K=$K slots=$S F=$F"
done; done; done
