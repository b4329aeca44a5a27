#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
HYPHY_HIP_VERBOSE=1 timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>&1 | grep "schedule tuner" | tail -2
for m in 8 12; do
HYPHY_HIP_REPEATS=1 HYPHY_HIP_CHAIN_M=$m HYPHY_HIP_TIMELINE=$OUT/tl_trunk_m$m.txt timeout 300 python bench.py --workload mg94_64x10k --steps 3 --warmup 3 --no-cpu-baseline --no-traffic > /dev/null 2>&1
(echo "# mg94_64x10k REP trunk, chain cut m = $m, trace build of prune_wave_kernel<REP>"; python tools/timeline_waves.py $OUT/tl_trunk_m$m.txt 624) > $OUT/phases_trunk_m$m.txt
rm -f $OUT/tl_trunk_m$m.txt
done
cat $OUT/phases_trunk_m8.txt
