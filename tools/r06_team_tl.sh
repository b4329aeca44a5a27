#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for wl in mg94_64x10k mg94_128x100k; do
HYPHY_HIP_REPEATS=1 HYPHY_HIP_REP_TIMELINE=$OUT/team_tl_$wl.txt timeout 300 python bench.py --workload $wl --steps 3 --warmup 3 --no-cpu-baseline --no-traffic > /dev/null 2> $OUT/team_tl_$wl.err
python tools/rep_team_timeline.py $OUT/team_tl_$wl.txt | tee $OUT/team_phases_$wl.txt
done
