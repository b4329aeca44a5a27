#!/bin/bash
# GPU box: SQ counters of the PRODUCTION pruning schedule (tuner off, the cut the tuner picks forced), separate --pmc passes.
# usage: tools/pmc_production.sh <out dir under gpurun_out>
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmcprod}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { wl=$1; shift
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    env "$@" timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 20 --warmup 6 --no-cpu-baseline --no-traffic > /dev/null 2>&1
  done
}
run mg94_64x10k HYPHY_HIP_CHAIN_M=12
run mg94_128x100k HYPHY_HIP_CHAIN_M=40 HYPHY_HIP_WAVE_VARIANT=2 HYPHY_HIP_SLOTS=2
find $OUT -name "*counter_collection.csv" | wc -l
