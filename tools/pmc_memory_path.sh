#!/bin/bash
# Runs on the GPU box: memory-path PMC passes for the pruning kernel (separate passes, kernel-trace only).
# Usage: HYPHY_HIP_KERNEL=0|1 tools/pmc_memory_path.sh tag
TAG=${1:-k0}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum" \
           "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_CYCLE_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_NORMAL_WRITEBACK_sum"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
done
python - <<PY
import csv,glob,collections,json
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        import re as _re; m=_re.search(r'(\w+_kernel)', r['Kernel_Name']); k=m.group(1) if m else r['Kernel_Name'][:40]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
out={k:{c:sum(v)/len(v) for c,v in d.items()} for k,d in acc.items()}
json.dump(out,open('$OUT/means.json','w'),indent=1,sort_keys=True)
for k in out:
    if 'prune' in k: print(k, json.dumps(out[k],indent=1,sort_keys=True))
PY
