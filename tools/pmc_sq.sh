#!/bin/bash
# SQ-level PMC passes for the pruning kernel (each pass bounded by timeout). Usage: tools/pmc_sq.sh tag [bench args]
TAG=${1:-x}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcsq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_IFETCH_LEVEL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > /dev/null 2>&1
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
    if 'prune' in k: print(k, json.dumps(out[k],indent=0,sort_keys=True))
PY
