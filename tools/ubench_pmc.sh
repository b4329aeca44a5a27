#!/bin/bash
# µbench under PMC: true GFX clock cycles (GRBM_GUI_ACTIVE) per kernel next to its wall time
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/ub -- $GRAFT_REPO_ROOT/tools/ubench_mfma_f64 > $GRAFT_REPO_ROOT/gpurun_out/ubench_pmc_stdout.txt 2>&1
python - <<PY
import csv,glob,collections
rows=collections.OrderedDict()
for f in glob.glob('/tmp/ub/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        d=rows.setdefault(r['Dispatch_Id'],{'grid':r['Grid_Size'],'name':r['Kernel_Name'][:40],'t':(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3})
        d[r['Counter_Name']]=float(r['Counter_Value'])
for k,d in rows.items():
    g=d.get('GRBM_GUI_ACTIVE',0)/8
    print(k,d['name'],d['grid'],'%.1f us'%d['t'],'gfxclk %.2f GHz'%(g/d['t']/1e3 if d['t'] else 0),'mfma',d.get('SQ_INSTS_MFMA'),'busy/inst %.1f'%(d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(1,d.get('SQ_INSTS_MFMA',1))), 'cycles/MFMA/SIMD %.1f'%(g/max(1,d.get('SQ_INSTS_MFMA',1)/1024)))
PY
