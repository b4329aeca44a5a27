cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for K in 0 1; do
export HYPHY_HIP_KERNEL=$K
HYPHY_HIP_VERBOSE=1 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep "level" | sort | uniq -c
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$K -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/kt$K/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'prune' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
g=[r['Grid_Size_X']+'x'+r['Grid_Size_Y']+'x'+r['Grid_Size_Z'] for r in rows]
print('K=$K last 8 launches (us, grid):', [(round(x,1),y) for x,y in zip(d[-8:],g[-8:])])
PY
done
