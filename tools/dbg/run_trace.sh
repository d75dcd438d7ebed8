O=$GRAFT_REPO_ROOT/gpurun_out/r2m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
PCS_ICP_ONE_LAUNCH=0 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o tr -- python3 $GRAFT_REPO_ROOT/tools/stage_times.py --repeat 2 > $O/prof_run.txt 2>&1
ls -la /tmp/prof >> $O/prof_run.txt
python3 - <<'PY'
import csv, os
src='/tmp/prof/tr_kernel_trace.csv'
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r2m/trace_small.csv'
rows=list(csv.DictReader(open(src)))
print(len(rows), rows[0].keys())
keep=['Kernel_Name','Start_Timestamp','End_Timestamp','Grid_Size_X','Grid_Size_Y','Grid_Size_Z','Workgroup_Size_X','Queue_Id','Stream_Id']
keep=[k for k in keep if k in rows[0]]
with open(out,'w') as f:
    w=csv.writer(f); w.writerow(keep)
    for r in rows:
        w.writerow([ (r[k][:48] if k=='Kernel_Name' else r[k]) for k in keep])
PY
