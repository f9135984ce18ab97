set -u
export TMPDIR=/tmp
R=$(pwd); cd /tmp
PYTHONPATH=$R timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06zc_trace -o p -- python $R/tools/debug/prof_dyn.py > $R/gpurun_out/r06zc_trace.log 2>&1
cd $R
grep "hspf repair\]" gpurun_out/r06zc_trace.log | tail -12
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r06zc_trace/p_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last run: find the last kr_seed and print from there
idx=[i for i,r in enumerate(rows) if 'kr_seed' in r['Kernel_Name']]
i0=idx[-1]
prev=None
for r in rows[i0-8:i0+60]:
    s=int(r['Start_Timestamp']);e=int(r['End_Timestamp'])
    nm=r['Kernel_Name'].split('(')[0].replace('hspf::','').replace('void ','')[:36]
    print(f"{nm:36s} {(e-s)/1e3:8.1f} us  gap {((s-prev)/1e3 if prev else 0):7.1f}  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
    prev=e
PY
find gpurun_out/r06zc_trace -name "*kernel_trace.csv" -size +8M -delete
