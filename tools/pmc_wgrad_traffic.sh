cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in "" "--x3"; do
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_wg$mode -o p -- python3 $R/tools/prof_conv_f32.py $mode --iters 3 --what wgrad 2>&1 | grep -i error
done
python3 - <<'PY'
import csv, os
R=os.environ['GRAFT_REPO_ROOT']
for m in ('', '--x3'):
    d=R+'/gpurun_out/pmc_wg'+m+'/'
    rows=[r for r in csv.DictReader(open(d+'p_counter_collection.csv')) if 'wgrad' in r['Kernel_Name']]
    kt=[r for r in csv.DictReader(open(d+'p_kernel_trace.csv')) if 'wgrad' in r['Kernel_Name']]
    print(m or 'native', 'FETCH_SIZE x2 MB:', [round(float(r['Counter_Value'])*2/1e3,1) for r in rows], 'us:', [round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in kt])
PY
