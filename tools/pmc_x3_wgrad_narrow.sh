cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc4_a -o p -- python3 $R/tools/prof_conv_f32.py 64 56 64 3 1 1 --x3 --iters 3 --what wgrad 2>&1 | grep -i error
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc4_b -o p -- python3 $R/tools/prof_conv_f32.py 64 56 64 3 1 1 --x3 --iters 3 --what wgrad 2>&1 | grep -i error
python3 - <<'PY'
import csv, os, collections
R=os.environ['GRAFT_REPO_ROOT']
a={}
for g in 'ab':
    rows=[r for r in csv.DictReader(open(R+'/gpurun_out/pmc4_%s/p_counter_collection.csv'%g)) if 'x3_wgrad' in r['Kernel_Name']]
    agg=collections.defaultdict(list)
    for r in rows: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): a[k]=sum(v)/len(v)
kt=[r for r in csv.DictReader(open(R+'/gpurun_out/pmc4_a/p_kernel_trace.csv')) if 'x3_wgrad' in r['Kernel_Name']]
print('dur us', [(int(r['End_Timestamp'])-int(r['Start_Timestamp']))//1000 for r in kt], kt[0]['Kernel_Name'][:80], 'grid', kt[0].get('Grid_Size'), kt[0].get('Workgroup_Size'))
for k,v in sorted(a.items()): print('%-28s %.4g' % (k, v))
PY
