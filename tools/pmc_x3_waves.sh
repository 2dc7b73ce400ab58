cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC --output-format csv -d $R/gpurun_out/pmc3_a -o p -- python3 $R/tools/prof_conv_f32.py --x3 --iters 20 --what fwd 2>&1 | grep -i error
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc3_b -o p -- python3 $R/tools/prof_conv_f32.py --x3 --iters 20 --what fwd 2>&1 | grep -i error
python3 - <<'PY'
import csv, os, collections
R=os.environ['GRAFT_REPO_ROOT']
a={}
for g in 'ab':
    rows=[r for r in csv.DictReader(open(R+'/gpurun_out/pmc3_%s/p_counter_collection.csv'%g)) if 'x3_act' in r['Kernel_Name']]
    agg=collections.defaultdict(list)
    for r in rows[len(rows)//2:]: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): a[k]=sum(v)/len(v)
kt=[r for r in csv.DictReader(open(R+'/gpurun_out/pmc3_a/p_kernel_trace.csv')) if 'x3_act' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in kt][10:]
print('dur us', round(sum(d)/len(d),1))
for k,v in sorted(a.items()): print('%-28s %.4g' % (k, v))
wc=a['SQ_WAVE_CYCLES']
print('per wave-cycle: wait_any %.2f wait_inst %.2f active_inst %.2f valu %.2f lds %.2f misc %.2f' % (a['SQ_WAIT_ANY']/wc, a['SQ_WAIT_INST_ANY']/wc, a['SQ_ACTIVE_INST_ANY']/wc, a['SQ_ACTIVE_INST_VALU']/wc, a['SQ_ACTIVE_INST_LDS']/wc, a['SQ_ACTIVE_INST_MISC']/wc))
PY
