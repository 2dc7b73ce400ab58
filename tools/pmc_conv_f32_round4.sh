# round 4: where the waves of one fp32 convolution kernel spend their cycles, and what it moves (rocprofv3 --pmc in separate passes, never with a trace).
#   SHAPE="Cin H Cout R stride pad" WHAT="fwd dgrad wgrad" bash tools/pmc_conv_f32_round4.sh      (extra environment, e.g. LEC_WGRAD_SHIFT=0, reaches the kernels)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_r4
rm -rf $O; mkdir -p $O
SHAPE="${SHAPE:-256 14 256 3 1 1}"
WHAT="${WHAT:-wgrad}"
for what in $WHAT; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/a_$what -o p -- python3 $R/tools/prof_conv_f32.py $SHAPE --iters 3 --what $what > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/b_$what -o p -- python3 $R/tools/prof_conv_f32.py $SHAPE --iters 3 --what $what > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/c_$what -o p -- python3 $R/tools/prof_conv_f32.py $SHAPE --iters 3 --what $what > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_ATOMIC_sum --output-format csv -d $O/d_$what -o p -- python3 $R/tools/prof_conv_f32.py $SHAPE --iters 3 --what $what > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for what in ("$what",):
    tot=collections.defaultdict(float); n=collections.defaultdict(int)
    for sub in ('a','b','c','d'):
        for f in glob.glob('$O/%s_%s/**/*counter_collection.csv' % (sub, what), recursive=True):
            for r in csv.DictReader(open(f)):
                if 'conv_f32' in r['Kernel_Name']:
                    tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    avg={k: tot[k]/max(n[k],1) for k in tot}
    print("$SHAPE", what, {k: '%.4g' % v for k, v in sorted(avg.items())})
    if 'SQ_WAVE_CYCLES' in avg:
        wc=avg['SQ_WAVE_CYCLES']
        print('   parked (s_waitcnt / barrier) %.1f %%, issue stall %.1f %% (of which LDS issue %.1f %%), issuing %.1f %% of wave-cycles; matrix pipe busy %.1f %% of kernel cycles' % (
            100*avg['SQ_WAIT_ANY']/wc, 100*avg['SQ_WAIT_INST_ANY']/wc, 100*avg.get('SQ_WAIT_INST_LDS',0)/wc, 100*avg['SQ_ACTIVE_INST_ANY']/wc, 100*avg['SQ_VALU_MFMA_BUSY_CYCLES']/1024/(avg['GRBM_GUI_ACTIVE']/8)))
    if 'SQ_INST_LEVEL_VMEM' in avg:
        print('   mean latency: global load %.0f cycles (%d loads), LDS %.0f cycles; VALU (non-MFMA) per MFMA %.2f; LDS bank conflict cycles %.3g' % (
            avg['SQ_INST_LEVEL_VMEM']/max(avg['SQ_INSTS_VMEM_RD'],1), avg['SQ_INSTS_VMEM_RD'], avg['SQ_INST_LEVEL_LDS']/max(avg['SQ_INSTS_LDS'],1), (avg['SQ_INSTS_VALU']-avg['SQ_INSTS_MFMA'])/avg['SQ_INSTS_MFMA'], avg['SQ_LDS_BANK_CONFLICT']))
    if 'FETCH_SIZE' in avg:
        print('   per launch: HBM read %.1f MB (FETCH_SIZE KiB x 2, the gfx950 correction), written %.1f MB; L2 hit rate %.3f; L2 atomics %.3g' % (
            avg['FETCH_SIZE']*1024*2/1e6, avg.get('WRITE_SIZE',0)*1024/1e6, avg.get('TCC_HIT_sum',0)/max(avg.get('TCC_HIT_sum',0)+avg.get('TCC_MISS_sum',0),1), avg.get('TCC_ATOMIC_sum',0)))
PY
done
