# SQ / LDS counters of ONE bf16 convolution layer (forward, data gradient, weight gradient): where do the waves of the implicit-GEMM kernels wait?
#   bash tools/pmc_conv_bf16.sh "<layer name substring of tools/bench_conv_bf16.py>"   -> gpurun_out/pmc_conv_bf16/*.csv + summary on stdout
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_conv_bf16
rm -rf $O; mkdir -p $O
L="$1"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU" \
         "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -o c -- python3 $R/tools/bench_conv_bf16.py --rows 512 --no-lib --iters 3 --only "$L" > $O/p$i.log 2>&1
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70]
        if 'conv_bf16' not in k: continue
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print('==', k)
    for c, v in sorted(d.items()):
        print('   %-32s n=%d  mean %.4g' % (c, len(v), sum(v) / len(v)))
PY
