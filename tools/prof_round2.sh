# round-2 profiles: kernel traces of the bench step (native fp32 and the split mode), PMC passes of the split kernels on one layer
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in native x3; do
  rm -rf $R/gpurun_out/kt_$m
  rocprofv3 --kernel-trace -d $R/gpurun_out/kt_$m -o t -- python3 $R/bench.py --steps 6 --warmup 3 --secondary none --no-cpu-baseline --no-stress --through-trainer 0 --conv-f32 $m > $R/gpurun_out/kt_$m.json 2> $R/gpurun_out/kt_$m.err
  python3 $R/tools/summarize_rocpd.py $(ls $R/gpurun_out/kt_$m/*/*.db $R/gpurun_out/kt_$m/*.db 2>/dev/null | head -1) --steps 3 --grid 16384 > $R/gpurun_out/kt_${m}_summary.md 2>> $R/gpurun_out/kt_$m.err
  rm -rf $R/gpurun_out/kt_$m
done
for what in fwd dgrad wgrad; do
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc2_a_$what -o p -- python3 $R/tools/prof_conv_f32.py --x3 --iters 3 --what $what 2>&1 | grep -i "error" 
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc2_b_$what -o p -- python3 $R/tools/prof_conv_f32.py --x3 --iters 3 --what $what 2>&1 | grep -i "error"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc2_c_$what -o p -- python3 $R/tools/prof_conv_f32.py --x3 --iters 3 --what $what 2>&1 | grep -i "error"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc2_d_$what -o p -- python3 $R/tools/prof_conv_f32.py --x3 --iters 3 --what $what 2>&1 | grep -i "error"
done
ls $R/gpurun_out | grep "kt_\|pmc2" | head -30
