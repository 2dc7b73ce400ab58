cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in main dbg2 dbg4; do
  if [ $v != main ]; then export LEC_LIB_PATH=$R/variants/liblecone_$v.so; fi
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_x3_$v -o p -- python3 $R/tools/prof_conv_f32.py --x3 --iters 3 --what fwd 2>&1 | grep -i error
done
ls $R/gpurun_out/pmc_x3_main
