# round 4: the fused cone-loss kernel (joint_loss_kernel) under rocprofv3 at four sizes: kernel trace (durations) and, in SEPARATE --pmc passes,
# HBM traffic (FETCH_SIZE x 2 on gfx950, WRITE_SIZE), the L2's atomic traffic, and the waves' wait / issue picture -- the evidence behind
# bench.py's roofline_cone "binding resource".  The program comes directly after `--`; counters never share a run with a trace.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/cone_pmc4
rm -rf $O; mkdir -p $O
for shape in "256 5 10 2000" "256 256 10 50000" "4096 256 10 50000" "256 256 128 50000"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -o k -- python3 $R/tools/prof_cone.py $shape > $O/kt_$tag.log 2>&1
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" "TCC_ATOMIC TCC_REQ_sum TCC_HIT_sum" "TCP_TCC_ATOMIC_WITHOUT_RET_REQ TCC_EA0_ATOMIC" "GRBM_GUI_ACTIVE SQ_WAIT_ANY"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $O/pm${i}_$tag -o c -- python3 $R/tools/prof_cone.py $shape > $O/pm${i}_$tag.log 2>&1
  done
done
python3 $R/tools/make_cone_pmc_round4.py $O > $O/r04_cone_pmc.md 2> $O/make.err
cp $O/r04_cone_pmc.md $O/r04_cone_pmc.json $R/gpurun_out/ 2>/dev/null
find $O -name "*.csv" -size +2M -delete
cat $O/r04_cone_pmc.md; tail -3 $O/make.err
