# The fused cone-loss kernel (joint_loss_kernel) under rocprofv3 at four sizes: kernel trace (durations) and, in SEPARATE --pmc passes, HBM traffic
# (FETCH_SIZE x 2 on gfx950, WRITE_SIZE) and the waves' wait / issue picture -- the evidence behind bench.py's roofline_cone / roofline_stress `traffic`.
#   bash tools/prof_cone_pmc.sh <name, e.g. r06_cone_pmc>   ->  gpurun_out/<name>.md / .json   (copy into profiles/)
# The program comes directly after `--`; counters never share a run with a trace; every profiler run sits under `timeout`.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
name=${1:-cone_pmc}
O=$R/gpurun_out/cone_pmc
rm -rf $O; mkdir -p $O
for shape in "256 5 10 2000" "256 256 10 50000" "4096 256 10 50000" "256 256 128 50000"; do
  tag=$(echo $shape | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -o k -- python3 $R/tools/prof_cone.py $shape > $O/kt_$tag.log 2>&1
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAIT_ANY" "TCC_REQ_sum TCC_HIT_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pm${i}_$tag -o c -- python3 $R/tools/prof_cone.py $shape > $O/pm${i}_$tag.log 2>&1
  done
done
python3 $R/tools/make_cone_pmc.py $O $name > $O/$name.md 2> $O/make.err
cp $O/$name.md $O/$name.json $R/gpurun_out/ 2>/dev/null
find $O -name "*.csv" -size +2M -delete
cat $O/$name.md; tail -3 $O/make.err
