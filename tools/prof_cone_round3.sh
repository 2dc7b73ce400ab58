# round 3: the fused cone-loss kernel (joint_loss_kernel) under rocprofv3 at three sizes -- kernel trace (durations) and, in SEPARATE
# passes, FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md, HBM section: FETCH_SIZE x 2 on gfx950).  The program comes directly after `--`.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/cone_pmc
rm -rf $O; mkdir -p $O
for shape in "256 5 10 2000" "256 256 10 50000" "4096 256 10 50000" "256 256 128 50000"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -o k -- python3 $R/tools/prof_cone.py $shape > $O/kt_$tag.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fs_$tag -o f -- python3 $R/tools/prof_cone.py $shape > $O/fs_$tag.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/ws_$tag -o w -- python3 $R/tools/prof_cone.py $shape > $O/ws_$tag.log 2>&1
done
python3 $R/tools/make_cone_pmc_round3.py $O > $O/r03_cone_pmc.md 2> $O/make.err
cp $O/r03_cone_pmc.md $O/r03_cone_pmc.json $R/gpurun_out/ 2>/dev/null
find $O -name "*.csv" -size +2M -delete
head -30 $O/r03_cone_pmc.md; tail -3 $O/make.err
