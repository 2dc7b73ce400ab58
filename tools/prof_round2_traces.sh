# round-2 kernel traces of the bench step in the launch mode bench.py picks (native fp32 and the split mode) -> steady-state summaries
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in native x3; do
  rm -rf $R/gpurun_out/kt_$m
  rocprofv3 --kernel-trace -d $R/gpurun_out/kt_$m -o t -- python3 $R/bench.py --steps 6 --warmup 3 --secondary none --no-cpu-baseline --no-stress --through-trainer 0 --launch eager --conv-f32 $m > $R/gpurun_out/kt_$m.json 2> $R/gpurun_out/kt_$m.err
  python3 $R/tools/summarize_rocpd.py $(ls $R/gpurun_out/kt_$m/*/*.db $R/gpurun_out/kt_$m/*.db 2>/dev/null | head -1) --steps 3 --grid 16384 > $R/gpurun_out/kt_${m}_summary.md 2>> $R/gpurun_out/kt_$m.err
  rm -rf $R/gpurun_out/kt_$m
done
head -12 $R/gpurun_out/kt_native_summary.md; grep launch_mode $R/gpurun_out/kt_native.json | head -2 | cut -c1-300
