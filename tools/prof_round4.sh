# round-4 profiles of the fp32 bench step (two concurrent half-batch passes): kernel trace -> steady-state table with per-family UNIONS of launch
# intervals (what bench.py's roofline.frac is made of); FETCH_SIZE / WRITE_SIZE of every kernel (separate --pmc passes) -> step traffic by family.
# Eager launches (a replayed graph shows the same kernels; counters per dispatch need them launched one by one).  The program comes directly after `--`.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4prof
rm -rf $O; mkdir -p $O
ARGS="--steps 6 --warmup 3 --secondary none --no-cpu-baseline --no-stress --through-trainer 0 --launch eager"
rocprofv3 --kernel-trace -d $O/kt -o t -- python3 $R/bench.py $ARGS > $O/kt.json 2> $O/kt.err
TF=$(python3 -c "import json,sys; d=json.load(open('$O/kt.json')); print(d['roofline']['alg_flops_per_step']/1e12)")
python3 $R/tools/summarize_rocpd.py $(ls $O/kt/*/*.db $O/kt/*.db 2>/dev/null | head -1) --steps 3 --grid 16384 --conv-tflop-per-step $TF > $O/r04_bench_cfg3_f32_steady_state.md 2>> $O/kt.err
python3 -c "
import json; d=json.load(open('$O/kt.json')); r=d['roofline']
print('\nbench.py of the SAME run (under the profiler, eager launches): ms_per_step %.3f; roofline.frac %.4f (union of HIP-event intervals: %.3f ms busy per step), frac over the step wall time %.4f' % (d['ms_per_step'], r['frac'], r['ms_per_step_family_busy'], r['frac_lower_bound_flops_over_step_wall_time']))" >> $O/r04_bench_cfg3_f32_steady_state.md
rm -rf $O/kt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/st_$c -o f -- python3 $R/bench.py --steps 2 --warmup 1 --launch eager --secondary none --no-cpu-baseline --no-stress --through-trainer 0 > /dev/null 2> $O/st_$c.err
  python3 $R/tools/summarize_pmc.py $(ls $O/st_$c/*counter_collection.csv $O/st_$c/*/*counter_collection.csv 2>/dev/null | head -1) --prefix "" > $O/st_$c.json
  rm -rf $O/st_$c
done
head -30 $O/r04_bench_cfg3_f32_steady_state.md; tail -12 $O/r04_bench_cfg3_f32_steady_state.md; ls -la $O; tail -2 $O/kt.err
