# round-5 profiles: kernel traces (rocprofv3 --kernel-trace, rocpd output) of the step at every BASELINE config that has a bench line -- cfg3 fp32 (the headline),
# cfg3 bf16 (the secondary = config 5's inner loop), cfg2 fp32, cfg5 bf16 (chunked) -- summarised per step, plus FETCH_SIZE / WRITE_SIZE passes of the bf16 step.
# Eager launches (counters per dispatch need kernels launched one by one; a replayed graph shows the same kernels).  The program comes directly after `--`.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5prof
rm -rf $O; mkdir -p $O
COMMON="--secondary none --no-cpu-baseline --no-stress --through-trainer 0 --launch eager"
run_kt () {   # tag, marker, marker grid (0 = any), steps summarised, skip-last, bench args...
  tag=$1; marker=$2; grid=$3; nsteps=$4; skip=$5; shift 5
  rocprofv3 --kernel-trace -d $O/kt_$tag -o t -- python3 $R/bench.py $COMMON "$@" > $O/$tag.json 2> $O/$tag.err
  DB=$(ls $O/kt_$tag/*/*.db $O/kt_$tag/*.db 2>/dev/null | head -1)
  TF=$(python3 -c "import json; d=json.load(open('$O/$tag.json')); r=d.get('roofline_conv') or {}; print(r.get('alg_flops_per_step', 0)/1e12)")
  python3 $R/tools/summarize_rocpd.py $DB --steps $nsteps --skip-last $skip --marker $marker --grid $grid --conv-tflop-per-step $TF > $O/r05_bench_${tag}_steady_state.md 2>> $O/$tag.err
  python3 -c "
import json; d=json.load(open('$O/$tag.json'))
print('\nbench.py of the SAME run (under the profiler, eager launches): %s images/s, ms_per_step %.3f, dtype %s' % (d['value'], d['ms_per_step'], d['dtype']))" >> $O/r05_bench_${tag}_steady_state.md
  rm -rf $O/kt_$tag
}
run_kt cfg3_f32 joint_loss_kernel 0 3 3 --steps 6 --warmup 3
run_kt cfg3_bf16 joint_loss_kernel 0 3 2 --steps 6 --warmup 3 --dtype bf16
run_kt cfg2_f32 joint_loss_kernel 0 3 3 --steps 6 --warmup 3 --workload cfg2
run_kt cfg5_bf16 table_adam_kernel 0 2 0 --steps 3 --warmup 2 --workload cfg5 --dtype bf16
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/st_$c -o f -- python3 $R/bench.py --steps 2 --warmup 1 $COMMON --dtype bf16 > /dev/null 2> $O/st_$c.err
  python3 $R/tools/summarize_pmc.py $(ls $O/st_$c/*counter_collection.csv $O/st_$c/*/*counter_collection.csv 2>/dev/null | head -1) --prefix "" > $O/bf16_$c.json
  rm -rf $O/st_$c
done
ls -la $O; for f in $O/*.err; do echo == $f; tail -2 $f; done
