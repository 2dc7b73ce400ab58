# round-3 profiles of the fp32 bench step (two concurrent half-batch passes): kernel trace -> steady-state table; FETCH_SIZE / WRITE_SIZE of
# every kernel (separate --pmc passes) -> step traffic by family.  Eager launches (a replayed graph shows the same kernels; counters per
# dispatch need them launched one by one).  The program comes directly after `--`.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3prof
rm -rf $O; mkdir -p $O
ARGS="--steps 6 --warmup 3 --secondary none --no-cpu-baseline --no-stress --through-trainer 0 --launch eager"
rocprofv3 --kernel-trace -d $O/kt -o t -- python3 $R/bench.py $ARGS > $O/kt.json 2> $O/kt.err
python3 $R/tools/summarize_rocpd.py $(ls $O/kt/*/*.db $O/kt/*.db 2>/dev/null | head -1) --steps 3 --grid 16384 > $O/r03_bench_cfg3_f32_steady_state.md 2>> $O/kt.err
rm -rf $O/kt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/st_$c -o f -- python3 $R/bench.py --steps 2 --warmup 1 --launch eager --secondary none --no-cpu-baseline --no-stress --through-trainer 0 > /dev/null 2> $O/st_$c.err
  python3 $R/tools/summarize_pmc.py $(ls $O/st_$c/*counter_collection.csv $O/st_$c/*/*counter_collection.csv 2>/dev/null | head -1) --prefix "" > $O/st_$c.json
  rm -rf $O/st_$c
done
head -30 $O/r03_bench_cfg3_f32_steady_state.md; ls -la $O; tail -2 $O/kt.err
