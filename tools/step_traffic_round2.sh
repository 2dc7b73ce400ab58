# HBM traffic of every kernel of the fp32 bench step (both convolution modes): FETCH_SIZE / WRITE_SIZE in separate --pmc passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in native x3; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/st_${m}_$c
    rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/st_${m}_$c -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --secondary none --no-cpu-baseline --no-stress --through-trainer 0 --conv-f32 $m > /dev/null 2> $R/gpurun_out/st_${m}_$c.err
    python3 $R/tools/summarize_pmc.py $(ls $R/gpurun_out/st_${m}_$c/*counter_collection.csv $R/gpurun_out/st_${m}_$c/*/*counter_collection.csv 2>/dev/null | head -1) --prefix "" > $R/gpurun_out/st_${m}_$c.json
    rm -rf $R/gpurun_out/st_${m}_$c
  done
done
ls -la $R/gpurun_out/st_*.json
