# HBM traffic of the fp32 bench step by kernel family (round 5): separate --pmc passes over eager launches of bench.py; summaries -> gpurun_out/r5prof/st_<counter>.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5prof
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/stf_$c -o f -- python3 $R/bench.py --steps 2 --warmup 1 --launch eager --secondary none --no-cpu-baseline --no-stress --through-trainer 0 > $O/stf_$c.out 2> $O/stf_$c.err
  python3 $R/tools/summarize_pmc.py $(ls $O/stf_$c/*counter_collection.csv $O/stf_$c/*/*counter_collection.csv 2>/dev/null | head -1) --prefix "" > $O/st_$c.json
  rm -rf $O/stf_$c
done
ls -la $O/st_*.json; tail -1 $O/stf_FETCH_SIZE.out | cut -c1-200
