# HBM traffic of a bench.py step by kernel family: separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over eager launches; one parameterised script for every config / round.
#   bash tools/step_traffic.sh <tag> <marker kernel> [bench.py args...]   ->  gpurun_out/traffic/<tag>_{FETCH_SIZE,WRITE_SIZE}.json, then
#   python tools/make_step_traffic.py <tag> <ms per step> <profiles/name>
# Every rocprofv3 run sits under `timeout` (a counter pass that hangs must not take the box's limit with it).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/traffic
mkdir -p $O
tag=$1; marker=$2; shift 2
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/raw_${tag}_$c
  timeout 420 rocprofv3 --pmc $c --output-format csv -d $O/raw_${tag}_$c -o f -- python3 $R/bench.py --steps 2 --warmup 1 --launch eager --secondary none --no-cpu-baseline --no-stress --through-trainer 0 --through-trainer-files 0 "$@" > $O/${tag}_$c.out 2> $O/${tag}_$c.err
  python3 $R/tools/summarize_pmc.py $(ls $O/raw_${tag}_$c/*counter_collection.csv $O/raw_${tag}_$c/*/*counter_collection.csv 2>/dev/null | head -1) --prefix "" --marker $marker > $O/${tag}_$c.json
  rm -rf $O/raw_${tag}_$c
done
ls -la $O/${tag}_*.json
