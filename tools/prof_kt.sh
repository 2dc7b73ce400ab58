# Kernel trace of a bench.py step, summarised per step (one parameterised script: replaces prof_round{2..5}.sh's per-round copies).
#   bash tools/prof_kt.sh <tag> <marker kernel> <steps summarised> <trailing steps skipped> [bench.py args...]
# writes gpurun_out/prof/<tag>_steady_state.md (+ <tag>.json, the bench line of the same run).  Eager launches (a replayed graph shows the same kernels);
# the program comes directly after `--` (never a wrapper: the profiler's preloaded library has initialised the GPU).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
tag=$1; marker=$2; nsteps=$3; skip=$4; shift 4
COMMON="--secondary none --no-cpu-baseline --no-stress --through-trainer 0 --through-trainer-files 0 --launch eager"
rm -rf $O/kt_$tag
rocprofv3 --kernel-trace -d $O/kt_$tag -o t -- python3 $R/bench.py $COMMON "$@" > $O/$tag.json 2> $O/$tag.err
DB=$(ls $O/kt_$tag/*/*.db $O/kt_$tag/*.db 2>/dev/null | head -1)
TF=$(python3 -c "import json; d=json.load(open('$O/$tag.json')); r=d.get('roofline_conv') or {}; print(r.get('alg_flops_per_step', 0)/1e12)")
PK=$(python3 -c "import json; d=json.load(open('$O/$tag.json')); print(2500.0 if d.get('dtype') == 'bf16' else 157.3)")   # dense matrix peak of the step's arithmetic type
python3 $R/tools/summarize_rocpd.py $DB --steps $nsteps --skip-last $skip --marker $marker --grid 0 --conv-tflop-per-step $TF --peak-tflops $PK > $O/${tag}_steady_state.md 2>> $O/$tag.err
python3 -c "
import json; d=json.load(open('$O/$tag.json'))
print('\nbench.py of the SAME run (under the profiler, eager launches): %s images/s, ms_per_step %.3f, dtype %s, library_conv_launches %s' % (d['value'], d['ms_per_step'], d['dtype'], d.get('library_conv_launches_per_step')))" >> $O/${tag}_steady_state.md
rm -rf $O/kt_$tag
tail -3 $O/$tag.err
