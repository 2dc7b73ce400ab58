# kernel trace of the fp32 bench step -> the intervals with no convolution in flight (tools/step_gaps_rocpd.py) + the steady-state table.
# MODE=graph (default: the replayed hipGraph, what the headline runs) or eager.  The program comes directly after `--`.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/gaps
rm -rf $O; mkdir -p $O
MODE="${MODE:-graph}"
ARGS="--steps 6 --warmup 3 --secondary none --no-cpu-baseline --no-stress --through-trainer 0 --through-trainer-files 0 --launch $MODE"
rocprofv3 --kernel-trace -d $O/kt -o t -- python3 $R/bench.py $ARGS > $O/kt.json 2> $O/kt.err
DB=$(ls $O/kt/*/*.db $O/kt/*.db 2>/dev/null | head -1)
python3 $R/tools/step_gaps_rocpd.py $DB --steps 3 --grid 16384 --top 14 > $O/step_gaps_$MODE.md 2>> $O/kt.err
python3 $R/tools/summarize_rocpd.py $DB --steps 3 --grid 16384 > $O/steady_$MODE.md 2>> $O/kt.err
rm -rf $O/kt
head -60 $O/step_gaps_$MODE.md | cut -c1-1200; tail -3 $O/kt.err
