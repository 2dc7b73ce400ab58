# same-box A/B of the fp32 bench step between builds of liblecone.so and environment settings: bash tools/ab_lib_step.sh "<lib or -> [VAR=value ...]" ...
R=$GRAFT_REPO_ROOT
for spec in "$@"; do
  set -- $spec; lib=$1; shift
  ( if [ "$lib" != "-" ]; then export LEC_LIB_PATH=$R/$lib; fi
    for kv in "$@"; do export "$kv"; done
    python bench.py --steps 30 --warmup 8 --through-trainer 0 --secondary none --no-stress --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('RUN %-70s ms/step %.2f (median %.2f)  conv frac %.4f  bn in-step frac %.4f' % ('$spec', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac'], d['roofline_bn']['frac']))" )
done
