# same-box A/B of the fp32 bench step under environment settings: bash tools/ab_env_step.sh VAR=a VAR=b ...   ("-" = defaults); alternating order is the caller's
for setting in "$@"; do
  if [ "$setting" != "-" ]; then export "$setting"; fi
  python bench.py --steps 30 --warmup 8 --through-trainer 0 --secondary none --no-stress --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('RUN %-28s ms/step %.2f (median %.2f)  conv frac %.4f  bn in-step frac %.4f' % ('$setting', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac'], d['roofline_bn']['frac']))"
  if [ "$setting" != "-" ]; then unset "${setting%%=*}"; fi
done
