# fused-loss launch time per setting, HIP events over graph-replayed launches.  usage: bash tools/sweep_cone_wpb.sh [VAR=value ...]   (each argument one run; "-" = defaults)
for setting in ${@:--}; do
  if [ "$setting" != "-" ]; then export "$setting"; fi
  python - <<'PY'
import sys, os
sys.path.insert(0, 'tools')
import bench_cone
out=[]
for (b,k,d,n) in [(256,256,10,50000),(4096,256,10,50000),(256,256,128,50000),(256,5,10,2000),(128,5,10,723),(4096,64,128,50000)]:
    r=bench_cone.time_joint(b,k,d,n,b,iters=30); rf=bench_cone.time_joint(b,k,d,n,b,iters=30,grad=False)
    out.append('%dx%dx%d: %.1f us = %.3f of 8 TB/s (fwd %.1f)'%(b,k,d,r['us'],r['GBps']/8000,rf['us']))
print('RUN', {k: v for k, v in os.environ.items() if k.startswith('LEC_JOINT')}, ' | '.join(out))
PY
  if [ "$setting" != "-" ]; then unset "${setting%%=*}"; fi
done
