# fused-loss launch time per waves-per-block setting (0 = the geometry's rule), HIP events over graph-replayed launches: bash tools/sweep_cone_wpb.sh [wpb ...]
for w in ${@:-0 4 8}; do
  LEC_JOINT_WPB=$w python - <<'PY'
import sys, os
sys.path.insert(0, 'tools')
import bench_cone
out=[]
for (b,k,d,n) in [(256,256,10,50000),(4096,256,10,50000),(256,256,128,50000),(256,5,10,2000),(128,5,10,723),(4096,64,128,50000)]:
    r=bench_cone.time_joint(b,k,d,n,b,iters=30); rf=bench_cone.time_joint(b,k,d,n,b,iters=30,grad=False)
    out.append('%dx%dx%d: %.1f us = %.3f of 8 TB/s (fwd %.1f)'%(b,k,d,r['us'],r['GBps']/8000,rf['us']))
print('WPB', os.environ['LEC_JOINT_WPB'], ' | '.join(out))
PY
done
