# round 4: the XCD-contiguous tile walk of the fp32 forward / data-gradient kernels (LEC_CF_XCD=1) against the default
# round-robin order, inside the two-pass bench step: HBM BYTES per family (FETCH_SIZE x 2, WRITE_SIZE; separate --pmc passes) and the step's time /
# the BatchNorm family's in-step time from an unprofiled run.  The program comes directly after `--`.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/xcd_ab
rm -rf $O; mkdir -p $O
ARGS="--launch eager --secondary none --no-cpu-baseline --no-stress --through-trainer 0"
for x in 0 1; do
  export LEC_CF_XCD=$x
  python3 $R/bench.py --steps 12 --warmup 4 --secondary none --no-cpu-baseline --no-stress --through-trainer 0 > $O/bench_xcd$x.json 2> $O/bench_xcd$x.err
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/st_${c}_$x -o f -- python3 $R/bench.py --steps 2 --warmup 1 $ARGS > /dev/null 2> $O/st_${c}_$x.err
    python3 $R/tools/summarize_pmc.py $(ls $O/st_${c}_$x/*counter_collection.csv $O/st_${c}_$x/*/*counter_collection.csv 2>/dev/null | head -1) --prefix "" > $O/st_${c}_$x.json
    rm -rf $O/st_${c}_$x
  done
done
python3 - <<PY
import json, collections
O='$O'
def fam(k):
    if 'lec::bn_' in k: return 'BatchNorm family'
    if 'conv_f32_wgrad' in k: return 'conv weight gradients'
    if 'conv_f32_act' in k: return 'conv forward / data gradient'
    return 'other'
print('| LEC_CF_XCD | ms/step (graph replay) | roofline_bn in-step ms | conv fwd/dgrad read GB | ... written | BatchNorm read GB | wgrad read GB | total read + written GB |')
print('|---|---|---|---|---|---|---|---|')
for x in (0, 1):
    b = json.load(open('%s/bench_xcd%d.json' % (O, x)))
    F = json.load(open('%s/st_FETCH_SIZE_%d.json' % (O, x)))['bytes_per_step']; W = json.load(open('%s/st_WRITE_SIZE_%d.json' % (O, x)))['bytes_per_step']
    f = collections.defaultdict(float); w = collections.defaultdict(float)
    for k, v in F.items(): f[fam(k)] += 2 * v / 1e9
    for k, v in W.items(): w[fam(k)] += v / 1e9
    print('| %d | %.2f | %.2f | %.1f | %.1f | %.1f | %.1f | %.1f |' % (x, b['ms_per_step'], b['roofline_bn']['ms_per_step'], f['conv forward / data gradient'], w['conv forward / data gradient'],
          f['BatchNorm family'], f['conv weight gradients'], sum(f.values()) + sum(w.values())))
PY
