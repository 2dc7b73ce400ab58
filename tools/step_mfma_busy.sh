# Matrix-pipe utilisation of a bench.py step by kernel family: one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) over eager launches; under counter
# collection the dispatches run one at a time, so these are per-kernel figures, not the overlapped step's.
#   bash tools/step_mfma_busy.sh <tag> [bench.py args...]   ->  gpurun_out/mfma/<tag>.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/mfma
mkdir -p $O
tag=$1; shift
rm -rf $O/raw_$tag
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/raw_$tag -o f -- python3 $R/bench.py --steps 2 --warmup 1 --launch eager --secondary none --no-cpu-baseline --no-stress --through-trainer 0 --through-trainer-files 0 "$@" > $O/$tag.out 2> $O/$tag.err
python3 - $(ls $O/raw_$tag/*counter_collection.csv $O/raw_$tag/*/*counter_collection.csv 2>/dev/null | head -1) "$tag" > $O/$tag.md <<'PY'
import csv, sys, collections
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    rows[(r['Dispatch_Id'], r['Kernel_Name'])][r['Counter_Name']] = float(r['Counter_Value'])
def fam(k):
    if 'lec::bn_' in k: return 'BatchNorm family'
    if 'wgrad' in k: return 'convolution weight gradients'
    if 'stem' in k: return 'stem kernels'
    if 'conv' in k and 'lec::' in k: return 'convolution forward / data gradient'
    if 'lec::' in k: return 'other liblecone'
    return 'library / framework'
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for (d, k), c in rows.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
        a = agg[fam(k)]; a[0] += c['SQ_VALU_MFMA_BUSY_CYCLES']; a[1] += c['GRBM_GUI_ACTIVE']; a[2] += 1
print('# Matrix-pipe utilisation by kernel family: %s (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, one MI355X, dispatches serialised by the counter collection)\n' % sys.argv[2])
print('busy = SQ_VALU_MFMA_BUSY_CYCLES / (1 024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), summed over the family\'s dispatches of the whole run (set-up, warm-up and 2 timed eager steps).\n')
print('| kernel family | dispatches | matrix pipe busy |\n|---|---|---|')
for k, (m, g, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('| %s | %d | %.1f %% |' % (k, n, 100.0 * m / (1024.0 * g / 8.0) if g else 0.0))
PY
rm -rf $O/raw_$tag
cat $O/$tag.md
