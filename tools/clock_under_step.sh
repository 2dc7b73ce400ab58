cd $GRAFT_REPO_ROOT
python bench.py --steps 400 --warmup 10 --through-trainer 0 --secondary none --no-stress --no-cpu-baseline > gpurun_out/clk_bench.json 2> gpurun_out/clk_bench.err &
BP=$!
sleep 14
for i in $(seq 1 30); do
  rocm-smi --showclocks --showpower --showuse --json 2>/dev/null | python -c "
import json,sys
try:
    d=json.load(sys.stdin); c=d.get('card0',{})
    print('SMI', {k:v for k,v in c.items() if any(s in k.lower() for s in ('sclk','mclk','power','busy','fclk'))})
except Exception as e: print('SMI err', e)"
  sleep 0.5
done
wait $BP
python -c "
import json; d=json.load(open('gpurun_out/clk_bench.json')); print('BENCH', d['ms_per_step'])"
