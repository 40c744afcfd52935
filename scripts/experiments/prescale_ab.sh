export PYTHONPATH=.
timeout 600 python -m pytest tests/test_gpu_wan_kernels.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -3
for w in c3 c4 c5; do for f in 0 1; do
  ALG_ATTN_PRESCALE=$f timeout 700 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${w}_pre$f.json 2> gpurun_out/bench_${w}_pre$f.err
  python -c "
import json
d=json.loads(open('gpurun_out/bench_${w}_pre$f.json').read().strip().splitlines()[-1])
print('$w prescale=$f', round(d['value'],4), round(d['ms_per_step']), round(d['roofline']['achieved']), round(d['roofline']['frac'],4))"
done; done
