for c in 0 1 2 4 8 16; do
python bench.py --workload resnet50_me --steps 3 --warmup 1 --no-cpu-baseline --chunk $c 2>&1 | grep -v amdgpu | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('chunk', d['config']['chunk_samples'], 'samples/s', d['value'], 'ms', d['ms_per_step'], 'conv frac', r['frac'], {k:(v['launches'], v['achieved']) for k,v in r['by_kernel'].items()}, r['profile_ms'])"
done
