cp hyper-vla_amd/lib/libhvla.so /tmp/libhvla_orig.so
for rep in 1 2; do for v in "$@"; do
  cp tmp_variants/lib_$v.so hyper-vla_amd/lib/libhvla.so
  for b in 1 256; do timeout 300 python bench.py --batch $b --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', 'B=$b', 'step', d['ms_per_step'], 'p50', d['p50_step_latency_ms'], 'policy_only', d['policy_only']['ms_per_step'])"; done
done; done
cp /tmp/libhvla_orig.so hyper-vla_amd/lib/libhvla.so
