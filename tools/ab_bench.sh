#!/bin/bash
# Diagnostic (GPU box): same-box A/B of libhvla.so variants built by tools/build_variants.sh.
#   bash tools/ab_bench.sh "<bench.py args>" BASE V1 V2 ...     (each variant twice, interleaved)
ARGS=$1; shift
cp hyper-vla_amd/lib/libhvla.so /tmp/libhvla_orig.so
for rep in 1 2; do
  for v in "$@"; do
    cp tmp_variants/lib_$v.so hyper-vla_amd/lib/libhvla.so
    timeout 300 python bench.py $ARGS --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('$v', 'ms/step', d['ms_per_step'], 'p50', d['p50_step_latency_ms'], {a: round(b,3) for a,b in k.items()})"
  done
done
cp /tmp/libhvla_orig.so hyper-vla_amd/lib/libhvla.so
