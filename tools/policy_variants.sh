#!/bin/bash
# Diagnostic: libhvla_bench.so variants that differ in policy.hip only (bench flavour = the time-stamp code compiled in), for
# tools/policy_determinism_probe.py.   bash tools/policy_variants.sh "FLAGS_A" "FLAGS_B" ...   -> tmp_variants/libp_<n>.so
set -e
cd "$(dirname "$0")/../hyper-vla_amd/csrc"
make -s bench >/dev/null
mkdir -p ../../tmp_variants
n=0
for flags in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -DHVLA_BENCH_HOOKS $flags -c policy.hip -o /tmp/policy_v$n.o
    OBJ=""; for o in api hypernet encoder policy selftest train t5 resize; do if [ $o = policy ]; then OBJ="$OBJ /tmp/policy_v$n.o"; else OBJ="$OBJ build_bench/$o.o"; fi; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tmp_variants/libp_$n.so $OBJ; echo "libp_$n.so: $flags" ) &
  n=$((n+1))
done
wait
