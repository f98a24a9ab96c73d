#!/bin/bash
# Diagnostic: build libhvla.so variants with -DHVLA_EXP_<name> on ONE source file into tmp_variants/lib_<name>.so, to be copied
# over hyper-vla_amd/lib/libhvla.so on the GPU box for same-box A/B runs:  bash tools/build_variants.sh encoder LNNT ATTNT ...
set -e
cd "$(dirname "$0")/../hyper-vla_amd/csrc"
SRC=$1; shift
mkdir -p ../../tmp_variants
for v in BASE "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -DHVLA_EXP_$v -DHVLA_ABL_$v -c $SRC.hip -o /tmp/${SRC}_$v.o
    OBJ=""; for o in api hypernet encoder policy selftest train t5 resize; do if [ $o = $SRC ]; then OBJ="$OBJ /tmp/${SRC}_$v.o"; else OBJ="$OBJ build/$o.o"; fi; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tmp_variants/lib_$v.so $OBJ ) &
done
wait
ls -la ../../tmp_variants
