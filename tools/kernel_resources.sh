#!/bin/bash
# Diagnostic (CPU, no GPU needed): registers, spills, scratch and static LDS of every kernel of one source file.
#   bash tools/kernel_resources.sh encoder [name filter] [extra hipcc flags]
set -e
SRC=${1:-encoder}; FILT=${2:-.}; shift || true; shift || true
D=$(mktemp -d); cd $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c /root/repo/hyper-vla_amd/csrc/$SRC.hip -save-temps -o x.o 2>/dev/null
python3 - "$FILT" <<'PY'
import re, sys, glob, subprocess
s = open(glob.glob('*gfx950.s')[0]).read()
md = s[s.index('amdhsa.kernels:'):]
for e in md.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', e).group(1)
    dem = subprocess.run(['/usr/bin/c++filt', name], capture_output=True, text=True).stdout.strip()
    if not re.search(sys.argv[1], dem): continue
    g = lambda k: re.search(r'\.' + k + r':\s+(\d+)', e).group(1)
    print(f"{dem[:100]:100s} vgpr {g('vgpr_count'):>3} sgpr {g('sgpr_count'):>3} spill {g('vgpr_spill_count'):>3} sgpr-spill {g('sgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size')}")
PY
# the policy kernel's tile hand-over must contain the written-out fence (__syncthreads() emits no vmcnt wait on gfx950)
if [ "$SRC" = policy ]; then
  n=$(grep -c "s_waitcnt vmcnt(0) lgkmcnt(0)" *gfx950.s || true)
  echo "policy.hip: $n x 's_waitcnt vmcnt(0) lgkmcnt(0)' in the ISA (the tile fence of acquire(): must be > 0)"
fi
echo "(asm in $D)"
