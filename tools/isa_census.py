"""Diagnostic (CPU): instruction classes per basic block of one kernel in a -save-temps assembly (tools/kernel_resources.sh prints the
directory): how many VALU / transcendental / packed / MFMA / LDS / memory / scalar instructions a wave issues where.

    python tools/isa_census.py <dir with *gfx950.s> <mangled-name substring> [min block size]
"""
import glob, re, sys
from collections import Counter
s = open(glob.glob(sys.argv[1] + '/*gfx950.s')[0]).read()
i = s.index('\n' + [l for l in s.split('\n') if l.startswith('_Z') and sys.argv[2] in l and l.rstrip().endswith(sys.argv[2] + l[l.index(sys.argv[2]) + len(sys.argv[2]):].rstrip())][0].split(':')[0] + ':')
body = s[i:s.index('s_endpgm', i)]
minb = int(sys.argv[3]) if len(sys.argv) > 3 else 15
blocks, cur = [], ['entry', []]
for l in body.split('\n'):
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append(cur); cur = [l.strip()[:60], []]
    else:
        cur[1].append(l)
blocks.append(cur)
tot = Counter()
for name, ls in blocks:
    ins = [x.strip() for x in ls if x.strip() and not x.strip().startswith((';', '.'))]
    c = Counter()
    for t in ins:
        op = t.split()[0]
        k = ('mfma' if op.startswith('v_mfma') else 'trans' if op.startswith(('v_exp', 'v_rcp', 'v_rsq', 'v_log', 'v_sqrt')) else 'pk' if op.startswith('v_pk')
             else 'valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else 'wait' if op.startswith('s_waitcnt') else 'salu' if op.startswith('s_')
             else 'vmem' if op.startswith(('global', 'buffer', 'flat')) else 'other')
        c[k] += 1
    tot += c
    if len(ins) >= minb:
        print(f"{name:60s} {len(ins):4d} {dict(c)}")
print('static total', sum(tot.values()), dict(tot))
