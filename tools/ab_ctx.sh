#!/bin/bash
# Diagnostic (GPU box): same-box A/B of the context encoder between libhvla.so variants (tools/build_variants.sh hypernet ...):
# rocprofv3 kernel statistics of ten create_tasks calls at B = 256 per variant, twice, interleaved.
cp hyper-vla_amd/lib/libhvla.so /tmp/libhvla_orig.so
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in "$@"; do
  cp $GRAFT_REPO_ROOT/tmp_variants/lib_$v.so $GRAFT_REPO_ROOT/hyper-vla_amd/lib/libhvla.so
  rm -rf /tmp/ctxp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ctxp -- python3 $GRAFT_REPO_ROOT/tools/create_tasks_profile.py 256 > /dev/null 2>&1
  python3 - "$v" <<'PY'
import csv, glob, sys
for f in glob.glob('/tmp/ctxp/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'ctx_encoder' in r['Name'] or 'weightgen' in r['Name']:
            print(sys.argv[1], r['Name'][:40], 'calls', r['Calls'], 'avg us %.1f' % (float(r['AverageNs']) / 1e3))
PY
done; done
cp /tmp/libhvla_orig.so $GRAFT_REPO_ROOT/hyper-vla_amd/lib/libhvla.so
