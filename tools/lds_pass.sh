#!/bin/bash
# one PMC pass (LDS counters only) over a short bench run, per library variant: tools/lds_pass.sh <tag> [libs...]
tag=${1:-lds}; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  n=$(basename $L .so)
  OWW_LIB=$GRAFT_REPO_ROOT/$L rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $out/$n -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-parity --no-extras > $out/$n.log 2>&1
  python - $out/$n <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in acc.items():
    if v.get('SQ_LDS_IDX_ACTIVE', 0) > 0:
        print(sys.argv[1].split('/')[-1], k, 'conflict/active = %.3f' % (v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE']), 'active', v['SQ_LDS_IDX_ACTIVE'])
PY
done
find $out -type f ! -name '*.log' -delete
