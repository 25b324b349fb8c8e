#!/bin/bash
# PMC comparison of stage B variants
cd /tmp && export TMPDIR=/tmp
for L in libowwhip_b2.so libowwhip_b2m.so; do
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
    out=/tmp/pb/$L/$(echo $grp | cut -c1-12)
    mkdir -p $out
    OWW_LIB=$GRAFT_REPO_ROOT/openwakeword_amd/$L rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-parity --no-extras > $out/log.txt 2>&1
    f=$(find $out -name '*counter_collection.csv' | head -1)
    python - "$f" "$L" <<'PY'
import csv,sys,collections
f,L=sys.argv[1],sys.argv[2]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name']
    if 'RCfg<24, 48' not in k: continue
    if int(r['Grid_Size'])<8000000: continue
    acc[k[:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(L, {c: '%.3e'%(sum(x)/len(x)) for c,x in v.items()})
PY
  done
done
