#!/bin/bash
# one extra PMC pass: LDS activity / bank conflicts per kernel (kernel-trace + pmc only, like tools/pmc.sh)
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_lds
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $out/pmc6 -- $BENCH > $out/pmc6.log 2>&1
cd $GRAFT_REPO_ROOT
find $out -type f ! -name '*.csv' ! -name '*.log' -delete
python profiles/summarize.py $(find $out/pmc6 -name '*kernel_trace.csv' | head -1) $(find $out/pmc6 -name '*counter_collection.csv') > $out/summary.txt 2>&1
tail -25 $out/summary.txt; tail -3 $out/pmc6.log | cut -c1-200
