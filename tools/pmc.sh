#!/bin/bash
# PMC passes over a short bench run (one rocprofv3 invocation per counter group: --pmc is never combined with
# tracing domains other than --kernel-trace).  usage: tools/pmc.sh <tag> [bench args...]
tag=${1:-pmc}; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-parity --no-extras $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $BENCH > $out/trace.log 2>&1
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_TC_INST_REQ" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           ${PMC_EXTRA:+"$PMC_EXTRA"} ${PMC_EXTRA2:+"$PMC_EXTRA2"}; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc$i -- $BENCH > $out/pmc$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
find $out -type f ! -name '*.csv' ! -name '*.log' -delete
python profiles/summarize.py $(find $out/trace -name '*kernel_trace.csv' | head -1) $(find $out/pmc* -name '*counter_collection.csv' | sort) > $out/summary.txt 2>&1
python profiles/summarize.py --traffic $out/traffic.json $(find $out/pmc4 $out/pmc5 -name '*counter_collection.csv' | sort)
python profiles/summarize.py --instr $out/instr.json $(find $out/pmc* -name '*counter_collection.csv' | sort)
cat $out/summary.txt
tail -3 $out/trace.log $out/pmc1.log | cut -c1-300
du -sh $out
