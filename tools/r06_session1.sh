#!/bin/bash
# round 6, GPU session 1: box class probe, hand-over pricing, packed-VALU micro-benchmark, PMC passes of configs[1]
out=gpurun_out/r06s1; mkdir -p $out
rocm-smi --showclocks --showpower > $out/smi_idle.txt 2>&1
tools/ubench/pk_ubench > $out/pk_ubench.txt 2>&1
python bench.py --streams 4096 --heads hey_jarvis --steps 1000 --warmup 300 --no-cpu-baseline --no-parity --no-extras > $out/c1.json 2> $out/c1.err
tools/price_handover.sh $out > $out/price.log 2>&1
rocprofv3 -L > $out/counters.txt 2>&1
PMC_EXTRA="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" PMC_EXTRA2="SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAVE_CYCLES" tools/pmc.sh r06c1 --streams 4096 --heads hey_jarvis > $out/pmc_c1.log 2>&1
cp gpurun_out/r06c1/summary.txt $out/pmc_c1_summary.txt 2>/dev/null
cp gpurun_out/r06c1/instr.json $out/pmc_c1_instr.json 2>/dev/null
grep -c . $out/counters.txt; cat $out/pk_ubench.txt; cat $out/price_handover.txt; python -c "import json; d=json.loads(open('$out/c1.json').readline()); print('c1', d['ms_per_step'], d['kernel_ms'])"
