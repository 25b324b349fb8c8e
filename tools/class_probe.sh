#!/bin/bash
# Which class of box did this gpurun call land on?  (profiles/r04_box_class2.jsonl: on the second class the 4,096 x hey_jarvis step takes
# 0.47-0.57 ms instead of 0.30-0.32 at the same sclk.)  On a second-class box the PMC passes of configs[1] are taken at once -- such a
# box cannot be asked for -- into gpurun_out/class2/ (VERDICT r05 next 4).
out=gpurun_out/class_probe; mkdir -p $out
python bench.py --streams 4096 --heads hey_jarvis --steps 1000 --warmup 300 --no-cpu-baseline --no-parity --no-extras > $out/c1.json 2> $out/c1.err
ms=$(python -c "import json; print(json.loads(open('$out/c1.json').readline())['ms_per_step'])")
echo "class probe: configs[1] step = $ms ms" | tee $out/probe.txt
if python -c "import sys; sys.exit(0 if float('$ms') > 0.42 else 1)"; then
  echo "second-class box: taking the configs[1] PMC passes" | tee -a $out/probe.txt
  mkdir -p gpurun_out/class2
  cp $out/c1.json gpurun_out/class2/c1.json
  rocm-smi --showclocks --showpower > gpurun_out/class2/smi.txt 2>&1
  PMC_EXTRA="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" PMC_EXTRA2="SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAVE_CYCLES" tools/pmc.sh r06c1_class2 --streams 4096 --heads hey_jarvis > gpurun_out/class2/pmc.log 2>&1
  cp gpurun_out/r06c1_class2/summary.txt gpurun_out/class2/pmc_c1_summary.txt 2>/dev/null
  for sw in 512 0; do
    OWW_SMALL_WGS=$sw python bench.py --streams 4096 --heads hey_jarvis --steps 1000 --warmup 300 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('OWW_SMALL_WGS=$sw', d['ms_per_step'], d['kernel_ms'])" | tee -a gpurun_out/class2/ring_ab.txt
  done
fi
