#!/bin/bash
# round 6, GPU session 2: deep weight rings at small batches (OWW_DEEP_WGS), bit-exactness first
out=gpurun_out/r06s2; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -k "ring_depth or large_batch or block_pipelined" 2>&1 | tail -5 > $out/tests.log
cat $out/tests.log
for rep in 1 2; do
for deep in 0 128 256 512 1024; do
  for cfg in "4096 hey_jarvis 1000 300" "16384 alexa,hey_mycroft,hey_jarvis 300 100"; do
    set -- $cfg
    OWW_DEEP_WGS=$deep python bench.py --streams $1 --heads $2 --steps $3 --warmup $4 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms']; print('deep=$deep S=$1 step %.4f  A %.4f B %.4f C %.4f D %.4f E %.4f heads %.4f' % (d['ms_per_step'], k['stageA'], k['stageB'], k['stageC'], k['stageD'], k['stageE'], k['heads']))" | tee -a $out/deep_ring.txt
  done
done
done
