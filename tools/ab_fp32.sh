#!/bin/bash
# same-box alternating A/B of every openwakeword_amd/libowwhip*.so on the exact-fp32 family: tools/ab_fp32.sh <out file> [reps]
cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/ab_fp32.txt}; mkdir -p $(dirname $out)
for rep in $(seq 1 ${2:-3}); do
for L in openwakeword_amd/libowwhip.so openwakeword_amd/libowwhip_*.so; do
  OWW_LIB=$PWD/$L python bench.py --fp32 --steps 15 --warmup 4 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms']; print('$rep %-24s fp32 step %.4f  mel %.4f A %.4f B %.4f C %.4f D %.4f E %.4f heads %.4f' % ('$(basename $L)', d['ms_per_step'], k['mel'], k['stageA'], k['stageB'], k['stageC'], k['stageD'], k['stageE'], k['heads']))" | tee -a $out
done
done
