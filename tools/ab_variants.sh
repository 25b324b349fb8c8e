#!/bin/bash
# same-box alternating A/B of every openwakeword_amd/libowwhip*.so: tools/ab_variants.sh <out file> [reps]
cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/ab.txt}; mkdir -p $(dirname $out)
for rep in $(seq 1 ${2:-3}); do
for L in openwakeword_amd/libowwhip.so openwakeword_amd/libowwhip_*.so; do
  OWW_LIB=$PWD/$L python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms']; print('$rep %-28s step %.4f  A %.4f B %.4f C %.4f D %.4f E %.4f heads %.4f' % ('$(basename $L)', d['ms_per_step'], k['stageA'], k['stageB'], k['stageC'], k['stageD'], k['stageE'], k['heads']))" | tee -a $out
done
done
