#!/bin/bash
# Same-box alternating comparison of two complete builds on the headline step, configs[1], configs[2] and the default six models:
# tools/ab_builds.sh <out file> [reps]   (every openwakeword_amd/libowwhip*.so takes part)
cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/ab_builds.txt}; mkdir -p $(dirname $out)
run() { OWW_LIB=$PWD/$1 python bench.py $3 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms']; print('$4 %-20s %-5s step %.4f  A %.4f B %.4f C %.4f D %.4f E %.4f heads %.4f' % ('$(basename $1)', '$2', d['ms_per_step'], k['stageA'], k['stageB'], k['stageC'], k['stageD'], k['stageE'], k['heads']))" | tee -a $out; }
for rep in $(seq 1 ${2:-3}); do
for L in openwakeword_amd/libowwhip.so openwakeword_amd/libowwhip_*.so; do
  run $L big "--steps 50 --warmup 10" $rep
  run $L c2 "--streams 65536 --steps 60 --warmup 10" $rep
  run $L c1 "--streams 4096 --heads hey_jarvis --steps 400 --warmup 100" $rep
  run $L six "--heads alexa,hey_mycroft,hey_jarvis,hey_rhasspy,timer,weather --steps 30 --warmup 10" $rep
done
done
