#!/bin/bash
# What would fusing the front end / stage A launch with stage B save at most?  
# Builds of the same kernels whose A -> B hand-over rows are folded onto 4 MB (L2-resident): bit 0 = stage A's stores, bit 1 = stage B's
# loads, 3 = both.  Every instruction still executes; only the 8 KB written + 8 KB read per stream-step stop reaching HBM.  Alternating
# runs on ONE box (tools/ab.sh order), 131,072 streams x 3 heads; scores of the pricing builds are garbage (--no-parity).
# usage (GPU box): tools/price_handover.sh <outdir>      -- the variant libraries must have been built before the gpurun call:
#   python tools/build_variants.py priceA=OWH_PRICE_HANDOVER=1 priceB=OWH_PRICE_HANDOVER=2 priceAB=OWH_PRICE_HANDOVER=3
out=${1:-gpurun_out/price}; mkdir -p $out
for rep in 1 2 3; do
  for L in libowwhip.so libowwhip_priceA.so libowwhip_priceB.so libowwhip_priceAB.so; do
    OWW_LIB=$PWD/openwakeword_amd/$L python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms']; print('$rep $L step %.4f  A %.4f B %.4f C %.4f D %.4f E %.4f heads %.4f' % (d['ms_per_step'], k['stageA'], k['stageB'], k['stageC'], k['stageD'], k['stageE'], k['heads']))" | tee -a $out/price_handover.txt
  done
done
