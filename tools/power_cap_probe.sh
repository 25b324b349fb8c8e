#!/bin/bash
# Power-cap probe: the headline step under the board's default power limit and under lower caps, with the
# per-stage kernel times of bench.py's hipEvent pass and rocm-smi's power / clock samples beside it.  If a stage's time follows the
# cap while its instruction stream is unchanged, the stage is bound by power (energy per step), not by issue slots.
#   tools/power_cap_probe.sh <outdir> [caps...]     (caps in watts; 0 = leave the default)
out=${1:-gpurun_out/power}; shift
caps=${@:-0 750 500}
mkdir -p "$out"
rocm-smi --showpower --showclocks --showmaxpower --showperflevel > "$out/smi_before.txt" 2>&1
for cap in $caps; do
  if [ "$cap" != 0 ]; then rocm-smi --setpoweroverdrive "$cap" --autorespond y > "$out/setcap_$cap.txt" 2>&1; fi
  (while true; do date +%s.%N; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk"; sleep 0.4; done) > "$out/smi_$cap.log" &
  sm=$!
  python bench.py --steps 1500 --warmup 200 --no-extras --no-cpu-baseline --no-parity > "$out/bench_cap$cap.json" 2> "$out/bench_cap$cap.err"
  kill $sm; wait $sm 2>/dev/null
done
rocm-smi --resetpoweroverdrive > "$out/reset.txt" 2>&1
rocm-smi --showpower --showmaxpower > "$out/smi_after.txt" 2>&1
