#!/bin/bash
# rocm-smi socket power / sclk samples (every ~0.4 s) while a bench.py configuration runs: tools/power_sample.sh <tag> <bench args...>
out=gpurun_out/power; mkdir -p $out; tag=$1; shift
(while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk"; sleep 0.3; done) > $out/smi_$tag.log &
sm=$!
python bench.py "$@" --no-extras --no-cpu-baseline --no-parity > $out/bench_$tag.json 2> $out/bench_$tag.err
kill $sm; wait $sm 2>/dev/null
python - "$out" "$tag" <<'PY'
import json, re, sys
out, tag = sys.argv[1:3]
txt = open(f"{out}/smi_{tag}.log").read()
pw = [float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", txt)]
ck = [int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)]
busy = [p for p in pw if p > 0.6 * max(pw)] if pw else []
bck = [c for c in ck if c > 0.6 * max(ck)] if ck else []
o = json.loads([l for l in open(f"{out}/bench_{tag}.json").read().splitlines() if l.startswith("{")][-1])
print(json.dumps({"tag": tag, "ms_per_step": o["ms_per_step"], "kernel_ms": o.get("kernel_ms"), "n_samples": len(pw),
                  "socket_power_w_busy_mean": round(sum(busy) / max(len(busy), 1), 1), "socket_power_w_max": max(pw or [0]),
                  "sclk_mhz_busy_mean": round(sum(bck) / max(len(bck), 1), 1), "sclk_mhz_max": max(ck or [0])}))
PY
