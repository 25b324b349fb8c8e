#!/bin/bash
# Round-end evidence on one GPU box: GPU test tier, default and driver-style bench lines, a 50-step kernel trace, the PMC passes,
# the single-event-loop socket load.  usage (via gpurun): bash tools/final_evidence.sh <tag>
tag=${1:-r05}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
ulimit -n 65535 2>/dev/null
(time timeout 900 python -m pytest tests -m gpu -x -q --durations=8) > $out/pytest_gpu_final.log 2>&1
tail -14 $out/pytest_gpu_final.log
timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_20x5.json 2> $out/bench_driver.err
python - $out/bench_default.json $out/bench_driver_20x5.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).readline())
        print(f.split("/")[-1], d["ms_per_step"], d["value"], d["kernel_ms"], "c1", d["configs"]["c1_4096x1"]["ms_per_step"], "c2", d["configs"]["c2_65536x3"]["ms_per_step"],
              "sust", d["sustained"]["ms_per_step"], "fp32", d["fp32_exact"]["ms_per_step"], "vad", d["vad_fused"]["ms_per_step"], "1m", d["resident_1m"].get("ms_per_step"),
              "parity", d["parity"]["max_abs_err"], "roof", d["roofline"]["kernel"], d["roofline"]["frac"], "host", d["host_pcm"]["ms_per_step"],
              "masked50", d["masked_step"]["participation_50pct"]["ms_per_step"], d["masked_step"]["packed"]["participation_50pct"]["ms_per_step"], "cpu", d["cpu_baseline"]["value"])
    except Exception as e:
        print(f, "unreadable:", repr(e)[:200])
PY
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace50 -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-parity --no-extras > $out/trace50.log 2>&1 )
find $out/trace50 -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats_50steps.csv
head -12 $out/kernel_stats_50steps.csv | cut -c1-200
bash tools/pmc.sh ${tag}pmc > $out/pmc.log 2>&1
tail -45 $out/pmc.log | cut -c1-220
for n in 1024 2048; do
  timeout 200 python tools/serve_load.py --clients $n --seconds 12 --procs 6 --out $out/serve_load_final_$n.json > /dev/null 2> $out/serve_load_final_$n.err
  python -c "
import json,sys; d=json.loads(open(sys.argv[1]).readline()); print(d['clients'], d['pump'], d['server_latency_ms'], d['client_latency_ms'], d['backlog_chunks_at_end'], d['bit_exact_sample'])" $out/serve_load_final_$n.json
done
find $out -name '*.db' -delete 2>/dev/null
du -sh $out
