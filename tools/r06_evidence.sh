#!/bin/bash
# Round-6 evidence on one GPU box: box-class probe, GPU test tier, default and driver-style bench lines, 50-step kernel traces (3 heads and
# the reference's default six models), the PMC passes of the build.  usage (via gpurun): bash tools/r06_evidence.sh <tag>
tag=${1:-r06}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
ulimit -n 65535 2>/dev/null
bash tools/class_probe.sh
cp gpurun_out/class_probe/probe.txt $out/class_probe.txt
bash tools/pmc.sh ${tag}pmc > $out/pmc.log 2>&1
tail -5 $out/pmc.log | cut -c1-200
cp gpurun_out/${tag}pmc/instr.json profiles/r06_instr.json; cp gpurun_out/${tag}pmc/traffic.json profiles/r06_traffic.json
cp profiles/r06_instr.json profiles/r06_traffic.json $out/
cp gpurun_out/${tag}pmc/summary.txt $out/pmc_summary.txt
find gpurun_out/${tag}pmc/trace -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
(time timeout 1200 python -m pytest tests -m gpu -x -q --durations=8) > $out/pytest_gpu_final.log 2>&1
tail -14 $out/pytest_gpu_final.log
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_20x5.json 2> $out/bench_driver.err
python - $out/bench_default.json $out/bench_driver_20x5.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).readline())
        print(f.split("/")[-1], d["ms_per_step"], d["value"], d["kernel_ms"], "c1", d["configs"]["c1_4096x1"]["ms_per_step"], "c2", d["configs"]["c2_65536x3"]["ms_per_step"],
              "six", d["configs"]["default6_131072x6"]["ms_per_step"], d["configs"]["default6_131072x6"]["parity"]["max_abs_err"],
              "sust", d["sustained"]["ms_per_step"], "fp32", d["fp32_exact"]["ms_per_step"], "vad", d["vad_fused"]["ms_per_step"], "1m", d["resident_1m"].get("ms_per_step"),
              "parity", d["parity"]["max_abs_err"], "roof", d["roofline"]["kernel"], d["roofline"]["frac"], "host", d["host_pcm"]["ms_per_step"],
              "embed", d["embed_clips"]["value"], d["embed_clips"]["roofline"]["frac"], d["embed_clips"]["parity"]["max_abs_err"], "cpu", d["cpu_baseline"]["value"])
        print("   composite", json.dumps(d["roofline"].get("composite"))[:300])
    except Exception as e:
        print(f, "unreadable:", repr(e)[:200])
PY
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace50 -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-parity --no-extras > $out/trace50.log 2>&1 )
find $out/trace50 -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats_50steps.csv
head -12 $out/kernel_stats_50steps.csv | cut -c1-200
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace6 -- python $GRAFT_REPO_ROOT/bench.py --heads alexa,hey_mycroft,hey_jarvis,hey_rhasspy,timer,weather --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-parity --no-extras > $out/trace6.log 2>&1 )
find $out/trace6 -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats_default6_30steps.csv
head -12 $out/kernel_stats_default6_30steps.csv | cut -c1-200
bash tools/power_sample.sh r06big --steps 200 --warmup 50 | tail -1 | tee $out/power.jsonl
bash tools/power_sample.sh r06six --heads alexa,hey_mycroft,hey_jarvis,hey_rhasspy,timer,weather --steps 200 --warmup 50 | tail -1 | tee -a $out/power.jsonl
find $out gpurun_out/${tag}pmc -name '*.db' -delete 2>/dev/null
rm -rf $out/trace50 $out/trace6
du -sh $out
