#!/bin/bash
# Last checks of a round on one GPU box: the GPU test tier, the PMC passes of the final build (their summaries carry the build's source
# hash: bench.py prices its composite only from them), then the default bench line.  usage (via gpurun): bash tools/final_check.sh <tag>
tag=${1:-r05h}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out; cd $GRAFT_REPO_ROOT; ulimit -n 65535 2>/dev/null
(time timeout 900 python -m pytest tests -m gpu -x -q --durations=5) > $out/pytest_gpu_final.log 2>&1; tail -10 $out/pytest_gpu_final.log
bash tools/pmc.sh ${tag}pmc > $out/pmc.log 2>&1; tail -4 $out/pmc.log | cut -c1-200
cp gpurun_out/${tag}pmc/instr.json profiles/_instr_tmp.json; cp gpurun_out/${tag}pmc/traffic.json profiles/_traffic_tmp.json
# (the bench run below must find the fresh passes under profiles/: copy them in place on the box)
cp gpurun_out/${tag}pmc/instr.json profiles/r05_instr.json; cp gpurun_out/${tag}pmc/traffic.json profiles/r05_traffic.json; rm -f profiles/_instr_tmp.json profiles/_traffic_tmp.json
timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err
python - $out/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).readline())
print(d["ms_per_step"], d["value"], d["kernel_ms"], "c1", d["configs"]["c1_4096x1"]["ms_per_step"], "roof", d["roofline"]["kernel"], d["roofline"]["frac"])
print(json.dumps(d["roofline"]["composite"])[:300])
PY
find gpurun_out/${tag}pmc -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
