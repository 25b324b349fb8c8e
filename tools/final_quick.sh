#!/bin/bash
# A short round-end pass on one GPU box when the budget does not cover tools/final_check.sh: the GPU tests of what changed (argument 2 =
# pytest -k expression), the PMC passes + default bench line of the build (as in final_check.sh), and the step with the multiclass
# `timer` head (generic heads kernel) at the full batch.  usage (via gpurun): bash tools/final_quick.sh <tag> '<-k expression>'
tag=${1:-r05q}; sel=${2:-generic}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out; cd $GRAFT_REPO_ROOT; ulimit -n 65535 2>/dev/null
(time timeout 150 python -m pytest tests -m gpu -x -q -k "$sel") > $out/pytest_gpu_selected.log 2>&1
grep -E "passed|failed|error" $out/pytest_gpu_selected.log | tail -2
if ! grep -qE "^[0-9]+ passed" $out/pytest_gpu_selected.log || grep -qE "failed|error" $out/pytest_gpu_selected.log; then tail -30 $out/pytest_gpu_selected.log; echo "TESTS NOT GREEN: stopping"; exit 1; fi
for H in timer alexa,timer; do
  timeout 100 python bench.py --streams 131072 --heads $H --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-parity 2>/dev/null | head -1 | python -c "
import sys, json; d = json.loads(sys.stdin.readline()); print('$H', d['ms_per_step'], d['kernel_ms']['heads'])"
done 2>&1 | tee $out/generic_heads_full_batch.txt
bash tools/pmc.sh ${tag}pmc > $out/pmc.log 2>&1
cp gpurun_out/${tag}pmc/instr.json profiles/r05_instr.json; cp gpurun_out/${tag}pmc/traffic.json profiles/r05_traffic.json
timeout 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err
python - $out/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).readline())
print(d["ms_per_step"], d["value"], d["kernel_ms"], "c1", d["configs"]["c1_4096x1"]["ms_per_step"], "roof", d["roofline"]["kernel"], d["roofline"]["frac"])
print(json.dumps(d["roofline"]["composite"])[:200])
PY
find gpurun_out/${tag}pmc -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find gpurun_out -name '*.db' -delete 2>/dev/null
