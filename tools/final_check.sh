out=$GRAFT_REPO_ROOT/gpurun_out/r05g; mkdir -p $out; cd $GRAFT_REPO_ROOT; ulimit -n 65535 2>/dev/null
(time timeout 900 python -m pytest tests -m gpu -x -q --durations=6) > $out/pytest_gpu_final.log 2>&1; tail -12 $out/pytest_gpu_final.log
timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err
python - $out/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).readline())
print(d["ms_per_step"], d["value"], d["kernel_ms"], "c1", d["configs"]["c1_4096x1"]["ms_per_step"], "roof", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"])
print(json.dumps(d["roofline"]["composite"])[:900])
PY
(timeout 400 python tests/soak_create.py --cycles 2000 --thread-cycles 100) 2>&1 | grep -E "soak" | tail -2
(OWW_GUARD_ALLOC=2 timeout 300 python tests/soak_create.py --cycles 1000 --thread-cycles 50 --max-streams 4096) 2>&1 | grep -E "soak" | tail -2
(OWW_GUARD_ALLOC=1 timeout 200 python tests/soak_create.py --cycles 500 --thread-cycles 0 --max-streams 4096) 2>&1 | grep -E "soak" | tail -2
