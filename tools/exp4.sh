mkdir -p gpurun_out/exp4
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/exp4/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/exp4/pytest.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc > gpurun_out/exp4/bench_dragon.json 2> gpurun_out/exp4/bench_dragon.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/exp4/bench_dragon.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('pipelined'), d.get('batched'))
PY
timeout -k 10 300 python bench.py --workload dragon_4k --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --batch 0 > gpurun_out/exp4/bench_4k.json 2> gpurun_out/exp4/bench_4k.err; echo "bench4k rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/exp4/bench_4k.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('pipelined'))
PY
timeout -k 10 200 python tools/share_all.py --workload dragon --indices 0,3 > gpurun_out/exp4/share.txt 2>&1; tail -5 gpurun_out/exp4/share.txt
