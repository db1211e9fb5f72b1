# round 4: the -m gpu suite once more at the very last HEAD (code changed after the final-validation job: budgets, env record, tests)
mkdir -p gpurun_out/r04n; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r04n/smoke.txt
timeout 1300 python -m pytest tests -m gpu -q --durations=8 -rf 2>&1 | grep -v "^SKIPPED" | tail -16 | tee gpurun_out/r04n/gpu_suite.txt
python bench.py > gpurun_out/r04n/bench_default.json 2> gpurun_out/r04n/bench_default.err || tail -5 gpurun_out/r04n/bench_default.err
python -c "
import json; d=json.load(open('gpurun_out/r04n/bench_default.json')); print('default bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
