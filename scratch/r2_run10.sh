#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/r2_gpu_tests10.log 2>&1; tail -6 $O/r2_gpu_tests10.log | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 600 python bench.py --no-cpu-baseline --infer-size 0 --infer-large 0 --steps 100 > $O/r2_bench10.log 2>&1; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r2_bench10.log') if x.startswith('{"metric"')]
if not l: print(open('gpurun_out/r2_bench10.log').read()[-2000:])
else:
    d=json.loads(l[0])
    print({k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step','kernel_launches_per_step')}, d['roofline']['frac'], d['train_bf16']['value'])
PY
timeout -k 10 300 python bench.py --no-cpu-baseline --infer-size 0 --infer-large 0 --steps 50 --graph off --also-dtype none 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('eager', d['value'], d['host_enqueue_ms_per_step'])"
