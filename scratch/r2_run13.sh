#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/r2_gpu_tests13.log 2>&1; tail -5 $O/r2_gpu_tests13.log | cut -c1-400
timeout -k 10 600 python bench.py --steps 100 --no-cpu-baseline --infer-size 0 > $O/r2_bench13.log 2>&1; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r2_bench13.log') if x.startswith('{"metric"')]
if not l: print(open('gpurun_out/r2_bench13.log').read()[-2000:])
else:
    d=json.loads(l[0])
    print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], 'bf16', d['train_bf16']['value'], d['train_bf16']['ms_per_step'])
    print('large', d['inference_large']['value'], d['inference_large']['f16_operands'])
PY
