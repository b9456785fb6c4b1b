#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/r2_gpu_tests9.log 2>&1; tail -6 $O/r2_gpu_tests9.log | cut -c1-400
timeout -k 10 600 python bench.py --cpu-seconds 6 > $O/r2_bench9.log 2>&1; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r2_bench9.log') if x.startswith('{"metric"')]
if not l: print(open('gpurun_out/r2_bench9.log').read()[-2000:])
else:
    d=json.loads(l[0])
    print({k:d[k] for k in ('value','ms_per_step','dtype','host_cpu_ms_per_step','whole_step_mfma_frac')})
    print(d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
    print('bf16:', {k:d['train_bf16'][k] for k in ('value','ms_per_step','final_loss','conv_launches_on_16bit_kernels_per_step')})
    print('infer', d['inference']['value'], 'large', d['inference_large']['value'], d['inference_large']['nms']['ms'])
    print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['inference']['value'])
PY
