#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_pipeline.py -m gpu -q -x > $O/r2_tests16.log 2>&1; tail -3 $O/r2_tests16.log | cut -c1-300
timeout -k 10 600 python bench.py --steps 100 --no-cpu-baseline --infer-size 0 --infer-large 0 > $O/r2_bench16.log 2>&1; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r2_bench16.log') if x.startswith('{"metric"')]
d=json.loads(l[0]); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], 'bf16', d['train_bf16']['value'], d['train_bf16']['ms_per_step'])
PY
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/$O/trace_a -o t -- python3 $R/bench.py --steps 10 --warmup 2 --graph off --no-cpu-baseline --infer-size 0 --infer-large 0 --also-dtype none --event-steps 0 > $R/$O/trace_a.log 2>&1
python3 $R/scratch/kstats.py $(ls $R/$O/trace_a/*/*.db $R/$O/trace_a/*.db 2>/dev/null | head -1) 17 14 | cut -c1-150
rm -rf $R/$O/trace_a
