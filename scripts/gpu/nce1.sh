#!/bin/bash
# NCE on the matrix cores + LDS-staged scatter: parity, timing A/B, step A/B
mkdir -p gpurun_out/r04n; O=gpurun_out/r04n
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "nce_parity or gather_scatter or trainer_iteration or hardest" 2>&1 | tail -40 > $O/pytest_nce.log
for n in 4096 1024 8192; do
  timeout 120 env PYTHONPATH=. python scripts/loss_bench.py $n >> $O/loss_bench.txt 2>&1
  PCMI_NCE_X3=0 timeout 120 env PYTHONPATH=. python scripts/loss_bench.py $n >> $O/loss_bench_valu.txt 2>&1
done
for i in 1; do
  timeout 200 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('x3  ', d['value'], d['ms_per_step'])" >> $O/ab.txt
  PCMI_NCE_X3=0 timeout 200 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('valu', d['value'], d['ms_per_step'])" >> $O/ab.txt
done
cat $O/pytest_nce.log $O/loss_bench.txt $O/loss_bench_valu.txt $O/ab.txt
