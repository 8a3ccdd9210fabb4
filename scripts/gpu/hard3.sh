#!/bin/bash
mkdir -p gpurun_out/r04n; O=gpurun_out/r04n
R=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -k "hardest or scatter or pdist or trainer_iteration or keyset" 2>&1 | tail -8 > $O/pytest_hard3.log
rm -f $O/hard3.txt
for i in 1 2 3 4 5; do
  timeout 200 python bench.py --loss hardest --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hardest', d['value'], d['ms_per_step'], d['config'].get('host_phase_ms_per_step'))" >> $O/hard3.txt
done
timeout 200 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nce', d['value'], d['ms_per_step'], d['config'].get('host_phase_ms_per_step'))" >> $O/hard3.txt
python - >> $O/hard3.txt <<'PY'
import numpy as np, time
for N,k in ((349520,8192),(349480,8192),(840000,4096)):
    t=time.perf_counter()
    for _ in range(5): np.random.choice(N, k, replace=False)
    print('np.random.choice(%d, %d, replace=False): %.2f ms' % (N, k, (time.perf_counter()-t)/5*1e3))
PY
cat $O/pytest_hard3.log $O/hard3.txt
