#!/bin/bash
# Round 3, GPU call 13: 32 -> 32 channel convolutions with the weights resident in LDS (csrc/spconv32r.hip)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r03o}
mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "conv32r or spconv_matches or conv16_matches" 2>&1 | tail -4 | tee $O/tests.txt
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline"
run() { local label=$1; shift; env "$@" 2>> $O/bench.err | tail -1 > "$O/run_${label// /_}.json"; python -c "
import json
try:
  d=json.load(open('$O/run_${label// /_}.json')); r=d['roofline']
  print('$label |', d['value'], 'pairs/s', d['ms_per_step'], 'ms |', ' '.join('%s=%.4f(%.3f)' % (k['kernel'][-30:].replace(' ','_'), k['ms'], k['frac']) for k in d.get('kernels', []) if '32->32' in k['kernel']))
except Exception as e: print('$label failed', e)" | tee -a $O/runs.txt; }
run "tiles a" PCMI_CONV32R=0 timeout 150 $B
run "resident>=8192 a" timeout 150 $B
run "tiles b" PCMI_CONV32R=0 timeout 150 $B
run "resident>=8192 b" timeout 150 $B
run "resident>=2048" PCMI_CONV32R=2048 timeout 150 $B
run "resident>=20000" PCMI_CONV32R=20000 timeout 150 $B
tail -3 $O/bench.err
echo done
