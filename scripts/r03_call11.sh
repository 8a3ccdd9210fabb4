#!/bin/bash
# Round 3, GPU call 11: software-pipelined tile-stationary weight-gradient kernel (operands requested one unit ahead,
# 3 offsets per workgroup) against the previous build (pointcontrast_amd/libpcmi_prev.so).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03l
mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -m gpu -k "wgrad or gradients or trainer_iteration" 2>&1 | tail -3 | tee $O/tests.txt
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline"
run() { local label=$1; shift; env "$@" 2>> $O/bench.err | tail -1 > "$O/run_${label// /_}.json"; python -c "
import json
try:
  d=json.load(open('$O/run_${label// /_}.json')); r=d['roofline']
  print('$label |', d['value'], 'pairs/s', d['ms_per_step'], 'ms | dominant', r['ms'], 'ms |', ' '.join('%s=%.4f' % (k['kernel'][:12].replace(' ','_')+k['kernel'][-22:].replace(' ','_'), k['ms']) for k in d.get('kernels', []) if 'wgrad' in k['kernel']))
except Exception as e: print('$label failed', e)" | tee -a $O/runs.txt; }
run "prev a" PCMI_LIB=pointcontrast_amd/libpcmi_prev.so timeout 150 $B
run "new a" timeout 150 $B
run "prev b" PCMI_LIB=pointcontrast_amd/libpcmi_prev.so timeout 150 $B
run "new b" timeout 150 $B
run "new x3t>=8192" PCMI_WGRAD_X3T=8192 timeout 150 $B
run "new x3t>=4096" PCMI_WGRAD_X3T=4096 timeout 150 $B
tail -3 $O/bench.err
echo done
