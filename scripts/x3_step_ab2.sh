#!/bin/bash
# Second whole-step A/B: combinations of the split-precision kernel's launch switches, each twice (the 20-step runs are
# noisy at the +-3 % level).  One line per run in gpurun_out/$TAG/ab.txt.
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r02d}
O=gpurun_out/$TAG
mkdir -p $O
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline"
run() {
  local label=$1; shift
  local line
  line=$(env "$@" timeout 120 $B 2>> $O/ab.err | tail -1)
  echo "$label | $(echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'pairs/s', d['ms_per_step'], 'ms/step')" 2>/dev/null || echo "FAILED: $line" | cut -c1-200)" | tee -a $O/ab.txt
}
for rep in 1 2; do
run "fp32"                       PCMI_CONV16_X3=0
run "x3"                         PCMI_CONV16_X3=1
run "x3 nt2"                     PCMI_CONV16_X3=1 PCMI_X3_MAXNT=2
run "x3 sk0"                     PCMI_CONV16_X3=1 PCMI_SPCONV_STREAMK=0
run "x3 nt2 sk0"                 PCMI_CONV16_X3=1 PCMI_X3_MAXNT=2 PCMI_SPCONV_STREAMK=0
run "x3 nt2 reg"                 PCMI_CONV16_X3=1 PCMI_X3_MAXNT=2 PCMI_X3_DMA=0
run "x3 nt2 wgrad-prio-normal"   PCMI_CONV16_X3=1 PCMI_X3_MAXNT=2 PCMI_WGRAD_PRIORITY=0
done
echo done
