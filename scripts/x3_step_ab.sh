#!/bin/bash
# Whole-step A/B of the split-precision conv kernel (PCMI_CONV16_X3) under the switches that change how it shares the chip
# with the weight-gradient stream; one bench line per configuration in gpurun_out/$TAG/ab.txt, plus a rocprofv3 kernel
# trace of the X3 step.
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r02c}
O=gpurun_out/$TAG
mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
run() {  # label, env...
  local label=$1; shift
  local line
  line=$(env "$@" timeout 120 $B 2>> $O/ab.err | tail -1)
  echo "$label | $(echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'pairs/s', d['ms_per_step'], 'ms/step', d['config'].get('gpu_phase_ms_per_step', ''))" 2>/dev/null || echo "FAILED: $line" | cut -c1-200)" | tee -a $O/ab.txt
}
( cd /tmp && PCMI_CONV16_X3=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_x3" -o bench -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/$O/prof_x3.log" 2>&1 )
run "fp32"                          PCMI_CONV16_X3=0
run "x3"                            PCMI_CONV16_X3=1
run "fp32 no-wgrad (diagnostic)"    PCMI_CONV16_X3=0 PCMI_DEBUG_SKIP_WGRAD=1
run "x3   no-wgrad (diagnostic)"    PCMI_CONV16_X3=1 PCMI_DEBUG_SKIP_WGRAD=1
run "fp32 streamk=0"                PCMI_CONV16_X3=0 PCMI_SPCONV_STREAMK=0
run "x3   streamk=0"                PCMI_CONV16_X3=1 PCMI_SPCONV_STREAMK=0
run "x3   reg-staged weights"       PCMI_CONV16_X3=1 PCMI_X3_DMA=0
run "x3   maxnt=2"                  PCMI_CONV16_X3=1 PCMI_X3_MAXNT=2
run "fp32 wgrad on the main stream" PCMI_CONV16_X3=0 PCMI_WGRAD_SIDE_STREAM=0
run "x3   wgrad on the main stream" PCMI_CONV16_X3=1 PCMI_WGRAD_SIDE_STREAM=0
echo done
