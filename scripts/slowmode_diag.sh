#!/bin/bash
# Is the occasional slow run of bench.py (whole process ~2x slower, any configuration) a state of the GPU (clock / power /
# temperature) rather than of the process?  Samples sclk, power and temperature from sysfs every 100 ms while bench.py runs
# N times back to back; gpurun_out/$TAG/{runs.txt,gpu_state.tsv}.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r02f}
O=gpurun_out/$TAG
mkdir -p $O
if [ "${RUN_TEST:-1}" = "1" ]; then
  timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "prepacked" > $O/pytest_prepack.log 2>&1
  echo "prepack test exit $?" | tee -a $O/runs.txt; tail -2 $O/pytest_prepack.log
fi
HW=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
DEVD=$(dirname $(dirname $HW) 2>/dev/null)
echo "hwmon $HW" | tee -a $O/runs.txt
ls $HW 2>/dev/null | tr '\n' ' ' >> $O/runs.txt; echo >> $O/runs.txt
( while true; do
    echo -e "$(date +%s.%N)\t$(cat $HW/freq1_input 2>/dev/null)\t$(cat $HW/freq2_input 2>/dev/null)\t$(cat $HW/power1_average 2>/dev/null || cat $HW/power1_input 2>/dev/null)\t$(cat $HW/temp1_input 2>/dev/null)\t$(cat $HW/temp2_input 2>/dev/null)\t$(cat $DEVD/gpu_busy_percent 2>/dev/null)"
    sleep 0.1
  done ) > $O/gpu_state.tsv 2>/dev/null &
POLL=$!
for i in $(seq 1 ${N:-7}); do
  t0=$(date +%s.%N)
  line=$(timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>> $O/diag.err | tail -1)
  t1=$(date +%s.%N)
  echo "run $i start $t0 end $t1 | $(echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'pairs/s', d['ms_per_step'], 'ms/step', d['config'].get('host_phase_ms_per_step',{}).get('forward'))" 2>/dev/null)" | tee -a $O/runs.txt
done
kill $POLL
rocm-smi --showperflevel --showclocks --showpower 2>/dev/null | head -30 >> $O/runs.txt
echo done
