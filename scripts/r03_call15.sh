#!/bin/bash
# Round 3, GPU call 15: flakiness check -- the whole GPU suite twice more and smoke() five times on one box
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03v
mkdir -p $O
for i in 1 2; do
  timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_$i.log 2>&1
  echo "suite $i exit $? $(grep -E 'passed|failed' $O/pytest_$i.log | tail -1)" | tee -a $O/summary.txt
  grep -E "^FAILED|^ERROR" $O/pytest_$i.log | head | tee -a $O/summary.txt
done
for i in 1 2 3 4 5; do
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$i.log 2>&1; echo "smoke $i exit $? $(tail -1 $O/smoke_$i.log)" | tee -a $O/summary.txt
done
echo done
