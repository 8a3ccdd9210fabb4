#!/bin/bash
# round-2 GPU session B: kernels changed (BN fused final, wgrad direct/arrive) -> targeted parity, host profile, A/Bs
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo "build failed"; tail -5 gpurun_out/build.log; }
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 -x \
  -k "${TEST_K:-batchnorm or bn_eval or spconv_parity or stem_conv or spconv_golden or engine_matches or adjointness or refsrc or (trainer_iteration and nce) or rccl}" \
  > gpurun_out/pytest_b.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_b.log
grep -E "passed|failed|error|exit|FAILED|Error" gpurun_out/pytest_b.log | tail -15
PCMI_HOST_PROFILE=1 timeout 300 python bench.py --steps 40 --warmup 5 --no-roofline --no-cpu-baseline > gpurun_out/bench_b_default.log 2>&1
tail -1 gpurun_out/bench_b_default.log | cut -c1-300; grep "host profile" gpurun_out/bench_b_default.log | tail -3
PCMI_SPCONV_STREAMK=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline > gpurun_out/bench_b_nosk.log 2>&1
tail -1 gpurun_out/bench_b_nosk.log | cut -c1-300
timeout 300 python - > gpurun_out/cprofile_b.log 2>&1 <<'PY'
import cProfile, pstats, sys, io
sys.argv = ["bench.py", "--steps", "30", "--warmup", "5", "--no-roofline", "--no-cpu-baseline"]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue())
PY
grep -A34 "tottime" gpurun_out/cprofile_b.log | cut -c1-160 | head -40
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_b" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/prof_b.log" 2>&1
cd "$GRAFT_REPO_ROOT"
find gpurun_out/prof_b -name "*kernel_trace*" -size +8M -delete
echo done
