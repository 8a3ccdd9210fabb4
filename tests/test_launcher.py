"""The rank launcher (pointcontrast_amd/lib/multiprocessing.py = pc/lib/multiprocessing.py:36-56 +
pc/lib/error_handler.py:22-59): N spawned ranks with the rendezvous environment and an initialised process group
(gloo here), a child's traceback re-raised in the parent, the siblings stopped; the script form bench.py uses for a
bare ``--gpus N``; and the entry points' own launch branches (bench.py, pointcontrast_amd/ddp_train.py)."""
import json
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe(out_dir):
  import torch
  import torch.distributed as dist
  rank = dist.get_rank()
  t = torch.tensor([float(rank + 1)])
  dist.all_reduce(t)
  rec = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
  rec.update(sum=float(t), backend=dist.get_backend(), world=dist.get_world_size(), pid=os.getpid())
  with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
    json.dump(rec, f)


def _fail_on_rank1(out_dir):
  import torch.distributed as dist
  with open(os.path.join(out_dir, "pid%d" % dist.get_rank()), "w") as f:
    f.write(str(os.getpid()))
  if dist.get_rank() == 1:
    raise ValueError("boom on rank 1")
  time.sleep(120)  # rank 0 would sit in the next collective for ever: the parent has to stop it


def _die_without_traceback(out_dir):
  import torch.distributed as dist
  if dist.get_rank() == 0:
    os._exit(7)
  time.sleep(120)


def _huge_traceback(out_dir):
  import torch.distributed as dist
  if dist.get_rank() == 1:
    raise ValueError("x" * (1 << 20))  # 1 MiB of message: far more than a pipe buffer holds
  time.sleep(120)


def _exit_zero(out_dir):
  import torch.distributed as dist
  with open(os.path.join(out_dir, "ran%d" % dist.get_rank()), "w") as f:
    f.write("1")
  sys.exit(0)


def _exit_five(out_dir):
  import torch.distributed as dist
  if dist.get_rank() == 0:
    sys.exit(5)
  time.sleep(120)


def _alive(pid):
  try:
    os.kill(pid, 0)
    return True
  except OSError:
    return False


def test_multi_proc_run_sets_up_two_gloo_ranks(tmp_path):
  from pointcontrast_amd.lib import multiprocessing as mpu
  mpu.multi_proc_run(2, fun=_probe, fun_args=(str(tmp_path),))
  recs = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
  for r, rec in enumerate(recs):
    assert rec["RANK"] == rec["LOCAL_RANK"] == str(r) and rec["WORLD_SIZE"] == "2" and rec["world"] == 2
    assert rec["MASTER_ADDR"] == "127.0.0.1" and rec["backend"] == "gloo"
    assert rec["sum"] == 3.0  # 1 + 2: the two ranks really met
  assert recs[0]["MASTER_PORT"] == recs[1]["MASTER_PORT"] and recs[0]["pid"] != recs[1]["pid"]
  assert "WORLD_SIZE" not in os.environ  # the parent's environment is untouched


def test_child_traceback_reaches_the_parent_and_siblings_are_stopped(tmp_path):
  from pointcontrast_amd.lib import multiprocessing as mpu
  t0 = time.time()
  with pytest.raises(mpu.ChildException) as ei:
    mpu.multi_proc_run(2, fun=_fail_on_rank1, fun_args=(str(tmp_path),))
  assert time.time() - t0 < 60, "the parent waited for the sleeping rank"
  assert ei.value.rank == 1 and "ValueError: boom on rank 1" in str(ei.value) and "Traceback" in str(ei.value)
  pid0 = int(open(tmp_path / "pid0").read())
  for _ in range(50):
    if not _alive(pid0):
      break
    time.sleep(0.1)
  assert not _alive(pid0), "rank 0 survived its sibling's failure"


def test_child_that_dies_in_native_code_is_reported_by_exit_code(tmp_path):
  from pointcontrast_amd.lib import multiprocessing as mpu
  with pytest.raises(mpu.ChildException) as ei:
    mpu.multi_proc_run(2, fun=_die_without_traceback, fun_args=(str(tmp_path),))
  assert ei.value.rank == 0 and ei.value.exitcode == 7 and "code 7" in str(ei.value)


def test_traceback_larger_than_the_pipe_buffer_does_not_hang(tmp_path):
  """The parent reads a child's pipe while it waits, not after the child has exited (a 64 KiB pipe
  buffer blocked the child's send for ever and the run hung instead of raising)."""
  from pointcontrast_amd.lib import multiprocessing as mpu
  t0 = time.time()
  with pytest.raises(mpu.ChildException) as ei:
    mpu.multi_proc_run(2, fun=_huge_traceback, fun_args=(str(tmp_path),))
  assert time.time() - t0 < 60
  assert ei.value.rank == 1 and len(str(ei.value)) > (1 << 20) and "ValueError" in str(ei.value)


def test_sys_exit_zero_is_a_clean_exit_and_a_nonzero_code_is_kept(tmp_path):
  from pointcontrast_amd.lib import multiprocessing as mpu
  mpu.multi_proc_run(2, fun=_exit_zero, fun_args=(str(tmp_path),))  # must not raise
  assert os.path.exists(tmp_path / "ran0") and os.path.exists(tmp_path / "ran1")
  with pytest.raises(mpu.ChildException) as ei:
    mpu.multi_proc_run(2, fun=_exit_five, fun_args=(str(tmp_path),))
  assert ei.value.rank == 0 and ei.value.exitcode == 5 and "sys.exit(5)" in str(ei.value)


SCRIPT = """
import json, os, sys, time
r = int(os.environ["RANK"])
json.dump({k: os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")},
          open(os.path.join(sys.argv[1], "r%d.json" % r), "w"))
if sys.argv[2] == "fail" and r == 1:
  sys.exit(3)
if sys.argv[2] == "fail":
  time.sleep(120)
"""


def test_script_launcher_passes_environment_and_exit_codes(tmp_path):
  from pointcontrast_amd.lib import multiprocessing as mpu
  script = tmp_path / "s.py"
  script.write_text(SCRIPT)
  assert mpu.launch_script_ranks(3, [str(script), str(tmp_path), "ok"]) == 0
  recs = [json.load(open(tmp_path / ("r%d.json" % r))) for r in range(3)]
  assert [r["RANK"] for r in recs] == ["0", "1", "2"] and {r["WORLD_SIZE"] for r in recs} == {"3"}
  assert len({r["MASTER_PORT"] for r in recs}) == 1 and {r["MASTER_ADDR"] for r in recs} == {"127.0.0.1"}
  t0 = time.time()
  assert mpu.launch_script_ranks(2, [str(script), str(tmp_path), "fail"]) == 3
  assert time.time() - t0 < 60


def test_bench_starts_its_own_ranks_when_no_launcher_did(monkeypatch):
  """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: builds once, then launches itself twice."""
  sys.path.insert(0, ROOT)
  import bench
  from pointcontrast_amd.lib import multiprocessing as mpu
  calls = []
  monkeypatch.setattr(mpu, "launch_script_ranks", lambda n, argv, env=None: calls.append((n, list(argv))) or 5)
  monkeypatch.delenv("WORLD_SIZE", raising=False)
  monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"])
  with pytest.raises(SystemExit) as ei:
    bench.main()
  assert ei.value.code == 5  # a failing rank's exit code is the launcher's
  assert calls == [(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"])]


def test_ddp_train_starts_num_gpus_ranks(monkeypatch, built_lib):
  from pointcontrast_amd import ddp_train
  from pointcontrast_amd.lib import multiprocessing as mpu
  calls = []
  monkeypatch.setattr(mpu, "multi_proc_run", lambda n, fun, fun_args=(), fun_kwargs=None: calls.append((n, fun, fun_args)))
  monkeypatch.delenv("WORLD_SIZE", raising=False)
  ddp_train.main(["misc.num_gpus=4", "trainer.batch_size=16"])
  assert calls == [(4, ddp_train.single_proc_run, (["misc.num_gpus=4", "trainer.batch_size=16"],))]
