"""Starts the ranks of a single-node run: one process per GPU.

Surface of pc/lib/multiprocessing.py (``multi_proc_run(num_proc, fun, fun_args, fun_kwargs)``, :36-56; the child
wrapper ``run``, :19-33) and of pc/lib/error_handler.py (``ChildException``: a child's traceback re-raised in the
parent, the siblings stopped, :14-59).  Differences, all forced by the platform:
  * children are *spawned*, not forked: a forked child of a process that has touched HIP cannot use the device;
  * the rendezvous is RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT in each child's
    environment (what torch.distributed.run would set) instead of the reference's fixed tcp://localhost:10001, so two
    runs on one host do not collide and the same worker code runs under either launcher;
  * the parent watches process sentinels and a pipe per child (no signal handler, no listener thread): the first child
    to fail is reported, the others are terminated by PID, and ChildException carries the traceback text.  A child
    that dies without a Python exception (a fault in native code, a kill) is reported by its exit code.
"""
import multiprocessing as mp
import os
import socket
import sys
import traceback
from multiprocessing.connection import wait as mp_wait


class ChildException(Exception):
  """An exception of a child process, re-raised in the parent with the child's traceback as its message."""

  def __init__(self, child_trace, rank=None, exitcode=None):
    super().__init__(child_trace)
    self.rank, self.exitcode = rank, exitcode


def free_port():
  with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


def rank_variables(rank, world_size, port):
  """The rendezvous variables of rank `rank` (single node: LOCAL_RANK == RANK), as torch.distributed.run sets them."""
  return dict(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size), LOCAL_WORLD_SIZE=str(world_size),
              MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))


def rank_environment(rank, world_size, port, base=None):
  """A full environment for rank `rank`: `base` (default: this process's) + the rendezvous variables."""
  env = dict(os.environ if base is None else base)
  env.update(rank_variables(rank, world_size, port))
  env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
  return env


def run(proc_rank, world_size, port, error_pipe, fun, fun_args, fun_kwargs, init_group=True):
  """Child side: rendezvous environment, process group, the function, teardown; a traceback goes to the parent."""
  from . import distributed as du
  code = 0
  try:
    os.environ.update(rank_variables(proc_rank, world_size, port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if init_group:
      du.init_process_group(proc_rank, world_size)
    fun(*fun_args, **fun_kwargs)
  except KeyboardInterrupt:
    pass  # stopped by the parent
  except SystemExit as e:  # sys.exit(n) inside fun: not a failure by itself -- its code is the child's exit code
    code = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
    if code != 0:
      try:
        error_pipe.send("rank %d called sys.exit(%r)" % (proc_rank, e.code))
      except (OSError, ValueError):
        pass
  except BaseException:
    code = 1
    try:
      # (the parent reads the pipe WHILE it waits for the process -- mp_wait on sentinels and pipes -- so a traceback
      #  larger than the pipe buffer does not block this send for ever)
      error_pipe.send(traceback.format_exc())
    except (OSError, ValueError):
      pass
  if code != 0:
    try:
      du.destroy_process_group()
    finally:
      sys.stdout.flush()
      sys.stderr.flush()
      error_pipe.close()
      os._exit(code)  # not sys.exit: a rank stuck in a collective's helper threads must not keep the process alive
  try:
    du.destroy_process_group()
  finally:
    error_pipe.close()


def _stop(procs):
  for p in procs:
    if p.is_alive():
      p.terminate()  # SIGTERM to that PID only
  for p in procs:
    p.join(10)
    if p.is_alive():
      p.kill()
      p.join()


def multi_proc_run(num_proc, fun, fun_args=(), fun_kwargs=None, init_group=True):
  """Runs fun(*fun_args, **fun_kwargs) in `num_proc` spawned processes with the process group initialised
  (nccl = RCCL on a GPU box, gloo otherwise); returns when all have finished, raises ChildException as soon as one
  fails (the others are terminated).  `fun` must be importable from the child (module-level)."""
  ctx = mp.get_context("spawn")
  port = free_port()
  procs, pipes = [], []
  for i in range(num_proc):
    rx, tx = ctx.Pipe(duplex=False)
    p = ctx.Process(target=run, args=(i, num_proc, port, tx, fun, tuple(fun_args), dict(fun_kwargs or {}), init_group))
    p.start()
    tx.close()  # the child's end stays open in the child only: EOF on rx = the child is gone
    procs.append(p)
    pipes.append(rx)
  pending = dict((p.sentinel, i) for i, p in enumerate(procs))
  by_pipe = dict((rx, i) for i, rx in enumerate(pipes))
  traces = {}  # rank -> what its pipe delivered (received as soon as it is readable, not after the child has exited)
  open_pipes = set(pipes)

  def drain(rx):
    i = by_pipe[rx]
    try:
      traces[i] = rx.recv()
    except (EOFError, OSError):
      pass  # closed without a message: a clean exit, or a death in native code
    open_pipes.discard(rx)

  try:
    while pending:
      for s in mp_wait(list(pending) + list(open_pipes)):
        if s in by_pipe:
          if s in open_pipes:
            drain(s)
          continue
        i = pending.pop(s)
        p = procs[i]
        p.join()
        if p.exitcode != 0:
          if pipes[i] in open_pipes and pipes[i].poll(0.5):
            drain(pipes[i])
          _stop(procs)
          raise ChildException(traces.get(i) or "rank %d exited with code %s without a Python traceback" % (i, p.exitcode),
                               rank=i, exitcode=p.exitcode)
  except BaseException:
    _stop(procs)
    raise
  finally:
    for rx in pipes:
      rx.close()


def launch_script_ranks(num_proc, argv, env=None):
  """The same for a *script*: starts `num_proc` copies of ``python argv...`` with the rendezvous environment of their
  rank and waits.  Returns 0 when all exit 0; else stops the others and returns the failing rank's exit code (its
  stderr -- shared with the parent -- already carries the traceback).  bench.py uses this for ``--gpus N`` without an
  external launcher: rank 0's stdout is the parent's, so the one JSON line arrives where the caller reads it."""
  import subprocess
  import time
  port = free_port()
  procs = [subprocess.Popen([sys.executable] + list(argv), env=rank_environment(i, num_proc, port, base=env))
           for i in range(num_proc)]
  rc = 0
  try:
    live = set(range(num_proc))
    while live and rc == 0:
      for i in sorted(live):
        r = procs[i].poll()
        if r is not None:
          live.discard(i)
          if r != 0:
            rc = r
            print("[launch] rank %d exited with code %d: stopping the other ranks" % (i, r), file=sys.stderr, flush=True)
            break
      if live and rc == 0:
        time.sleep(0.05)
  finally:
    for p in procs:
      if p.poll() is None:
        p.terminate()
    for p in procs:
      try:
        p.wait(10)
      except subprocess.TimeoutExpired:
        p.kill()
        p.wait()
  return rc
