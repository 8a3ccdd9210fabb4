"""Entry point of the pre-training run (pc/ddp_train.py): one process per GPU.

  single GPU : python -m pointcontrast_amd.ddp_train trainer.trainer=PointNCELossTrainer misc.nceT=0.4
  N GPUs     : python -m pointcontrast_amd.ddp_train misc.num_gpus=N trainer.batch_size=32 ...
               (the reference's form, pc/ddp_train.py:57-59: the entry point starts its own N ranks through
               lib/multiprocessing.multi_proc_run), or under a launcher that sets RANK / WORLD_SIZE:
               python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
               -m pointcontrast_amd.ddp_train trainer.batch_size=32 ...
Overrides use the reference's ``group.key=value`` syntax (scripts/ddp_local.sh)."""
import logging
import os

# compute, plan, pair-forward, weight-gradient and RCCL streams should each get a hardware queue (ROCm default: 4)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import sys

import numpy as np
import torch

from .lib import ddp_trainer, distributed as du, multiprocessing as mpu
from .lib.config import get_config
from .lib.ddp_data_loaders import make_data_loader


def get_trainer(name):
  if name == "HardestContrastiveLossTrainer":
    return ddp_trainer.HardestContrastiveLossTrainer
  if name == "PointNCELossTrainer":
    return ddp_trainer.PointNCELossTrainer
  raise ValueError("Trainer %s not found" % name)


def single_proc_run(overrides):
  """One rank (pc/ddp_train.py:61-72): under multi_proc_run the process group already exists; under an external
  launcher (WORLD_SIZE in the environment) it is created here."""
  logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s", datefmt="%m/%d %H:%M:%S",
                      handlers=[logging.StreamHandler(sys.stdout)])
  config = get_config(overrides)
  torch.manual_seed(config.misc.seed)  # same seed on every rank, pc/ddp_train.py:28-29
  np.random.seed(config.misc.seed)
  world = int(os.environ.get("WORLD_SIZE", 1))
  if world > 1:
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
      du.init_process_group()
    config.misc.num_gpus = world
  os.makedirs(config.misc.out_dir, exist_ok=True)
  os.chdir(config.misc.out_dir)
  loader = make_data_loader(config, config.trainer.batch_size, num_threads=config.misc.train_num_thread)
  trainer = get_trainer(config.trainer.trainer)(config=config, data_loader=loader)
  trainer.train()


def main(argv=None):
  overrides = [a for a in (argv if argv is not None else sys.argv[1:]) if "=" in a]
  n = int(get_config(overrides).misc.num_gpus)
  if "WORLD_SIZE" not in os.environ and n > 1:
    # the reference's launch (pc/ddp_train.py:57-59): build the library once, then one spawned process per GPU; a
    # child's traceback is re-raised here as ChildException and the other ranks are stopped
    from . import build
    build.build_lib()
    mpu.multi_proc_run(n, fun=single_proc_run, fun_args=(overrides,))
    return
  try:
    single_proc_run(overrides)
  finally:
    du.destroy_process_group()


if __name__ == "__main__":
  main()
