"""Entry point of the pre-training run (pc/ddp_train.py): one process per GPU.

  single GPU : python -m pointcontrast_amd.ddp_train trainer.trainer=PointNCELossTrainer misc.nceT=0.4
  N GPUs     : python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
               -m pointcontrast_amd.ddp_train misc.num_gpus=N trainer.batch_size=32 ...
Overrides use the reference's ``group.key=value`` syntax (scripts/ddp_local.sh)."""
import logging
import os

# compute, plan, pair-forward, weight-gradient and RCCL streams should each get a hardware queue (ROCm default: 4)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import sys

import numpy as np
import torch

from .lib import ddp_trainer, distributed as du
from .lib.config import get_config
from .lib.ddp_data_loaders import make_data_loader


def get_trainer(name):
  if name == "HardestContrastiveLossTrainer":
    return ddp_trainer.HardestContrastiveLossTrainer
  if name == "PointNCELossTrainer":
    return ddp_trainer.PointNCELossTrainer
  raise ValueError("Trainer %s not found" % name)


def main(argv=None):
  logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s", datefmt="%m/%d %H:%M:%S",
                      handlers=[logging.StreamHandler(sys.stdout)])
  config = get_config([a for a in (argv if argv is not None else sys.argv[1:]) if "=" in a])
  torch.manual_seed(config.misc.seed)  # same seed on every rank, pc/ddp_train.py:28-29
  np.random.seed(config.misc.seed)
  world = int(os.environ.get("WORLD_SIZE", 1))
  if world > 1:
    du.init_process_group()
    config.misc.num_gpus = world
  os.makedirs(config.misc.out_dir, exist_ok=True)
  os.chdir(config.misc.out_dir)
  loader = make_data_loader(config, config.trainer.batch_size, num_threads=config.misc.train_num_thread)
  trainer = get_trainer(config.trainer.trainer)(config=config, data_loader=loader)
  trainer.train()
  du.destroy_process_group()


if __name__ == "__main__":
  main()
