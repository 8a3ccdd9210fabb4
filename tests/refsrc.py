"""Imports the reference's own model source (pc/model/*.py under /root/reference, UNMODIFIED) over a
stand-in `MinkowskiEngine` module.  Only usable where /root/reference exists (this container); the
tests that need it skip elsewhere and work from the fixtures committed under tests/golden/."""
import collections
import collections.abc
import importlib
import os
import sys
import zlib

import torch

REF_PC = "/root/reference/pretrain/pointcontrast"


def reference_available():
  return os.path.isfile(os.path.join(REF_PC, "model", "res16unet.py"))


def _purge():
  for k in [k for k in sys.modules if k == "model" or k.startswith("model.") or k == "MinkowskiEngine" or
            k.startswith("MinkowskiEngine.")]:
    del sys.modules[k]


def import_reference_models(install):
  """`install()` must register sys.modules["MinkowskiEngine"] (+ ".MinkowskiOps").  Returns the reference's
  `model` package (model.load_model, model.res16unet, ...), bound to that stand-in."""
  assert reference_available(), "%s is not present" % REF_PC
  _purge()
  install()
  if not hasattr(collections, "Sequence"):  # removed in Python 3.10; pc/model/modules/common.py:77 still uses it
    collections.Sequence = collections.abc.Sequence
  sys.path.insert(0, REF_PC)
  try:
    pkg = importlib.import_module("model")
    importlib.import_module("model.res16unet")
  finally:
    sys.path.remove(REF_PC)
  mods = {k: v for k, v in sys.modules.items() if k == "model" or k.startswith("model.")}
  for m in mods.values():
    assert os.path.abspath(m.__file__).startswith(REF_PC), m.__file__
  _purge()  # the returned module objects stay alive; the names are free for the next stand-in
  return pkg


def fill_deterministic(model, seed=0):
  """Seeded, name-keyed parameter values (same numbers for the oracle, the reference source and the device model,
  on any host): conv kernels / biases ~ U(-b, b) with ME's bound b = 1/sqrt(fan), BN gamma = 1 + 0.2 U(-1,1),
  beta = 0.2 U(-1,1); running statistics stay at their initial values."""
  with torch.no_grad():
    for name, p in sorted(model.state_dict().items()):
      if not p.dtype.is_floating_point or "running_" in name:
        continue
      g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
      u = torch.rand(p.shape, generator=g, dtype=torch.float32) * 2 - 1
      if name.endswith(".bn.weight"):
        v = 1 + 0.2 * u
      elif name.endswith(".bn.bias"):
        v = 0.2 * u
      else:
        fan = p.numel() // p.shape[-1] if name.endswith(".kernel") else p.shape[-1]
        v = u / max(fan, 1) ** 0.5
      p.copy_(v.to(p.dtype))


def import_reference_trainer(install):
  """pc/lib/ddp_trainer.py (+ its lib.* siblings and the model package), imported unmodified with `MinkowskiEngine`
  resolved through `install()`.  The two packages it imports that are absent here and irrelevant to the arithmetic
  (omegaconf, tensorboardX) are satisfied with empty stand-ins; `torch.autograd.set_detect_anomaly(True)`, which the
  file switches on at import (pc/lib/ddp_trainer.py:36), is switched off again."""
  import types
  assert reference_available()
  _purge()
  for k in [k for k in sys.modules if k == "lib" or k.startswith("lib.")]:
    del sys.modules[k]
  install()
  if not hasattr(collections, "Sequence"):
    collections.Sequence = collections.abc.Sequence
  stubs = {}
  for name, attr in (("omegaconf", "OmegaConf"), ("tensorboardX", "SummaryWriter")):
    if name not in sys.modules:
      m = types.ModuleType(name)
      setattr(m, attr, type(attr, (), {}))
      sys.modules[name] = stubs[name] = m
  sys.path.insert(0, REF_PC)
  try:
    mod = importlib.import_module("lib.ddp_trainer")
  finally:
    sys.path.remove(REF_PC)
    torch.autograd.set_detect_anomaly(False)
    for name in stubs:
      del sys.modules[name]
  assert os.path.abspath(mod.__file__).startswith(REF_PC)
  _purge()
  for k in [k for k in sys.modules if k == "lib" or k.startswith("lib.")]:
    del sys.modules[k]
  return mod
