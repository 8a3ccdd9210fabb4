"""Options of the executor / trainer that must not change one bit of a training step:
  * the measurement mode (include/pcmi.h: pcmi_net_time_all / _timed_ms / _timed_groups_ms / _timed_launches -- bench.py's
    families[] and per-layer times): every op of the program comes back with a forward and a backward time and a launch
    count, the convolutions with a weight-gradient time or as part of a grouped launch;
  * the optimiser per gradient bucket (misc.bucket_sgd);
  * the ReLU pattern of the fused BatchNorm outputs as one bit per element for the backward pass (csrc/norm.hip: relu_bits)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _trainer(seed=3, extra=(), pairs=2, crop=0.9):
  from pointcontrast_amd.lib import synthetic
  from pointcontrast_amd.lib.config import get_config
  from pointcontrast_amd.lib.ddp_data_loaders import FixedBatchLoader, default_collate_pair_fn
  from pointcontrast_amd.lib.ddp_trainer import PointNCELossTrainer
  cfg = get_config(["net.model=Res16UNet34C", "misc.nceT=0.4", "misc.npos=512", "opt.lr=0.1", "misc.engine=native"] + list(extra))
  rng = np.random.RandomState(seed)
  batch = default_collate_pair_fn([synthetic.make_pair_item(rng, 0.025, crop=crop) for _ in range(pairs)])
  loader = FixedBatchLoader([batch], batch_size=pairs)
  torch.manual_seed(seed)
  return PointNCELossTrainer(cfg, loader), loader, batch


def _draws(batch, step):
  pp = batch["correspondences"].numpy()
  nq = len(np.unique(pp[:, 0]))
  return dict(uniform=torch.rand(nq, generator=torch.Generator().manual_seed(step)),
              sampled_inds=np.random.RandomState(step).choice(nq, min(512, nq), replace=False))


def test_timing_every_op_leaves_the_step_bit_identical_and_reports_every_op():
  from pointcontrast_amd.lib.timer import AverageMeter, Timer
  runs = []
  for timed in (False, True):
    trainer, loader, batch = _trainer()
    eng = trainer.engine
    if timed:
      eng.time_all(3)
    it, timers = iter(loader), [AverageMeter(), Timer(), Timer()]
    losses = [float(trainer._train_iter(it, timers, draws=_draws(batch, s))["loss"]) for s in range(3)]
    torch.cuda.synchronize()
    if timed:
      recs, groups, counts = eng.timed_ms(3), eng.timed_groups_ms(3), eng.timed_launches(3)
      eng.time_all(0)
      assert len(recs) == 3 and len(recs[0][0]) == eng.n_ops
      for s in range(3):
        fwd, bwd, wg = recs[s]
        cf, cb, cw, cg = counts[s]
        grouped = 0
        for q, o in enumerate(eng._ops):
          assert fwd[q] >= 0 and cf[q] >= 1, "op %d: no forward record in set %d" % (q, s)
          if o["type"] == 0 and o["in_"] == 0:
            assert bwd[q] < 0, "the input layer has no backward-data launch"
          else:
            assert bwd[q] >= 0 and cb[q] >= 1, "op %d: no backward record in set %d" % (q, s)
          if o["type"] == 0:
            if wg[q] >= 0:
              assert cw[q] >= 1
            else:
              grouped += 1
          else:
            assert wg[q] < 0
        # convolutions without a record of their own were launched in groups: at least one grouped launch, timed
        assert (grouped > 0) == (len(groups[s]) > 0), (grouped, groups[s])
        assert all(g > 0 for g in groups[s]) and sum(cg) >= len(groups[s])
      tot = sum(t for t in recs[1][0] if t >= 0) + sum(t for t in recs[1][1] if t >= 0)
      print("Res16UNet34C, 2 pairs: chain ops of set 1 add up to %.2f ms, %d grouped weight-gradient launches" % (tot, len(groups[1])))
    runs.append((losses, trainer.flat.w.clone()))
  assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
  assert torch.equal(runs[0][1], runs[1][1]), "timing the ops changed the weights"


def test_time_ops_accepts_any_op_and_stops():
  trainer, loader, batch = _trainer()
  eng = trainer.engine
  bn = [i for i, o in enumerate(eng._ops) if o["type"] == 1][:3]
  eng.time_ops(bn, n_sets=2)
  from pointcontrast_amd.lib.timer import AverageMeter, Timer
  it, timers = iter(loader), [AverageMeter(), Timer(), Timer()]
  for s in range(2):
    trainer._train_iter(it, timers, draws=_draws(batch, s))
  torch.cuda.synchronize()
  recs = eng.timed_ms(2)
  eng.time_ops([])
  for fwd, bwd, wg in recs:
    assert all(t >= 0 for t in fwd) and all(t >= 0 for t in bwd) and all(t < 0 for t in wg)


def test_per_bucket_sgd_is_bit_identical_to_the_single_launch():
  """misc.bucket_sgd=True (off by default: a measured loss on one GPU, profiles/r06e_*): the optimiser steps every gradient bucket on the communication stream as soon as the
  executor reports it final, beside the rest of the backward pass (GradReducer.after_bucket -> FlatSGD.step_range).
  SGD is elementwise, so three iterations must leave weights and momentum IDENTICAL to the single launch behind the
  pass -- and the hook must really have fired for every bucket."""
  from pointcontrast_amd.lib.timer import AverageMeter, Timer
  runs = []
  for on in (True, False):
    trainer, loader, batch = _trainer(extra=["misc.bucket_sgd=%s" % on])
    assert (trainer.reducer.after_bucket is not None) == on
    calls = []
    if on:
      assert len(trainer.reducer.buckets) >= 3
      orig = trainer.optimizer.step_range
      trainer.optimizer.step_range = lambda lo, hi: (calls.append((lo, hi)), orig(lo, hi))[1]
    it, timers = iter(loader), [AverageMeter(), Timer(), Timer()]
    losses = [float(trainer._train_iter(it, timers, draws=_draws(batch, s))["loss"]) for s in range(3)]
    torch.cuda.synchronize()
    if on:
      assert len(calls) == 3 * len(trainer.reducer.buckets), calls
      assert sorted(calls[:len(trainer.reducer.buckets)]) == sorted((lo, hi) for lo, hi, _ in trainer.reducer.buckets)
    runs.append((losses, trainer.flat.w.clone(), trainer.flat.v.clone()))
  assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
  assert torch.equal(runs[0][1], runs[1][1]), "per-bucket SGD changed the weights"
  assert torch.equal(runs[0][2], runs[1][2]), "per-bucket SGD changed the momentum"


@pytest.mark.parametrize("pairs,crop", [(2, 0.9), (4, None)])
def test_relu_bits_leave_the_step_bit_identical(pairs, crop, monkeypatch):
  """PCMI_BN_RELU_BITS (default on): the forward BatchNorm apply kernels write (y > 0) as one bit per element beside y and
  the backward statistics / apply kernels read that instead of the fp32 tensor -- 4 bytes where they read 128.  Same
  predicate on the same stored values, so two iterations must leave IDENTICAL weights and momentum with the bits and with
  the fp32 masks.  Second case: the full bench batch (175k rows at level 1: the 48-register statistics kernel, the
  three-launch kernels, the one-launch kernels of the coarse levels all on their bit-reading paths)."""
  from pointcontrast_amd.lib.timer import AverageMeter, Timer
  runs = []
  for bits in ("1", "0"):
    monkeypatch.setenv("PCMI_BN_RELU_BITS", bits)  # read per pass by the executor
    trainer, loader, batch = _trainer(pairs=pairs, crop=crop)
    it, timers = iter(loader), [AverageMeter(), Timer(), Timer()]
    losses = [float(trainer._train_iter(it, timers, draws=_draws(batch, s))["loss"]) for s in range(2)]
    torch.cuda.synchronize()
    runs.append((losses, trainer.flat.w.clone(), trainer.flat.v.clone(), batch["sinput0_C"].shape[0] + batch["sinput1_C"].shape[0]))
  print("rows per joint pass:", runs[0][3])
  assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
  assert torch.equal(runs[0][1], runs[1][1]), "the bit pattern changed the weights"
  assert torch.equal(runs[0][2], runs[1][2]), "the bit pattern changed the momentum"


@pytest.mark.parametrize("switch", ["PCMI_FWD_BRANCH", "PCMI_X3_PACK_SIDE"])
@pytest.mark.parametrize("pairs,crop", [(2, 0.9), (4, None)])
def test_residual_branches_on_the_side_stream_leave_the_step_bit_identical(pairs, crop, switch, monkeypatch):
  """PCMI_FWD_BRANCH (default on): the 1x1 convolution + BatchNorm that compute a BasicBlock's residual (pc/model/resnet.py:
  downsample) run on the executor's side stream beside conv1 / bn1 / conv2 of the block and the BatchNorm that adds the
  residual waits for them.  PCMI_X3_PACK_SIDE (default on): the forward orientations of the split-precision weights are
  packed on that stream too and the pass waits for them in front of the first convolution that reads a pack.  Same kernels on
  the same operands in the same per-tensor order: two iterations must leave IDENTICAL losses, weights, momentum and BatchNorm
  running estimates with the switch on and off."""
  from pointcontrast_amd.lib.timer import AverageMeter, Timer
  runs = []
  for mode in ("1", "0"):
    monkeypatch.setenv(switch, mode)  # read per pass by the executor
    trainer, loader, batch = _trainer(pairs=pairs, crop=crop)
    it, timers = iter(loader), [AverageMeter(), Timer(), Timer()]
    losses = [float(trainer._train_iter(it, timers, draws=_draws(batch, s))["loss"]) for s in range(3)]
    torch.cuda.synchronize()
    bufs = torch.cat([b.detach().flatten().float() for b in trainer.model.buffers()])
    runs.append((losses, trainer.flat.w.clone(), trainer.flat.v.clone(), bufs.clone()))
  assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
  assert torch.equal(runs[0][1], runs[1][1]), "%s changed the weights" % switch
  assert torch.equal(runs[0][2], runs[1][2]), "%s changed the momentum" % switch
  assert torch.equal(runs[0][3], runs[1][3]), "%s changed the BatchNorm running estimates" % switch
