"""SGD on the flat parameter buffer: one fused libpcmi kernel per step instead of ~250 small
torch kernels (torch.optim.SGD at pc/lib/ddp_trainer.py:107-111: momentum = opt.momentum,
weight_decay = opt.weight_decay, dampening 0, no Nesterov; the downstream fine-tuning uses dampening
0.1, downstream/semseg/lib/solvers.py:52-60).  It IS a torch Optimizer, so
torch.optim.lr_scheduler.ExponentialLR (pc/lib/ddp_trainer.py:113) drives its lr and
state_dict() has torch's layout (param_groups + per-parameter momentum_buffer)."""
import torch

from .. import functional as PF


class FlatSGD(torch.optim.Optimizer):

  def __init__(self, flat, lr, momentum=0.0, weight_decay=0.0, grad_scale=1.0, dampening=0.0):
    self.flat = flat
    self.grad_scale = grad_scale
    super().__init__(flat.params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=dampening,
                                       nesterov=False))
    for i, p in enumerate(flat.params):
      self.state[p]["momentum_buffer"] = flat.view(flat.v, i)
    # torch creates a momentum buffer on a parameter's first step as a copy of the gradient (no dampening applied);
    # here the buffers exist (zero-filled) from the start, so "first step" is tracked explicitly
    self._fresh = True
    self._done = []  # [lo, hi) ranges of the flat buffers this step's step_range calls have already updated

  def zero_grad(self, set_to_none=False):
    self.flat.zero_grad()

  def _step_slice(self, lo, hi):
    g = self.param_groups[0]
    PF.sgd_step(self.flat.w[lo:hi], self.flat.g[lo:hi], self.flat.v[lo:hi], g["lr"], g["momentum"], g["weight_decay"],
                self.grad_scale, dampening=g.get("dampening", 0.0), first_step=self._fresh)

  @torch.no_grad()
  def step_range(self, lo, hi):
    """The step for the elements [lo, hi) only, on the CURRENT stream: SGD is elementwise, so a step taken in slices is
    bit for bit the step taken in one launch.  The trainer calls this per gradient bucket as soon as the bucket is final
    (GradReducer.after_bucket), beside the rest of the backward pass; step() then covers what is left and closes the step."""
    self._step_slice(int(lo), int(hi))
    self._done.append((int(lo), int(hi)))

  @torch.no_grad()
  def step(self, closure=None):
    pos = 0
    for lo, hi in sorted(self._done):  # the complement of the ranges already stepped
      if lo > pos:
        self._step_slice(pos, lo)
      pos = max(pos, hi)
    if pos < self.flat.numel:
      self._step_slice(pos, self.flat.numel)
    self._done = []
    self._fresh = False

  def state_dict(self):
    """torch's layout plus `pcmi_steps_taken` in the (single) param group: whether the optimiser has stepped.  torch
    encodes that as "the momentum buffers exist"; here they always do, so a checkpoint written before the first step
    would otherwise resume with dampening already applied to the first gradient."""
    sd = super().state_dict()
    sd["param_groups"][0]["pcmi_steps_taken"] = 0 if self._fresh else 1
    return sd

  def load_state_dict(self, state_dict):
    super().load_state_dict(state_dict)
    taken = self.param_groups[0].pop("pcmi_steps_taken", None)
    any_buf, any_nonzero = False, False
    with torch.no_grad():
      for i, p in enumerate(self.flat.params):
        buf = self.state[p].get("momentum_buffer")
        view = self.flat.view(self.flat.v, i)
        if buf is not None:
          any_buf = True
          if buf.data_ptr() != view.data_ptr():
            view.copy_(buf.to(view.device))
        self.state[p]["momentum_buffer"] = view
      if taken is None and any_buf:  # a checkpoint of torch.optim.SGD, or of a build without the flag: derive it
        any_nonzero = bool(torch.count_nonzero(self.flat.v).item())
    # fresh = no step has been taken by the run being resumed: the flag when the checkpoint carries it, else "no
    # momentum buffer holds anything" (torch writes none before the first step; this class wrote zero-filled ones)
    self._fresh = (taken == 0) if taken is not None else not any_nonzero
