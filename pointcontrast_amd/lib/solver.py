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

  def zero_grad(self, set_to_none=False):
    self.flat.zero_grad()

  @torch.no_grad()
  def step(self, closure=None):
    g = self.param_groups[0]
    PF.sgd_step(self.flat.w, self.flat.g, self.flat.v, g["lr"], g["momentum"], g["weight_decay"], self.grad_scale,
                dampening=g.get("dampening", 0.0), first_step=self._fresh)
    self._fresh = False

  def load_state_dict(self, state_dict):
    super().load_state_dict(state_dict)
    with torch.no_grad():
      for i, p in enumerate(self.flat.params):
        buf = self.state[p].get("momentum_buffer")
        view = self.flat.view(self.flat.v, i)
        if buf is not None and buf.data_ptr() != view.data_ptr():
          view.copy_(buf.to(view.device))
        self.state[p]["momentum_buffer"] = view
        self._fresh = self._fresh and buf is None  # loaded buffers: the run being resumed has stepped before
