"""SGD on the flat parameter buffer: one fused libpcmi kernel per step instead of ~250 small
torch kernels (torch.optim.SGD at pc/lib/ddp_trainer.py:107-111: momentum = opt.momentum,
weight_decay = opt.weight_decay, dampening 0, no Nesterov).  It IS a torch Optimizer, so
torch.optim.lr_scheduler.ExponentialLR (pc/lib/ddp_trainer.py:113) drives its lr and
state_dict() has torch's layout (param_groups + per-parameter momentum_buffer)."""
import torch

from .. import functional as PF


class FlatSGD(torch.optim.Optimizer):

  def __init__(self, flat, lr, momentum=0.0, weight_decay=0.0, grad_scale=1.0):
    self.flat = flat
    self.grad_scale = grad_scale
    super().__init__(flat.params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=0,
                                       nesterov=False))
    for i, p in enumerate(flat.params):
      self.state[p]["momentum_buffer"] = flat.view(flat.v, i)

  def zero_grad(self, set_to_none=False):
    self.flat.zero_grad()

  @torch.no_grad()
  def step(self, closure=None):
    g = self.param_groups[0]
    PF.sgd_step(self.flat.w, self.flat.g, self.flat.v, g["lr"], g["momentum"], g["weight_decay"], self.grad_scale)

  def load_state_dict(self, state_dict):
    super().load_state_dict(state_dict)
    with torch.no_grad():
      for i, p in enumerate(self.flat.params):
        buf = self.state[p].get("momentum_buffer")
        view = self.flat.view(self.flat.v, i)
        if buf is not None and buf.data_ptr() != view.data_ptr():
          view.copy_(buf.to(view.device))
        self.state[p]["momentum_buffer"] = view
