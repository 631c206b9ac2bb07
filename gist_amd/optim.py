"""Adam with torch.optim.Adam's semantics (coupled L2 weight decay, bias correction as
torch computes it), stepping through the HIP kernel gist_adam_f32.

Replaces torch.optim.Adam at cluster_gcn/cluster_gcn_ist_distrib.py:405-407,417 and
cluster_gcn/cluster_gcn.py:78-80,105.  If all parameters are views of one flat
arena (gist_amd.engine / gist_amd.ist lay them out that way) the whole model is
one launch; otherwise one launch per tensor.
"""
import torch

from . import ops  # noqa: F401  (registers torch.ops.gist.*)


class Adam(object):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.state = [None] * len(self.params)
        self.param_groups = [dict(params=self.params, lr=lr, betas=betas, eps=eps,
                                  weight_decay=weight_decay)]

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    @torch.no_grad()
    def step(self):
        self.step_count += 1
        lr = self.param_groups[0]['lr']
        # parameters that live in a step plan's arena (a GCN bound to its ClusterIter, gist_amd/module_engine.py) with
        # their gradients in its gradient arena: the whole model is one launch
        me = self.params[0].__dict__.get('_gist_me') if self.params else None
        if me is not None:
            from .module_engine import _REGISTRY
            me = _REGISTRY.get(me)
        if me is not None and me.owns(self.params) and me.homed() and me.grads_in_arena():
            me.optimizer_step(self)
            return
        for i, p in enumerate(self.params):
            if p.grad is None:
                continue
            if self.state[i] is None:
                self.state[i] = (torch.zeros_like(p.data, memory_format=torch.contiguous_format),
                                 torch.zeros_like(p.data, memory_format=torch.contiguous_format))
            m, v = self.state[i]
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            torch.ops.gist.adam_step_(p.data, g, m, v, self.step_count, lr, self.betas[0],
                                      self.betas[1], self.eps, self.weight_decay)
