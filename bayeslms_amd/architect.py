"""First-order architecture step of the search (reference steps/pytorchnn/architect.py).

``Architect.step`` keeps the reference's signature.  What it does (architect.py:46-75 with
``unrolled`` off): cross-entropy of the super-net on a VALIDATION window, gradient with respect to
the architecture logits only, one Adam(lr=arch_lr, weight_decay=arch_wdecay) update of them.

MI355X-first: the reference back-propagates into every network weight as well and throws those
gradients away at the next ``optimizer.zero_grad()`` (train_search_bayes.py:228); here the network
parameters are frozen for the duration of the step, so every weight-gradient GEMM (a third of the
backward FLOPs) is skipped and only the dgrad chain that reaches the branch mixes runs.  Adam's
moments live on the device and the update is one kernel (blm_adam_step).

``unrolled=True`` is not supported -- in the reference it cannot run either:
``_compute_unrolled_model`` calls ``model.new()``, which builds the non-search base class and then
asks it for ``arch_parameters()`` (model_search_bayes.py:316-320,592-596): AttributeError.
"""
import torch

from . import ops
from .model import repackage_hidden

__all__ = ["Architect", "repackage_hidden"]


class Architect(object):

    def __init__(self, model, ntokens, args):
        self.network_weight_decay = args.wdecay
        self.network_clip = args.clip
        self.model = model
        self.ntokens = ntokens
        self.lr, self.weight_decay = args.arch_lr, args.arch_wdecay
        self.betas, self.eps = (0.9, 0.999), 1e-8
        self.steps = 0
        self._moments = None  # (tensor identity, exp_avg, exp_avg_sq) per architecture tensor

    def _state(self):
        arch = self.model.arch_parameters()
        if self._moments is None or any(a is not t for a, (t, _, _) in zip(arch, self._moments)):
            self._moments = [(a, torch.zeros_like(a), torch.zeros_like(a)) for a in arch]
        return self._moments

    def step(self, input_train, target_train, input_valid, target_valid, network_optimizer, unrolled, hiddens_valid=None):
        if unrolled:
            self.model.new()  # raises exactly where the reference does
        for a in self.model.arch_parameters():
            a.grad = None
        self._backward_step(input_valid, target_valid, hiddens_valid)
        self.steps += 1
        for a, m, v in self._state():
            if a.grad is None:
                continue
            with torch.no_grad():
                ops.adam_step(a, a.grad.contiguous(), m, v, self.steps, self.lr, self.betas, self.eps, self.weight_decay)

    def _backward_step(self, input, target, hiddens=None):
        frozen = [p for p in self.model.parameters() if p.requires_grad]
        for p in frozen:
            p.requires_grad_(False)
        try:
            if hiddens is None:
                output = self.model(input)
            else:
                output, _ = self.model(input, repackage_hidden(hiddens))
            loss, _ = ops.cross_entropy(output.view(-1, self.ntokens), target, unit_grad=True)
            loss.backward()
        finally:
            for p in frozen:
                p.requires_grad_(True)
        return loss.detach()
