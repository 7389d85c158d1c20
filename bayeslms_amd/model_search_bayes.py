"""Architecture-search super-nets with the reference's class surface, on the HIP kernels.

Drop-in for ``steps/pytorchnn/model_search_bayes.py`` on the two paths ``train_search_bayes.py``
builds (:158-163): ``GaussTransModelSearch`` (every encoder layer mixes GELU(linear1(x)) with a
GPNN(x) by softmax'd architecture logits) and ``BayesLSTMModelSearch`` (every LSTM gate mixes the
standard gate with a ``Bayes`` affine map of [inp|hx]).  Same class names, positional constructor
arguments, ``arch_parameters()``, ``.weights`` attributes and ``state_dict()`` keys, so a checkpoint
written by either side loads in the other.

MI355X-first differences underneath: the branch mixes and their architecture-weight gradients are
streaming HIP kernels (csrc/search.hip); the LSTM search cell runs ONE (B,8H) recurrent GEMM per
step over the stacked [standard | Bayes] weights instead of five products and a dozen pointwise ops;
the architecture logits live on the GPU next to the model (the reference keeps them on the host,
model_search_bayes.py:322-330); weight-gradient GEMMs are skipped when only the architecture
gradient is wanted (architect.py:66-75).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import BayesLMError
from .model import (GPNN, BayesLinear, MultiheadAttention, NoiseState, PositionalEncoding, _LMHead, _ProjHolder, _RNNLM,
                    _Site, bind_state)

__all__ = ["differentiable_gumble_sample", "BayesTransSearchEncoderLayer", "BayesTransModel", "BayesTransModelSearch",
           "GaussTransSearchEncoderLayer", "GaussTransModel", "GaussTransModelSearch",
           "Bayes", "BayesLSTMSearchCell", "BayesLSTMSearch", "BayesLSTMModel", "BayesLSTMModelSearch"]

INITRANGE = 0.04
TEMPERATURE = 5


def differentiable_gumble_sample(logits, noise=None):
    """softmax((logits - log(-log(U))) / TEMPERATURE), U ~ U(0,1) (model_search_bayes.py:25-30).  Twelve
    numbers at most: plain tensor glue on the device.  ``noise``: inject U (parity tests)."""
    if noise is None:
        noise = torch.rand_like(logits)
    return torch.softmax((logits - torch.log(-torch.log(noise))) / TEMPERATURE, dim=-1)


def _arch_tensor(data, device):
    w = data.detach().to(device=device, dtype=torch.float32).clone().contiguous()
    w.requires_grad_(True)
    return w


# ----------------------------------------------------------------------------
# Transformer: standard | Bayesian linear2 search (not built by train_search_bayes.py; class surface kept)
# ----------------------------------------------------------------------------
class BayesTransSearchEncoderLayer(_Site):
    """Reference model_search_bayes.py:33-81.  FFN output = ffn_linear2(drop(h)) * p[0] + bayes_linear2(drop2(h)) *
    p[1], h = GELU(linear1(x)), drop2 a hard-coded 0.1 (:51); p = the layer's RAW logits pushed through the
    Gumbel softmax while ``gumble_flag`` is set (:61-65).  As in the reference, h is computed once per branch."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, bayes_pos=None):
        super().__init__()
        self.bayes_pos = bayes_pos
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = _ProjHolder(d_model, dim_feedforward)
        self.bayes_linear2 = BayesLinear(dim_feedforward, d_model)
        self.ffn_linear2 = _ProjHolder(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.p = dropout
        self.p2 = 0.1
        self.gumble_flag = True
        self.gumble_noise_override = None
        self.weights = None

    def probs(self):
        if self.weights is None:
            raise AttributeError("'BayesTransSearchEncoderLayer' object has no attribute 'weights'")
        p = self.weights
        if self.gumble_flag is True:
            p = differentiable_gumble_sample(p, self.gumble_noise_override)
        return p.reshape(-1)

    def forward(self, src, src_mask=None):
        a = self.self_attn(src, src, src, attn_mask=src_mask)[0]
        x = ops.add_dropout_ln(src, a, self.norm1.weight, self.norm1.bias, self.norm1.eps, self._drop(self.p, 1))
        bl = self.bayes_linear2
        ya = ops.ffn(x, self.linear1.weight, self.linear1.bias, self.ffn_linear2.weight, self.ffn_linear2.bias,
                     drop=self._drop(self.p, 0))
        yb = ops.ffn(x, self.linear1.weight, self.linear1.bias, bl.weight_mean, None, bl.weight_lgstd, bl.noise(),
                     bl.fused_kl_lambda, self._st().fused, self._drop(self.p2, 3))
        f = ops.mix2(ya, yb, self.probs())
        return ops.add_dropout_ln(x, f, self.norm2.weight, self.norm2.bias, self.norm2.eps, self._drop(self.p, 2))


class BayesTransModel(_LMHead):
    """Reference model_search_bayes.py:84-150."""

    def __init__(self, ntoken, ninp, nhead, nhid, nlayers, dropout=0.5, tie_weights=False):
        super().__init__()
        self.model_type = "Transformer"
        self.src_mask = None
        self.ninp = ninp
        self.nlayers = nlayers
        self.pos_encoder = PositionalEncoding(ninp, dropout)
        self.transformerlayers = nn.ModuleList(BayesTransSearchEncoderLayer(ninp, nhead, nhid, dropout)
                                               for _ in range(nlayers))
        self._init_io(ntoken, ninp, ninp, tie_weights)
        self.noise_state = bind_state(self, NoiseState())

    def forward(self, src, has_mask=True):
        if not has_mask:
            raise BayesLMError("has_mask=False: the fused attention kernel is causal only")
        x = ops.embed(src, self.encoder.weight, self.pos_encoder.table(), math.sqrt(self.ninp),
                      self.pos_encoder._drop(self.pos_encoder.p))
        for layer in self.transformerlayers:
            x = layer(x, src_mask=True)
        return self.decoder(x)


# ----------------------------------------------------------------------------
# Transformer: GELU | GPNN feed-forward search
# ----------------------------------------------------------------------------
class _SampledGPNN(GPNN):
    """The super-net's GPNN: ``sample`` is raised by train_search_bayes.py:236-237 for the network step.  The
    sampled forward itself (model.py:1872-1883) lives in the base class (``GPNN.sampled``)."""


class GaussTransSearchEncoderLayer(_Site):
    """Reference model_search_bayes.py:197-241.  ``self.weights`` ((1,2) logits, a row of the model's
    architecture tensor) is attached by ``GaussTransModelSearch._initialize_arch_parameters``."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, gauss_pos=3):
        super().__init__()
        self.gauss_pos = self.gpnn_type = gauss_pos
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = _ProjHolder(d_model, dim_feedforward)
        self.linear2 = _ProjHolder(dim_feedforward, d_model)
        if 0 <= gauss_pos <= 3:
            self.gpnn = _SampledGPNN(d_model, dim_feedforward, act_set=['tanh', 'sigmoid', 'relu', 'gelu'], gpnn_type=gauss_pos)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.p = dropout
        self.gumble_flag = False
        self.weights = None

    def probs(self):
        if self.weights is None:
            raise AttributeError("'GaussTransSearchEncoderLayer' object has no attribute 'weights'")
        p = torch.softmax(self.weights, dim=-1)
        if self.gumble_flag is True:
            p = differentiable_gumble_sample(p)
        return p.reshape(-1)

    def forward(self, src, src_mask=None):
        a = self.self_attn(src, src, src, attn_mask=src_mask)[0]
        x = ops.add_dropout_ln(src, a, self.norm1.weight, self.norm1.bias, self.norm1.eps, self._drop(self.p, 1))
        wg, bg, coef = self.gpnn.sampled()
        f = ops.search_ffn(x, self.linear1.weight, self.linear1.bias, wg, bg, coef, self.probs(), self.linear2.weight,
                           self.linear2.bias, self._drop(self.p, 0))
        return ops.add_dropout_ln(x, f, self.norm2.weight, self.norm2.bias, self.norm2.eps, self._drop(self.p, 2))


class GaussTransModel(_LMHead):
    """Reference model_search_bayes.py:244-305: nlayers search layers (gauss_pos 3), tied decoder."""

    def __init__(self, ntoken, ninp, nhead, nhid, nlayers, dropout=0.5, tie_weights=False):
        super().__init__()
        self.model_type = "Transformer"
        self.src_mask = None
        self.ninp = ninp
        self.nlayers = nlayers
        self.pos_encoder = PositionalEncoding(ninp, dropout)
        self.transformerlayers = nn.ModuleList(GaussTransSearchEncoderLayer(ninp, nhead, nhid, dropout)
                                               for _ in range(nlayers))
        self._init_io(ntoken, ninp, ninp, tie_weights)
        self.noise_state = bind_state(self, NoiseState())

    def forward(self, src, has_mask=True):
        if not has_mask:
            raise BayesLMError("has_mask=False: the fused attention kernel is causal only")
        x = ops.embed(src, self.encoder.weight, self.pos_encoder.table(), math.sqrt(self.ninp),
                      self.pos_encoder._drop(self.pos_encoder.p))
        for layer in self.transformerlayers:
            x = layer(x, src_mask=True)
        return self.decoder(x)


class _ArchMixin:
    """``model.weights`` / ``arch_parameters()`` of the *ModelSearch classes.  The architecture logits are a
    plain tensor (not an nn.Parameter: absent from ``parameters()`` and ``state_dict()``, as in the
    reference), kept on the device of the model."""

    def _attach(self):
        raise NotImplementedError

    def arch_parameters(self):
        return self._arch_parameters

    def set_arch(self, data):
        """Replace the architecture logits (values copied) and re-attach the per-layer rows."""
        dev = next(self.parameters()).device
        self.weights = _arch_tensor(torch.as_tensor(data), dev)
        self._arch_parameters = [self.weights]
        self._attach()

    def _apply(self, fn, *a, **k):  # .to(device) / .cuda() move the logits with the model
        out = super()._apply(fn, *a, **k)
        if getattr(self, "weights", None) is not None:
            dev = next(self.parameters()).device
            if self.weights.device != dev:
                self.set_arch(self.weights.detach())
        return out


class GaussTransModelSearch(_ArchMixin, GaussTransModel):
    """Reference model_search_bayes.py:308-334: architecture logits (nlayers,1,2), initialised to zero."""

    def __init__(self, *args):
        super().__init__(*args)
        self._args = args
        self._initialize_arch_parameters()

    def new(self):
        """The reference builds a BayesTransModel here and then asks it for ``arch_parameters()``, which that
        class does not have (model_search_bayes.py:316-320): ``--unrolled`` cannot run there either."""
        raise AttributeError("'BayesTransModel' object has no attribute 'arch_parameters'")

    def _initialize_arch_parameters(self):
        torch.randn(self.nlayers, 1, 2)  # drawn and then replaced by zeros in the reference (:323-324): same generator state after
        self.set_arch(torch.zeros(self.nlayers, 1, 2))

    def _attach(self):
        for i, layer in enumerate(self.transformerlayers):
            layer.weights = self.weights[i]


class BayesTransModelSearch(_ArchMixin, BayesTransModel):
    """Reference model_search_bayes.py:153-179: architecture logits (nlayers,1,2), zero-initialised, consumed raw
    (Gumbel softmax) by the layers."""

    def __init__(self, *args):
        super().__init__(*args)
        self._args = args
        self._initialize_arch_parameters()

    def new(self):
        raise AttributeError("'BayesTransModel' object has no attribute 'arch_parameters'")  # :161-165

    def _initialize_arch_parameters(self):
        torch.randn(self.nlayers, 1, 2)  # drawn and then replaced by zeros in the reference (:168-170): same generator state after
        self.set_arch(torch.zeros(self.nlayers, 1, 2))

    def _attach(self):
        for i, layer in enumerate(self.transformerlayers):
            layer.weights = self.weights[i]


# ----------------------------------------------------------------------------
# LSTM: standard | Bayes gate search
# ----------------------------------------------------------------------------
class Bayes(_Site):
    """Reference model_search_bayes.py:790-853: affine map of [inp|hx] with Gaussian weights and bias, sampled
    only in train mode when ``sample`` is set.  ``eps_override``: (weights eps, bias eps)."""

    def __init__(self, input_size, output_size):
        super().__init__()
        self.input_size, self.output_size = input_size, output_size
        s = 1.0 / math.sqrt(output_size)
        lo, hi = 2 * np.log(s), np.log(s)
        self.weights_mean = nn.Parameter(torch.empty(output_size, input_size).uniform_(-s, s))
        self.bias_mean = nn.Parameter(torch.zeros(output_size))
        self.sample = False
        self.weights_lgstd = nn.Parameter(torch.empty(output_size, input_size).uniform_(lo, hi))
        self.bias_lgstd = nn.Parameter(torch.empty(output_size).uniform_(lo, hi))
        # the reference constructor ends with sample_parameters() (:813, :832-835): two N(0,1) draws that are replaced
        # before anything reads them, made here too so that torch's generator leaves this constructor where theirs does
        torch.zeros(output_size, input_size).normal_()
        torch.zeros(output_size).normal_()
        self.eps_override = None

    def kl_divergence(self):
        kl = 0
        if self.sample:
            kl = kl + ops.kl_mean(self.weights_mean, self.weights_lgstd, minus_one=True)
            kl = kl + ops.kl_mean(self.bias_mean, self.bias_lgstd, minus_one=True)
        return kl

    def sample_parameters(self):
        """See GPNN.sample_parameters: nothing to store."""

    def draws_noise(self):
        return True  # once ``sample`` is raised

    def sampled(self):
        if not (self.training and self.sample):
            return self.weights_mean, self.bias_mean
        ew, eb = self.eps_override if self.eps_override is not None else (None, None)
        return (ops.sampled(self.weights_mean, self.weights_lgstd, self._noise(0, ew)),
                ops.sampled(self.bias_mean, self.bias_lgstd, self._noise(1, eb)))

    def forward(self, inp, hx=None):
        x = inp if hx is None else torch.cat([inp, hx], -1)
        w, b = self.sampled()
        return ops.linear(x, w, b)


class BayesLSTMSearchCell(_Site):
    """Reference model_search_bayes.py:636-710, including its bias quirk: ``bias_ih`` enters both the input
    and the recurrent product and ``bias_hh`` is never used (:690-691).  ``self.weights`` ((4,2) logits,
    rows i,f,g,o) is attached by ``BayesLSTMModelSearch``."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        cat = hidden_size + input_size
        self.bayes_ingate = Bayes(cat, hidden_size)
        self.bayes_forgate = Bayes(cat, hidden_size)
        self.bayes_cellgate = Bayes(cat, hidden_size)
        self.bayes_outgate = Bayes(cat, hidden_size)
        s = 1.0 / math.sqrt(hidden_size)
        self.weights_ih = nn.Parameter(torch.empty(hidden_size * 4, input_size).uniform_(-s, s))
        self.bias_ih = nn.Parameter(torch.zeros(hidden_size * 4))
        self.weights_hh = nn.Parameter(torch.empty(hidden_size * 4, hidden_size).uniform_(-s, s))
        self.bias_hh = nn.Parameter(torch.zeros(hidden_size * 4))
        self.weights = None

    def gates(self):
        return (self.bayes_ingate, self.bayes_forgate, self.bayes_cellgate, self.bayes_outgate)

    def forward(self, inputs, hid=None):
        if self.weights is None:
            raise AttributeError("'BayesLSTMSearchCell' object has no attribute 'weights'")
        if inputs.dim() == 2:
            inputs = inputs.unsqueeze(0)
        B, I = inputs.size(1), self.input_size
        if hid is None:
            z = torch.zeros(B, self.hidden_size, dtype=inputs.dtype, device=inputs.device)
            hid = (z, z)
        wb = [g.sampled() for g in self.gates()]
        # stacked operands of the single recurrent GEMM: rows [i f g o | i' f' g' o']
        w8_ih = torch.cat([self.weights_ih] + [w[:, :I] for w, _ in wb], 0)
        w8_hh = torch.cat([self.weights_hh] + [w[:, I:] for w, _ in wb], 0)
        bias8 = torch.cat([self.bias_ih * 2.0] + [b for _, b in wb], 0)
        probs = torch.softmax(self.weights, dim=-1)
        out, h, c = ops.lstm_search_layer(inputs, hid[0], hid[1], w8_ih, w8_hh, bias8, probs)
        return out, (h, c)


class BayesLSTMSearch(_Site):
    """Reference model_search_bayes.py:610-634: two search cells, BOTH built with ``input_size`` inputs (so
    input_size must equal hidden_size), num_layers / dropout accepted and unused."""

    def __init__(self, input_size, hidden_size, num_layers=1, bias=True, dropout=0.):
        super().__init__()
        self.input_size, self.hidden_size, self.bias = input_size, hidden_size, bias
        self.num_layers, self.dropout = num_layers, float(dropout)
        self.rnn = nn.ModuleList([BayesLSTMSearchCell(input_size, hidden_size), BayesLSTMSearchCell(input_size, hidden_size)])

    def forward(self, inputs, hidden=None):
        out0, hid0 = self.rnn[0](inputs, (hidden[0][0], hidden[1][0]))
        out1, hid1 = self.rnn[1](out0, (hidden[0][1], hidden[1][1]))
        return out1, (torch.stack([hid0[0], hid1[0]]), torch.stack([hid0[1], hid1[1]]))


class BayesLSTMModel(_RNNLM):
    """Reference model_search_bayes.py:532-581: drop(emb) -> BayesLSTMSearch -> drop -> decoder."""

    def __init__(self, rnn_type, ntoken, ninp, nhid, nlayers, dropout=0.5, tie_weights=False):
        super().__init__()
        if rnn_type not in ("LSTM", "GRU"):  # :542-550, the nn.RNN branch is not built here
            raise ValueError("An invalid option for `--model` was supplied: this engine builds 'LSTM'")
        if tie_weights and nhid != ninp:
            raise ValueError("When using the tied flag, nhid must be equal to emsize.")
        self.rnn_type, self.nhid, self.nlayers, self.p = rnn_type, nhid, nlayers, dropout
        self._make_encoder(ntoken, ninp)
        self.rnn = BayesLSTMSearch(ninp, nhid, nlayers, dropout=dropout)
        self._init_io(ntoken, ninp, nhid, tie_weights)
        self.noise_state = bind_state(self, NoiseState())

    def forward(self, x, hidden):
        emb = ops.embed(x, self.encoder.weight, None, 1.0, self._drop(self.p, 0))
        out, hidden = self.rnn(emb, hidden)
        out = ops.dropout(out, self._drop(self.p, 1))
        return self.decoder(out), hidden


class BayesLSTMModelSearch(_ArchMixin, BayesLSTMModel):
    """Reference model_search_bayes.py:584-607: architecture logits (nlayers,4,2) ~ 1e-3 * N(0,1); rows 0 and
    1 go to the two cells (so nlayers must be >= 2)."""

    def __init__(self, *args):
        super().__init__(*args)
        self._args = args
        self._initialize_arch_parameters()

    def new(self):
        """As GaussTransModelSearch.new: BayesLSTMModel has no ``arch_parameters`` (model_search_bayes.py:592-596)."""
        raise AttributeError("'BayesLSTMModel' object has no attribute 'arch_parameters'")

    def _initialize_arch_parameters(self):
        self.set_arch(torch.randn(self.nlayers, 4, 2).mul_(1e-3))

    def _attach(self):
        self.rnn.rnn[0].weights = self.weights[0]
        self.rnn.rnn[1].weights = self.weights[1]
