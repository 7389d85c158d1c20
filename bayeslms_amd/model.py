"""Model zoo with the reference's class surface, running on the HIP kernels.

Drop-in for ``steps/pytorchnn/model.py`` of AmourWaltz/BayesLMs on the hot path:
same class names, positional constructor arguments, ``forward`` / ``init_hidden``
/ ``kl_divergence`` / ``embed_kl_divergence`` methods, attribute paths used by
train.py (``model.rnn``, ``model.transformerlayers[i].linear2|self_attn.o_net``)
and ``state_dict()`` key names and shapes, so a
``model.pt`` written by either side loads in the other.

What is different underneath: every forward/backward op is a hand-written
gfx950 kernel behind the C ABI (bayeslms_amd/ops.py); eps of the variational
tensors and all dropout masks come from a counter-based Philox stream keyed by
(seed, tensor/site id, step) instead of torch's generator, so backward and all
data-parallel ranks regenerate them with no storage.  Tensors must live on the
GPU; there is no CPU path.
"""
import copy
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import BayesLMError
from .ops import Drop, NoiseSpec

__all__ = ["NoiseState", "variational_sites", "PositionalEncoding", "MultiheadAttention", "BayesMultiheadAttention", "BayesLinear",
           "StandardTransformerEncoderLayer", "BayesTransformerEncoderLayer", "BayesTransformerModel",
           "TransformerModel", "GPNN", "GaussTransformerEncoderLayer", "GaussTransformerModel", "RNNModel", "BayesRNNModel", "Bayes2LSTM", "repackage_hidden",
           "VTransformerEncoderLayer", "VTransformerModel", "GPLSTMCell", "GPLSTM", "GaussRNNModel", "VNN", "VLSTMCell",
           "VariationalLSTM", "VariationalRNNModel"]


class NoiseState:
    """Shared by all modules of one model: Philox key, optimisation step and this rank's window of
    global batch columns.  ``fused`` selects eps generation inside the GEMM tile loader."""

    def __init__(self, seed=1111):
        self.seed = int(seed)
        self.step = 0
        self.col_offset = 0
        self.global_cols = 0
        self.fused = False
        self.dropout_off = False  # Monte-Carlo weight sampling at inference: noise on, dropout off
        # Nobody manages the step (a reference-shaped training loop around the shim, INTEGRATION.md level 1): every training-mode
        # forward in grad mode moves on to the next step's noise / dropout streams, as torch's generator would.  The first explicit
        # set_step() -- engine.Trainer, the Monte-Carlo scorer, tests -- takes the counter over.
        self.auto_step = True
        # "philox" (default): eps / dropout from the counter-based streams keyed by (seed, site, step), generated inside the kernels.
        # "torch": a parity mode -- every draw the reference makes from torch's CPU generator in a training run is made here, by the
        # same call, in the same order, and uploaded: the variational eps of the Bayesian / Variational / GP families
        # (``new_zeros(shape).normal_(0, std)`` per tensor; the GPNN's per-forward sample_parameters() draws, used or not; GPNN2's
        # fresh frequencies at every call) and every dropout mask (``torch.dropout`` on ones: embedding / positional encoding,
        # nn.LSTM's inter-layer, output, attention probabilities (B * h, T, T), dropout1, feed-forward, dropout2), the blocks then
        # run unfused with the masks multiplied in between.  A run from the same ``--seed`` then sees the reference's own noise and
        # follows its CPU run (train --noise-source torch, BLM_NOISE_SOURCE=torch)
        self.source = os.environ.get("BLM_NOISE_SOURCE", "philox")  # an unchanged reference script selects it from outside
        if self.source not in ("philox", "torch"):
            raise ValueError("BLM_NOISE_SOURCE must be 'philox' or 'torch', not %r" % self.source)


class _Site(nn.Module):
    """Mixin: access to the model's NoiseState and stable per-module stream ids."""
    _state = None
    _site_base = 0

    def _st(self):
        if self._state is None:
            self._state = NoiseState()
        return self._state

    def _drop(self, p, k=0):
        st = self._st()
        if not self.training or p <= 0.0 or st.dropout_off:
            return ops.NO_DROP
        return Drop(float(p), st.seed, self._site_base + k, st.step, st.col_offset, st.global_cols)

    def _torch_drop(self, p):
        """Is this dropout site served by torch's generator in the current forward?  (NoiseState.source "torch", training mode.)"""
        st = self._st()
        return st.source == "torch" and self.training and p > 0.0 and not st.dropout_off

    def _dropout(self, x, p, k=0):
        """drop(x) at dropout site k of this module: the engine's kernel with the Philox mask, or -- NoiseState.source "torch" --
        x times the mask torch's CPU dropout would have drawn here (a parity mode: the mask crosses the bus every step)."""
        if self._torch_drop(p):
            st = self._st()
            return x * torch_dropout_mask(x.shape, p, x.device, st.col_offset, st.global_cols)
        return ops.dropout(x, self._drop(p, k))

    def _attn_drop(self, p, T, B, nhead, device):
        """The dropout of the attention probabilities: the Philox site, or -- NoiseState.source "torch" -- the mask torch's CPU
        dropout draws for the reference's (B * h, T, T) probability tensor (model.py:905-914), handed to the vector-ALU attention
        kernels as it is (under data parallelism: the global batch's heads, this rank's window selected by col_offset)."""
        if self._torch_drop(p):
            st = self._st()
            G = max(int(st.global_cols), B)
            keep = torch.dropout(torch.ones(G * nhead, T, T), float(p), True).to(device)
            return Drop(float(p), col_offset=int(st.col_offset), global_cols=G, keep=keep)
        return self._drop(p)

    def _embed_dropout(self, ids, weight, p, k=0):
        """drop(embedding(ids)) of the LSTM language models (model.py:217-219): one fused launch, or gather then torch's mask."""
        if self._torch_drop(p):
            return self._dropout(ops.embed(ids, weight, None, 1.0, ops.NO_DROP), p, k)
        return ops.embed(ids, weight, None, 1.0, self._drop(p, k))

    def _noise(self, k=0, override=None, like=None):
        """NoiseSpec of variational tensor k of this module, or None in eval mode (mean weights).  ``like``: the log-sigma tensor
        the draw belongs to -- under NoiseState.source "torch" the eps is drawn here, now, from torch's CPU generator the way
        the reference draws it (model.py:1087, :671: ``lgstd.new_zeros(*lgstd.size()).normal_()``)."""
        if override is not None:
            return NoiseSpec(eps=override)
        if not self.training:
            return None
        st = self._st()
        if st.source == "torch" and like is not None:
            return NoiseSpec(eps=torch_eps(like.shape, like.device))
        return NoiseSpec(None, st.seed, self._site_base + k, st.step)


def torch_eps(shape, device, std=1.0):
    """One eps tensor from torch's CPU generator, drawn as the reference draws it on the CPU path (same call, same shape, so the
    same values from the same generator state), then moved to the device."""
    return torch.zeros(*shape).normal_(0, std).to(device)


def torch_dropout_mask(shape, p, device, col_offset=0, global_cols=0):
    """The mask (0 or 1 / (1 - p)) torch's CPU dropout would have applied to a (T, B, D) tensor: ``torch.dropout`` on ones, i.e. the
    same bernoulli_ of the same shape from the same generator (nn.Dropout and nn.LSTM's inter-layer dropout both end there).  Under
    data parallelism the reference's single process would have drawn the mask of the GLOBAL batch: every rank draws that and keeps
    its columns [col_offset, col_offset + B)."""
    T, B, D = shape
    full = torch.dropout(torch.ones(T, max(int(global_cols), B), D), float(p), True)
    return full[:, col_offset:col_offset + B].contiguous().to(device)


def _advance_step(module, _inputs, _output):
    st = module._st()
    if st.auto_step and module.training and torch.is_grad_enabled():
        st.step = (st.step + 1) & 0xFFFFFFFF


def bind_state(model, state):
    """Give every sub-module the model's NoiseState and a unique id range (16 ids per module)."""
    for idx, m in enumerate(model.modules()):
        if isinstance(m, _Site):
            m._state = state
            m._site_base = 16 * idx
    if 16 * (idx + 1) >= (1 << 28):  # ids live in the low 28 bits of the Philox stream word (_lib.STREAM_*)
        raise BayesLMError("model has too many modules for the noise-stream id field")
    if isinstance(model, _Site):  # the root: see NoiseState.auto_step
        model.register_forward_hook(_advance_step)
    return state


def variational_sites(model):
    """The modules of ``model`` that draw noise in a training-mode forward (variational weights, GP coefficients, random
    frequencies, VNN noise rows) -- GPNNs counted as they are once their ``sample`` flag is raised.  An empty list means
    that Monte-Carlo weight sampling has nothing to sample (the scorer refuses ``--mc-samples`` then)."""
    return [m for m in model.modules() if callable(getattr(m, "draws_noise", None)) and m.draws_noise()]


def repackage_hidden(h):
    """Detach hidden states from their history (train.py:291-295)."""
    if isinstance(h, torch.Tensor):
        return h.detach()
    return tuple(repackage_hidden(v) for v in h)


# ----------------------------------------------------------------------------
# Transformer parts
# ----------------------------------------------------------------------------
class PositionalEncoding(_Site):
    """Sinusoidal table + dropout (reference model.py:76-117); buffer ``pe`` is (max_len, 1, d)."""

    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.p = dropout
        pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        freq = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        table = torch.zeros(max_len, d_model)
        table[:, 0::2] = torch.sin(pos * freq)
        table[:, 1::2] = torch.cos(pos * freq)
        self.register_buffer("pe", table.unsqueeze(1))

    def table(self):
        return self.pe.view(self.pe.shape[0], self.pe.shape[2])

    def forward(self, x):
        if self._torch_drop(self.p):
            return self._dropout(ops.add_pe(x, self.table(), ops.NO_DROP), self.p)
        return ops.add_pe(x, self.table(), self._drop(self.p))

    def embed(self, src, weight, scale):
        """drop(embedding(src) * scale + pe): one launch, or -- torch's masks -- the launch without dropout, then the mask."""
        if self._torch_drop(self.p):
            return self._dropout(ops.embed(src, weight, self.table(), scale, ops.NO_DROP), self.p)
        return ops.embed(src, weight, self.table(), scale, self._drop(self.p))


class BayesLinear(_Site):
    """y = x (mu + exp(lgstd) eps)^T, no bias (reference model.py:1049-1134).

    ``eps_override``: inject eps (parity tests).  ``fused_kl_lambda``: when > 0 the wgrad epilogue
    adds lambda * dKL/d(mu, lgstd) itself, so the caller must NOT also backprop kl_divergence()."""

    def __init__(self, in_features, out_features, bias=False):
        super().__init__()
        if bias:
            raise BayesLMError("BayesLinear(bias=True) is never built by the reference recipes and is not supported")
        self.in_features, self.out_features = in_features, out_features
        self.weight_mean = nn.Parameter(torch.empty(out_features, in_features))
        self.weight_lgstd = nn.Parameter(torch.empty(out_features, in_features))
        self.use_bias = False
        self.sample = True
        self.eps_override = None
        self.fused_kl_lambda = 0.0
        self.reset_parameters()

    def reset_parameters(self):
        s = 1.0 / math.sqrt(self.out_features + 1)  # model.py:1070-1073
        self.weight_mean.data.uniform_(-s, s)
        self.weight_lgstd.data.uniform_(2 * np.log(s), np.log(s))

    def draws_noise(self):
        return bool(self.sample)

    def noise(self):
        if not (self.training and self.sample):
            return None
        return self._noise(0, self.eps_override, self.weight_lgstd)

    def kl_divergence(self, prior=None):
        if prior is not None:
            raise BayesLMError("kl_divergence(prior=...) is dead code in the reference (model.py:1120-1122)")
        return ops.kl_mean(self.weight_mean, self.weight_lgstd)

    def forward(self, input):
        return ops.bayes_linear(input, self.weight_mean, self.weight_lgstd, self.noise(), self.fused_kl_lambda,
                                self._st().fused and self._st().source == "philox")

    def extra_repr(self):
        return "in_features={}, out_features={}, bias=False".format(self.in_features, self.out_features)


class _ProjHolder(nn.Module):
    """weight/bias container with nn.Linear's default initialisation (keys '<name>.weight|bias')."""

    def __init__(self, in_f, out_f):
        super().__init__()
        lin = nn.Linear(in_f, out_f)
        self.weight, self.bias = lin.weight, lin.bias
        self.in_features, self.out_features = in_f, out_f

    rows = None  # inference only (the n-best scorer): apply the projection to these flat rows of x, not to all of them
    # inference only (scorer, evaluate()): with targets set, the DECODER returns the per-row NLL of its logits against them
    # instead of the logits themselves, which are never stored (ops.linear_nll)
    nll_targets = None
    # inference only (two-model scoring): the DECODER hands back its input rows instead of logits -- the interpolated decoder +
    # cross-entropy launch (ops.linear_nll_interp) takes both models' rows at once
    return_input = False
    is_decoder = False  # set by _LMHead._init_io for the vocabulary projection

    def forward(self, x, link=None):
        if self.rows is not None:
            if torch.is_grad_enabled():
                raise BayesLMError("_ProjHolder.rows is an inference-only row selection")
            x = x.reshape(-1, x.shape[-1]).index_select(0, self.rows)
        if self.return_input:
            if torch.is_grad_enabled():
                raise BayesLMError("_ProjHolder.return_input is an inference-only path")
            return x
        if self.nll_targets is not None:
            if torch.is_grad_enabled():
                raise BayesLMError("_ProjHolder.nll_targets is an inference-only path")
            return ops.linear_nll(x, self.weight, self.bias, self.nll_targets)
        out = ops.linear(x, self.weight, self.bias, link)
        return ops.as_logits(out) if self.is_decoder else out  # grad mode: F.cross_entropy on it runs the engine's kernels (ops.Logits)


def _need_causal(attn_mask):
    if attn_mask is None:
        raise BayesLMError("the fused attention kernel is causal; the reference LMs always pass the "
                           "square subsequent mask (model.py:1277-1281)")


class MultiheadAttention(_Site):
    """Fused-QKV causal self-attention (reference model.py:836-928).  The head-averaged attention
    weights the reference also returns are discarded by every caller (``[0]`` at model.py:1040,1163);
    the fused kernel never materialises them, so the second return value is None."""

    def __init__(self, embed_dim, num_heads, dropout=0., bias=True, add_bias_kv=False, add_zero_attn=False,
                 kdim=None, vdim=None):
        super().__init__()
        if embed_dim % num_heads:
            raise AssertionError("embed_dim must be divisible by num_heads")
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        self.qkv_net = _ProjHolder(embed_dim, 3 * embed_dim)
        self.o_net = _ProjHolder(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.qkv_net.weight)  # model.py:863-869
        nn.init.constant_(self.qkv_net.bias, 0.)
        nn.init.constant_(self.o_net.bias, 0.)

    def forward(self, query, key=None, value=None, key_padding_mask=None, need_weights=True, attn_mask=None, _link=None):
        _need_causal(attn_mask)
        qkv = self.qkv_net(query, _link)  # _link: ops.ResidualLink of the enclosing post-LN block (new, optional)
        pk = ops.packing()  # scorer: compact real-token rows everywhere but inside the attention core
        if pk is not None:
            a = pk.attention(qkv, None, None, self.num_heads)  # inference: no dropout
        else:
            a = ops.attention(qkv, self.num_heads, self._attn_drop(self.dropout, qkv.shape[0], qkv.shape[1], self.num_heads, qkv.device))
        return self.o_net(a), None


class BayesMultiheadAttention(_Site):
    """Separate q/k/v projections, Bayesian output projection (reference model.py:931-1019; its
    parameter reset is skipped there, :961, so the projections keep nn.Linear's default init)."""

    def __init__(self, embed_dim, num_heads, dropout=0., bias=True, add_bias_kv=False, add_zero_attn=False,
                 kdim=None, vdim=None):
        super().__init__()
        if embed_dim % num_heads:
            raise AssertionError("embed_dim must be divisible by num_heads")
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        self.q_net = _ProjHolder(embed_dim, embed_dim)
        self.k_net = _ProjHolder(embed_dim, embed_dim)
        self.v_net = _ProjHolder(embed_dim, embed_dim)
        self.o_net = BayesLinear(embed_dim, embed_dim)

    def forward(self, query, key=None, value=None, key_padding_mask=None, need_weights=True, attn_mask=None, _link=None):
        _need_causal(attn_mask)
        key = query if key is None else key
        value = query if value is None else value
        # three projections read the block input: its gradient has several producers, so the residual link (valid only
        # when the linked op is the ONLY other consumer of x) is not used here
        q, k, v = self.q_net(query), self.k_net(key), self.v_net(value)
        pk = ops.packing()
        if pk is not None:
            a = pk.attention(q, k, v, self.num_heads)  # inference: no dropout
        else:
            a = ops.attention_qkv(q, k, v, self.num_heads, self._attn_drop(self.dropout, q.shape[0], q.shape[1], self.num_heads, q.device))
        return self.o_net(a), None


class _PostLNLayer(_Site):
    """x = LN1(x + drop(attn(x)));  x = LN2(x + drop(lin2(drop(gelu(lin1(x))))))."""

    def _build(self, d_model, dim_feedforward, dropout, bayes_ffn):
        self.linear1 = _ProjHolder(d_model, dim_feedforward)
        self.linear2 = BayesLinear(dim_feedforward, d_model) if bayes_ffn else _ProjHolder(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.p = dropout

    def _forward_torch_masks(self, src, src_mask, inner):
        """The block with every dropout mask drawn by torch's CPU dropout in the reference's order (NoiseState.source "torch", a
        parity mode): attention probabilities (inside self_attn; the Bayesian o_net's eps after it), dropout1, the feed-forward's
        inner dropout (linear2's eps after it), dropout2 (model.py:1160-1176).  Unfused: the engine's products, attention,
        activation and add + LayerNorm kernels with the masks multiplied in between."""
        a = self.self_attn(src, src, src, attn_mask=src_mask)[0]
        x = ops.add_dropout_ln(src, self._dropout(a, self.p, 1), self.norm1.weight, self.norm1.bias, self.norm1.eps, ops.NO_DROP)
        f = self.linear2(self._dropout(inner(x), self.p, 0))
        return ops.add_dropout_ln(x, self._dropout(f, self.p, 2), self.norm2.weight, self.norm2.bias, self.norm2.eps, ops.NO_DROP)

    def _gelu_linear1(self, x):
        """GELU(linear1(x)) as its own tensor: the product, then the activation-mixture kernel with the mixture (0, 0, 0, 1) on
        (tanh, sigmoid, relu, gelu) -- the erf form, as F.gelu."""
        z = ops.linear(x, self.linear1.weight, self.linear1.bias)
        coef = torch.zeros(4, z.shape[-1], device=z.device)
        coef[3] = 1.0
        return ops.gp_mix(z, coef)

    def forward(self, src, src_mask=None):
        if self._torch_drop(self.p):
            return self._forward_torch_masks(src, src_mask, self._gelu_linear1)
        lk1, lk2 = ops.ResidualLink(), ops.ResidualLink()  # residual + branch gradients meet inside the dgrad GEMMs
        a = self.self_attn(src, src, src, attn_mask=src_mask, _link=lk1)[0]
        x = ops.add_dropout_ln(src, a, self.norm1.weight, self.norm1.bias, self.norm1.eps, self._drop(self.p, 1), lk1)
        l2 = self.linear2
        if isinstance(l2, BayesLinear):
            f = ops.ffn(x, self.linear1.weight, self.linear1.bias, l2.weight_mean, None, l2.weight_lgstd, l2.noise(),
                        l2.fused_kl_lambda, self._st().fused, self._drop(self.p, 0), lk2)
        else:
            f = ops.ffn(x, self.linear1.weight, self.linear1.bias, l2.weight, l2.bias, drop=self._drop(self.p, 0), link=lk2)
        return ops.add_dropout_ln(x, f, self.norm2.weight, self.norm2.bias, self.norm2.eps, self._drop(self.p, 2), lk2)


class StandardTransformerEncoderLayer(_PostLNLayer):
    """Reference model.py:1022-1046."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self._build(d_model, dim_feedforward, dropout, False)


class BayesTransformerEncoderLayer(_PostLNLayer):
    """Reference model.py:1137-1176: 'FFN' -> Bayesian linear2, 'MHA' -> Bayesian o_net."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, bayes_pos=None):
        super().__init__()
        self.bayes_pos = bayes_pos
        att = BayesMultiheadAttention if bayes_pos == "MHA" else MultiheadAttention
        self.self_attn = att(d_model, nhead, dropout=dropout)
        self._build(d_model, dim_feedforward, dropout, bayes_pos == "FFN")


class _LMHead(_Site):
    """Shared embedding / decoder plumbing of the language models."""

    def _make_encoder(self, ntoken, ninp):
        self.encoder = nn.Embedding(ntoken, ninp)

    def _init_io(self, ntoken, ninp, nout, tie_weights):
        """Embedding (unless the caller made it already: the RNN models create it BEFORE the recurrent module, the
        Transformers after the layers -- the reference's orders, which fix both the state_dict key order and the order of
        the RNG draws), vocabulary projection, then ``init_weights``."""
        if not hasattr(self, "encoder"):
            self._make_encoder(ntoken, ninp)
        self.decoder = _ProjHolder(nout, ntoken)
        self.decoder.is_decoder = True
        if tie_weights:
            self.decoder.weight = self.encoder.weight
        nn.init.uniform_(self.encoder.weight, -0.1, 0.1)  # model.py:1264-1268 / :211-215
        nn.init.zeros_(self.decoder.bias)
        nn.init.uniform_(self.decoder.weight, -0.1, 0.1)

    # training-loop controls (new; the reference draws from torch's global generator instead)
    def set_step(self, step):
        self.noise_state.step = int(step)
        self.noise_state.auto_step = False  # the caller manages the step from here on

    def set_seed(self, seed):
        self.noise_state.seed = int(seed)

    def set_columns(self, col_offset, global_cols):
        self.noise_state.col_offset, self.noise_state.global_cols = int(col_offset), int(global_cols)

    def set_fused_sampling(self, on):
        self.noise_state.fused = bool(on)

    def set_noise_source(self, source):
        """"philox" (default) | "torch": see NoiseState.source."""
        if source not in ("philox", "torch"):
            raise ValueError("noise source must be 'philox' or 'torch'")
        self.noise_state.source = source


class BayesTransformerModel(_LMHead):
    """Reference model.py:1179-1309.  Only layer 0 is Bayesian and it is built with a hard-coded
    dropout of 0.2 (model.py:1202,1207); any other ``bayes_pos`` string builds zero layers."""
    supports_packed = True  # ops.packed_tokens (scorer): everything outside the attention core is token-wise


    def __init__(self, ntoken, ninp, nhead, nhid, nlayers, dropout=0.5, tie_weights=False, bayes_pos=None):
        super().__init__()
        self.model_type = "Transformer"
        self.src_mask = None
        self.ninp = ninp
        self.pos_encoder = PositionalEncoding(ninp, dropout)
        self.transformerlayers = nn.ModuleList()
        if bayes_pos in ("none", "EMB"):
            for _ in range(nlayers):
                self.transformerlayers.append(StandardTransformerEncoderLayer(ninp, nhead, nhid, dropout))
        elif bayes_pos in ("FFN", "MHA"):
            self.transformerlayers.append(BayesTransformerEncoderLayer(ninp, nhead, nhid, dropout=0.2, bayes_pos=bayes_pos))
            for _ in range(nlayers - 1):
                self.transformerlayers.append(StandardTransformerEncoderLayer(ninp, nhead, nhid, dropout))
        self.bayes_embed = bayes_pos == "EMB"
        self._init_io(ntoken, ninp, ninp, tie_weights)
        if self.bayes_embed:
            s = 1.0 / math.sqrt(ninp + 1)  # model.py:1269-1272
            self.embed_mean = nn.Parameter(torch.empty(ninp, ninp).uniform_(-s, s))
            self.embed_lgstd = nn.Parameter(torch.empty(ninp, ninp).uniform_(2 * np.log(s), np.log(s)))
            self.embed_eps_override = None
        self.noise_state = bind_state(self, NoiseState())

    def embed_kl_divergence(self):
        return ops.kl_mean(self.embed_mean, self.embed_lgstd)

    def draws_noise(self):
        return self.bayes_embed

    def forward(self, src, has_mask=True):
        if not has_mask:
            raise BayesLMError("has_mask=False: the fused attention kernel is causal only")
        self.src_mask = True
        scale = math.sqrt(self.ninp)
        if self.bayes_embed:
            x = ops.embed(src, self.encoder.weight, None, scale)
            noise = self._noise(0, self.embed_eps_override, self.embed_lgstd) if self.training else None
            W = ops.sampled(self.embed_mean, self.embed_lgstd, noise) if noise is not None else self.embed_mean
            x = ops.linear(x, W)
            x = self.pos_encoder(x)
        else:
            # gather * sqrt(d) + positional table + dropout in one kernel (model.py:1284,1293)
            x = self.pos_encoder.embed(src, self.encoder.weight, scale)
        if ops.packing() is not None:
            x = ops.packing().pack(x)
        for layer in self.transformerlayers:
            x = layer(x, src_mask=self.src_mask)
        if self.bayes_embed:
            x = ops.linear(x, self.embed_mean.t())  # model.py:1302-1304, mean weights
        return self.decoder(x)


class GPNN(_Site):
    """Parameters of the reference's GPNN (model.py:1780-1906): an affine map followed by a learnt
    mixture of activations, sum_i act_i(z) * coef[i].  ``gpnn_type``: 0 deterministic, 1 Bayesian
    coefficients, 2 Bayesian weights, 3 both -- which decides which ``*_lgstd`` tensors exist, enter
    the KL (with the '-1', model.py:1816-1826) and are sampled.  ``self.sample`` is False by default and
    train.py never raises it (model.py:1799), so under the reference's training entry point the forward
    uses the mean tensors; with ``sample`` raised a TRAINING forward uses coef / weights / bias =
    mean + exp(lgstd) * eps (model.py:1871-1883), ONE draw per forward of the enclosing layer / cell
    (``sample_parameters()`` at model.py:2280-2281 and :1721-1723).  eps comes from the Philox streams
    (seed, this module's tensor ids 0 = coef, 1 = weights, 2 = bias, step), or from ``eps_override``
    (dict with any of "coef" / "weights" / "bias": parity tests against the reference's own draw)."""
    eps_override = None

    def __init__(self, input_size, output_size, act_set=('sigmoid', 'tanh', 'relu'), gpnn_type=0):
        super().__init__()
        self.input_size, self.output_size, self.gpnn_type, self.act_set = input_size, output_size, gpnn_type, list(act_set)
        s = 1.0 / math.sqrt(output_size)
        lo, hi = 2 * np.log(s), np.log(s)
        self.weights_mean = nn.Parameter(torch.empty(output_size, input_size).uniform_(-s, s))
        self.bias_mean = nn.Parameter(torch.zeros(output_size))
        self.coef_mean = nn.Parameter(torch.empty(len(act_set), output_size).uniform_(0, 1))
        self.sample = False
        if gpnn_type in (1, 3):
            self.coef_lgstd = nn.Parameter(torch.empty(len(act_set), output_size).uniform_(lo, hi))
        if gpnn_type in (2, 3):
            self.weights_lgstd = nn.Parameter(torch.empty(output_size, input_size).uniform_(lo, hi))
            self.bias_lgstd = nn.Parameter(torch.empty(output_size).uniform_(lo, hi))
        # the reference constructor ends with sample_parameters() (model.py:1812, :1853-1861): N(0,1) draws nobody
        # reads before the next one replaces them, but they advance torch's generator, so every module built after
        # this one is initialised from a different point of the stream -- made here too, and dropped
        if gpnn_type in (1, 3):
            torch.zeros(len(act_set), output_size).normal_()
        if gpnn_type in (2, 3):
            torch.zeros(output_size, input_size).normal_()
            torch.zeros(output_size).normal_()

    _SLOT = {"tanh": 0, "sigmoid": 1, "relu": 2, "gelu": 3}

    def coef4(self, coef=None):
        """coefficient rows (default: coef_mean) placed in the kernels' fixed slot order (tanh, sigmoid, relu, gelu)."""
        coef = self.coef_mean if coef is None else coef
        rows = [None] * 4
        for i, a in enumerate(self.act_set):
            rows[self._SLOT[a]] = coef[i]
        zero = torch.zeros_like(coef[0])
        return torch.stack([r if r is not None else zero for r in rows])

    def draws_noise(self):
        return self.gpnn_type in (1, 2, 3)

    def sampling(self):
        """Does a forward in the current mode draw anything?  (model.py:1872,1878-1879)"""
        return bool(self.training and self.sample and self.gpnn_type in (1, 2, 3))

    def sampled(self):
        """-> (weights, bias, coef) of this forward: the mean tensors, or -- training with ``sample`` raised -- every
        tensor that has an lgstd as mean + exp(lgstd) * eps.  All draws of the module in ONE launch per direction
        (ops.variational_group: the samples forward, d mean / d lgstd backward)."""
        drawn = None
        if self._st().source == "torch" and self.gpnn_type in (1, 2, 3):
            # the reference's sample_parameters() of this forward (model.py:1853-1861; called by the enclosing layer / cell in train
            # AND eval mode, :1721-1723, :2280-2281): coef, weights, bias from torch's CPU generator -- used below when sampling,
            # dropped otherwise, but the generator moves either way (what draws next, e.g. a dropout mask, sees the same state)
            drawn = {}
            if self.gpnn_type in (1, 3):
                drawn["coef"] = torch_eps(self.coef_mean.shape, "cpu")
            if self.gpnn_type in (2, 3):
                drawn["weights"] = torch_eps(self.weights_mean.shape, "cpu")
                drawn["bias"] = torch_eps(self.bias_mean.shape, "cpu")
        if not self.sampling():
            return self.weights_mean, self.bias_mean, self.coef_mean
        e = self.eps_override or {k: v.to(self.weights_mean.device) for k, v in (drawn or {}).items()}
        names = (["coef"] if self.gpnn_type in (1, 3) else []) + (["weights", "bias"] if self.gpnn_type in (2, 3) else [])
        ids = {"coef": 0, "weights": 1, "bias": 2}
        specs = [(getattr(self, n + "_mean"), getattr(self, n + "_lgstd"), self._noise(ids[n], e.get(n)), 0, 0.0, 0.0)
                 for n in names]
        ws, _ = ops.variational_group(specs)
        got = dict(zip(names, ws))
        return got.get("weights", self.weights_mean), got.get("bias", self.bias_mean), got.get("coef", self.coef_mean)

    def sample_parameters(self):
        """The reference redraws its eps buffers here (model.py:1855-1861); eps is a Philox stream keyed by
        (seed, tensor id, step) in this engine, so there is nothing to store."""

    def forward(self, inp, hx=None):
        """Generic (unfused) form used by the LSTM cells: sum_i act_i(W [inp|hx] + b) coef[i]."""
        x = inp if hx is None else torch.cat([inp, hx], -1)
        w, b, coef = self.sampled()
        return ops.gp_mix(ops.linear(x, w, b), self.coef4(coef))

    def kl_divergence(self, prior=None):
        if prior is not None:
            raise BayesLMError("GPNN.kl_divergence(prior=...) returns 0 in the reference; not supported")
        kl = 0
        if self.gpnn_type in (1, 3):
            kl = kl + ops.kl_mean(self.coef_mean, self.coef_lgstd, minus_one=True)
        if self.gpnn_type in (2, 3):
            kl = kl + ops.kl_mean(self.weights_mean, self.weights_lgstd, minus_one=True)
            kl = kl + ops.kl_mean(self.bias_mean, self.bias_lgstd, minus_one=True)
        return kl


class GPNN2(_Site):
    """Reference model.py:2036-2102 (``--T_gauss_pos 4``): random-feature GP.  features = x @ frequency
    with frequency = frequency_mean + eps * exp(frequency_lgstd) in train mode (one N(0,1) draw of
    shape (input_dim, n_MC_terms)); output = coef((features + sum_act act(features)) / sqrt(n_MC))
    over the act set {sigmoid, tanh, relu, gelu} (summed: the order is irrelevant).  The frequencies are
    sampled by the weight-sampling kernel in their own layout, the product is a plain GEMM, the
    activation sum the GPNN mixture kernel with unit coefficients.  train.py adds no KL for
    this position (train.py:360); ``kl_divergence`` needs ``reset_prior()`` first, as in the reference."""

    def __init__(self, input_dim, output_dim, n_MC_terms=150, act_set=('sigmoid', 'tanh', 'relu', 'gelu'), skip_act=True,
                 deterministic=False, update_prior=True):
        super().__init__()
        self.input_dim, self.output_dim, self.n_MC_terms = input_dim, output_dim, n_MC_terms
        self.act_set, self.skip_act, self.deterministic, self.update_prior = set(act_set), skip_act, deterministic, update_prior
        bad = self.act_set - set(GPNN._SLOT)
        if bad:
            raise BayesLMError("GPNN2 activations %s are not built by this engine" % sorted(bad))
        stdv = 1.0 / math.sqrt(n_MC_terms)
        self.frequency_mean = nn.Parameter(torch.empty(input_dim, n_MC_terms))
        self.frequency_lgstd = nn.Parameter(torch.empty(input_dim, n_MC_terms))
        self.coef = _ProjHolder(n_MC_terms, output_dim)  # drawn BEFORE the frequencies (model.py:2053-2059)
        with torch.no_grad():
            self.frequency_mean.uniform_(-stdv, stdv)
            self.frequency_lgstd.uniform_(2 * np.log(stdv), np.log(stdv))
        self.eps_override = None  # (input_dim, n_MC_terms), the reference's layout
        self.frequency_mean_prior = self.frequency_lgstd_prior = None

    def forward(self, x, call=0):
        """``call``: index of this call inside one forward of the parent (the time step of the GP-LSTM
        loop) -- the reference draws fresh frequencies at EVERY call, so the Philox counter gets the call
        index next to the training step; ``eps_override`` may be one tensor or a list indexed by it."""
        freq = self.frequency_mean
        if self.training and not self.deterministic:  # elementwise sampling in the parameters' own layout
            ov = self.eps_override
            if isinstance(ov, (list, tuple)):
                ov = ov[call]
            if ov is None and self._st().source == "torch":  # the reference's own draw (model.py:2064-2066), from torch's generator
                ov = torch_eps(self.frequency_lgstd.shape, self.frequency_lgstd.device)
            if ov is not None:
                noise = NoiseSpec(eps=ov)
            else:
                st = self._st()
                noise = NoiseSpec(None, st.seed, self._site_base, (st.step * 1024 + call) & 0xFFFFFFFF)
            freq = ops.sampled(self.frequency_mean, self.frequency_lgstd, noise)
        z = ops.linear(x, freq.t().contiguous())  # x @ frequency; the (n_MC, input_dim) copy is 77 k floats
        ones = torch.zeros(4, self.n_MC_terms, device=z.device, dtype=torch.float32)
        for a in self.act_set:
            ones[GPNN._SLOT[a]] = 1.0
        mix = ops.gp_mix(z, ones)
        a = (z + mix) if self.skip_act else mix
        return self.coef(a * (1.0 / math.sqrt(self.n_MC_terms)))

    def draws_noise(self):
        return not self.deterministic

    def step_noises(self, T):
        """The noise of calls 0..T-1 of one forward exactly as ``forward(x, call)`` would draw it (None: mean frequencies)."""
        if not (self.training and not self.deterministic):
            return None
        out = []
        for call in range(T):
            ov = self.eps_override
            if isinstance(ov, (list, tuple)):
                ov = ov[call]
            if ov is None and self._st().source == "torch":  # one draw per call, in call order (nothing else draws inside the time loop)
                ov = torch_eps(self.frequency_lgstd.shape, self.frequency_lgstd.device)
            if ov is not None:
                out.append(NoiseSpec(eps=ov))
            else:
                st = self._st()
                out.append(NoiseSpec(None, st.seed, self._site_base, (st.step * 1024 + call) & 0xFFFFFFFF))
        return out

    def reset_prior(self):
        self.frequency_mean_prior = torch.zeros_like(self.frequency_mean.data)
        self.frequency_lgstd_prior = torch.zeros_like(self.frequency_lgstd.data)
        if self.update_prior:
            self.frequency_mean_prior = self.frequency_mean.data.clone()
            self.frequency_lgstd_prior = self.frequency_lgstd.data.clone()

    def kl_divergence(self):
        if self.frequency_mean_prior is None:
            raise BayesLMError("GPNN2.kl_divergence needs reset_prior() first (reference model.py:2078-2096)")
        var, var_p = torch.exp(2 * self.frequency_lgstd), torch.exp(2 * self.frequency_lgstd_prior)
        ms = (self.frequency_mean - self.frequency_mean_prior) ** 2. / var_p
        ls = 2 * (self.frequency_lgstd_prior - self.frequency_lgstd) / self.frequency_mean.size(1)
        return torch.sum(ms + var / var_p - ls - 1) / 2.  # small (input_dim, n_MC) glue, never called by train.py


class GaussTransformerEncoderLayer(_Site):
    """Reference model.py:2250-2295: GPNN replaces GELU(linear1(x)); ``linear1`` exists in the
    state_dict but is unused (model.py:2257,2283)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, gauss_pos=None):
        super().__init__()
        if not (0 <= gauss_pos <= 4):
            raise BayesLMError("GaussTransformerEncoderLayer: gauss_pos must be 0..4")
        self.gauss_pos = self.gpnn_type = gauss_pos
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = _ProjHolder(d_model, dim_feedforward)
        self.linear2 = _ProjHolder(dim_feedforward, d_model)
        if gauss_pos < 4:
            self.gpnn = GPNN(d_model, dim_feedforward, act_set=['tanh', 'sigmoid', 'relu', 'gelu'], gpnn_type=gauss_pos)
        else:
            self.gpnn = GPNN2(d_model, dim_feedforward, act_set=['tanh', 'sigmoid', 'relu', 'gelu'])
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.p = dropout

    def forward(self, src, src_mask=None):
        if self._torch_drop(self.p):
            # same block, the GPNN (or GPNN2) in GELU(linear1)'s place; its sample_parameters() draws come right before it (model.py:2280-2283)
            return _PostLNLayer._forward_torch_masks(self, src, src_mask, self.gpnn)
        lk1, lk2 = ops.ResidualLink(), ops.ResidualLink()
        a = self.self_attn(src, src, src, attn_mask=src_mask, _link=lk1)[0]
        x = ops.add_dropout_ln(src, a, self.norm1.weight, self.norm1.bias, self.norm1.eps, self._drop(self.p, 1), lk1)
        g = self.gpnn
        if self.gauss_pos == 4:  # GPNN2: 150 random features, then its own Linear to dim_feedforward
            f = self.linear2(ops.dropout(g(x), self._drop(self.p, 0)))
            lk2 = None
        else:  # model.py:2280-2283: one draw of the GPNN's tensors per forward when its sample flag is raised
            wg, bg, coef = g.sampled()
            f = ops.ffn_gp(x, wg, bg, coef, self.linear2.weight, self.linear2.bias, self._drop(self.p, 0), lk2)
        return ops.add_dropout_ln(x, f, self.norm2.weight, self.norm2.bias, self.norm2.eps, self._drop(self.p, 2), lk2)


class GaussTransformerModel(_LMHead):
    """Reference model.py:2298-2364: layer 0 is the GP layer for gauss_pos 0..3 (built with the
    model's dropout), gauss_pos > 4 builds an all-standard stack."""
    supports_packed = True  # ops.packed_tokens (scorer): everything outside the attention core is token-wise


    def __init__(self, ntoken, ninp, nhead, nhid, nlayers, dropout=0.5, tie_weights=False, gauss_pos=4):
        super().__init__()
        self.model_type = "Transformer"
        self.src_mask = None
        self.ninp = ninp
        self.pos_encoder = PositionalEncoding(ninp, dropout)
        self.transformerlayers = nn.ModuleList()
        if gauss_pos <= 4:
            self.transformerlayers.append(GaussTransformerEncoderLayer(ninp, nhead, nhid, dropout, gauss_pos=gauss_pos))
            for _ in range(nlayers - 1):
                self.transformerlayers.append(StandardTransformerEncoderLayer(ninp, nhead, nhid, dropout))
        else:
            for _ in range(nlayers):
                self.transformerlayers.append(StandardTransformerEncoderLayer(ninp, nhead, nhid, dropout))
        self._init_io(ntoken, ninp, ninp, tie_weights)
        self.noise_state = bind_state(self, NoiseState())

    def forward(self, src, has_mask=True):
        if not has_mask:
            raise BayesLMError("has_mask=False: the fused attention kernel is causal only")
        x = self.pos_encoder.embed(src, self.encoder.weight, math.sqrt(self.ninp))
        if ops.packing() is not None:
            x = ops.packing().pack(x)
        for layer in self.transformerlayers:
            x = layer(x, src_mask=True)
        return self.decoder(x)


class VTransformerEncoderLayer(StandardTransformerEncoderLayer):
    """Reference model.py:2741-2805.  Carries the four (100, 1, d) ``hiddens_*`` tensors (left at
    their torch.rand initialisation there, :2756-2759).  The reference's noise branch only fires in
    train mode at exactly T == 100 and then dereferences ``self.hiddens``, which does not exist:
    the same AttributeError is raised here; at every other length the layer is a standard one."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1):
        super().__init__(d_model, nhead, dim_feedforward, dropout)
        for name in ("hiddens_mean_p", "hiddens_lgstd_p", "hiddens_mean", "hiddens_lgstd"):
            self.register_parameter(name, nn.Parameter(torch.rand(100, 1, d_model)))

    def kl_divergence(self):
        raise BayesLMError("VTransformerEncoderLayer.kl_divergence broadcasts (T,B,d) against (100,1,d) and only "
                           "reaches its formula in the branch that already crashed in forward (reference model.py:2770-2779)")

    def forward(self, src, src_mask=None):
        if self.training and src.size(0) == 100:
            raise AttributeError("'VTransformerEncoderLayer' object has no attribute 'hiddens'")
        return super().forward(src, src_mask)


class VTransformerModel(_LMHead):
    """Reference model.py:2808-2897, including its layer-count arithmetic: v_pos 2 and 3 build
    nlayers-1 layers, any other value (e.g. the README's 11) builds none."""

    def __init__(self, ntoken, ninp, nhead, nhid, nlayers, dropout=0.5, tie_weights=False, v_pos=0):
        super().__init__()
        self.model_type = "Transformer"
        self.src_mask = None
        self.ninp = ninp
        self.pos_encoder = PositionalEncoding(ninp, dropout)
        std = lambda: StandardTransformerEncoderLayer(ninp, nhead, nhid, dropout)  # noqa: E731
        var = lambda: VTransformerEncoderLayer(ninp, nhead, nhid, dropout)  # noqa: E731
        layers = []
        if v_pos == 0:
            layers = [std() for _ in range(nlayers)]
        elif v_pos == 1:
            layers = [var()] + [std() for _ in range(nlayers - 1)]
        elif v_pos == 2:
            layers = [std(), var()] + [std() for _ in range(nlayers - 3)]
        elif v_pos == 3:
            layers = [var(), var()] + [std() for _ in range(nlayers - 3)]
        self.transformerlayers = nn.ModuleList(layers)
        self._init_io(ntoken, ninp, ninp, tie_weights)
        self.noise_state = bind_state(self, NoiseState())

    def forward(self, src, has_mask=True):
        if not has_mask:
            raise BayesLMError("has_mask=False: the fused attention kernel is causal only")
        x = self.pos_encoder.embed(src, self.encoder.weight, math.sqrt(self.ninp))
        for layer in self.transformerlayers:
            x = layer(x, src_mask=True)
        return self.decoder(x)


class _TorchMHAParams(_Site):
    """nn.MultiheadAttention's parameter names (in_proj_weight, in_proj_bias, out_proj.*)."""

    def __init__(self, d_model, nhead, dropout):
        super().__init__()
        self.num_heads, self.dropout = nhead, dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = _ProjHolder(d_model, d_model)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.)

    def forward(self, query, key=None, value=None, attn_mask=None, **_):
        _need_causal(attn_mask)
        qkv = ops.linear(query, self.in_proj_weight, self.in_proj_bias)
        pk = ops.packing()
        if pk is not None:
            return self.out_proj(pk.attention(qkv, None, None, self.num_heads)), None  # inference: no dropout
        return self.out_proj(ops.attention(qkv, self.num_heads, self._attn_drop(self.dropout, qkv.shape[0], qkv.shape[1], self.num_heads,
                                                                                qkv.device))), None


class _TorchEncoderLayer(_PostLNLayer):
    def __init__(self, d_model, nhead, dim_feedforward, dropout):
        super().__init__()
        self.self_attn = _TorchMHAParams(d_model, nhead, dropout)
        self._build(d_model, dim_feedforward, dropout, False)


class _TorchEncoder(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward, dropout, nlayers):
        super().__init__()
        # nn.TransformerEncoder clones ONE constructed layer (reference model.py:134-136): every layer starts from layer
        # 0's draw, and the generator advances by one layer only
        first = _TorchEncoderLayer(d_model, nhead, dim_feedforward, dropout)
        self.layers = nn.ModuleList([first] + [copy.deepcopy(first) for _ in range(nlayers - 1)] if nlayers > 0 else [])


class TransformerModel(_LMHead):
    """Baseline (reference model.py:120-171): the state_dict keys are nn.TransformerEncoder's
    (``transformerlayers.layers.N.self_attn.in_proj_weight`` ...), post-LN, GELU."""
    supports_packed = True  # ops.packed_tokens (scorer): everything outside the attention core is token-wise


    def __init__(self, ntoken, ninp, nhead, nhid, nlayers, dropout=0.5, activation="relu", tie_weights=False):
        super().__init__()
        if activation != "gelu":
            raise BayesLMError("only activation='gelu' is built by the reference entry points (train.py:196-199)")
        self.model_type = "Transformer"
        self.src_mask = None
        self.ninp = ninp
        self.pos_encoder = PositionalEncoding(ninp, dropout)
        self.transformerlayers = _TorchEncoder(ninp, nhead, nhid, dropout, nlayers)
        self._init_io(ntoken, ninp, ninp, tie_weights)
        self.noise_state = bind_state(self, NoiseState())

    def forward(self, src, has_mask=True):
        if not has_mask:
            raise BayesLMError("has_mask=False: the fused attention kernel is causal only")
        x = self.pos_encoder.embed(src, self.encoder.weight, math.sqrt(self.ninp))
        if ops.packing() is not None:
            x = ops.packing().pack(x)
        for layer in self.transformerlayers.layers:
            x = layer(x, src_mask=True)
        return self.decoder(x)


# ----------------------------------------------------------------------------
# LSTM models
# ----------------------------------------------------------------------------
class Bayes2LSTM(_Site):
    """Two LSTM layers whose gate rows [(pos-1)H, pos*H) of all 8 tensors are variational
    (reference model.py:585-828).  position 0 = plain LSTM; 5 is accepted like the reference
    (lgstd tensors exist at full gate size, are never sampled)."""

    def __init__(self, input_size, hidden_size, num_layers=1, position=0, bias=True, dropout=0., bayes_pos=0):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.bias, self.dropout, self.position = bias, float(dropout), position
        G, H = 4 * hidden_size, hidden_size
        s = 1.0 / math.sqrt(H)
        # Registration AND draw order follow the reference constructor (model.py:598-636 creates, :642-665 resets), so that
        # the same torch seed gives the same initial state_dict: the lgstd tensors are born from torch.rand (draws that
        # positions 1-4 then overwrite), the means are uniform in the order ih, hh, bias_hh, bias_ih.
        for layer in (1, 2):
            for name, shape in (("weight_ih", (G, input_size)), ("weight_hh", (G, H)), ("bias_ih", (G,)), ("bias_hh", (G,))):
                self.register_parameter("%s_mean_%d" % (name, layer), nn.Parameter(torch.empty(*shape)))
        if 1 <= position <= 5:
            R = H if position <= 4 else G
            for layer in (1, 2):
                for name, shape in (("weight_hh", (R, H)), ("weight_ih", (R, input_size)), ("bias_hh", (R,)), ("bias_ih", (R,))):
                    self.register_parameter("%s_lgstd_%d" % (name, layer), nn.Parameter(torch.rand(*shape)))
        with torch.no_grad():
            for layer in (1, 2):
                for name in ("weight_ih", "weight_hh", "bias_hh", "bias_ih"):
                    getattr(self, "%s_mean_%d" % (name, layer)).uniform_(-s, s)
            if 1 <= position <= 4:  # position 5 keeps its torch.rand values (model.py:625-633, never reset)
                for layer in (1, 2):
                    for name in self._ORDER:
                        getattr(self, "%s_lgstd_%d" % (name, layer)).uniform_(2 * math.log(s), math.log(s))
        self.eps_override = None  # list of 8 tensors in the reference's draw order (model.py:668-703)
        self._kl_cache = None     # (KL of the last training forward, noise step, parameter versions): see _weights

    _ORDER = ("weight_hh", "weight_ih", "bias_hh", "bias_ih")

    def draws_noise(self):
        return 1 <= self.position <= 4  # position 5 has lgstd tensors that are never sampled (model.py:716)

    def _weights(self):
        """The 8 tensors _VF.lstm gets (model.py:705-732), sampled on the gate rows in train mode."""
        out = {}
        pos, H = self.position, self.hidden_size
        self._kl_cache = None
        if 1 <= pos <= 4 and self.training:
            # one launch samples all 8 tensors and leaves the KL term of layer 1 (kl_divergence below) behind
            E = self.input_size
            cnt = {"weight": float(H * (H + E)), "bias": float(2 * H)}
            specs, keys = [], []
            for k, (layer, name) in enumerate((ly, nm) for ly in (1, 2) for nm in self._ORDER):
                mu = getattr(self, "%s_mean_%d" % (name, layer))
                lg = getattr(self, "%s_lgstd_%d" % (name, layer))
                ov = self.eps_override[k] if self.eps_override is not None else None
                klw = lg.numel() / cnt[name.split("_")[0]] if layer == 1 else 0.0
                specs.append((mu, lg, self._noise(k, ov, lg), (pos - 1) * H, klw, 0.0))
                keys.append((name, layer))
            ws, kl = ops.variational_group(specs)
            if torch.is_grad_enabled():
                self._kl_cache = (kl, self._st().step, tuple(sp[j]._version for sp in specs[:4] for j in (0, 1)))
            return dict(zip(keys, ws))
        k = 0
        for layer in (1, 2):
            for name in self._ORDER:
                mu = getattr(self, "%s_mean_%d" % (name, layer))
                if 1 <= pos <= 4 and self.training:
                    lg = getattr(self, "%s_lgstd_%d" % (name, layer))
                    ov = self.eps_override[k] if self.eps_override is not None else None
                    out[(name, layer)] = ops.sampled(mu, lg, self._noise(k, ov, lg), (pos - 1) * H)
                else:
                    out[(name, layer)] = mu
                k += 1
        return out

    def kl_divergence(self, prior=None):
        pos, H, E = self.position, self.hidden_size, self.input_size
        if prior is not None or not (1 <= pos <= 5):
            raise BayesLMError("Bayes2LSTM.kl_divergence is defined for position 1..5 without prior "
                               "(reference model.py:734-775: other branches are dead or raise)")
        if pos == 5:
            # model.py:746-755, as written: layer 1's [hh|ih] tensors PLUS [hh of layer 2 | ih of layer 1
            # again], means and log-sigmas alike, then the mean-form KL.  Nothing is ever sampled at this
            # position; plain tensor glue on the parameters, not a hot path.
            wm = torch.cat([self.weight_hh_mean_1, self.weight_ih_mean_1], -1) + torch.cat([self.weight_hh_mean_2, self.weight_ih_mean_1], -1)
            wl = torch.cat([self.weight_hh_lgstd_1, self.weight_ih_lgstd_1], -1) + torch.cat([self.weight_hh_lgstd_2, self.weight_ih_lgstd_1], -1)
            bm = torch.cat([self.bias_hh_mean_1, self.bias_ih_mean_1], -1) + torch.cat([self.bias_hh_mean_2, self.bias_ih_mean_1], -1)
            bl = torch.cat([self.bias_hh_lgstd_1, self.bias_ih_lgstd_1], -1) + torch.cat([self.bias_hh_lgstd_2, self.bias_ih_lgstd_1], -1)
            return torch.mean(wm ** 2. - wl * 2. + torch.exp(wl * 2)) / 2. + torch.mean(bm ** 2. - bl * 2. + torch.exp(bl * 2)) / 2.
        lo = (pos - 1) * H
        nw, nb = H * (H + E), 2 * H  # the reference concatenates hh|ih before taking the mean
        cache, self._kl_cache = getattr(self, "_kl_cache", None), None
        if cache is not None and pos <= 4 and self.training and torch.is_grad_enabled() and cache[1] == self._st().step:
            ps = [getattr(self, "%s_%s_1" % (name, kind)) for name in self._ORDER for kind in ("mean", "lgstd")]
            if cache[2] == tuple(p._version for p in ps):  # same parameters as the forward that sampled them
                return cache[0]
        kl = ops.kl_mean(self.weight_hh_mean_1, self.weight_hh_lgstd_1, lo, count=nw)
        kl = kl + ops.kl_mean(self.weight_ih_mean_1, self.weight_ih_lgstd_1, lo, count=nw)
        kl = kl + ops.kl_mean(self.bias_hh_mean_1, self.bias_hh_lgstd_1, lo, count=nb)
        kl = kl + ops.kl_mean(self.bias_ih_mean_1, self.bias_ih_lgstd_1, lo, count=nb)
        return kl

    def forward(self, inputs, hx=None):
        T, B, _ = inputs.shape
        if hx is None:
            z = torch.zeros(self.num_layers, B, self.hidden_size, dtype=inputs.dtype, device=inputs.device)
            hx = (z, z)
        w = self._weights()
        h0, c0 = hx
        l1 = (w[("weight_ih", 1)], w[("weight_hh", 1)], w[("bias_ih", 1)], w[("bias_hh", 1)])
        l2 = (w[("weight_ih", 2)], w[("weight_hh", 2)], w[("bias_ih", 2)], w[("bias_hh", 2)])
        if ops.lstm_stack2_ok(inputs, l1[1], l2[1], l2[0]):  # the two layers as a wavefront on two streams
            y, (h1, h2), (c1, c2) = ops.lstm_stack2(inputs, h0, c0, l1, l2)
            return y, (torch.stack([h1, h2]), torch.stack([c1, c2]))
        y, h1, c1 = ops.lstm_layer(inputs, h0[0], c0[0], w[("weight_ih", 1)], w[("weight_hh", 1)], w[("bias_ih", 1)],
                                   w[("bias_hh", 1)])
        y, h2, c2 = ops.lstm_layer(y, h0[1], c0[1], w[("weight_ih", 2)], w[("weight_hh", 2)], w[("bias_ih", 2)],
                                   w[("bias_hh", 2)])
        return y, (torch.stack([h1, h2]), torch.stack([c1, c2]))


class _RNNLM(_LMHead):
    def _check_type(self, rnn_type):
        if rnn_type != "LSTM":
            raise ValueError("An invalid option for `--model` was supplied: this engine builds 'LSTM' "
                             "(the reference recipes, run_nnlm_ami_lstm.sh) and 'Transformer'")

    def init_hidden(self, bsz):
        w = next(self.parameters())
        return (w.new_zeros(self.nlayers, bsz, self.nhid), w.new_zeros(self.nlayers, bsz, self.nhid))


class BayesRNNModel(_RNNLM):
    """Reference model.py:179-229: drop(emb) -> Bayes2LSTM -> drop -> tied decoder."""

    def __init__(self, rnn_type, ntoken, ninp, nhid, nlayers, dropout=0.5, tie_weights=False, bayes_pos=0):
        super().__init__()
        self._check_type(rnn_type)
        if tie_weights and nhid != ninp:
            raise ValueError("When using the tied flag, nhid must be equal to emsize.")
        self.rnn_type, self.nhid, self.nlayers, self.p = rnn_type, nhid, nlayers, dropout
        self._make_encoder(ntoken, ninp)
        self.rnn = Bayes2LSTM(ninp, nhid, nlayers, position=bayes_pos, dropout=dropout)
        self._init_io(ntoken, ninp, nhid, tie_weights)
        self.noise_state = bind_state(self, NoiseState())

    def forward(self, x, hidden):
        emb = self._embed_dropout(x, self.encoder.weight, self.p, 0)
        out, hidden = self.rnn(emb, hidden)
        out = self._dropout(out, self.p, 1)
        return self.decoder(out), hidden


class _LSTMParams(_Site):
    """nn.LSTM's parameter names (weight_ih_l0 ...), inter-layer dropout in train mode."""

    def __init__(self, ninp, nhid, nlayers, dropout):
        super().__init__()
        self.nlayers, self.p = nlayers, dropout
        s = 1.0 / math.sqrt(nhid)
        for k in range(nlayers):
            inp = ninp if k == 0 else nhid
            self.register_parameter("weight_ih_l%d" % k, nn.Parameter(torch.empty(4 * nhid, inp).uniform_(-s, s)))
            self.register_parameter("weight_hh_l%d" % k, nn.Parameter(torch.empty(4 * nhid, nhid).uniform_(-s, s)))
            self.register_parameter("bias_ih_l%d" % k, nn.Parameter(torch.empty(4 * nhid).uniform_(-s, s)))
            self.register_parameter("bias_hh_l%d" % k, nn.Parameter(torch.empty(4 * nhid).uniform_(-s, s)))

    def forward(self, x, hx):
        h0, c0 = hx
        if self.nlayers == 2 and not self._torch_drop(self.p):  # torch's mask covers a whole layer output: the layers run one after the other
            l1 = tuple(getattr(self, "%s_l0" % n) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
            l2 = tuple(getattr(self, "%s_l1" % n) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
            if ops.lstm_stack2_ok(x, l1[1], l2[1], l2[0]):  # wavefront on two streams, inter-layer dropout per chunk
                y, (h1, h2), (c1, c2) = ops.lstm_stack2(x, h0, c0, l1, l2, self._drop(self.p, 0))
                return y, (torch.stack([h1, h2]), torch.stack([c1, c2]))
        hs, cs = [], []
        for k in range(self.nlayers):
            x, h, c = ops.lstm_layer(x, h0[k], c0[k], getattr(self, "weight_ih_l%d" % k), getattr(self, "weight_hh_l%d" % k),
                                     getattr(self, "bias_ih_l%d" % k), getattr(self, "bias_hh_l%d" % k))
            if k + 1 < self.nlayers:  # nn.LSTM's inter-layer dropout: on every layer's output but the last
                x = self._dropout(x, self.p, k)
            hs.append(h)
            cs.append(c)
        return x, (torch.stack(hs), torch.stack(cs))


class RNNModel(_RNNLM):
    """Baseline LSTM LM (reference model.py:23-73, nn.LSTM with inter-layer dropout)."""

    def __init__(self, rnn_type, ntoken, ninp, nhid, nlayers, dropout=0.5, tie_weights=False):
        super().__init__()
        self._check_type(rnn_type)
        if tie_weights and nhid != ninp:
            raise ValueError("When using the tied flag, nhid must be equal to emsize.")
        self.rnn_type, self.nhid, self.nlayers, self.p = rnn_type, nhid, nlayers, dropout
        self._make_encoder(ntoken, ninp)
        self.rnn = _LSTMParams(ninp, nhid, nlayers, dropout)
        self._init_io(ntoken, ninp, nhid, tie_weights)
        self.noise_state = bind_state(self, NoiseState())

    def forward(self, x, hidden):
        emb = self._embed_dropout(x, self.encoder.weight, self.p, 0)
        out, hidden = self.rnn(emb, hidden)
        out = self._dropout(out, self.p, 1)
        return self.decoder(out), hidden


# ----------------------------------------------------------------------------
# GP / Variational LSTMs (Python time loops, as in the reference)
# ----------------------------------------------------------------------------
class _LoopCell(_Site):
    """weights_ih/weights_hh/bias_ih/bias_hh with the reference's init (model.py:1712-1717) and its
    quirk: ``bias_ih`` is added twice, ``bias_hh`` is never used (model.py:1750-1752, 2519)."""

    def _params(self, input_size, hidden_size):
        s = 1.0 / math.sqrt(hidden_size)
        self.input_size, self.hidden_size = input_size, hidden_size
        self.weights_ih = nn.Parameter(torch.empty(4 * hidden_size, input_size).uniform_(-s, s))
        self.bias_ih = nn.Parameter(torch.zeros(4 * hidden_size))
        self.weights_hh = nn.Parameter(torch.empty(4 * hidden_size, hidden_size).uniform_(-s, s))
        self.bias_hh = nn.Parameter(torch.zeros(4 * hidden_size))


class GPLSTMCell(_LoopCell):
    """Reference model.py:1674-1777.  gate_type 1-4: that gate's activation is a GPNN of [inp|h];
    5: the cell state passes through a GPNN first; 6 / 7: the hidden / input projection of all four
    gates is a GPNN.  gpnn_type 0-3; 4 = GPNN2 random features on the gate's pre-activation etc.
    (``_forward_gpnn2``)."""

    def __init__(self, input_size, hidden_size, gate_type=0, gpnn_type=0):
        super().__init__()
        if gpnn_type > 4:
            raise BayesLMError("GPLSTMCell: gpnn_type must be 0..4")
        self.gate_type, self.gpnn_type = gate_type, gpnn_type
        H, E = hidden_size, input_size
        if gpnn_type == 4:  # GPNN2 on the gate's pre-activation / cell state / a whole projection (model.py:1698-1702)
            if 0 < gate_type <= 5:
                self.gpnn = GPNN2(H, H, act_set=['sigmoid', 'relu', 'tanh'])
            elif 5 < gate_type <= 7:
                self.gpnn = GPNN2(H, 4 * H, act_set=['sigmoid', 'relu', 'tanh'])
        elif gate_type == 3:
            self.gpnn = GPNN(H + E, H, gpnn_type=gpnn_type)
        elif gate_type in (1, 4):
            self.gpnn = GPNN(H + E, H, act_set=['sigmoid', 'tanh', 'relu'], gpnn_type=gpnn_type)
        elif gate_type == 2:
            self.gpnn = GPNN(H + E, H, act_set=['sigmoid'], gpnn_type=gpnn_type)
        elif gate_type == 5:
            self.gpnn = GPNN(E, H, gpnn_type=gpnn_type)
        elif 5 < gate_type <= 7:
            self.gpnn = GPNN(E, 4 * H, gpnn_type=gpnn_type)
        self._params(input_size, hidden_size)

    def forward(self, inputs, hid=None):
        if inputs.dim() == 2:
            inputs = inputs.unsqueeze(0)
        T, B, _ = inputs.shape
        if hid is None:
            z = torch.zeros(B, self.hidden_size, dtype=inputs.dtype, device=inputs.device)
            hid = (z, z)
        hx, cx = hid
        gt = self.gate_type
        if self.gpnn_type == 4 and gt > 0:
            return self._forward_gpnn2(inputs, hx, cx)
        if not 1 <= gt <= 7:  # no GPNN is built for other gate types (model.py:1686-1697): a plain cell
            y, hT, cT = ops.lstm_layer(inputs, hx, cx, self.weights_ih, self.weights_hh, self.bias_ih, self.bias_ih)
            return y, (hT, cT)
        # ONE draw of the GPNN's tensors for all steps of this call when its sample flag is raised (model.py:1721-1723)
        Wg, bg, cf = self.gpnn.sampled()
        c4 = self.gpnn.coef4(cf)
        fused = ops.lstm_recurrent_gp_supported(self.hidden_size, self.weights_hh)
        if 1 <= gt <= 4 and fused:
            # GPNN on one gate: the whole layer on the fused step kernels.  The GPNN's affine map over
            # [inp|h] splits into an input part (batched over T with the other gates' input GEMM) and a
            # hidden part that takes that gate's row block of the recurrent weight; the mixture is the
            # gate's activation inside the step kernel.  bias_ih enters twice, as in the reference.
            g, E, H = gt - 1, self.input_size, self.hidden_size
            xw_std = ops.linear(inputs, self.weights_ih, 2.0 * self.bias_ih)
            xw_gp = ops.linear(inputs, Wg[:, :E], bg)
            xw = torch.cat([xw_std[..., :g * H], xw_gp, xw_std[..., (g + 1) * H:]], -1)
            w_rec = torch.cat([self.weights_hh[:g * H], Wg[:, E:], self.weights_hh[(g + 1) * H:]], 0)
            y, hT, cT = ops.lstm_recurrent_gp(xw, hx, cx, w_rec, c4, g)
            return y, (hT, cT)
        if gt in (6, 7) and fused:
            if gt == 6:  # the hidden projection of all four gates is the GPNN of h (no bias_ih on that side)
                xw = ops.linear(inputs, self.weights_ih, self.bias_ih)
                y, hT, cT = ops.lstm_recurrent_gp(xw, hx, cx, Wg, c4, 4, bg)
            else:        # the input projection is the GPNN of the inputs; the hidden side carries bias_ih
                xw = ops.gp_mix(ops.linear(inputs, Wg, bg), c4) + self.bias_ih
                y, hT, cT = ops.lstm_recurrent_gp(xw, hx, cx, self.weights_hh)
            return y, (hT, cT)
        if gt == 5 and self.hidden_size % 64 == 0 and self.input_size == self.hidden_size and fused:
            # the cell state enters every step through the GPNN (model.py:1759-1760): a second recurrent product
            # c_{t-1} Wg^T per step, launched in front of the fused step kernel, which applies bias + mixture.  The
            # GPNN is built on input_size inputs but fed the H-wide cell state (model.py:1694,1760): any other shape
            # falls through to the step-wise loop below, which raises the reference's shape error.
            xw = ops.linear(inputs, self.weights_ih, 2.0 * self.bias_ih)  # bias_ih enters on both sides, as in the reference
            y, hT, cT = ops.lstm_recurrent_gp(xw, hx, cx, self.weights_hh, c4, 5, bg, Wg)
            return y, (hT, cT)

        def gp(v):
            return ops.gp_mix(ops.linear(v, Wg, bg), c4)

        # input-side projection of all steps in one GEMM (the reference does it per step)
        xw_all = gp(inputs) if gt == 7 else ops.linear(inputs, self.weights_ih, self.bias_ih)
        outs = []
        for t in range(T):
            hw = gp(hx) if gt == 6 else ops.linear(hx, self.weights_hh, self.bias_ih)
            if gt == 5:
                cx = gp(cx)
            if 1 <= gt <= 4:
                hx, cx = ops.lstm_cell(xw_all[t], hw, cx, gp(torch.cat([inputs[t], hx], -1)), gt - 1)
            else:
                hx, cx = ops.lstm_cell(xw_all[t], hw, cx)
            outs.append(hx)
        return torch.stack(outs, 0), (hx, cx)


    def _forward_gpnn2(self, inputs, hx, cx):
        """gpnn_type 4 (reference model.py:1744-1771): GPNN2 with fresh frequencies at every time step, on
        the overridden gate's pre-activation (gate types 1-4), on the cell state (5) or as the hidden /
        input projection (6 / 7).  Step-wise, like the reference."""
        gt, H = self.gate_type, self.hidden_size
        T = inputs.shape[0]
        gp = self.gpnn
        if 1 <= gt <= 6 and gp.skip_act and ops.lstm_recurrent_gpnn2_supported(H, gp.n_MC_terms):
            # the whole layer from one autograd node, 4-6 skinny launches per step (ops._LSTMRecurrentGPNN2); bias_ih enters
            # on both sides where the reference adds F.linear(hx, weights_hh, bias_ih) (gate types 1-5), once for type 6
            acts = sum(1 << GPNN._SLOT[a] for a in gp.act_set)
            xw = ops.linear(inputs, self.weights_ih, self.bias_ih if gt == 6 else 2.0 * self.bias_ih)
            mode, gate = (0, gt - 1) if gt <= 4 else ((1, 0) if gt == 5 else (2, 0))
            y, hT, cT = ops.lstm_recurrent_gpnn2(xw, hx, cx, None if gt == 6 else self.weights_hh, gp.coef.weight, gp.coef.bias,
                                                 gp.frequency_mean, gp.frequency_lgstd, gp.step_noises(T), gate, acts, mode)
            return y, (hT, cT)
        if (gt == 7 and gp.skip_act and ops.lstm_recurrent_gpnn2_supported(self.input_size, gp.n_MC_terms)
                and ops.lstm_recurrent_gp_supported(H, self.weights_hh)):
            # the input projection of every step in a handful of batched launches (ops._GPNN2Steps), then the plain fused
            # recurrence on it; bias_ih rides on the hidden side in the reference (model.py:1748)
            acts = sum(1 << GPNN._SLOT[a] for a in gp.act_set)
            xw = ops.gpnn2_steps(inputs, gp.coef.weight, gp.coef.bias + self.bias_ih, gp.frequency_mean, gp.frequency_lgstd,
                                 gp.step_noises(T), acts)
            y, hT, cT = ops.lstm_recurrent_gp(xw, hx, cx, self.weights_hh)
            return y, (hT, cT)
        xw_all = None if gt == 7 else ops.linear(inputs, self.weights_ih, self.bias_ih)
        outs = []
        zero = None
        for t in range(T):
            if gt == 6:
                xw, hw = xw_all[t], self.gpnn(hx, t)
            elif gt == 7:
                xw, hw = self.gpnn(inputs[t], t), ops.linear(hx, self.weights_hh, self.bias_ih)
            else:
                xw, hw = xw_all[t], ops.linear(hx, self.weights_hh, self.bias_ih)
            if gt == 5:
                cx = self.gpnn(cx, t)
            if 1 <= gt <= 4:
                pre = xw + hw  # the gate's pre-activation is the GPNN2's input
                if zero is None:
                    zero = torch.zeros_like(pre)
                hx, cx = ops.lstm_cell(pre, zero, cx, self.gpnn(pre[:, (gt - 1) * H:gt * H].contiguous(), t), gt - 1)
            else:
                hx, cx = ops.lstm_cell(xw, hw, cx)
            outs.append(hx)
        return torch.stack(outs, 0), (hx, cx)


class GPLSTM(_Site):
    """Reference model.py:1609-1671: which of the two layers is a GP cell is decided by the LENGTH of
    the ``gpnn_type`` string ('00' plain; 'gt' GP cell then nn.LSTM; 'gtx' nn.LSTM then GP cell;
    'gtg2x' two GP cells with gates g and g2)."""

    def __init__(self, input_size, hidden_size, num_layers=1, bias=True, dropout=0., gpnn_type='00'):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers, self.gpnn_type = input_size, hidden_size, num_layers, gpnn_type
        g = gpnn_type
        cells = []
        if int(g[0]) != 0:
            if len(g) == 2:
                cells = [GPLSTMCell(input_size, hidden_size, int(g[0]), int(g[1])),
                         _LSTMParams(hidden_size, hidden_size, num_layers - 1, 0.0)]
            elif len(g) == 3:
                cells = [_LSTMParams(hidden_size, hidden_size, num_layers - 1, 0.0),
                         GPLSTMCell(input_size, hidden_size, int(g[0]), int(g[1]))]
            else:
                cells = [GPLSTMCell(input_size, hidden_size, int(g[0]), int(g[1])),
                         GPLSTMCell(input_size, hidden_size, int(g[2]), int(g[1]))]
        else:
            cells = [_LSTMParams(hidden_size, hidden_size, num_layers, float(dropout))]
        self.rnn = nn.ModuleList(cells)

    def forward(self, inputs, hidden=None):
        g = self.gpnn_type
        h0, c0 = hidden
        if int(g[0]) == 0:
            return self.rnn[0](inputs, hidden)
        if len(g) == 2:
            y, (h1, c1) = self.rnn[0](inputs, (h0[0], c0[0]))
            y, (hr, cr) = self.rnn[1](y, (h0[1:], c0[1:]))
            return y, (torch.cat([h1.unsqueeze(0), hr], 0), torch.cat([c1.unsqueeze(0), cr], 0))
        if len(g) == 3:
            y, (hr, cr) = self.rnn[0](inputs, (h0[:1], c0[:1]))
            y, (h1, c1) = self.rnn[1](y, (h0[1], c0[1]))
            return y, (torch.cat([hr, h1.unsqueeze(0)], 0), torch.cat([cr, c1.unsqueeze(0)], 0))
        y, (h1, c1) = self.rnn[0](inputs, (h0[0], c0[0]))
        y, (h2, c2) = self.rnn[1](y, (h0[1], c0[1]))
        return y, (torch.stack([h1, h2]), torch.stack([c1, c2]))


class GaussRNNModel(_RNNLM):
    """Reference model.py:1317-1366."""

    def __init__(self, rnn_type, ntoken, ninp, nhid, nlayers, dropout=0.5, tie_weights=False, gauss_pos='00'):
        super().__init__()
        self._check_type(rnn_type)
        if tie_weights and nhid != ninp:
            raise ValueError("When using the tied flag, nhid must be equal to emsize.")
        self.rnn_type, self.nhid, self.nlayers, self.p = rnn_type, nhid, nlayers, dropout
        self._make_encoder(ntoken, ninp)
        self.rnn = GPLSTM(ninp, nhid, nlayers, dropout=dropout, gpnn_type=gauss_pos)
        self._init_io(ntoken, ninp, nhid, tie_weights)
        self.noise_state = bind_state(self, NoiseState())

    def forward(self, x, hidden):
        emb = self._embed_dropout(x, self.encoder.weight, self.p, 0)
        out, hidden = self.rnn(emb, hidden)
        out = self._dropout(out.contiguous(), self.p, 1)
        return self.decoder(out), hidden


class VNN(_Site):
    """Reference model.py:2534-2579: after every time step h += eps * exp(hidden_lgstd) with
    eps ~ N(0, 0.1) of shape (1, H) (shared by the batch columns), in train mode only.  The KL
    (model.py:2545-2550) uses the LAST step's h as the mean and contains exp(2*h) (sic)."""

    def __init__(self, input_size):
        super().__init__()
        self.input_size = input_size
        self.sample = True
        s = 1.0 / math.sqrt(input_size)
        self.hidden_lgstd = nn.Parameter(torch.empty(1, input_size).uniform_(2 * np.log(s), np.log(s)))
        self.hidden_mean = None

    def kl_divergence(self, prior=None):
        hm, lg = self.hidden_mean, self.hidden_lgstd
        return torch.mean(hm ** 2 - lg * 2. + torch.exp(hm * 2) - 1) / 2.  # small (B,H) glue, as written in the reference

    def noise_rows(self, T, eps=None):
        """(T, H) rows eps_t * exp(lgstd), eps_t ~ N(0, 0.1) from the Philox stream (or injected)."""
        H = self.input_size
        if eps is None and self._st().source == "torch":
            # one (1, H) draw of N(0, 0.1) per time step, as the reference's per-step call makes it (model.py:2555-2561)
            eps = torch.cat([torch_eps((1, H), "cpu", 0.1) for _ in range(T)], 0).to(self.hidden_lgstd.device)
        if eps is None:
            st = self._st()
            eps = ops.philox_normal(T * H, st.seed, ops.L.STREAM_WEIGHT + self._site_base, st.step,
                                    self.hidden_lgstd.device).view(T, H) * 0.1
        return eps * torch.exp(self.hidden_lgstd)


class VLSTMCell(_LoopCell):
    """Reference model.py:2471-2531."""

    def __init__(self, input_size, hidden_size, vnn_type=0):
        super().__init__()
        self.vnn_type = vnn_type
        self.vnn = VNN(input_size)
        self._params(input_size, hidden_size)
        self.eps_override = None  # (T, H) injected noise for parity tests

    def draws_noise(self):
        return self.vnn_type == 1 and bool(self.vnn.sample)

    def forward(self, inputs, hid=None):
        if inputs.dim() == 2:
            inputs = inputs.unsqueeze(0)
        T, B, _ = inputs.shape
        if hid is None:
            z = torch.zeros(B, self.hidden_size, dtype=inputs.dtype, device=inputs.device)
            hid = (z, z)
        hx, cx = hid
        noisy = self.vnn_type == 1 and self.training and self.vnn.sample
        rows = self.vnn.noise_rows(T, self.eps_override) if noisy else None
        # the whole layer in the fused LSTM path (input GEMM batched over T, one launch per time step in
        # each direction); the reference's quirk -- bias_ih added twice, bias_hh unused -- is the bias pair
        # (bias_ih, bias_ih); the per-step noise row is added to h inside the step kernel
        y, hT, cT = ops.lstm_layer(inputs, hx, cx, self.weights_ih, self.weights_hh, self.bias_ih, self.bias_ih, rows)
        if self.vnn_type == 1:
            self.vnn.hidden_mean = hT - rows[T - 1] if noisy else hT  # the last step's h BEFORE its noise
        return y, (hT, cT)


class VariationalLSTM(_Site):
    """Reference model.py:2426-2468: always two VLSTMCells; ``vlstm_type`` '00'/'01'/'10'/'11'."""

    def __init__(self, input_size, hidden_size, num_layers=1, bias=True, dropout=0., vlstm_type='00'):
        super().__init__()
        self.vlstm_type = vlstm_type
        self.rnn = nn.ModuleList([VLSTMCell(input_size, hidden_size, vnn_type=int(vlstm_type[0])),
                                  VLSTMCell(input_size, hidden_size, vnn_type=int(vlstm_type[1]))])

    def forward(self, inputs, hidden=None):
        h0, c0 = hidden
        y, (h1, c1) = self.rnn[0](inputs, (h0[0], c0[0]))
        y, (h2, c2) = self.rnn[1](y, (h0[1], c0[1]))
        return y, (torch.stack([h1, h2]), torch.stack([c1, c2]))


class VariationalRNNModel(_RNNLM):
    """Reference model.py:2373-2423."""

    def __init__(self, rnn_type, ntoken, ninp, nhid, nlayers, dropout=0.5, tie_weights=False, v_pos='00'):
        super().__init__()
        self._check_type(rnn_type)
        if tie_weights and nhid != ninp:
            raise ValueError("When using the tied flag, nhid must be equal to emsize.")
        self.rnn_type, self.nhid, self.nlayers, self.p = rnn_type, nhid, nlayers, dropout
        self._make_encoder(ntoken, ninp)
        self.rnn = VariationalLSTM(ninp, nhid, nlayers, dropout=dropout, vlstm_type=v_pos)
        self._init_io(ntoken, ninp, nhid, tie_weights)
        self.noise_state = bind_state(self, NoiseState())

    def forward(self, x, hidden):
        emb = self._embed_dropout(x, self.encoder.weight, self.p, 0)
        out, hidden = self.rnn(emb, hidden)
        out = self._dropout(out.contiguous(), self.p, 1)
        return self.decoder(out), hidden
