"""Pure-PyTorch fp32 CPU restatement of the steps/pytorchnn hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every function is stateless and
works on a flat ``state_dict`` (name -> tensor) whose key names are the
reference's, so a checkpoint written by either side can
be fed to the other.  Citations are ``/root/reference/steps/pytorchnn/<file>:line``.

Noise (eps) is always an explicit argument: ``None`` means eval mode / mean
weights, a tensor means "this is the N(0,1) draw the layer would have made".
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# variational weights + KL
# ----------------------------------------------------------------------------


def sampled_weight(mu, lgstd, eps):
    """W = mu + exp(lgstd) * eps   (model.py:1083-1107; eval -> W = mu)."""
    if eps is None:
        return mu
    return mu + torch.exp(lgstd) * eps


def bayes_linear(x, mu, lgstd, eps=None):
    """BayesLinear.forward, bias=False default (model.py:1050,1127-1129)."""
    return F.linear(x, sampled_weight(mu, lgstd, eps))


def kl_mean_form(mu, lgstd):
    """mean(mu^2 - 2 lgstd + exp(2 lgstd)) / 2 -- no '-1', mean not sum
    (model.py:1115, :762-765, :1255)."""
    return torch.mean(mu ** 2.0 - lgstd * 2.0 + torch.exp(lgstd * 2.0)) / 2.0


def kl_mean_form_minus1(mu, lgstd):
    """GPNN flavour, with the '-1' (model.py:1816-1826)."""
    return torch.mean(mu ** 2 - lgstd * 2.0 + torch.exp(lgstd * 2.0) - 1) / 2.0


# ----------------------------------------------------------------------------
# Transformer pieces
# ----------------------------------------------------------------------------


def positional_table(max_len, d_model):
    """Sinusoidal table (max_len, 1, d) (model.py:97-103)."""
    pe = torch.zeros(max_len, d_model)
    pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.unsqueeze(1)


def causal_mask(T):
    """Additive float mask, -inf above the diagonal (model.py:1258-1262)."""
    m = torch.full((T, T), float("-inf"))
    return torch.triu(m, diagonal=1)


def attention_core(q, k, v, nhead, mask, drop_p=0.0):
    """q,k,v: (T,B,d) already projected.  Head index = b*nhead + head
    (model.py:889-920).  Returns (T,B,d) before o_net.  ``drop_p`` > 0: the reference's dropout on the attention
    probabilities (:913), torch's generator -- only the CPU-baseline timing of bench.py turns it on."""
    T, B, d = q.shape
    hd = d // nhead
    q = q * (float(hd) ** -0.5)
    q = q.contiguous().view(T, B * nhead, hd).transpose(0, 1)
    k = k.contiguous().view(-1, B * nhead, hd).transpose(0, 1)
    v = v.contiguous().view(-1, B * nhead, hd).transpose(0, 1)
    s = torch.bmm(q, k.transpose(1, 2))
    if mask is not None:
        s = s + mask.unsqueeze(0)
    p = torch.softmax(s, dim=-1)
    if drop_p > 0.0:
        p = F.dropout(p, drop_p, True)
    o = torch.bmm(p, v)
    return o.transpose(0, 1).contiguous().view(T, B, d)


def mha(x, sd, pre, nhead, mask, eps=None, drop_p=0.0):
    """MultiheadAttention (fused qkv_net, model.py:871-928) or
    BayesMultiheadAttention (separate q/k/v nets, Bayesian o_net,
    model.py:971-1019), chosen by which keys exist under ``pre``."""
    if pre + "qkv_net.weight" in sd:
        qkv = F.linear(x, sd[pre + "qkv_net.weight"], sd[pre + "qkv_net.bias"])
        q, k, v = qkv.chunk(3, dim=-1)
    elif pre + "in_proj_weight" in sd:  # nn.MultiheadAttention (TransformerModel baseline)
        qkv = F.linear(x, sd[pre + "in_proj_weight"], sd[pre + "in_proj_bias"])
        q, k, v = qkv.chunk(3, dim=-1)
    else:
        q = F.linear(x, sd[pre + "q_net.weight"], sd[pre + "q_net.bias"])
        k = F.linear(x, sd[pre + "k_net.weight"], sd[pre + "k_net.bias"])
        v = F.linear(x, sd[pre + "v_net.weight"], sd[pre + "v_net.bias"])
    a = attention_core(q, k, v, nhead, mask, drop_p)
    if pre + "o_net.weight_mean" in sd:
        return bayes_linear(a, sd[pre + "o_net.weight_mean"], sd[pre + "o_net.weight_lgstd"], eps)
    if pre + "out_proj.weight" in sd:
        return F.linear(a, sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"])
    return F.linear(a, sd[pre + "o_net.weight"], sd[pre + "o_net.bias"])


def gp_mixture(z, coef, acts):
    """sum_i act_i(z) * coef[i]  (model.py:1885-1899)."""
    out = 0
    for i, a in enumerate(acts):
        out = out + getattr(F, a)(z) * coef[i]
    return out


def gpnn2(x, sd, pre, eps=None, gelu=False):
    """GPNN2.forward (model.py:2061-2076): frequency = mean + eps * exp(lgstd) in train mode (eps
    (input_dim, n_MC), the module's single N(0,1) draw), features = x @ frequency, output =
    coef((features + sum_act act(features)) / sqrt(n_MC)) with the act set {sigmoid, tanh, relu, gelu}."""
    fm, fl = sd[pre + "frequency_mean"], sd[pre + "frequency_lgstd"]
    freq = fm if eps is None else fm + eps * torch.exp(fl)
    z = x.matmul(freq)
    a = z + torch.sigmoid(z) + torch.tanh(z) + F.relu(z)
    if gelu:  # the Transformer layer's set has gelu (model.py:2264), the LSTM cells' does not (:1699-1702)
        a = a + F.gelu(z)
    return F.linear(a / math.sqrt(fm.shape[1]), sd[pre + "coef.weight"], sd[pre + "coef.bias"])


def encoder_layer(x, sd, pre, nhead, mask, eps=None, drop_p=0.0):
    """Post-LN block (model.py:1037-1046, 1162-1176, 2274-2295).  ``eps`` is the
    single draw this layer makes: for the FFN position it belongs to linear2,
    for the MHA position to o_net.  A layer with ``gpnn.*`` keys is the
    GaussTransformerEncoderLayer: gpnn replaces GELU(linear1(x)); under
    train.py GPNN.sample stays False so its forward is deterministic."""
    d = x.shape[-1]
    att_eps = eps if pre + "self_attn.o_net.weight_mean" in sd else None
    dr = (lambda t: F.dropout(t, drop_p, True)) if drop_p > 0.0 else (lambda t: t)  # model.py:1041-1045 sites
    a = mha(x, sd, pre + "self_attn.", nhead, mask, att_eps, drop_p)
    x = F.layer_norm(x + dr(a), (d,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
    if pre + "gpnn.weights_mean" in sd:  # eps: None, or the dict of gpnn_params (GPNN.sample raised, model.py:1871-1883)
        h = gpnn(x, sd, pre + "gpnn.", ["tanh", "sigmoid", "relu", "gelu"], eps)
    elif pre + "gpnn.frequency_mean" in sd:  # gauss_pos 4: GPNN2 random features (model.py:2036-2076)
        h = gpnn2(x, sd, pre + "gpnn.", eps, gelu=True)
    else:
        h = F.gelu(F.linear(x, sd[pre + "linear1.weight"], sd[pre + "linear1.bias"]))
    h = dr(h)
    if pre + "linear2.weight_mean" in sd:
        f = bayes_linear(h, sd[pre + "linear2.weight_mean"], sd[pre + "linear2.weight_lgstd"], eps)
    else:
        f = F.linear(h, sd[pre + "linear2.weight"], sd[pre + "linear2.bias"])
    return F.layer_norm(x + dr(f), (d,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])


def transformer_lm(src, sd, nhead, eps=None, drop_p=0.0):
    """BayesTransformerModel / TransformerModel / GaussTransformerModel forward
    with dropout off (model.py:1274-1309, 159-171).  src (T,B) int64 -> logits
    (T,B,V).  ``eps``: the one N(0,1) draw of the Bayesian tensor (layer-0
    linear2 / o_net, or embed for 'EMB'); None = eval."""
    T = src.shape[0]
    d = sd["encoder.weight"].shape[1]
    x = F.embedding(src, sd["encoder.weight"]) * math.sqrt(d)
    if "embed_mean" in sd:  # 'EMB' (model.py:1286-1290)
        x = F.linear(x, sampled_weight(sd["embed_mean"], sd["embed_lgstd"], eps))
    x = x + sd["pos_encoder.pe"][:T]
    if drop_p > 0.0:  # PositionalEncoding's dropout (model.py:116-117); parity paths run with drop_p = 0
        x = F.dropout(x, drop_p, True)
    mask = causal_mask(T)
    # nn.TransformerEncoder keys are transformerlayers.layers.N.*, ours .N.*
    base = "transformerlayers.layers." if any(k.startswith("transformerlayers.layers.") for k in sd) \
        else "transformerlayers."
    i = 0
    while (base + "%d.norm1.weight" % i) in sd:
        x = encoder_layer(x, sd, base + "%d." % i, nhead, mask, eps if i == 0 else None, drop_p)
        i += 1
    if "embed_mean" in sd:  # model.py:1302-1304 (mean weights, transposed)
        x = F.linear(x, sd["embed_mean"].t())
    return F.linear(x, sd["decoder.weight"], sd["decoder.bias"])


def kl_transformer(sd, bayes_pos):
    """Which tensor's KL train.py adds (train.py:340-356)."""
    if bayes_pos == "FFN":
        p = "transformerlayers.0.linear2."
        return kl_mean_form(sd[p + "weight_mean"], sd[p + "weight_lgstd"])
    if bayes_pos == "MHA":
        p = "transformerlayers.0.self_attn.o_net."
        return kl_mean_form(sd[p + "weight_mean"], sd[p + "weight_lgstd"])
    if bayes_pos == "EMB":
        return kl_mean_form(sd["embed_mean"], sd["embed_lgstd"])
    return torch.zeros(())


# ----------------------------------------------------------------------------
# LSTM pieces
# ----------------------------------------------------------------------------

LSTM_EPS_ORDER = ("weight_hh_lgstd_1", "weight_ih_lgstd_1", "bias_hh_lgstd_1", "bias_ih_lgstd_1",
                  "weight_hh_lgstd_2", "weight_ih_lgstd_2", "bias_hh_lgstd_2", "bias_ih_lgstd_2")


def bayes2lstm_weights(sd, pre, pos, eps8=None):
    """flat_parameters (model.py:705-732): noise only on gate rows
    [(pos-1)H, pos*H) of all 8 tensors; eps8 follows LSTM_EPS_ORDER
    (draw order of model.py:668-703)."""
    out = {}
    for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
        for layer in (1, 2):
            out["%s_%d" % (name, layer)] = sd["%s%s_mean_%d" % (pre, name, layer)].clone()
    if 1 <= pos <= 4 and eps8 is not None:
        for key, eps in zip(LSTM_EPS_ORDER, eps8):
            lg = sd[pre + key]
            H = lg.shape[0]
            tgt = key.replace("_lgstd", "")
            out[tgt][(pos - 1) * H:pos * H] += torch.exp(lg) * eps
    return out


def lstm_layer(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """One layer over T steps; gates i,f,g,o; c'=sig(f)c+sig(i)tanh(g);
    h'=sig(o)tanh(c')  (what _VF.lstm computes, model.py:812)."""
    outs = []
    xw = F.linear(x, w_ih, b_ih)
    for t in range(x.shape[0]):
        g = xw[t] + F.linear(h, w_hh, b_hh)
        i, f, gg, o = g.chunk(4, dim=-1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, 0), h, c


def bayes_rnn_lm(x, hidden, sd, pos, eps8=None):
    """BayesRNNModel.forward, dropout off (model.py:217-222).  x (T,B) int64,
    hidden = (h0,c0) each (2,B,H).  Returns logits (T,B,V), (hT,cT)."""
    emb = F.embedding(x, sd["encoder.weight"])
    w = bayes2lstm_weights(sd, "rnn.", pos, eps8)
    h0, c0 = hidden
    y, h1, c1 = lstm_layer(emb, h0[0], c0[0], w["weight_ih_1"], w["weight_hh_1"], w["bias_ih_1"], w["bias_hh_1"])
    y, h2, c2 = lstm_layer(y, h0[1], c0[1], w["weight_ih_2"], w["weight_hh_2"], w["bias_ih_2"], w["bias_hh_2"])
    logits = F.linear(y, sd["decoder.weight"], sd["decoder.bias"])
    return logits, (torch.stack([h1, h2]), torch.stack([c1, c2]))


def rnn_lm(x, hidden, sd):
    """RNNModel with nn.LSTM, eval mode (model.py:61-66): keys rnn.weight_ih_l{k}."""
    y = F.embedding(x, sd["encoder.weight"])
    h0, c0 = hidden
    hs, cs = [], []
    layer = 0
    while "rnn.weight_ih_l%d" % layer in sd:
        y, h, c = lstm_layer(y, h0[layer], c0[layer], sd["rnn.weight_ih_l%d" % layer], sd["rnn.weight_hh_l%d" % layer],
                             sd["rnn.bias_ih_l%d" % layer], sd["rnn.bias_hh_l%d" % layer])
        hs.append(h)
        cs.append(c)
        layer += 1
    logits = F.linear(y, sd["decoder.weight"], sd["decoder.bias"])
    return logits, (torch.stack(hs), torch.stack(cs))


def rnn_lm_train(x, hidden, sd, p):
    """RNNModel.forward in training mode (model.py:61-66 with nn.LSTM(dropout=p), :35): dropout (torch's generator) on
    the embeddings, between the LSTM layers and on the output.  Used for the CPU baseline timing of BASELINE configs[0]."""
    y = F.dropout(F.embedding(x, sd["encoder.weight"]), p, True)
    h0, c0 = hidden
    hs, cs = [], []
    layer = 0
    while "rnn.weight_ih_l%d" % layer in sd:
        if layer > 0:
            y = F.dropout(y, p, True)
        y, h, c = lstm_layer(y, h0[layer], c0[layer], sd["rnn.weight_ih_l%d" % layer], sd["rnn.weight_hh_l%d" % layer],
                             sd["rnn.bias_ih_l%d" % layer], sd["rnn.bias_hh_l%d" % layer])
        hs.append(h)
        cs.append(c)
        layer += 1
    logits = F.linear(F.dropout(y, p, True), sd["decoder.weight"], sd["decoder.bias"])
    return logits, (torch.stack(hs), torch.stack(cs))


def kl_bayes2lstm(sd, pre, pos):
    """Bayes2LSTM.kl_divergence (model.py:734-765).  pos 1..4: layer-1 tensors only; weights and
    biases each get their own mean.  pos 5 (:746-755, as written): layer 1's [hh|ih] plus
    [hh of layer 2 | ih of layer 1 again], for means and log-sigmas alike."""
    if pos == 5:
        c = lambda a, b: torch.cat([sd[pre + a], sd[pre + b]], -1)  # noqa: E731
        wm = c("weight_hh_mean_1", "weight_ih_mean_1") + c("weight_hh_mean_2", "weight_ih_mean_1")
        wl = c("weight_hh_lgstd_1", "weight_ih_lgstd_1") + c("weight_hh_lgstd_2", "weight_ih_lgstd_1")
        bm = c("bias_hh_mean_1", "bias_ih_mean_1") + c("bias_hh_mean_2", "bias_ih_mean_1")
        bl = c("bias_hh_lgstd_1", "bias_ih_lgstd_1") + c("bias_hh_lgstd_2", "bias_ih_lgstd_1")
        return kl_mean_form(wm, wl) + kl_mean_form(bm, bl)
    if not (1 <= pos <= 4):
        return torch.zeros(())
    H = sd[pre + "weight_hh_lgstd_1"].shape[0]
    r = slice((pos - 1) * H, pos * H)
    wm = torch.cat([sd[pre + "weight_hh_mean_1"][r], sd[pre + "weight_ih_mean_1"][r]], -1)
    wl = torch.cat([sd[pre + "weight_hh_lgstd_1"], sd[pre + "weight_ih_lgstd_1"]], -1)
    bm = torch.cat([sd[pre + "bias_hh_mean_1"][r], sd[pre + "bias_ih_mean_1"][r]], -1)
    bl = torch.cat([sd[pre + "bias_hh_lgstd_1"], sd[pre + "bias_ih_lgstd_1"]], -1)
    return kl_mean_form(wm, wl) + kl_mean_form(bm, bl)


# ----------------------------------------------------------------------------
# loss / step / scoring
# ----------------------------------------------------------------------------


def cross_entropy_mean(logits, targets):
    """nn.CrossEntropyLoss() on (M,V) (train.py:233,332)."""
    return F.cross_entropy(logits.reshape(-1, logits.shape[-1]), targets.reshape(-1))


def token_nll(logits, targets):
    """Per-token NLL (M,), the quantity PPL and n-best scores are sums of."""
    return F.cross_entropy(logits.reshape(-1, logits.shape[-1]), targets.reshape(-1), reduction="none")


def sentence_score(logits, targets):
    """len * CE_mean = sum of token NLL (compute_sentence_scores_bayes_jianwei.py:168-170)."""
    return targets.numel() * cross_entropy_mean(logits, targets)


def clip_and_sgd(params, grads, bufs, lr, clip, momentum=0.9):
    """clip_grad_norm_(params, clip) then SGD(momentum, wd=0)
    (train.py:419-420,466).  ``bufs`` entries may be None (first step).
    Distinct tensors only (a tied weight appears once).  In place."""
    total = torch.sqrt(sum((g.detach() ** 2).sum() for g in grads))
    coef = torch.clamp(clip / (total + 1e-6), max=1.0)
    for i, (p, g) in enumerate(zip(params, grads)):
        g = g * coef
        if bufs[i] is None:
            bufs[i] = g.clone()
        else:
            bufs[i].mul_(momentum).add_(g)
        p.data.add_(bufs[i], alpha=-lr)
    return total


def transformer_train_loss(src, targets, sd, nhead, bayes_pos, eps, kl_scale, drop_p=0.0):
    """loss = CE_mean + KL * seq_len/len(train_data) (train.py:332-412)."""
    logits = transformer_lm(src, sd, nhead, eps, drop_p)
    mle = cross_entropy_mean(logits, targets)
    kl = kl_transformer(sd, bayes_pos) * kl_scale
    return mle + kl, mle, kl


# ----------------------------------------------------------------------------
# GP / Variational LSTMs (Python time loops in the reference)
# ----------------------------------------------------------------------------
def gpnn_params(sd, pre, eps=None):
    """(weights, bias, coef) of one GPNN forward (model.py:1871-1883).  ``eps`` None: the mean tensors (eval mode, or
    ``sample`` False as under train.py); a dict with any of "coef" / "weights" / "bias": training with ``sample`` raised --
    each tensor that HAS an lgstd (gpnn_type 1: coef; 2: weights + bias; 3: all) is mean + exp(lgstd) * eps."""
    out = []
    for name in ("weights", "bias", "coef"):
        t = sd[pre + name + "_mean"]
        if eps is not None and pre + name + "_lgstd" in sd:
            t = t + torch.exp(sd[pre + name + "_lgstd"]) * eps[name]
        out.append(t)
    return out


def gpnn(x, sd, pre, acts, eps=None):
    """GPNN.forward (model.py:1863-1902); ``eps``: see gpnn_params."""
    w, b, coef = gpnn_params(sd, pre, eps)
    return gp_mixture(F.linear(x, w, b), coef, acts)


def _gp_acts(gate_type):
    return ["sigmoid"] if gate_type == 2 else ["sigmoid", "tanh", "relu"]  # model.py:1688-1697


def gp_lstm_cell(x, h, c, sd, pre, gate_type, eps=None):
    """GPLSTMCell.forward (model.py:1720-1777): bias_ih is added twice, bias_hh unused.  gpnn_type 0-3
    (GPNN, keys gpnn.weights_mean ...): gate 1-4 is a GPNN of [inp|h]; 5: c = GPNN(c) first; 6/7: the
    hidden/input projection is a GPNN.  gpnn_type 4 (GPNN2, keys gpnn.frequency_mean ...): gate 1-4 is
    GPNN2 of that gate's PRE-ACTIVATION, 5: c = GPNN2(c), 6/7 as above with GPNN2; every call draws new
    frequencies in train mode: ``eps`` = list of T tensors (input_dim, n_MC) or None (eval).  For a GPNN cell ``eps`` may be
    the dict of gpnn_params: GPNN.sample raised, ONE draw for all steps of this call (model.py:1721-1723)."""
    acts = _gp_acts(gate_type)
    w_ih, b_ih, w_hh = sd[pre + "weights_ih"], sd[pre + "bias_ih"], sd[pre + "weights_hh"]
    two = pre + "gpnn.frequency_mean" in sd
    outs = []
    for t in range(x.shape[0]):
        inp = x[t]
        e = None if (eps is None or not two) else eps[t]
        gp_of = (lambda v: gpnn2(v, sd, pre + "gpnn.", e)) if two else (lambda v: gpnn(v, sd, pre + "gpnn.", acts, eps))
        if gate_type == 6:
            gates = F.linear(inp, w_ih, b_ih) + gp_of(h)
        elif gate_type == 7:
            gates = gp_of(inp) + F.linear(h, w_hh, b_ih)
        else:
            gates = F.linear(inp, w_ih, b_ih) + F.linear(h, w_hh, b_ih)
        i, f, g, o = gates.chunk(4, 1)
        if two:
            i = gp_of(i) if gate_type == 1 else torch.sigmoid(i)
            f = gp_of(f) if gate_type == 2 else torch.sigmoid(f)
            g = gp_of(g) if gate_type == 3 else torch.tanh(g)
            o = gp_of(o) if gate_type == 4 else torch.sigmoid(o)
        else:
            gp = gp_of(torch.cat([inp, h], -1)) if 1 <= gate_type <= 4 else None
            i = gp if gate_type == 1 else torch.sigmoid(i)
            f = gp if gate_type == 2 else torch.sigmoid(f)
            g = gp if gate_type == 3 else torch.tanh(g)
            o = gp if gate_type == 4 else torch.sigmoid(o)
        if gate_type == 5:
            c = gp_of(c)
        c = f * c + i * g
        h = o * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, 0), h, c


def _nn_lstm(x, h0, c0, sd, pre):
    """nn.LSTM stack stored under ``pre`` (weight_ih_l{k} ...), eval / dropout 0."""
    hs, cs, k = [], [], 0
    while pre + "weight_ih_l%d" % k in sd:
        x, h, c = lstm_layer(x, h0[k], c0[k], sd[pre + "weight_ih_l%d" % k], sd[pre + "weight_hh_l%d" % k],
                             sd[pre + "bias_ih_l%d" % k], sd[pre + "bias_hh_l%d" % k])
        hs.append(h)
        cs.append(c)
        k += 1
    return x, torch.stack(hs), torch.stack(cs)


def gauss_rnn_lm(x, hidden, sd, gauss_pos, eps=None):
    """GaussRNNModel.forward, dropout off (model.py:1355-1360 -> GPLSTM.forward :1638-1671).
    ``eps`` = {cell index: [T tensors]} for GPNN2 cells in train mode, {cell index: dict of gpnn_params} for GPNN cells
    whose ``sample`` flag is raised."""
    y = F.embedding(x, sd["encoder.weight"])
    h0, c0 = hidden
    g = gauss_pos
    eps = eps or {}
    if int(g[0]) == 0:
        y, hs, cs = _nn_lstm(y, h0, c0, sd, "rnn.rnn.0.")
    elif len(g) == 2:
        y, h1, c1 = gp_lstm_cell(y, h0[0], c0[0], sd, "rnn.rnn.0.", int(g[0]), eps.get(0))
        y, hr, cr = _nn_lstm(y, h0[1:], c0[1:], sd, "rnn.rnn.1.")
        hs, cs = torch.cat([h1.unsqueeze(0), hr]), torch.cat([c1.unsqueeze(0), cr])
    elif len(g) == 3:
        y, hr, cr = _nn_lstm(y, h0[:1], c0[:1], sd, "rnn.rnn.0.")
        y, h1, c1 = gp_lstm_cell(y, h0[1], c0[1], sd, "rnn.rnn.1.", int(g[0]), eps.get(1))
        hs, cs = torch.cat([hr, h1.unsqueeze(0)]), torch.cat([cr, c1.unsqueeze(0)])
    else:
        y, h1, c1 = gp_lstm_cell(y, h0[0], c0[0], sd, "rnn.rnn.0.", int(g[0]), eps.get(0))
        y, h2, c2 = gp_lstm_cell(y, h0[1], c0[1], sd, "rnn.rnn.1.", int(g[2]), eps.get(1))
        hs, cs = torch.stack([h1, h2]), torch.stack([c1, c2])
    return F.linear(y, sd["decoder.weight"], sd["decoder.bias"]), (hs, cs)


def kl_gauss_rnn(sd, gauss_pos):
    """train.py:366-376 + GPNN.kl_divergence (model.py:1816-1826, with the '-1')."""
    g = gauss_pos

    def one(pre, t):
        kl = torch.zeros(())
        if t in (1, 3):
            kl = kl + kl_mean_form_minus1(sd[pre + "coef_mean"], sd[pre + "coef_lgstd"])
        if t in (2, 3):
            kl = kl + kl_mean_form_minus1(sd[pre + "weights_mean"], sd[pre + "weights_lgstd"])
            kl = kl + kl_mean_form_minus1(sd[pre + "bias_mean"], sd[pre + "bias_lgstd"])
        return kl
    if not (int(g[0]) > 0 and 0 < int(g[1]) <= 3):
        return torch.zeros(())
    t = int(g[1])
    if len(g) < 3:
        return one("rnn.rnn.0.gpnn.", t)
    if len(g) == 3:
        return one("rnn.rnn.1.gpnn.", t)
    return one("rnn.rnn.0.gpnn.", t) + one("rnn.rnn.1.gpnn.", t)


def v_lstm_cell(x, h, c, sd, pre, eps_rows=None):
    """VLSTMCell.forward (model.py:2493-2531) + VNN (2571-2577): after each step h += eps_t *
    exp(hidden_lgstd); eps_rows (T,H) or None.  Also returns the last pre-noise h (VNN.hidden_mean)."""
    w_ih, b_ih, w_hh = sd[pre + "weights_ih"], sd[pre + "bias_ih"], sd[pre + "weights_hh"]
    outs, hm = [], None
    for t in range(x.shape[0]):
        gates = F.linear(x[t], w_ih, b_ih) + F.linear(h, w_hh, b_ih)
        i, f, g, o = gates.chunk(4, 1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        hm = h
        if eps_rows is not None:
            h = h + eps_rows[t] * torch.exp(sd[pre + "vnn.hidden_lgstd"])
        outs.append(h)
    return torch.stack(outs, 0), h, c, hm


def variational_rnn_lm(x, hidden, sd, v_pos, eps=None):
    """VariationalRNNModel.forward (model.py:2414-2419); eps = {cell: (T,H)} for train mode.
    Returns logits, hidden and the KL train.py would add (train.py:379-382; VNN.kl_divergence
    model.py:2545-2550 uses exp(2*hidden_mean), as written)."""
    y = F.embedding(x, sd["encoder.weight"])
    h0, c0 = hidden
    eps = eps or {}
    kl = torch.zeros(())
    hs, cs = [], []
    for cidx in (0, 1):
        pre = "rnn.rnn.%d." % cidx
        y, h, c, hm = v_lstm_cell(y, h0[cidx], c0[cidx], sd, pre, eps.get(cidx))
        hs.append(h)
        cs.append(c)
        if int(v_pos[cidx]) == 1:
            lg = sd[pre + "vnn.hidden_lgstd"]
            kl = kl + torch.mean(hm ** 2 - lg * 2.0 + torch.exp(hm * 2) - 1) / 2.0
    return F.linear(y, sd["decoder.weight"], sd["decoder.bias"]), (torch.stack(hs), torch.stack(cs)), kl


# ----------------------------------------------------------------------------
# Monte-Carlo weight-sample scoring (BASELINE.json configs[4]; NOT in the reference, which scores with mean
# weights, compute_sentence_scores_bayes_jianwei.py:225 -- SURVEY.md 8(e) defines it: S forward passes with the
# variational weights sampled, dropout off, sentence PROBABILITIES averaged)
# ----------------------------------------------------------------------------
def mc_sentence_score(nll_samples):
    """nll_samples: (S,) sentence NLLs (sum of token NLL) under S weight samples -> -log(mean_s exp(-NLL_s))."""
    v = torch.as_tensor(nll_samples, dtype=torch.float64)
    return float(-(torch.logsumexp(-v, 0) - math.log(v.numel())))


def mc_scores(nbest, vocab, sd, family, S, seed, tensor_ids, nhead=4, pos=3, get_input_and_target=None):
    """Oracle of compute_scores_batched(mc_samples=S).  ``family`` / ``pos`` / ``tensor_ids`` (Philox tensor ids = the
    engine's module numbering, host-side information):
      "tlm_ffn"     eps of layer-0 linear2                                    ids [linear2]
      "lstm_bayes"  the 8 tensors in LSTM_EPS_ORDER, pos = --L_bayes_pos      ids of the 8 tensors
      "tlm_gauss"   layer-0 GPNN with its sample flag raised: coef / weights / bias = ids[0] + 0 / 1 / 2 (a GPNN2 layer:
                    the one frequency draw of a forward, id ids[0], Philox step s * 1024)
      "lstm_gauss"  GP-LSTM cells (GPNN types 1-3), pos = --L_gauss_pos       ids {cell: id of its GPNN}
      "lstm_var"    VNN noise rows eps_t ~ N(0, 0.1), pos = --L_v_pos         ids {cell: id of its VNN}; row t of a
                    hypothesis is elements [t H, (t+1) H) of the stream, whatever the padded batch length
    Sample s uses philox.normal(n, seed, STREAM_WEIGHT + id, step = s): ONE model per sample for the whole list.
    LSTM: the state carried between utterances is the mean-weight state after the first hypothesis (:271-274).
    -> [(key-n, score)]"""
    from . import philox as P

    def draw(like, tid, smp):
        return torch.from_numpy(P.normal(like.numel(), seed, P.STREAM_WEIGHT + tid, smp)).view_as(like)

    def gp_eps(pre, tid, smp):
        return {n: draw(sd[pre + n + "_lgstd"], tid + k, smp) for k, n in enumerate(("coef", "weights", "bias"))
                if pre + n + "_lgstd" in sd}

    is_rnn = family.startswith("lstm")
    eps = []
    for smp in range(S):
        if family == "lstm_bayes":
            eps.append([draw(sd["rnn." + k], tid, smp) for k, tid in zip(LSTM_EPS_ORDER, tensor_ids)])
        elif family == "tlm_ffn":
            eps.append(draw(sd["transformerlayers.0.linear2.weight_lgstd"], tensor_ids[0], smp))
        elif family == "tlm_gauss":
            pre = "transformerlayers.0.gpnn."
            if pre + "frequency_mean" in sd:
                eps.append(draw(sd[pre + "frequency_lgstd"], tensor_ids[0], smp * 1024))
            else:
                eps.append(gp_eps(pre, tensor_ids[0], smp))
        elif family == "lstm_gauss":
            eps.append({c: gp_eps("rnn.rnn.%d.gpnn." % c, tid, smp) for c, tid in tensor_ids.items()})
        elif family == "lstm_var":
            eps.append(None)  # drawn per hypothesis length below
        else:
            raise ValueError(family)
    key0 = {"lstm_bayes": "rnn.weight_hh_mean_1", "lstm_gauss": "encoder.weight", "lstm_var": "encoder.weight"}

    def forward(xs, hid, smp):
        """logits of one hypothesis under sample ``smp`` (None: mean weights) -> (logits, hidden)"""
        e = None if smp is None else eps[smp]
        if family == "lstm_bayes":
            return bayes_rnn_lm(xs, hid, sd, pos, e)
        if family == "lstm_gauss":
            return gauss_rnn_lm(xs, hid, sd, pos, e)
        if family == "lstm_var":
            rows = None
            if smp is not None:
                T = xs.shape[0]
                rows = {c: torch.from_numpy(P.normal(T * H, seed, P.STREAM_WEIGHT + tid, smp)).view(T, H) * 0.1
                        for c, tid in tensor_ids.items()}
            logits, hidden, _ = variational_rnn_lm(xs, hid, sd, pos, rows)
            return logits, hidden
        return transformer_lm(xs, sd, nhead, e), None

    out = []
    H = sd[key0[family]].shape[1] if is_rnn else 0
    hid = (torch.zeros(2, 1, H), torch.zeros(2, 1, H)) if is_rnn else None
    for key, hyps in nbest.items():
        first = None
        for n, hyp in enumerate(hyps, 1):
            x, t = get_input_and_target(hyp, vocab)
            xs, ts = torch.tensor(x).view(-1, 1), torch.tensor(t)
            nlls = [float(sentence_score(forward(xs, hid, smp)[0], ts)) for smp in range(S)]
            if is_rnn and first is None:
                first = forward(xs, hid, None)[1]
            out.append(("%s-%d" % (key, n), mc_sentence_score(nlls)))
        if is_rnn:
            hid = first
    return out
