#!/usr/bin/env python3
"""N-best sentence scoring with a trained LM -- same command line and file contract as the
reference's steps/pytorchnn/compute_sentence_scores_bayes_jianwei.py (stage 6 of
lmrescore_nbest_pytorchnn_cuda.sh:199-218): n-best file ``uttid-n w1 w2 ...`` in, ``uttid-n %.4f``
out, score = len * CE_mean = sum of token NLL under mean weights (model.eval(), :225), optional logit
interpolation with a second model (:157-168), LSTM hidden state carried to the next utterance from
the FIRST hypothesis of the previous one (:271-274).
"""
import argparse
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from ._lib import BayesLMError


def load_nbest(path):
    """uttid-n hyp...  -> {uttid: [hyp, ...]} in file order (reference :20-51; an empty
    hypothesis becomes ' ')."""
    nbest = OrderedDict()
    with open(path, 'r', encoding='utf-8') as f:
        for line in f:
            line = line.strip()
            parts = line.split(' ', 1)
            key, hyp = (parts[0], parts[1]) if len(parts) == 2 else (line, ' ')
            nbest.setdefault(key.rsplit('-', 1)[0], []).append(hyp)
    return nbest


def read_vocab(path):
    word2idx = {}
    with open(path, 'r', encoding='utf-8') as f:
        for line in f:
            fields = line.split()
            assert len(fields) == 2
            if fields[0] not in word2idx:
                word2idx[fields[0]] = len(word2idx)
    return word2idx


def get_input_and_target(hyp, vocab):
    """'<s> ' + hyp -> input ids; hyp + ' <s>' -> target ids; OOV -> <unk> (reference :87-120)."""
    unk = vocab.get('<unk>')

    def ids(text):
        out = []
        for w in text.split():
            i = vocab.get(w, unk)
            if i is None:
                raise KeyError('<unk>')
            out.append(i)
        return out
    return ids('<s> ' + hyp), ids(hyp + ' <s>')


def build_models(args, ntokens):
    """Model dispatch of the reference (:373-449): dropout 0.5 (irrelevant in eval), tied."""
    from . import model as M
    m2 = None
    if args.model == 'Transformer':
        if args.uncertainty == 'none':
            m1 = M.TransformerModel(ntokens, args.emsize, args.nhead, args.nhid, args.nlayers, 0.5, "gelu", True)
        elif args.uncertainty == 'Bayesian':
            m1 = M.BayesTransformerModel(ntokens, args.emsize, args.nhead, args.nhid, args.nlayers, 0.5, True, args.T_bayes_pos)
        elif args.uncertainty == 'Gaussian':
            m1 = M.GaussTransformerModel(ntokens, args.emsize, args.nhead, args.nhid, args.nlayers, 0.5, True, args.T_gauss_pos)
        elif args.uncertainty == 'Variational':
            m1 = M.VTransformerModel(ntokens, args.emsize, args.nhead, args.nhid, args.nlayers, 0.5, True, args.T_v_pos)
        else:
            raise SystemExit("unknown --uncertainty %s" % args.uncertainty)
        if args.interpolation_flag == 1 and args.uncertainty != 'none':
            m2 = M.BayesTransformerModel(ntokens, args.emsize, args.nhead, args.nhid, args.nlayers, 0.5, True, 'none')
    else:
        if args.uncertainty == 'none':
            m1 = M.RNNModel(args.model, ntokens, args.emsize, args.nhid, args.nlayers, 0.5, True)
        elif args.uncertainty == 'Bayesian':
            m1 = M.BayesRNNModel(args.model, ntokens, args.emsize, args.nhid, args.nlayers, 0.5, True, args.L_bayes_pos)
        elif args.uncertainty == 'Gaussian':  # tie_weights False here, as in the reference (:428-429)
            m1 = M.GaussRNNModel(args.model, ntokens, args.emsize, args.nhid, args.nlayers, 0.5, False, args.L_gauss_pos)
        elif args.uncertainty == 'Variational':
            m1 = M.VariationalRNNModel(args.model, ntokens, args.emsize, args.nhid, args.nlayers, 0.5, True, args.L_v_pos)
        else:
            raise SystemExit("unknown --uncertainty %s" % args.uncertainty)
        if args.interpolation_flag == 1 and args.uncertainty != 'none':
            m2 = M.BayesRNNModel(args.model, ntokens, args.emsize, args.nhid, args.nlayers, 0.5, False, 0)
    return m1, m2


def load_partial(model, path):
    """Keys filtered by name: extra keys ignored, missing keys keep their init (reference :457-462)."""
    sd = torch.load(path, map_location='cpu')
    own = model.state_dict()
    own.update({k: v for k, v in sd.items() if k in own and tuple(v.shape) == tuple(own[k].shape)})
    model.load_state_dict(own)


def sentence_score(model, ids, tgt, model_type, hidden, device, model_2=None, hidden_2=None, alpha=0.0):
    from . import ops
    data = torch.tensor(ids, dtype=torch.int64, device=device).view(-1, 1)
    target = torch.tensor(tgt, dtype=torch.int64, device=device)
    with torch.no_grad():
        if model_type == 'Transformer':
            out = model(data)
        else:
            out, hidden = model(data, hidden)
        if model_2 is not None:
            if model_type == 'Transformer':
                out2 = model_2(data)
            else:
                out2, hidden_2 = model_2(data, hidden_2)
            # interpolates LOGITS (reference :163); the mixture is formed inside the CE kernel
            loss, _ = ops.cross_entropy_interp(out.view(-1, out.shape[-1]), out2.view(-1, out2.shape[-1]), alpha, target)
        else:
            loss, _ = ops.cross_entropy(out.view(-1, out.shape[-1]), target)
    return len(ids) * float(loss), hidden, hidden_2


def compute_scores(nbest, model, vocab, model_type, device, model_2=None, alpha=0.0):
    model.eval()
    if model_2 is not None:
        model_2.eval()
    scores = OrderedDict()
    hidden = model.init_hidden(1) if model_type != 'Transformer' else None
    hidden_2 = model_2.init_hidden(1) if (model_2 is not None and model_type != 'Transformer') else None
    for key, hyps in nbest.items():
        first, first_2 = None, None
        for hyp in hyps:
            x, t = get_input_and_target(hyp, vocab)
            s, h_new, h2_new = sentence_score(model, x, t, model_type, hidden, device, model_2, hidden_2, alpha)
            if first is None:
                first, first_2 = h_new, h2_new
            scores.setdefault(key, []).append((hyp, s))
        if model_type != 'Transformer':
            hidden, hidden_2 = first, first_2
    return scores


_FUSED_NLL = os.environ.get("BLM_SCORER_FUSED_NLL", "1") != "0"  # 0: decoder logits materialised + CE kernel (A/B)
_PACKED = os.environ.get("BLM_SCORER_PACKED", "1") != "0"  # 0: padded (T, N) activations in the Transformer stacks (A/B)


_INTERP = {}  # (id(model), id(model_2), alpha) -> ops.InterpDecoder of the scoring run in progress


def _interp_decoder(model, model_2, alpha):
    from . import ops
    key = (id(model), id(model_2), float(alpha))
    dec = _INTERP.get(key)
    d1, d2 = model.decoder, model_2.decoder
    if dec is None or dec.w1.data_ptr() != d1.weight.data_ptr() or dec.w2.data_ptr() != d2.weight.data_ptr():
        _INTERP.clear()  # one pair at a time: the packed [W1 | W2] is V x (K1 + K2) floats
        dec = _INTERP[key] = ops.InterpDecoder(d1.weight, d1.bias, d2.weight, d2.bias, alpha)
    return dec


def _batch_nll(model, data, target_flat, model_type, hidden, model_2, hidden_2, alpha, rows=None):
    """Per-token NLL of a padded batch of hypotheses (all columns start from the same state): (T, N), or -- with
    ``rows`` (flat indices t*N + n of the real tokens) -- one value per selected row, the decoder being applied to those
    rows only (model._ProjHolder.rows)."""
    from . import ops
    import contextlib
    # Transformers whose operations outside the attention core are all token-wise keep ONLY the real tokens' rows through
    # the whole stack (ops.packed_tokens): the rows arrive at the decoder already selected, in the order of ``rows``
    packed = (rows is not None and model_type == 'Transformer' and _PACKED and getattr(model, "supports_packed", False)
              and (model_2 is None or getattr(model_2, "supports_packed", False)))
    for m in (model, model_2):
        if m is not None:
            m.decoder.rows = None if packed else rows
    # one model: the decoder returns the NLL itself, its (rows, V) logits are never stored (ops.linear_nll); two models
    # interpolate LOGITS (reference :163) and keep the materialised pair + the two-input CE kernel
    fused = (model_2 is None and _FUSED_NLL and not torch.is_grad_enabled() and hasattr(model.decoder, "nll_targets")
             and ops.linear_nll_supported(model.decoder.weight, model.decoder.bias))
    # two models: the reference interpolates LOGITS (:163); alpha (x1 W1^T + b1) + (1 - alpha) (x2 W2^T + b2) is one product
    # over packed operands, so one decoder + cross-entropy launch takes both and no logits are stored (ops.linear_nll_interp)
    fused2 = (model_2 is not None and _FUSED_NLL and not torch.is_grad_enabled()
              and all(hasattr(m.decoder, "return_input") for m in (model, model_2))
              and ops.linear_nll_interp_supported(model.decoder.weight, model.decoder.bias, model_2.decoder.weight,
                                                  model_2.decoder.bias))
    if fused:
        model.decoder.nll_targets = target_flat
    if fused2:
        model.decoder.return_input = model_2.decoder.return_input = True
    try:
        with (ops.packed_tokens(rows, data.shape[0], data.shape[1]) if packed else contextlib.nullcontext()):
            if model_type == 'Transformer':
                out = model(data)
            else:
                out, _ = model(data, hidden)
            out2 = None
            if model_2 is not None:
                out2 = model_2(data) if model_type == 'Transformer' else model_2(data, hidden_2)[0]
        if fused:
            nll = out
        elif fused2:
            nll = ops.linear_nll_interp(out, out2, _interp_decoder(model, model_2, alpha), target_flat)
        elif model_2 is not None:
            _, nll = ops.cross_entropy_interp(out.view(-1, out.shape[-1]), out2.view(-1, out2.shape[-1]), alpha, target_flat)
        else:
            _, nll = ops.cross_entropy(out.view(-1, out.shape[-1]), target_flat)
    finally:
        for m in (model, model_2):
            if m is not None:
                m.decoder.rows = None
        if fused:
            model.decoder.nll_targets = None
        if fused2:
            model.decoder.return_input = model_2.decoder.return_input = False
    return nll if rows is not None else nll.view(data.shape[0], data.shape[1])


def _carry(model, x0, hidden):
    """State after running x0 (T,1) from ``hidden`` in eval mode: the recurrent stack only -- every LSTM family here is
    embedding -> self.rnn -> dropout -> decoder (model.py:217-229 and its siblings), and the decoder's (T,V) logits
    are not needed for the state."""
    from . import ops
    if hasattr(model, "rnn") and hasattr(model, "encoder") and not model.training:
        emb = ops.embed(x0, model.encoder.weight, None, 1.0, ops.NO_DROP)
        return model.rnn(emb, hidden)[1]
    return model(x0, hidden)[1]


def _carry_chain(model, stream_ids, offs, hidden, max_tokens=8192):
    """-> [state entering utterance u] for the concatenated first hypotheses ``stream_ids`` (utterance u = tokens
    offs[u]:offs[u+1]).  The chain is one long B = 1 sequence: it is run in segments of whole utterances (<= max_tokens
    tokens) through the recurrent stack, and the states at the utterance boundaries are tapped out of the fused LSTM
    layers (ops.state_tap) -- one call per segment instead of one per utterance.  Cells that run step-wise do not
    report their states: those models walk the chain utterance by utterance."""
    from . import ops
    n = len(offs) - 1
    carries = []
    u = 0
    nlayers = hidden[0].shape[0]
    tappable = hasattr(model, "rnn") and hasattr(model, "encoder") and not model.training
    while u < n:
        v = u + 1
        while v < n and offs[v + 1] - offs[u] <= max_tokens:
            v += 1
        if tappable:
            seg = stream_ids[offs[u]:offs[v]].view(-1, 1)
            bounds = torch.as_tensor([offs[k + 1] - offs[u] - 1 for k in range(u, v)], dtype=torch.int64, device=seg.device)
            with ops.state_tap(bounds) as tap:
                emb = ops.embed(seg, model.encoder.weight, None, 1.0, ops.NO_DROP)
                last = model.rnn(emb, hidden)[1]
            if len(tap.layers) == nlayers:
                hb = torch.stack([h for h, _ in tap.layers], 1)  # (n_utt, L, 1, H): an utterance's (L, 1, H) state is a contiguous
                cb = torch.stack([c for _, c in tap.layers], 1)  # view of it, not a copy per utterance (two launches each before)
                carries.append(hidden)
                for k in range(v - u - 1):
                    carries.append((hb[k], cb[k]))
                hidden = last
                u = v
                continue
            tappable = False  # a step-wise cell: no taps -- redo this segment (and the rest) utterance by utterance
        for k in range(u, v):
            carries.append(hidden)
            hidden = _carry(model, stream_ids[offs[k]:offs[k + 1]].view(-1, 1), hidden)
        u = v
    return carries


def compute_scores_batched(nbest, model, vocab, model_type, device, model_2=None, alpha=0.0, mc_samples=0, seed=1111,
                           batch_tokens=None):
    """the N hypotheses of an utterance are padded into ONE (T_max, N) batch instead
    of N separate launches.  Exact for causal Transformers (padding sits after every real token) and
    for LSTMs (all hypotheses of an utterance start from the same carried state; the carry is the state after
    a B = 1 pass over the previous utterance's first hypothesis exactly as the reference does, :271-274).
    Consecutive utterances are packed into one batch up to ``batch_tokens`` padded tokens (20-best lists alone are
    launch bound): Transformers carry no state; for LSTMs the carries depend on first hypotheses only, so they
    are computed first in one sequential pass and every column of a packed batch starts from its own carry.

    mc_samples = S > 0 (new, default off; not in the reference, which scores with mean weights, :225):
    S passes with the variational weights sampled (dropout off) and the sentence PROBABILITIES
    averaged, score = -log(mean_s exp(-NLL_s)).  Sample s is ONE model for the whole n-best list: its
    eps comes from Philox step s of every variational tensor's stream (seed, tensor id, step = s), so the scores do not
    depend on how utterances are packed into batches; an LSTM's carried state stays the mean-weight one."""
    # The loop builds hundreds of thousands of short-lived, acyclic Python objects (token lists, views); the cyclic collector's
    # full passes over them stall the host while the GPU waits -- 20000 hypotheses: 177 ms with the collector off against
    # 190-225 ms with it on (the first call of a process, with a small heap, hides it).  Off for the duration of the call.
    import gc
    gc_was = gc.isenabled()
    gc.disable()
    try:
        return _compute_scores_batched(nbest, model, vocab, model_type, device, model_2, alpha, mc_samples, seed, batch_tokens)
    finally:
        if gc_was:
            gc.enable()


def _compute_scores_batched(nbest, model, vocab, model_type, device, model_2, alpha, mc_samples, seed, batch_tokens):
    model.eval()
    if model_2 is not None:
        model_2.eval()
    scores = OrderedDict()
    is_rnn = model_type != 'Transformer'
    if not batch_tokens:  # measured (tools/bench_scorer.py): Transformers 16384 (+4 %, +11 % with Monte-Carlo samples), LSTMs 8192
        batch_tokens = 8192 if is_rnn else 16384
    hidden = model.init_hidden(1) if is_rnn else None
    hidden_2 = model_2.init_hidden(1) if (model_2 is not None and is_rnn) else None
    S = max(1, int(mc_samples))
    raised = []
    if mc_samples > 0:
        from .model import variational_sites
        sites = variational_sites(model)
        if not sites:
            # e.g. --uncertainty none, Gaussian type 0, Variational '00', VTransformer (whose noise branch cannot run,
            # model.py:2800): S passes would be S identical mean-weight passes at S times the cost
            raise BayesLMError("--mc-samples %d: %s has no variational tensor to sample (mean-weight scoring is "
                               "--mc-samples 0)" % (mc_samples, type(model).__name__))
        for m in sites:  # optional sampling flags (GPNN.sample, model.py:1799: False unless somebody raises it)
            if getattr(m, "sample", True) is False:
                m.sample = True
                raised.append(m)
        model.train()
        model.noise_state.dropout_off = True
        model.set_seed(seed)

    def score_group(group, hidden, hidden_2):
        """group = [(key, hyps, pairs)]; one padded batch over all their hypotheses."""
        pairs = [p for _, _, ps in group for p in ps]
        lens = [len(x) for x, _ in pairs]
        if min(lens) < 1:  # the running-sum difference below indexes ends - 1: an empty row would read the batch total
            raise ValueError("score_group: a hypothesis without tokens (every n-best entry carries at least '<s>')")
        Tm, N = max(lens), len(pairs)
        # one host buffer [data | real-token rows | their targets | hypothesis ends] -> ONE host-to-device copy per batch.
        # Only the REAL tokens reach the decoder GEMM and the cross entropy (the bulk of the work: 2*d*V flops per token):
        # the decoder gathers rows t*N + n, n-major, so a hypothesis' tokens are contiguous and its score is a difference
        # of a running sum (padding was 40-50 % of a batch of AMI-shaped hypotheses).
        R = int(sum(lens))
        host = np.zeros(Tm * N + 2 * R + N, dtype=np.int64)
        dat = host[:Tm * N].reshape(Tm, N)
        sel, tsel, ends = host[Tm * N:Tm * N + R], host[Tm * N + R:Tm * N + 2 * R], host[Tm * N + 2 * R:]
        o = 0
        for n, (x, t) in enumerate(pairs):
            ln = lens[n]
            dat[:ln, n] = x
            sel[o:o + ln] = np.arange(ln, dtype=np.int64) * N + n
            tsel[o:o + ln] = t
            o += ln
            ends[n] = o
        dev_buf = torch.from_numpy(host).to(device, non_blocking=True)
        data = dev_buf[:Tm * N].view(Tm, N)
        d_sel, d_tsel, d_ends = dev_buf[Tm * N:Tm * N + R], dev_buf[Tm * N + R:Tm * N + 2 * R], dev_buf[Tm * N + 2 * R:]
        hN = h2N = None
        if is_rnn:  # every column starts from the state carried into ITS utterance
            counts = [len(ps) for _, _, ps in group]
            hN = tuple(torch.cat([hidden[u][i].expand(-1, c, -1) for u, c in enumerate(counts)], 1).contiguous() for i in (0, 1))
            if hidden_2 is not None:
                h2N = tuple(torch.cat([hidden_2[u][i].expand(-1, c, -1) for u, c in enumerate(counts)], 1).contiguous()
                            for i in (0, 1))
        sent = []
        for smp in range(S):
            if mc_samples > 0:
                model.set_step(smp)
            nll = _batch_nll(model, data, d_tsel, model_type, hN, model_2, h2N, alpha, rows=d_sel)  # (R,)
            run = torch.cumsum(nll.double(), 0)
            hi = run[d_ends - 1]
            sent.append((hi - torch.cat([hi.new_zeros(1), hi[:-1]])).float())
        if S == 1:
            tot = sent[0]
        else:
            tot = -(torch.logsumexp(-torch.stack(sent), 0) - torch.log(torch.tensor(float(S), device=device)))
        # the scores stay on the device: no host sync per batch (the LSTM path launches one small batch per
        # utterance); they are read back together in flush()
        pending.append((group, tot))
        if len(pending) >= 1024:
            flush()

    pending = []

    def flush():
        if not pending:
            return
        flat = torch.cat([t for _, t in pending]).tolist()
        o = 0
        for group, _ in pending:
            for key, hyps, ps in group:
                scores[key] = [(h, float(v)) for h, v in zip(hyps, flat[o:o + len(hyps)])]
                o += len(hyps)
        pending.clear()

    try:
        _score_all(nbest, model, model_2, vocab, device, is_rnn, hidden, hidden_2, batch_tokens, score_group, flush)
    finally:
        _INTERP.clear()  # the packed [W1 | W2] belongs to this scoring run
        if mc_samples > 0:
            model.noise_state.dropout_off = False
            model.eval()
            for m in raised:
                m.sample = False
    return scores


def _score_all(nbest, model, model_2, vocab, device, is_rnn, hidden, hidden_2, batch_tokens, score_group, flush):
    """The batching loop of compute_scores_batched: the carry chain of an LSTM first, then cross-utterance batches."""
    with torch.no_grad():
        items = list(nbest.items())
        carries, carries_2 = None, None
        if is_rnn and items:
            # Pass 1 -- the carry chain.  The state entering utterance u is the state after the FIRST hypothesis of
            # utterance u-1 alone, mean weights (reference :271-274): it depends on first hypotheses only, so the chain
            # is walked once (B = 1, one host-to-device copy for all of them) and every utterance's hypotheses are
            # scored afterwards in cross-utterance batches, each column starting from its utterance's carry.
            was_training = model.training
            model.eval()
            firsts = [get_input_and_target(hyps[0], vocab)[0] for _, hyps in items]
            offs = np.cumsum([0] + [len(f) for f in firsts])
            stream_ids = torch.from_numpy(np.concatenate([np.asarray(f, dtype=np.int64) for f in firsts])).to(device)
            carries = _carry_chain(model, stream_ids, offs, hidden)
            carries_2 = _carry_chain(model_2, stream_ids, offs, hidden_2) if model_2 is not None else None
            model.train(was_training)
        group, g_h, g_h2, g_cols, g_tmax = [], [], [], 0, 0
        for u, (key, hyps) in enumerate(items):  # tokenised lazily: the host work overlaps the batches already in flight
            pairs = [get_input_and_target(h, vocab) for h in hyps]
            tmax = max(len(x) for x, _ in pairs)
            if group and max(g_tmax, tmax) * (g_cols + len(pairs)) > batch_tokens:
                score_group(group, g_h if is_rnn else None, g_h2 if (is_rnn and model_2 is not None) else None)
                group, g_h, g_h2, g_cols, g_tmax = [], [], [], 0, 0
            group.append((key, hyps, pairs))
            if is_rnn:
                g_h.append(carries[u])
                if model_2 is not None:
                    g_h2.append(carries_2[u])
            g_cols, g_tmax = g_cols + len(pairs), max(g_tmax, tmax)
        if group:
            score_group(group, g_h if is_rnn else None, g_h2 if (is_rnn and model_2 is not None) else None)
        flush()


def write_scores(scores, path):
    with open(path, 'w', encoding='utf-8') as f:
        for key, lst in scores.items():
            for idx, (_, s) in enumerate(lst, 1):
                f.write('%s %.4f\n' % ('-'.join([key, str(idx)]), s))


def interpolate_scores(nolm_path, lmonly_path, nn_path, nnweight, out_path):
    """Stage 7 of lmrescore_nbest_pytorchnn_cuda.sh (:221-229) without the paste | awk hop: per n-best entry
    score = graph + nnweight * nn + (1 - nnweight) * lm, from the three 'key value' files the script pastes
    (lmwt.nolm, lmwt.lmonly, lmwt.nn).  Keys are taken from the first file; lines are matched by position, as paste
    does; numbers are printed the way awk prints them (OFMT %.6g).  Host text processing, offered so that the scorer
    process can leave lmwt.interp.<w> behind itself."""
    def col(path):
        with open(path, 'r', encoding='utf-8') as f:
            return [ln.split() for ln in f if ln.strip()]
    a, b, c = col(nolm_path), col(lmonly_path), col(nn_path)
    if not (len(a) == len(b) == len(c)):
        raise SystemExit("interpolate_scores: %d / %d / %d lines in %s, %s, %s" % (len(a), len(b), len(c), nolm_path,
                                                                                   lmonly_path, nn_path))
    w = float(nnweight)
    rows = []
    for i, (x, y, z) in enumerate(zip(a, b, c)):
        if not (x[0] == y[0] == z[0]):  # paste would silently mix the scores of different n-best entries
            raise SystemExit("interpolate_scores: line %d holds different keys: %s (%s), %s (%s), %s (%s)"
                             % (i + 1, x[0], nolm_path, y[0], lmonly_path, z[0], nn_path))
        rows.append((x[0], float(x[1]) + w * float(z[1]) + (1.0 - w) * float(y[1])))
    with open(out_path, 'w', encoding='utf-8') as f:  # written only once every line has been checked
        for key, score in rows:
            f.write("%s %s\n" % (key, _awk_num(score)))


def _awk_num(v):
    """awk's print conversion: integers print as integers, everything else with OFMT = %.6g; nan / inf as awk
    spells them."""
    if not math.isfinite(v):
        return "nan" if v != v else ("inf" if v > 0 else "-inf")
    return "%d" % v if v == int(v) and abs(v) < 1e16 else "%.6g" % v


def build_parser():
    p = argparse.ArgumentParser(description="Compute sentence scores of nbest lists with a trained neural LM (MI355X engine).")
    p.add_argument('--nbest-list', type=str, required=True)
    p.add_argument('--outfile', type=str, required=True)
    p.add_argument('--vocabulary', type=str, required=True)
    p.add_argument('--model-path', type=str, required=True)
    p.add_argument('--model', type=str, default='LSTM')
    p.add_argument('--emsize', type=int, default=1024)
    p.add_argument('--nhid', type=int, default=1024)
    p.add_argument('--nlayers', type=int, default=2)
    p.add_argument('--nhead', type=int, default=8)
    p.add_argument('--uncertainty', type=str, default='none')
    p.add_argument('--T_bayes_pos', type=str, default='none')
    p.add_argument('--L_bayes_pos', type=int, default=0)
    p.add_argument('--L_gauss_pos', type=str, default='00')
    p.add_argument('--T_gauss_pos', type=int, default=3)
    p.add_argument('--L_v_pos', type=str, default='11')
    p.add_argument('--T_v_pos', type=int, default=0)
    p.add_argument('--interpolation_flag', type=int, default=0)
    p.add_argument('--inter_path', type=str, default='')
    p.add_argument('--inter_alpha', type=float, default=0.8)
    # new, optional
    p.add_argument('--batched', type=int, default=1, help='1: all hypotheses of an utterance in one padded batch; '
                   '0: one launch per hypothesis like the reference')
    p.add_argument('--mc-samples', type=int, default=0, help='S > 0: average sentence probabilities over S weight samples')
    p.add_argument('--batch-tokens', type=int, default=0, help='padded tokens per batch across utterances (0: 16384 for Transformers, 8192 for LSTMs)')
    p.add_argument('--gemm-mode', type=str, default='f32', choices=['f32', 'bf16x6', 'bf16x3'],
                   help='opt-in split-bf16 arithmetic of the GEMM family (DESIGN.md section 7); default fp32 MFMA')
    p.add_argument('--interp-nolm', type=str, default='', help='lmwt.nolm of the rescoring script (graph scores): with '
                   '--interp-lmonly, --interp-nnweight and --interp-out the scorer also writes the stage-7 file')
    p.add_argument('--interp-lmonly', type=str, default='')
    p.add_argument('--interp-nnweight', type=float, default=0.8)
    p.add_argument('--interp-out', type=str, default='')
    p.add_argument('--job', type=int, default=0, help='the rescoring script\'s JOB index (1-based, `$cmd JOB=1:$nj`): job j '
                   'scores on GPU (j - 1) mod the number of visible GPUs, so the nj independent jobs of stage 6 spread over '
                   'a node (replicas, no exchange); 0: the GPU of LOCAL_RANK (default 0)')
    return p


def job_device_index(job, n_devices, local_rank=0):
    """GPU of one stage-6 job (lmrescore_nbest_pytorchnn_cuda.sh:199-203 starts nj of them, each over its own
    archives.JOB): the jobs share nothing, so they are laid round-robin over the visible GPUs."""
    if job < 0:
        raise SystemExit("--job is the rescoring script's 1-based JOB index (0: use LOCAL_RANK)")
    if n_devices < 1:
        raise SystemExit("bayeslms_amd scoring needs an MI355X: there is no CPU path")
    idx = (job - 1) % n_devices if job > 0 else int(local_rank)
    if not 0 <= idx < n_devices:
        raise SystemExit("LOCAL_RANK %d but %d GPU(s) visible" % (idx, n_devices))
    return idx


def main(argv=None):
    args = build_parser().parse_args(argv)
    for pth, what in ((args.nbest_list, "Nbest list"), (args.vocabulary, "Vocabulary"), (args.model_path, "Model")):
        assert os.path.exists(pth), "%s path does not exists." % what
    if not torch.cuda.is_available():
        raise SystemExit("bayeslms_amd scoring needs an MI355X: there is no CPU path")
    device = torch.device("cuda", job_device_index(args.job, torch.cuda.device_count(), os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)  # one process per GPU: kernels go to the current device's stream
    if args.gemm_mode != 'f32':
        from . import ops
        ops.set_gemm_mode(args.gemm_mode)
    vocab = read_vocab(args.vocabulary)
    model_1, model_2 = build_models(args, len(vocab))
    load_partial(model_1, args.model_path)
    model_1 = model_1.to(device)
    if model_2 is not None:
        assert os.path.exists(args.inter_path), "Interpolation model path does not exists."
        load_partial(model_2, args.inter_path)
        model_2 = model_2.to(device)
    nbest = load_nbest(args.nbest_list)
    if args.batched or args.mc_samples > 0:
        scores = compute_scores_batched(nbest, model_1, vocab, args.model, device, model_2, args.inter_alpha, args.mc_samples,
                                        batch_tokens=args.batch_tokens)
    else:
        scores = compute_scores(nbest, model_1, vocab, args.model, device, model_2, args.inter_alpha)
    write_scores(scores, args.outfile)
    print("Write to %s" % args.outfile)
    if args.interp_out:
        interpolate_scores(args.interp_nolm, args.interp_lmonly, args.outfile, args.interp_nnweight, args.interp_out)
        print("Write to %s" % args.interp_out)


if __name__ == '__main__':
    main()
