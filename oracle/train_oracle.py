"""TEST INFRASTRUCTURE -- CPU restatement of the reference's training / evaluation loop.

Restates steps/pytorchnn/train.py of AmourWaltz/BayesLMs over the functional models of bayes_oracle.py:
batchify / get_batch (:167-185, 299-303), the step (CE + KL / len(train_data) * seq_len, backward,
clip_grad_norm_, SGD momentum 0.9; :306-420), the per-interval log means (:422-437), evaluate (:441-458) and the
epoch loop with best-checkpoint / LR halving / fresh optimizer / reload (:464-519).  Pinned by
tests/golden/train_traj_*.npz: RNG-free runs of the reference's own train.py (make_golden.py f6_train_trajectory).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
import torch

from . import bayes_oracle as O


def batchify(ids, bsz):
    """train.py:167-179: (rows, bsz), column c = the c-th contiguous chunk."""
    nb = ids.size(0) // bsz
    return ids.narrow(0, 0, nb * bsz).view(bsz, -1).t().contiguous()


def get_batch(source, i, seq_len):
    n = min(seq_len, len(source) - 1 - i)  # train.py:299-303
    return source[i:i + n], source[i + 1:i + 1 + n].reshape(-1)


def kl_gpnn(sd, pre, t):
    """GPNN.kl_divergence (model.py:1816-1826) for gpnn_type t."""
    kl = torch.zeros(())
    if t in (1, 3):
        kl = kl + O.kl_mean_form_minus1(sd[pre + "coef_mean"], sd[pre + "coef_lgstd"])
    if t in (2, 3):
        kl = kl + O.kl_mean_form_minus1(sd[pre + "weights_mean"], sd[pre + "weights_lgstd"])
        kl = kl + O.kl_mean_form_minus1(sd[pre + "bias_mean"], sd[pre + "bias_lgstd"])
    return kl


def family(args):
    """argv-style dict -> (forward(sd, x, hidden) -> (logits, hidden), kl(sd) or None, is_rnn).  Mirrors the model
    dispatch of train.py:193-223 and the KL selection of :335-399 for the RNG-free configurations."""
    model, unc = args["model"], args["uncertainty"]
    nhead = int(args.get("nhead", 2))
    if model == "Transformer":
        def fwd(sd, x, hidden):
            return O.transformer_lm(x, sd, nhead), None
        kl = None
        if unc == "Gaussian" and 1 <= int(args["T_gauss_pos"]) <= 3:
            t = int(args["T_gauss_pos"])
            kl = lambda sd: kl_gpnn(sd, "transformerlayers.0.gpnn.", t)  # noqa: E731
        elif unc == "Bayesian":
            pos = args["T_bayes_pos"]
            kl = lambda sd: O.kl_transformer(sd, pos)  # noqa: E731
        return fwd, kl, False
    if unc == "none":
        return (lambda sd, x, h: O.rnn_lm(x, h, sd)), None, True
    if unc == "Bayesian":
        pos = int(args["L_bayes_pos"])
        kl = (lambda sd: O.kl_bayes2lstm(sd, "rnn.", pos)) if 1 <= pos <= 5 else None
        return (lambda sd, x, h: O.bayes_rnn_lm(x, h, sd, pos, None)), kl, True
    if unc == "Gaussian":
        g = args["L_gauss_pos"]
        kl = (lambda sd: O.kl_gauss_rnn(sd, g)) if (int(g[0]) > 0 and 0 < int(g[1]) <= 3) else None
        return (lambda sd, x, h: O.gauss_rnn_lm(x, h, sd, g)), kl, True
    if unc == "Variational":
        v = args["L_v_pos"]
        return (lambda sd, x, h: O.variational_rnn_lm(x, h, sd, v)[:2]), None, True  # '00': no noise, no KL
    raise ValueError((model, unc))


def evaluate(sd, fwd, is_rnn, source, seq_len, nlayers, nhid):
    """train.py:441-458."""
    total = 0.0
    bsz = source.shape[1]
    hidden = (torch.zeros(nlayers, bsz, nhid), torch.zeros(nlayers, bsz, nhid)) if is_rnn else None
    with torch.no_grad():
        for i in range(0, source.size(0) - 1, seq_len):
            data, tgt = get_batch(source, i, seq_len)
            logits, hidden = fwd(sd, data, hidden)
            total += len(data) * O.cross_entropy_mean(logits, tgt).item()
    return total / (len(source) - 1)


def train_run(sd, fwd, kl, is_rnn, train_data, val_data, test_data, *, seq_len, lr, clip, epochs, log_interval,
              nlayers=2, nhid=0):
    """The whole of train.py:464-546 on a flat state dict (``sd`` is updated in place; tensors that share storage
    in the reference -- decoder.weight is encoder.weight -- must be the same object here).
    -> dict(step_loss, interval_loss, valid_loss, snapshots, halved_epochs, sgd_lr, test_loss)."""
    names, seen = [], set()
    for k, v in sd.items():
        if v.dtype.is_floating_point and k != "pos_encoder.pe" and id(v) not in seen:
            seen.add(id(v))
            v.requires_grad_(True)
            names.append(k)
    params = [sd[k] for k in names]
    bufs = {}
    out = {"step_loss": [], "interval_loss": [], "valid_loss": [], "snapshots": [], "halved_epochs": [], "sgd_lr": [lr]}
    best, best_sd, counter = None, None, 0
    bsz = train_data.shape[1]
    for epoch in range(1, epochs + 1):
        total = 0.0
        hidden = (torch.zeros(nlayers, bsz, nhid), torch.zeros(nlayers, bsz, nhid)) if is_rnn else None
        for batch, i in enumerate(range(0, train_data.size(0) - 1, seq_len)):
            data, tgt = get_batch(train_data, i, seq_len)
            for p in params:
                p.grad = None
            if hidden is not None:
                hidden = tuple(h.detach() for h in hidden)
            logits, hidden = fwd(sd, data, hidden)
            loss = O.cross_entropy_mean(logits, tgt)
            if kl is not None:
                loss = loss + kl(sd) / len(train_data) * seq_len  # train.py:338
            loss.backward()
            # clip_grad_norm_ and SGD both skip parameters without a gradient
            live = [k for k in names if sd[k].grad is not None]
            b = [bufs.get(k) for k in live]
            O.clip_and_sgd([sd[k] for k in live], [sd[k].grad for k in live], b, lr, clip)
            bufs.update(dict(zip(live, b)))
            out["step_loss"].append(loss.item())
            total += loss.item()
            if batch % log_interval == 0 and batch > 0:
                out["interval_loss"].append(total / log_interval)
                total = 0.0
        val = evaluate(sd, fwd, is_rnn, val_data, seq_len, nlayers, nhid)
        out["valid_loss"].append(val)
        out["snapshots"].append({k: sd[k].detach().clone() for k in names})
        if not best or val < best:  # train.py:498-501
            best_sd = {k: sd[k].detach().clone() for k in names}
            best = val
        else:  # :502-508: halve, fresh SGD (momentum buffers gone), reload the best checkpoint
            lr /= 2.0
            bufs = {}
            with torch.no_grad():
                for k in names:
                    sd[k].copy_(best_sd[k])
            counter += 1
            out["halved_epochs"].append(epoch)
            out["sgd_lr"].append(lr)
        if counter == 8:
            break
    with torch.no_grad():
        for k in names:
            sd[k].copy_(best_sd[k])
    out["test_loss"] = evaluate(sd, fwd, is_rnn, test_data, seq_len, nlayers, nhid)
    out["snapshots"].append({k: sd[k].detach().clone() for k in names})
    out["final_lr"] = lr
    return out
