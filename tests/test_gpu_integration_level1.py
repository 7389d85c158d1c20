"""GPU: INTEGRATION.md level 1 -- `from bayeslms_amd.model import *` UNDER THE REFERENCE'S OWN LOOP SHAPE.

Every other GPU test drives the engine's Trainer (flat buffers, fused cross entropy, fused clip + SGD).  A maintainer who only
swaps `import model` keeps the reference's train.py as it is: torch's nn.CrossEntropyLoss on the logits, optimizer.zero_grad()
(gradients set to None), loss.backward(), torch.nn.utils.clip_grad_norm_, torch.optim.SGD(momentum 0.9), evaluate() with the
same criterion, best-checkpoint / LR-halving / fresh-optimizer / reload through state_dict (train.py:306-438, 441-458,
464-519).  This file restates that loop around the shim and holds it to the trajectories recorded from the reference's own
train.py (tests/golden/train_traj_*.npz) with the bars of the CLI test: same LR-halving epochs, valid / test loss 1e-4,
interval loss and final checkpoint 1e-3."""
import io
import math

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.optim as optim

from test_train_traj_oracle import load_traj, write_corpus

pytestmark = pytest.mark.gpu


def reference_shaped_run(cfg, data_dir, init, device):
    """What a maintainer's unchanged training script does around the shim module -- the steps of the
    reference's train.py:167-185 (column layout), :193-223 (constructors), :239-258 (partial state-dict load), :299-519 (windows,
    loss, clipping, SGD with momentum, evaluation, best checkpoint / LR halving / fresh optimizer / reload), written here in this
    file's own words.  Everything that is not the model is torch's own API: nn.CrossEntropyLoss on the logits, zero_grad()
    (gradients become None), clip_grad_norm_, optim.SGD, state_dict round trips through torch.save / torch.load."""
    import types
    shim = types.ModuleType("model")
    exec("from bayeslms_amd.model import *", shim.__dict__)  # the two-line shim, verbatim
    from bayeslms_amd import data as corpus_io  # same file formats as the reference's data.py

    text = corpus_io.Corpus(data_dir)
    vocab = len(text.dictionary)

    def columns(ids, width):  # the stream cut into `width` contiguous columns
        rows = ids.size(0) // width
        return ids[: rows * width].view(width, rows).t().contiguous().to(device)

    EVAL_COLS = 20
    streams = {"train": columns(text.train, cfg.batch_size), "valid": columns(text.valid, EVAL_COLS), "test": columns(text.test, EVAL_COLS)}
    recurrent = cfg.model != 'Transformer'
    if getattr(cfg, "seed", None) is not None:
        # no saved state: the script seeds torch's generator (train.py:124) and builds -- for `--uncertainty none` TWICE, keeping
        # the second model (train.py:196-199, :211-214)
        torch.manual_seed(cfg.seed)
        if cfg.uncertainty == 'none':
            build_through_shim(shim, cfg, vocab)
    net = build_through_shim(shim, cfg, vocab).to(device)
    extra_term = penalty_of(cfg)  # None, or net -> the divergence term this flag combination adds to the criterion
    per_window = float(cfg.seq_len) / float(streams["train"].size(0))  # ... / len(train_data) * seq_len
    weights = net.state_dict()
    weights.update({name: t for name, t in init.items() if name in weights})  # --prior True: keys filtered by name
    net.load_state_dict(weights)
    xent = nn.CrossEntropyLoss()

    def detach(state):
        return state.detach() if torch.is_tensor(state) else tuple(detach(s) for s in state)

    def windows(stream):
        last = stream.size(0) - 1
        for start in range(0, last, cfg.seq_len):
            n = min(cfg.seq_len, last - start)
            yield stream[start:start + n], stream[start + 1:start + 1 + n].reshape(-1)

    hist = {"interval_loss": [], "valid_loss": [], "halved_epochs": []}

    def one_epoch(opt):
        net.train()
        state = net.init_hidden(cfg.batch_size) if recurrent else None
        running = 0.0
        for k, (x, y) in enumerate(windows(streams["train"])):
            opt.zero_grad()
            if recurrent:
                logits, state = net(x, detach(state))
            else:
                logits = net(x)
            nll = xent(logits.view(-1, vocab), y)
            if extra_term is not None:  # a torch scalar with a grad_fn: autograd carries it into the .grad tensors
                nll = nll + extra_term(net) * per_window
            nll.backward()
            torch.nn.utils.clip_grad_norm_(net.parameters(), cfg.clip)
            opt.step()
            running += nll.item()
            if k > 0 and k % cfg.log_interval == 0:
                hist["interval_loss"].append(running / cfg.log_interval)
                running = 0.0

    def held_out_loss(stream):
        net.eval()
        state = net.init_hidden(EVAL_COLS) if recurrent else None
        acc = 0.0
        with torch.no_grad():
            for x, y in windows(stream):
                if recurrent:
                    logits, state = net(x, state)
                    state = detach(state)
                else:
                    logits = net(x)
                acc += len(x) * xent(logits.view(-1, vocab), y).item()
        return acc / (stream.size(0) - 1)

    def fresh_sgd(rate):
        return optim.SGD(net.parameters(), lr=rate, momentum=0.9, weight_decay=0)

    rate, best, halvings = cfg.lr, None, 0
    opt = fresh_sgd(rate)
    kept = io.BytesIO()
    for epoch in range(1, cfg.epochs + 1):
        one_epoch(opt)
        v = held_out_loss(streams["valid"])
        hist["valid_loss"].append(v)
        if best is None or v < best:  # (the reference's `not best` treats 0.0 like None; a loss never is)
            best, kept = v, io.BytesIO()
            torch.save(net.state_dict(), kept)
        else:
            rate, halvings = rate / 2.0, halvings + 1
            opt = fresh_sgd(rate)  # momentum buffers gone
            kept.seek(0)
            net.load_state_dict(torch.load(kept, map_location="cpu"))
            hist["halved_epochs"].append(epoch)
        if halvings == 8:
            break
    kept.seek(0)
    net.load_state_dict(torch.load(kept, map_location="cpu"))
    hist["test_loss"] = held_out_loss(streams["test"])
    hist["final"] = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    assert math.isfinite(hist["test_loss"])
    return hist


def build_through_shim(shim, cfg, vocab):
    """The constructor a maintainer's script reaches for each --model / --uncertainty pair (reference train.py:193-223), called
    positionally as the reference calls it."""
    kind = cfg.uncertainty
    if cfg.model == 'Transformer':
        head = (vocab, cfg.emsize, cfg.nhead, cfg.nhid, cfg.nlayers, cfg.dropout)
        if kind == 'none':
            return shim.TransformerModel(*head, "gelu", cfg.tied)
        table = {'Bayesian': (shim.BayesTransformerModel, cfg.T_bayes_pos), 'Gaussian': (shim.GaussTransformerModel, cfg.T_gauss_pos),
                 'Variational': (shim.VTransformerModel, cfg.T_v_pos)}
    else:
        head = (cfg.model, vocab, cfg.emsize, cfg.nhid, cfg.nlayers, cfg.dropout)
        if kind == 'none':
            return shim.RNNModel(*head, cfg.tied)
        table = {'Bayesian': (shim.BayesRNNModel, cfg.L_bayes_pos), 'Gaussian': (shim.GaussRNNModel, cfg.L_gauss_pos),
                 'Variational': (shim.VariationalRNNModel, cfg.L_v_pos)}
    ctor, where = table[kind]
    return ctor(*head, cfg.tied, where)


def penalty_of(cfg):
    """Which modules' kl_divergence() the reference's training loop adds for a flag combination (train.py:334-399), restated as a
    table: (model, uncertainty) -> the list of module paths, summed.  An empty list means the loss is the criterion alone."""
    paths = []
    if cfg.uncertainty == 'Bayesian' and cfg.model == 'LSTM':
        if cfg.L_bayes_pos in (1, 2, 3, 4, 5):
            paths = ["rnn"]
    elif cfg.uncertainty == 'Bayesian':
        paths = {'FFN': ["transformerlayers.0.linear2"], 'MHA': ["transformerlayers.0.self_attn.o_net"]}.get(cfg.T_bayes_pos, [])
        if cfg.T_bayes_pos == 'EMB':
            return lambda net: net.embed_kl_divergence()
    elif cfg.uncertainty == 'Gaussian' and cfg.model == 'Transformer':
        if cfg.T_gauss_pos in (1, 2, 3):
            paths = ["transformerlayers.0.gpnn"]
    elif cfg.uncertainty == 'Gaussian':
        digits = cfg.L_gauss_pos  # first digit: where in the cell, second: which tensors are random; 2 / 3 / 4 digits: which cells
        if int(digits[0]) > 0 and int(digits[1]) in (1, 2, 3):
            cells = {2: [0], 3: [1]}.get(len(digits), [0, 1])
            paths = ["rnn.rnn.%d.gpnn" % c for c in cells]
    elif cfg.uncertainty == 'Variational' and cfg.model == 'LSTM':
        paths = ["rnn.rnn.%d.vnn" % c for c in (0, 1) if cfg.L_v_pos[c] == '1']
    elif cfg.uncertainty == 'Variational':
        paths = ["transformerlayers.%d" % i for i in {1: [0], 2: [1], 3: [0, 1]}.get(int(cfg.T_v_pos), [])]
    if not paths:
        return None
    return lambda net: sum(net.get_submodule(p).kl_divergence() for p in paths)


TRAJECTORIES = ["tlm_none", "lstm_none", "lstm_bayes5", "tlm_gauss3", "lstm_gauss33", "lstm_var00"]
# started from `--seed 1111` alone; the `noisy` ones sample their weights every step: BLM_NOISE_SOURCE=torch in the environment of the
# unchanged script makes the shim draw each eps from torch's generator as the reference's own modules would have
FROM_SEED = ["seed_lstm_none", "seed_tlm_gauss3", "seed_noisy_lstm_bayes3", "seed_noisy_lstm_var11", "seed_noisy_tlm_bayes_emb",
             "seed_noisy_drop_lstm_none", "seed_noisy_drop_lstm_bayes3", "seed_noisy_drop_lstm_gauss33",  # --dropout 0.2 as well
             "seed_noisy_drop_tlm_none", "seed_noisy_drop_tlm_bayes_ffn", "seed_noisy_drop_tlm_gauss3"]


@pytest.mark.parametrize("tag", TRAJECTORIES + FROM_SEED)
def test_shim_under_the_reference_loop_reproduces_train_py(tag, tmp_path, monkeypatch):
    import argparse
    z, a, init, snaps = load_traj(tag)
    if "noisy" in tag:
        monkeypatch.setenv("BLM_NOISE_SOURCE", "torch")
    d = str(tmp_path)
    write_corpus(z, d)
    args = argparse.Namespace(model=a["model"], emsize=int(a["emsize"]), nhid=int(a["nhid"]), nlayers=int(a["nlayers"]),
                              nhead=int(a.get("nhead", 2)), dropout=float(a["dropout"]), tied=bool(a.get("tied", False)),
                              batch_size=int(a["batch_size"]), seq_len=int(a["seq_len"]), clip=float(a["clip"]), lr=float(a["lr"]),
                              epochs=int(a["epochs"]), log_interval=int(a["log_interval"]), uncertainty=a["uncertainty"],
                              T_bayes_pos=a.get("T_bayes_pos", "none"), L_bayes_pos=int(a.get("L_bayes_pos", 0)),
                              L_gauss_pos=a.get("L_gauss_pos", "00"), L_v_pos=a.get("L_v_pos", "11"),
                              T_gauss_pos=int(a.get("T_gauss_pos", 3)), T_v_pos=int(a.get("T_v_pos", 0)),
                              seed=int(a["seed"]) if tag.startswith("seed_") else None)
    assert bool(init) != tag.startswith("seed_")
    hist = reference_shaped_run(args, d, init, torch.device("cuda:0"))
    assert list(hist["halved_epochs"]) == list(z["halved_epochs"]), (hist["valid_loss"], list(z["valid_loss"]))
    assert np.allclose(hist["valid_loss"], z["valid_loss"], rtol=1e-4), (hist["valid_loss"], list(z["valid_loss"]))
    assert abs(hist["test_loss"] - float(z["test_loss"])) <= 1e-4 * float(z["test_loss"])
    assert np.allclose(hist["interval_loss"], z["interval_loss"], rtol=1e-3)
    for k, v in snaps[-1].items():
        scale = float(v.abs().max()) + 1e-12
        assert float((hist["final"][k] - v).abs().max()) <= 1e-3 * scale, k


def test_zero_grad_set_to_none_keeps_no_stale_gradient(tmp_path):
    """optimizer.zero_grad() sets every .grad to None; the in-place wgrad kernels then start from a zeroed buffer again
    (ops._grad_buf), so two identical steps from the same weights give identical gradients -- also for a parameter
    that gets NO gradient in the second step (it must read None, not the first step's values)."""
    from bayeslms_amd import model as M
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    m = M.TransformerModel(60, 32, 4, 64, 2, 0.0, "gelu", True).to(dev)
    crit = nn.CrossEntropyLoss()
    x = torch.randint(0, 60, (12, 5), device=dev)
    t = torch.randint(0, 60, (60,), device=dev)
    opt = optim.SGD(m.parameters(), lr=0.0)
    grads = []
    for _ in range(2):
        opt.zero_grad()
        assert all(p.grad is None for p in m.parameters())
        crit(m(x).view(-1, 60), t).backward()
        grads.append({k: p.grad.clone() for k, p in m.named_parameters()})
    for k in grads[0]:  # (float atomics in the LayerNorm / split-K reductions: equal to rounding, not bit for bit)
        a, b = grads[0][k], grads[1][k]
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-9, k


def test_torch_cross_entropy_on_model_logits_runs_the_engine_kernels(monkeypatch):
    """The unchanged reference loop calls nn.CrossEntropyLoss on `output.view(-1, ntokens)`.  In grad mode the decoder returns
    ops.Logits, on which F.cross_entropy takes the engine's one-pass kernels (non-destructively): same loss and gradient as torch's
    own chain on a plain copy, the logits keep their values after backward, and non-default losses fall through to torch."""
    from bayeslms_amd import model as M, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    V = 60
    m = M.TransformerModel(V, 32, 4, 64, 2, 0.0, "gelu", True).to(dev)
    x = torch.randint(0, V, (12, 5), device=dev)
    t = torch.randint(0, V, (60,), device=dev)
    calls = []
    real = ops._CrossEntropy.apply
    monkeypatch.setattr(ops._CrossEntropy, "apply", staticmethod(lambda *a: (calls.append(a[2:]), real(*a))[1]))
    m.train()
    out = m(x)
    assert type(out) is ops.Logits and type(out.view(-1, V)) is ops.Logits and type(out * 2.0) is torch.Tensor
    before = out.detach().clone()
    loss = nn.CrossEntropyLoss()(out.view(-1, V), t)
    assert calls == [(False, True)] and type(loss) is torch.Tensor
    loss.backward()
    assert torch.equal(out.detach().as_subclass(torch.Tensor), before)  # the user's logits are untouched
    g_engine = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad()
    plain = m(x).as_subclass(torch.Tensor)
    ref = torch.nn.functional.cross_entropy(plain.view(-1, V), t)  # torch's own log-softmax + NLL chain
    assert len(calls) == 1 and abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))
    ref.backward()
    for k, p in m.named_parameters():
        assert float((p.grad - g_engine[k]).abs().max()) <= 2e-5 * float(g_engine[k].abs().max()) + 1e-9, k
    # anything but the default loss is torch's business
    nn.CrossEntropyLoss(reduction="sum")(m(x).view(-1, V), t)
    nn.CrossEntropyLoss(label_smoothing=0.1)(m(x).view(-1, V), t)
    assert len(calls) == 1
    m.eval()
    with torch.no_grad():
        assert type(m(x)) is torch.Tensor


def test_unmanaged_training_forwards_draw_fresh_noise():
    """A reference-shaped loop never calls set_step: every training-mode forward in grad mode moves on to the next step's noise and
    dropout streams by itself (NoiseState.auto_step), eval / no_grad forwards do not, and the first explicit set_step takes over."""
    from bayeslms_amd import model as M
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    m = M.BayesTransformerModel(50, 32, 4, 64, 2, 0.3, True, "FFN").to(dev)
    x = torch.randint(0, 50, (9, 4), device=dev)
    m.train()
    a, b = m(x).detach().clone(), m(x).detach().clone()
    assert m.noise_state.step == 2 and not torch.equal(a, b)  # new eps and new masks
    with torch.no_grad():
        c, d = m(x), m(x)
    assert m.noise_state.step == 2 and torch.equal(c, d)      # no_grad: the Monte-Carlo scorer's kind of call manages its own step
    m.eval()
    m(x)
    assert m.noise_state.step == 2
    m.train()
    m.set_step(7)
    e, f = m(x).detach().clone(), m(x).detach().clone()
    assert m.noise_state.step == 7 and torch.equal(e, f)      # managed: the same step gives the same draw


@pytest.mark.parametrize("family", ["tlm_ffn", "lstm_bayes3"])
def test_missing_gradients_come_from_one_zeroed_slab_per_step(family):
    """`zero_grad()` (set_to_none) leaves every `.grad` None, and the in-place weight-gradient kernels then need zeroed targets: from
    the second step on they are views of ONE slab per step (ops._GradSlab: one fill instead of one per parameter -- 78 launches per
    step of the headline model).  Same losses and parameters as with a zeros_like per parameter, under torch's own clip + SGD with
    momentum; a fresh slab every step; a parameter that joins late or is dropped does not disturb the others."""
    from bayeslms_amd import model as M, ops
    dev = torch.device("cuda:0")
    V = 70

    def run(slab):
        ops.set_grad_slab(slab)
        torch.manual_seed(11)
        if family == "tlm_ffn":
            m = M.BayesTransformerModel(V, 32, 4, 64, 2, 0.1, True, "FFN").to(dev)
        else:
            m = M.BayesRNNModel("LSTM", V, 32, 32, 2, 0.1, True, 3).to(dev)
        g = torch.Generator().manual_seed(3)
        opt = torch.optim.SGD(m.parameters(), lr=0.5, momentum=0.9)
        losses, slabs = [], []
        hidden = m.init_hidden(6) if family != "tlm_ffn" else None
        for step in range(4):
            x = torch.randint(0, V, (10, 6), generator=g).to(dev)
            t = torch.randint(0, V, (60,), generator=g).to(dev)
            m.train()
            m.set_step(step)
            m.zero_grad()
            assert all(p.grad is None for p in m.parameters())
            if hidden is None:
                out = m(x)
            else:
                out, _ = m(x, tuple(h.detach() for h in hidden))
            loss = nn.CrossEntropyLoss()(out.view(-1, V), t)
            loss.backward()
            stores = {p.grad.untyped_storage().data_ptr() for p in m.parameters() if p.grad is not None}
            slabs.append(stores)
            torch.nn.utils.clip_grad_norm_(m.parameters(), 0.25)
            opt.step()
            losses.append(float(loss))
        return losses, slabs, {k: v.detach().clone() for k, v in m.state_dict().items()}
    try:
        l_on, s_on, p_on = run(True)
        l_off, s_off, p_off = run(False)
    finally:
        ops.set_grad_slab(True)
    n_par = len(p_on)
    assert len(s_on[0]) > 3 and len(s_off[1]) > 3            # first sight / slab off: a buffer per parameter
    # steps 1..3: the gradients written by the engine's kernels live in one storage (autograd-accumulated ones, if any, in their own)
    for k in (1, 2, 3):
        assert 2 * len(s_on[k]) <= len(s_off[k]), (k, len(s_on[k]), len(s_off[k]), n_par)
    for a, b in zip(l_on, l_off):
        assert abs(a - b) <= 1e-6 * abs(b), (l_on, l_off)
    for k in p_on:
        assert float((p_on[k] - p_off[k]).abs().max()) <= 1e-5 * float(p_off[k].abs().max()) + 1e-8, k
