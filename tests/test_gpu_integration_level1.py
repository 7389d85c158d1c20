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


def reference_shaped_run(args, data_dir, init, device):
    """train.py:167-185, 193-223, 239-258, 299-519 with `model` = the shim module.  Everything that
    is not the model is torch's own: criterion, zero_grad, clip_grad_norm_, optim.SGD, state_dict save / load."""
    import types
    shim = types.ModuleType("model")
    exec("from bayeslms_amd.model import *", shim.__dict__)  # the two-line shim, verbatim
    from bayeslms_amd import data as refdata  # data.py of the reference: Corpus / Dictionary, same file formats
    model = shim

    corpus = refdata.Corpus(data_dir)

    def batchify(data, bsz):  # train.py:167-179
        nbatch = data.size(0) // bsz
        data = data.narrow(0, 0, nbatch * bsz)
        return data.view(bsz, -1).t().contiguous().to(device)

    eval_batch_size = 20
    train_data = batchify(corpus.train, args.batch_size)
    val_data = batchify(corpus.valid, eval_batch_size)
    test_data = batchify(corpus.test, eval_batch_size)
    ntokens = len(corpus.dictionary)
    if args.model == 'Transformer':  # train.py:193-199 (--uncertainty none)
        net = model.TransformerModel(ntokens, args.emsize, args.nhead, args.nhid, args.nlayers, args.dropout, "gelu", args.tied).to(device)
    else:
        net = model.RNNModel(args.model, ntokens, args.emsize, args.nhid, args.nlayers, args.dropout, args.tied).to(device)
    criterion = nn.CrossEntropyLoss()
    model_dict = net.state_dict()  # --prior True, train.py:239-258
    model_dict.update({k: v for k, v in init.items() if k in model_dict})
    net.load_state_dict(model_dict)

    def repackage_hidden(h):
        return h.detach() if isinstance(h, torch.Tensor) else tuple(repackage_hidden(v) for v in h)

    def get_batch(source, i):
        seq_len = min(args.seq_len, len(source) - 1 - i)
        return source[i:i + seq_len], source[i + 1:i + 1 + seq_len].view(-1)

    hist = {"interval_loss": [], "valid_loss": [], "halved_epochs": []}

    def train(optimizer):
        net.train()
        total_loss = 0.
        hidden = net.init_hidden(args.batch_size) if args.model != 'Transformer' else None
        for batch, i in enumerate(range(0, train_data.size(0) - 1, args.seq_len)):
            data, targets = get_batch(train_data, i)
            optimizer.zero_grad()
            if args.model == 'Transformer':
                output = net(data)
            else:
                hidden = repackage_hidden(hidden)
                output, hidden = net(data, hidden)
            loss = criterion(output.view(-1, ntokens), targets)  # --uncertainty none: kl_loss = 0
            loss.backward()
            torch.nn.utils.clip_grad_norm_(net.parameters(), args.clip)
            optimizer.step()
            total_loss += loss.item()
            if batch % args.log_interval == 0 and batch > 0:
                hist["interval_loss"].append(total_loss / args.log_interval)
                total_loss = 0.

    def evaluate(source):
        net.eval()
        total_loss = 0.
        hidden = net.init_hidden(eval_batch_size) if args.model != 'Transformer' else None
        with torch.no_grad():
            for i in range(0, source.size(0) - 1, args.seq_len):
                data, targets = get_batch(source, i)
                if args.model == 'Transformer':
                    output = net(data)
                else:
                    output, hidden = net(data, hidden)
                    hidden = repackage_hidden(hidden)
                total_loss += len(data) * criterion(output.view(-1, ntokens), targets).item()
        return total_loss / (len(source) - 1)

    lr, best_val_loss, counter = args.lr, None, 0
    optimizer = optim.SGD(net.parameters(), lr=args.lr, momentum=0.9, weight_decay=0)
    saved = io.BytesIO()
    for epoch in range(1, args.epochs + 1):
        train(optimizer)
        val_loss = evaluate(val_data)
        hist["valid_loss"].append(val_loss)
        if not best_val_loss or val_loss < best_val_loss:
            saved = io.BytesIO()
            torch.save(net.state_dict(), saved)
            best_val_loss = val_loss
        else:
            lr /= 2.
            optimizer = optim.SGD(net.parameters(), lr=lr, momentum=0.9, weight_decay=0)
            saved.seek(0)
            net.load_state_dict(torch.load(saved, map_location=lambda storage, loc: storage))
            counter += 1
            hist["halved_epochs"].append(epoch)
        if counter == 8:
            break
    saved.seek(0)
    net.load_state_dict(torch.load(saved, map_location=lambda storage, loc: storage))
    hist["test_loss"] = evaluate(test_data)
    hist["final"] = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    assert math.isfinite(hist["test_loss"])
    return hist


@pytest.mark.parametrize("tag", ["tlm_none", "lstm_none"])
def test_shim_under_the_reference_loop_reproduces_train_py(tag, tmp_path):
    import argparse
    z, a, init, snaps = load_traj(tag)
    d = str(tmp_path)
    write_corpus(z, d)
    args = argparse.Namespace(model=a["model"], emsize=int(a["emsize"]), nhid=int(a["nhid"]), nlayers=int(a["nlayers"]),
                              nhead=int(a.get("nhead", 2)), dropout=float(a["dropout"]), tied=bool(a.get("tied", False)),
                              batch_size=int(a["batch_size"]), seq_len=int(a["seq_len"]), clip=float(a["clip"]), lr=float(a["lr"]),
                              epochs=int(a["epochs"]), log_interval=int(a["log_interval"]))
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
