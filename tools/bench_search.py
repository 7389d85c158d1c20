#!/usr/bin/env python3
"""Architecture-search step time at the reference's full sizes: one window =
Architect.step (validation CE -> Adam on the architecture logits) + network step (CE -> clip -> SGD with
weight decay).  KIND=tlm: GaussTransModelSearch d=512 ff=4096 8 heads 6 layers V=33000 T=128 B=64;
KIND=lstm: BayesLSTMModelSearch E=H=1024 V=33000 T=35 B=64.  Prints tokens/s of the training stream and the
split between the two halves of the window."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayeslms_amd import engine, model_search_bayes as S, train_search_bayes as TS  # noqa: E402
from bayeslms_amd.architect import Architect  # noqa: E402
from bayeslms_amd.model import repackage_hidden  # noqa: E402

kind = os.environ.get("KIND", "tlm")
steps, warm = int(os.environ.get("STEPS", "10")), int(os.environ.get("WARM", "5"))
dev = torch.device("cuda:0")
V, B = 33000, 64
torch.manual_seed(11)
if kind == "tlm":
    T = 128
    m = S.GaussTransModelSearch(V, 512, 8, 4096, 6, 0.2, True).to(dev)
    args = types.SimpleNamespace(model="Transformer", T_bayes_pos="FFN", uncertainty="none", L_bayes_pos=0)
else:
    T = 35
    m = S.BayesLSTMModelSearch("LSTM", V, 1024, 1024, 2, 0.2, True).to(dev)
    args = types.SimpleNamespace(model="LSTM", T_bayes_pos="none", uncertainty="none", L_bayes_pos=1)
TS.freeze_unused(args, m)
kl_fn = TS.kl_selector(args)
arch = Architect(m, V, types.SimpleNamespace(wdecay=5e-7, clip=1.0, arch_lr=3e-3, arch_wdecay=1e-3))
tr = engine.Trainer(m, lr=0.1, clip=1.0, kl_scale=T / 65536.0, weight_decay=TS.SGD_WEIGHT_DECAY)
data = torch.randint(0, V, (T + 1, B), device=dev)
x, y = data[:T], data[1:].reshape(-1)
hidden = m.init_hidden(B) if kind == "lstm" else None
hv = m.init_hidden(B) if kind == "lstm" else None
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
t_arch = t_net = 0.0
for s in range(warm + steps):
    if s == warm:
        torch.cuda.synchronize()
        t0 = time.time()
    m.train()
    m.set_step(2 * s + 1)
    ev[0].record()
    arch.step(x, y, x, y, None, False, hv)
    ev[1].record()
    if kind == "tlm":
        for layer in m.transformerlayers:
            layer.gpnn.sample = True
    else:
        hidden = repackage_hidden(hidden)
    loss, kl, hidden = tr.step(x, y, hidden, kl_fn, philox_step=2 * s)
    if kind == "tlm":
        for layer in m.transformerlayers:
            layer.gpnn.sample = False
    ev[2].record()
    if s >= warm:
        torch.cuda.synchronize()
        t_arch += ev[0].elapsed_time(ev[1])
        t_net += ev[1].elapsed_time(ev[2])
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print("%s search window: %.2f ms (architect %.2f ms, network %.2f ms) -> %.0f tokens/s, loss %.4f, arch softmax %s" % (
    kind, dt * 1e3, t_arch / steps, t_net / steps, T * B / dt, float(loss),
    torch.softmax(m.weights.detach(), -1).flatten().tolist()[:4]))
