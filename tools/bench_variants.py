"""Training tokens/s of the other model families at the BASELINE shapes (one MI355X, fp32, synthetic
stream, the same Trainer step as bench.py): Transformer d512 ff4096 L6 h8 T128 B64 V33000 with
--uncertainty none / Bayesian {FFN, MHA, EMB} / Gaussian (T_gauss_pos 3), and the 2x1024 LSTMs (T35 B64)
none / Bayesian pos 3 / Gaussian '33' / Variational '11'."""
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, ".")
from bayeslms_amd import engine, model as M, train as T  # noqa: E402
from bayeslms_amd.data import batchify, get_batch, synthetic_corpus  # noqa: E402


def run(name, model, kl_fn, seq, B, lr, steps=12, warm=4):
    dev = torch.device("cuda:0")
    V = 33000
    stream = synthetic_corpus(V, B * ((steps + warm) * seq + 1) + 17, seed=1111)
    train = batchify(stream, B, dev)
    model = model.to(dev)
    tr = engine.Trainer(model, lr=lr, clip=1.0, kl_scale=float(seq) / train.size(0), seed=1111)
    is_rnn = hasattr(model, "init_hidden")
    hidden = model.init_hidden(B) if is_rnn else None
    for i in range(warm + steps):
        if i == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        data, tgt = get_batch(train, i * seq, seq)
        if is_rnn:
            hidden = tuple(h.detach() for h in hidden)  # train.py:318 repackage_hidden
            loss, _, hidden = tr.step(data, tgt, kl_fn=kl_fn, hidden=hidden)
        else:
            loss, _, _ = tr.step(data, tgt, kl_fn=kl_fn)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name:46s} {dt * 1e3:7.2f} ms/step  {seq * B / dt:9.0f} tokens/s  loss {float(loss):.3f}", flush=True)


def main():
    V = 33000
    ns = lambda **k: SimpleNamespace(**{**dict(model="Transformer", uncertainty="none", T_bayes_pos="none", L_bayes_pos=0,  # noqa: E731
                                               T_gauss_pos=3, L_gauss_pos="00", L_v_pos="11", T_v_pos=0), **k})
    torch.manual_seed(1111)
    tf = [("Transformer none", lambda: M.TransformerModel(V, 512, 8, 4096, 6, 0.2, "gelu", True), ns()),
          ("Transformer Bayesian FFN (cfg3)", lambda: M.BayesTransformerModel(V, 512, 8, 4096, 6, 0.2, True, "FFN"),
           ns(uncertainty="Bayesian", T_bayes_pos="FFN")),
          ("Transformer Bayesian MHA", lambda: M.BayesTransformerModel(V, 512, 8, 4096, 6, 0.2, True, "MHA"),
           ns(uncertainty="Bayesian", T_bayes_pos="MHA")),
          ("Transformer Bayesian EMB", lambda: M.BayesTransformerModel(V, 512, 8, 4096, 6, 0.2, True, "EMB"),
           ns(uncertainty="Bayesian", T_bayes_pos="EMB")),
          ("Transformer Gaussian T_gauss_pos 3 (cfg5)", lambda: M.GaussTransformerModel(V, 512, 8, 4096, 6, 0.2, True, 3),
           ns(uncertainty="Gaussian", T_gauss_pos=3))]
    def sampled(build):  # round 4: GPNN.sample raised (train --gp-sample 1): the GP tensors re-sampled every step
        def f():
            m = build()
            for g in m.modules():
                if isinstance(g, M.GPNN):
                    g.sample = True
            return m
        return f
    tf.append(("Transformer Gaussian T_gauss_pos 3, GPNN.sample raised", sampled(tf[-1][1]), tf[-1][2]))
    if len(sys.argv) < 2 or sys.argv[1] != "lstm":
        for name, build, a in tf:
            run(name, build(), T.kl_selector(a), 128, 64, 0.1)
    rn = [("LSTM none", lambda: M.RNNModel("LSTM", V, 1024, 1024, 2, 0.2, True), ns(model="LSTM")),
          ("LSTM Bayesian L_bayes_pos 3 (cfg2)", lambda: M.BayesRNNModel("LSTM", V, 1024, 1024, 2, 0.2, True, 3),
           ns(model="LSTM", uncertainty="Bayesian", L_bayes_pos=3)),
          ("LSTM Gaussian L_gauss_pos 33", lambda: M.GaussRNNModel("LSTM", V, 1024, 1024, 2, 0.2, True, "33"),
           ns(model="LSTM", uncertainty="Gaussian", L_gauss_pos="33")),
          ("LSTM Gaussian L_gauss_pos 6360 (README example)", lambda: M.GaussRNNModel("LSTM", V, 1024, 1024, 2, 0.2, True, "6360"),
           ns(model="LSTM", uncertainty="Gaussian", L_gauss_pos="6360")),
          ("LSTM Variational L_v_pos 11", lambda: M.VariationalRNNModel("LSTM", V, 1024, 1024, 2, 0.2, True, "11"),
           ns(model="LSTM", uncertainty="Variational", L_v_pos="11")),
          ("LSTM Gaussian L_gauss_pos 53 (GPNN on the cell state)", lambda: M.GaussRNNModel("LSTM", V, 1024, 1024, 2, 0.2, True, "53"),
           ns(model="LSTM", uncertainty="Gaussian", L_gauss_pos="53")),
          # GPNN2 cells: fresh frequencies at every time step, 4-6 skinny launches per step from one autograd node
          ("LSTM Gaussian L_gauss_pos 34 (GPNN2 on the cell gate)", lambda: M.GaussRNNModel("LSTM", V, 1024, 1024, 2, 0.2, True, "34"),
           ns(model="LSTM", uncertainty="Gaussian", L_gauss_pos="34")),
          ("LSTM Gaussian L_gauss_pos 54 (GPNN2 on the cell state)", lambda: M.GaussRNNModel("LSTM", V, 1024, 1024, 2, 0.2, True, "54"),
           ns(model="LSTM", uncertainty="Gaussian", L_gauss_pos="54")),
          ("LSTM Gaussian L_gauss_pos 64 (GPNN2 hidden projection)", lambda: M.GaussRNNModel("LSTM", V, 1024, 1024, 2, 0.2, True, "64"),
           ns(model="LSTM", uncertainty="Gaussian", L_gauss_pos="64")),
          ("LSTM Gaussian L_gauss_pos 74 (GPNN2 input projection, batched over the window)", lambda: M.GaussRNNModel("LSTM", V, 1024, 1024, 2, 0.2, True, "74"),
           ns(model="LSTM", uncertainty="Gaussian", L_gauss_pos="74"))]
    for name in ("LSTM Gaussian L_gauss_pos 33", "LSTM Gaussian L_gauss_pos 6360 (README example)", "LSTM Gaussian L_gauss_pos 53 (GPNN on the cell state)"):
        _, build, a = next(r for r in rn if r[0] == name)
        rn.append((name + ", GPNN.sample raised", sampled(build), a))
    for name, build, a in rn:
        run(name, build(), T.kl_selector(a), 35, 64, 0.5)


if __name__ == "__main__":
    main()
