"""GPU: deterministic mode (ops.set_deterministic / blm_set_option("deterministic", 1) / BLM_DETERMINISTIC=1) -- the bitwise
double-run check SURVEY 5.2 planned.  With it, two runs from one seed give bit-identical losses and parameters: no K slices
meeting through float atomics in the GEMM family, column sums / GP coefficient gradients in one row chunk, KL sums through
block partials added by one block, the embedding gradient by one wave per vocabulary row in position order, the two-layer LSTM
on one stream.  The reference has no counterpart (cuDNN / cuBLAS atomics decide; train.py:125-131 only seeds)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


@pytest.fixture()
def det():
    from bayeslms_amd import ops
    ops.set_deterministic(True)
    yield ops
    ops.set_deterministic(False)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(family, dev, V):
    from bayeslms_amd import model as M
    torch.manual_seed(5)
    kl = None
    if family == "cfg3":      # BASELINE configs[2] at a small vocabulary: 6 layers, d_model 512, d_ff 4096, 8 heads
        m = M.BayesTransformerModel(V, 512, 8, 4096, 6, 0.2, True, "FFN")
        kl = lambda mm: mm.transformerlayers[0].linear2.kl_divergence()  # noqa: E731
        kl.fusable = True
    elif family == "cfg2":    # BASELINE configs[1] at a small vocabulary: Bayesian LSTM, cell gate, 2 x 1024
        m = M.BayesRNNModel("LSTM", V, 1024, 1024, 2, 0.2, True, 3)
        kl = lambda mm: mm.rnn.kl_divergence()  # noqa: E731
    elif family == "tlm_mha":
        m = M.BayesTransformerModel(V, 64, 4, 128, 2, 0.2, True, "MHA")
        kl = lambda mm: mm.transformerlayers[0].self_attn.o_net.kl_divergence()  # noqa: E731
        kl.fusable = True
    elif family == "tlm_emb":
        m = M.BayesTransformerModel(V, 64, 4, 128, 2, 0.2, False, "EMB")
        kl = lambda mm: mm.embed_kl_divergence()  # noqa: E731
    elif family == "tlm_gauss3":
        m = M.GaussTransformerModel(V, 64, 4, 128, 2, 0.2, True, 3)
        kl = lambda mm: mm.transformerlayers[0].gpnn.kl_divergence()  # noqa: E731
    elif family == "tlm_var3":
        m = M.VTransformerModel(V, 64, 4, 128, 3, 0.2, True, 3)  # (its kl_divergence raises as the reference's crashes, model.py:2770-2779)
    elif family == "rnn_none":
        m = M.RNNModel("LSTM", V, 64, 64, 2, 0.2, True)
    elif family == "cfg2_small":
        m = M.BayesRNNModel("LSTM", V, 64, 64, 2, 0.2, True, 3)
        kl = lambda mm: mm.rnn.kl_divergence()  # noqa: E731
    elif family == "rnn_gauss33":
        m = M.GaussRNNModel("LSTM", V, 64, 64, 2, 0.2, True, "33")
        kl = lambda mm: mm.rnn.rnn[0].gpnn.kl_divergence()  # noqa: E731
    elif family == "rnn_var11":
        m = M.VariationalRNNModel("LSTM", V, 64, 64, 2, 0.2, True, "11")
        kl = lambda mm: sum(mm.rnn.rnn[c].vnn.kl_divergence() for c in (0, 1))  # noqa: E731
    else:
        raise ValueError(family)
    return m.to(dev), kl


def _train(family, dev, V, T, B, steps, rank=0, world=1, gp_sample=False):
    """`steps` Trainer steps from a fixed seed -> (losses, flat parameter buffer on the host)."""
    from bayeslms_amd import data as D, engine
    from bayeslms_amd.model import GPNN, repackage_hidden
    m, kl_fn = _build(family, dev, V)
    if gp_sample:
        for g in m.modules():
            if isinstance(g, GPNN) and g.draws_noise():
                g.sample = True
    stream = torch.randint(0, V, (B * world * (steps * T + 1) + 5,), generator=torch.Generator().manual_seed(1))
    train = D.batchify(stream, B * world, dev, rank, world)
    tr = engine.Trainer(m, lr=0.5, clip=0.5, kl_scale=float(T) / train.size(0), seed=1111, rank=rank, world=world, bucket_bytes=1 << 20)
    hidden = m.init_hidden(B) if hasattr(m, "init_hidden") else None
    losses = []
    for i in range(steps):
        data, tgt = D.get_batch(train, i * T, T)
        if hidden is not None:
            hidden = repackage_hidden(hidden)
        loss, _, hidden = tr.step(data, tgt, hidden=hidden, kl_fn=kl_fn)
        losses.append(loss)
    losses = [float(x) for x in torch.stack(losses).cpu()]
    return losses, tr.flat.flat_param.detach().cpu().clone()


@pytest.mark.parametrize("family,T,B", [("cfg3", 128, 64), ("cfg2", 35, 64)])
def test_two_full_width_runs_from_one_seed_are_bit_identical(family, T, B, det):
    """two cfg3-shaped and two cfg2-shaped (small-V) 20-step runs -- dropout 0.2 on, KL on, clip + SGD momentum --
    give torch.equal parameters and equal loss sequences."""
    dev = torch.device("cuda:0")
    l1, p1 = _train(family, dev, 2000, T, B, 20)
    l2, p2 = _train(family, dev, 2000, T, B, 20)
    assert l1 == l2, [(a, b) for a, b in zip(l1, l2) if a != b][:3]
    assert torch.equal(p1, p2), float((p1 - p2).abs().max())
    assert all(x == x for x in l1) and len(set(l1)) == len(l1)  # finite, and every step moved the model


@pytest.mark.parametrize("family", ["tlm_mha", "tlm_emb", "tlm_gauss3", "tlm_var3", "rnn_none", "rnn_gauss33", "rnn_var11"])
def test_every_family_is_bit_identical_run_to_run(family, det):
    """The other reduction sites: the generic (D not 256 / 512 / 1024 / 2048) LayerNorm backward, GP coefficient gradients
    (also with GPNN.sample raised), the grouped sampling + KL kernel, the Variational noise rows, untied embeddings."""
    dev = torch.device("cuda:0")
    for gp in ((False, True) if "gauss" in family else (False,)):
        l1, p1 = _train(family, dev, 150, 12, 8, 6, gp_sample=gp)
        l2, p2 = _train(family, dev, 150, 12, 8, 6, gp_sample=gp)
        assert l1 == l2 and torch.equal(p1, p2), (family, gp, float((p1 - p2).abs().max()))


@pytest.mark.parametrize("T,B", [(36, 8), (72, 16)])
def test_layer_wavefront_is_bit_identical_to_the_sequential_layers(T, B, det):
    """A bitwise race check of the two- / three-stream layer wavefront (ops.lstm_stack2; ADVICE r4: at the default mode's 1e-6
    atomics noise a missing stream wait cannot be told from summation order).  In deterministic mode every per-chunk product has
    one K slice, so the wavefront FORCED on must reproduce the sequential layers bit for bit -- losses and every parameter
    after 8 training steps of the 2 x 64 LSTM and of the Bayesian LSTM (dropout on, clip + SGD momentum).  T 36: three chunks on
    two streams; T 72: nine chunks, the per-chunk GEMMs on their third stream."""
    dev = torch.device("cuda:0")
    out = {}
    for family in ("rnn_none", "cfg2_small"):
        for forced in (False, True):
            det.set_lstm_wavefront(forced)
            try:
                calls = []
                real = det.lstm_stack2
                det.lstm_stack2 = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
                out[(family, forced)] = _train(family, dev, 150, T, B, 8) + (len(calls),)
            finally:
                det.lstm_stack2 = real
                det.set_lstm_wavefront(None)
        (l0, p0, n0), (l1, p1, n1) = out[(family, False)], out[(family, True)]
        assert n0 == 0 and n1 == 8, (n0, n1)  # the sequential layers, then the wavefront in every step
        assert l0 == l1 and torch.equal(p0, p1), (family, float((p0 - p1).abs().max()))


def test_deterministic_mode_equals_the_default_mode_to_rounding(det):
    """Same arithmetic, another order of additions: 6 steps of a small model differ from the default mode's by rounding only."""
    dev = torch.device("cuda:0")
    ld, pd = _train("tlm_mha", dev, 150, 12, 8, 6)
    det.set_deterministic(False)
    l0, p0 = _train("tlm_mha", dev, 150, 12, 8, 6)
    assert max(abs(a - b) for a, b in zip(ld, l0)) < 1e-4 * abs(l0[0])
    assert float((pd - p0).abs().max()) < 1e-4 * float(p0.abs().max())


def test_planner_gives_one_slice_and_kernels_take_the_fixed_order_forms(det):
    """What the switch changes, kernel by kernel: every GEMM plan is one K slice (table, override and model plans alike);
    embedding gradient, KL sums, column sums and GP coefficient gradients are bit-identical call to call and equal to the
    default forms to rounding."""
    import ctypes as C
    from bayeslms_amd import _lib as L, ops
    dev = torch.device("cuda:0")
    lib = L.lib()
    # planner: a shape whose table plan is K-sliced (the decoder wgrad of the LSTM recipes) and an override
    a = L.GemmArgs()
    a.abi_version = L.ABI_VERSION
    a.op, a.M, a.N, a.K = L.GEMM_TN, 1024, 1024, 2240
    a.lda = a.ldb = a.ldc = 1024
    a.flags = L.GEMM_ACCUMULATE
    pl = L.GemmPlan()
    for forced in (0, 4):
        assert lib.blm_gemm_plan_override(0, forced) == 0
        assert lib.blm_gemm_plan_query(C.byref(a), C.byref(pl)) == 0 and pl.splits == 1, (forced, pl.splits)
        ops.set_deterministic(False)
        assert lib.blm_gemm_plan_query(C.byref(a), C.byref(pl)) == 0 and (pl.splits != 1 or forced == 0)
        ops.set_deterministic(True)
    assert lib.blm_gemm_plan_override(0, 0) == 0
    # embedding gradient with repeated ids, out-of-range ids skipped, D not a multiple of 64, more rows than one LDS chunk
    g = torch.Generator().manual_seed(3)
    for T, B, D, V in ((40, 16, 96, 50), (300, 32, 520, 700)):
        ids = torch.randint(0, V, (T, B), generator=g).to(dev)
        ids[0, 0], ids[1, 1] = -1, V + 5
        dy = torch.randn(T, B, D, generator=g).to(dev)
        outs = []
        for mode in (True, True, False):
            ops.set_deterministic(mode)
            acc = torch.ones(V, D, device=dev)
            L.check(lib.blm_embed_bwd(ids.data_ptr(), dy.data_ptr(), acc.data_ptr(), T, B, D, V, 0.5, 0.0, None, 0, B, L.stream()), "embed_bwd")
            outs.append(acc.cpu())
        ops.set_deterministic(True)
        ok = (ids >= 0) & (ids < V)
        want = torch.ones(V, D).index_add_(0, ids[ok].cpu().reshape(-1), 0.5 * dy[ok].cpu().reshape(-1, D))
        assert torch.equal(outs[0], outs[1]) and torch.allclose(outs[0], want, atol=1e-4) and torch.allclose(outs[2], want, atol=1e-4)
    # KL sum, column sums
    mu, lg = torch.randn(300, 130, generator=g).to(dev), (0.1 * torch.randn(300, 130, generator=g)).to(dev)
    kls = []
    for mode in (True, True, False):
        ops.set_deterministic(mode)
        kls.append(float(ops.kl_mean(mu, lg)))
    ops.set_deterministic(True)
    assert kls[0] == kls[1] and abs(kls[0] - kls[2]) < 1e-5 * abs(kls[2])
    x = torch.randn(5000, 130, generator=g).to(dev)
    sums = []
    for mode in (True, True, False):
        ops.set_deterministic(mode)
        o = torch.zeros(130, device=dev)
        L.check(lib.blm_colsum(x.data_ptr(), 130, o.data_ptr(), 5000, 130, 0, L.stream()), "colsum")
        sums.append(o.cpu())
    ops.set_deterministic(True)
    assert torch.equal(sums[0], sums[1]) and torch.allclose(sums[0], x.sum(0).cpu(), atol=2e-3) and torch.allclose(sums[2], sums[0], atol=2e-3)


def _dp_run(rank, world, port, ret, tag):
    import torch.distributed as dist
    from bayeslms_amd import ops
    torch.cuda.set_device(0)
    ops.set_deterministic(True)
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    ret[(tag, rank)] = _train("tlm_mha", torch.device("cuda:0"), 150, 12, 4, 6, rank, world)
    if world > 1:
        dist.destroy_process_group()


def test_two_rank_runs_are_bit_identical_to_each_other_and_equal_one_rank_to_rounding():
    """Two data-parallel runs of the same world size (2 ranks sharing this box's GPU, gloo) are bit-identical, rank to rank and
    run to run.  A one-rank run of the same global batch adds the batch columns in another order (each rank's wgrad sums its own
    columns, the all-reduce adds the two): equal to rounding, not bit for bit -- a property of float addition, not of the mode."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        for tag in ("a", "b"):
            mp.spawn(_dp_run, args=(2, _free_port(), ret, tag), nprocs=2, join=True)
        la, pa = ret[("a", 0)]
        assert torch.equal(pa, ret[("a", 1)][1])                      # replicas in lock step
        assert torch.equal(pa, ret[("b", 0)][1]) and la == ret[("b", 0)][0]  # run to run
        assert ret[("a", 1)][0] == ret[("b", 1)][0]
    os.environ["BLM_DETERMINISTIC"] = "1"
    try:
        from bayeslms_amd import ops
        ops.set_deterministic(True)
        l1, p1 = _train("tlm_mha", torch.device("cuda:0"), 150, 12, 8, 6)
    finally:
        os.environ.pop("BLM_DETERMINISTIC", None)
        ops.set_deterministic(False)
    assert float((p1 - pa).abs().max()) < 1e-4 * float(p1.abs().max())


def test_train_cli_deterministic_flag_gives_bit_identical_checkpoints(tmp_path):
    """`python -m bayeslms_amd.train --deterministic 1`, twice, on a reference trajectory's corpus WITH dropout (0.3, which the
    recorded RNG-free run does not have): identical log values and a bit-identical best checkpoint."""
    import numpy as np
    from bayeslms_amd import ops, train as T
    from test_train_traj_oracle import load_traj, write_corpus
    z, args, init, _ = load_traj("lstm_bayes5")
    d = str(tmp_path)
    write_corpus(z, d)
    prior = os.path.join(d, "prior")
    os.makedirs(prior)
    torch.save(init, os.path.join(prior, "model.pt"))
    argv = [str(a) for a in z["argv"]]
    argv[argv.index("--dropout") + 1] = "0.3"
    argv[argv.index("--epochs") + 1] = "2"
    runs = []
    try:
        for k in range(2):
            save = os.path.join(d, "m%d.pt" % k)
            hist = {}
            T.main(argv + ["--data", d, "--save", save, "--prior_path", prior, "--cuda", "--deterministic", "1"], history=hist)
            runs.append((hist, torch.load(save, map_location="cpu")))
    finally:
        ops.set_deterministic(False)
    (h0, f0), (h1, f1) = runs
    assert h0["interval_loss"] == h1["interval_loss"] and h0["valid_loss"] == h1["valid_loss"] and h0["test_loss"] == h1["test_loss"]
    assert all(torch.equal(f0[k], f1[k]) for k in f0) and len(h0["interval_loss"]) > 0 and np.isfinite(h0["test_loss"])
