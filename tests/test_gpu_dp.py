"""Data-parallel correctness on the GPU box: two ranks (sharing the one GPU, gloo transport) that
each own half of the global batch columns must reproduce the single-process run of the whole global
batch: same eps (no rank in the Philox key), dropout masks keyed by global column, gradients
averaged by the bucketed all-reduce, identical clip+SGD."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(dev):
    from bayeslms_amd import model as M
    torch.manual_seed(5)
    m = M.BayesTransformerModel(150, 32, 4, 64, 2, 0.2, True, "FFN")
    return m.to(dev)


def _kl(model):
    return model.transformerlayers[0].linear2.kl_divergence()


_kl.fusable = True


def _run(rank, world, port, ret):
    import torch.distributed as dist
    from bayeslms_amd import data as D, engine
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    stream = torch.randint(0, 150, (8 * 61,), generator=torch.Generator().manual_seed(1))
    train = D.batchify(stream, 8, dev, rank, world)
    m = _build(dev)
    tr = engine.Trainer(m, lr=0.2, clip=0.5, kl_scale=0.01, seed=1111, rank=rank, world=world, bucket_bytes=8192)
    losses = []
    for i in range(3):
        data, tgt = D.get_batch(train, i * 12, 12)
        loss, kl, _ = tr.step(data, tgt, kl_fn=_kl)
        losses.append(float(loss))
    ret[(world, rank)] = (losses, tr.flat.flat_param.detach().cpu().clone())
    if world > 1:
        dist.destroy_process_group()


def test_two_ranks_equal_one_rank_on_the_global_batch():
    with mp.Manager() as mgr:
        ret = mgr.dict()
        _spawn(1, ret)
        _spawn(2, ret)
        l1, p1 = ret[(1, 0)]
        l2a, p2a = ret[(2, 0)]
        l2b, p2b = ret[(2, 1)]
    assert torch.equal(p2a, p2b)  # ranks stay in lock step
    # mean loss over the global batch = mean of the two half-batch means
    for a, b, c in zip(l1, l2a, l2b):
        assert abs(a - 0.5 * (b + c)) < 2e-4 * abs(a)
    err = float((p1 - p2a).abs().max() / p1.abs().max())
    assert err < 1e-4, err


def _spawn(world, ret):
    port = _free_port()
    mp.spawn(_run, args=(world, port, ret), nprocs=world, join=True)


def _rccl_one_rank(port, ret):
    """RCCL itself (backend "nccl"), one rank: process-group init with device_id as bench.py does it,
    the reducer's side-stream async all-reduce pattern, barrier, and a Trainer step in 'world 2'
    arithmetic (the collective over a 1-rank group is the identity, so the result must equal the
    plain run with gradients halved by grad_scale)."""
    import torch.distributed as dist
    from bayeslms_amd import data as D, engine
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    try:
        x = torch.arange(1024, device=dev, dtype=torch.float32)
        side = torch.cuda.Stream()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        side.wait_event(ev)
        with torch.cuda.stream(side):
            h = dist.all_reduce(x[128:512], op=dist.ReduceOp.SUM, async_op=True)
        h.wait()
        torch.cuda.current_stream().wait_stream(side)
        dist.barrier()
        ok_identity = bool(torch.equal(x.cpu(), torch.arange(1024, dtype=torch.float32)))
        stream = torch.randint(0, 150, (8 * 61,), generator=torch.Generator().manual_seed(1))
        train = D.batchify(stream, 8, dev)
        m = _build(dev)
        tr = engine.Trainer(m, lr=0.2, clip=0.5, kl_scale=0.01, seed=1111, rank=0, world=2, bucket_bytes=8192)
        tr.reducer.world = 2  # grad-ready hooks on, bucketed collectives issued over RCCL (1-rank group: identity)
        losses = []
        for i in range(3):
            data, tgt = D.get_batch(train, i * 12, 12)
            loss, _, _ = tr.step(data, tgt, kl_fn=_kl)
            losses.append(float(loss))
        ret["rccl"] = (ok_identity, losses, bool(torch.isfinite(tr.flat.flat_param).all()))
    finally:
        dist.destroy_process_group()


def test_rccl_backend_single_rank_smoke():
    import torch.distributed as dist
    if not dist.is_nccl_available():
        pytest.skip("no RCCL in this torch build")
    with mp.Manager() as mgr:
        ret = mgr.dict()
        p = mp.get_context("spawn").Process(target=_rccl_one_rank, args=(_free_port(), ret))
        p.start()
        p.join(300)
        assert p.exitcode == 0
        ok, losses, finite = ret["rccl"]
        assert ok and finite and all(v == v for v in losses) and losses[-1] < losses[0] + 1.0
