"""Probe (round 5): does engine.LateRows agree on U -- the number of distinct token ids of the GLOBAL batch -- on every rank when W
ranks share one GPU over gloo?  `bench.py --gpus 4 --backend gloo` died with a gloo size mismatch in the compact all-reduce
(ranks held different U).  Every rank runs LateRows.begin() on fresh random ids for N steps; U (read as sink() reads it) is compared
with the count computed on the host from an object all-gather of the same ids.  usage: late_rows_gloo_probe.py [world] [steps]"""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(rank, world, port, steps, ret):
    import torch.distributed as dist
    from bayeslms_amd import engine
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    V, D, T, B = 33000, 512, 128, 16
    net = torch.nn.ModuleDict({"encoder": torch.nn.Embedding(V, D)}).to(dev)
    flat = engine.FlatBuffers(net)
    red = engine.GradReducer(flat, bucket_bytes=32 << 20)
    late = engine.LateRows(red, net["encoder"].weight)
    bad = []
    g = torch.Generator().manual_seed(100 + rank)
    work = torch.randn(4096, 4096, device=dev)
    for s in range(steps):
        ids = torch.randint(0, 2000 + 500 * (s % 7), (T, B), generator=g).to(dev)
        late.begin(ids)
        for _ in range(int(os.environ.get("LOAD", "3"))):  # something on the compute stream meanwhile, as the forward pass would be
            work = torch.tanh(work @ work * 1e-4)
        late.event.synchronize()
        u = int(late.count_host.item())
        torch.cuda.synchronize()
        u_mark = int(late.mark.sum().item())          # what the presence bitmap holds now
        u_ids = int(torch.unique(late.allids).numel())  # what the gathered ids hold now
        every = [None] * world
        dist.all_gather_object(every, ids.cpu())
        want = int(torch.unique(torch.cat([e.reshape(-1) for e in every])).numel())
        if u != want or u_mark != want or u_ids != want:
            bad.append((s, u, u_mark, u_ids, want))
    ret[rank] = bad
    dist.barrier()
    dist.destroy_process_group()


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(run, args=(world, port, steps, ret), nprocs=world, join=True)
        for r in range(world):
            print("rank", r, "mismatches (step, U read by sink, ones in mark, distinct in allids, wanted):", ret[r][:6], len(ret[r]), "of", steps)


if __name__ == "__main__":
    main()
