import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.multiprocessing as mp
import test_gpu_dp as D
if __name__ == "__main__":
    for family in D.FAMILIES:
        with mp.Manager() as mgr:
            ret = mgr.dict()
            D._spawn(1, ret, family)
            D._spawn(4, ret, family)
            l1, p1 = ret[(1, 0)]
            ps = [ret[(4, r)][1] for r in range(4)]
            same = all(torch.equal(ps[0], p) for p in ps[1:])
            err = float((p1 - ps[0]).abs().max() / p1.abs().max())
            print(family, "replicas identical:", same, "rel err vs 1 rank: %.2e" % err, flush=True)
