"""GPU: the data-parallel entry point users start -- `torchrun ... -m bayeslms_amd.train` -- as two ranks sharing this
box's GPU (`--dist-backend gloo`; RCCL refuses two ranks on one device), `--batch-size` = the fixture's GLOBAL batch:
the same LR-halving epochs, valid / test loss (1e-4) and final checkpoint (1e-3) as the single-process CLI AND as the
RNG-free trajectory of the reference's own train.py (tests/golden/train_traj_*.npz; train.py:306-438, 464-519).
The ranks are child processes of a `torch.distributed.run` child: nothing is forked from or exec'ed over this
(GPU-initialised) test process."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from test_train_traj_oracle import load_traj, write_corpus

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cli(world, argv, hist, extra=()):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    if world == 1:
        cmd = [sys.executable, "-m", "bayeslms_amd.train"]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "-m", "bayeslms_amd.train"]
    r = subprocess.run(cmd + argv + ["--history", hist] + list(extra), capture_output=True, text=True, timeout=900,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    return json.load(open(hist)), r.stdout


@pytest.mark.parametrize("tag", ["lstm_none", "tlm_none"])
def test_torchrun_two_rank_cli_equals_one_rank_and_reference_trajectory(tag, tmp_path):
    z, args, init, snaps = load_traj(tag)
    d = str(tmp_path)
    write_corpus(z, d)
    prior = os.path.join(d, "prior")
    os.makedirs(prior)
    torch.save(init, os.path.join(prior, "model.pt"))
    base = [str(a) for a in z["argv"]] + ["--data", d, "--prior_path", prior, "--cuda"]
    assert int(args["batch_size"]) % 2 == 0
    runs = {}
    settings = [(1, ()), (2, ("--dist-backend", "gloo")), (-2, ("--dist-backend", "gloo", "--dp-overlap", "0", "--dp-late-rows", "0"))]
    if int(args["batch_size"]) % 4 == 0:  # four ranks, ONE column each (the B <= 4 step kernels, one-column attention batches)
        settings.append((4, ("--dist-backend", "gloo")))
    for world, extra in settings:
        save = os.path.join(d, "model_w%d.pt" % world)
        hist, out = _cli(abs(world), base + ["--save", save], os.path.join(d, "hist_w%d.json" % world), extra)
        runs[world] = (hist, torch.load(save, map_location="cpu"), out)
    ref_final = snaps[-1]  # parameters train.py evaluated on the test set = its best checkpoint
    for world, (hist, final, out) in runs.items():
        assert list(hist["halved_epochs"]) == list(z["halved_epochs"]), (world, hist["valid_loss"], list(z["valid_loss"]))
        assert np.allclose(hist["valid_loss"], z["valid_loss"], rtol=1e-4), (world, hist["valid_loss"])
        assert abs(hist["test_loss"] - float(z["test_loss"])) <= 1e-4 * float(z["test_loss"]), world
        assert np.allclose(hist["interval_loss"], z["interval_loss"], rtol=1e-3), world  # mean over the GLOBAL batch
        for k, v in ref_final.items():
            assert float((final[k] - v).abs().max()) <= 1e-3 * (float(v.abs().max()) + 1e-12), (world, k)
        assert out.count("| end of epoch") == len(z["valid_loss"])  # rank 0 alone prints
    h1, f1, _ = runs[1]
    for world in [w for w in runs if w != 1]:
        h2, f2, _ = runs[world]
        assert np.allclose(h2["valid_loss"], h1["valid_loss"], rtol=1e-4) and h2["halved_epochs"] == h1["halved_epochs"]
        for k, v in f1.items():
            assert float((f2[k] - v).abs().max()) <= 1e-3 * (float(v.abs().max()) + 1e-12), (world, k)


@pytest.mark.parametrize("tag", ["seed_noisy_drop_lstm_bayes3", "seed_noisy_drop_tlm_bayes_ffn"])
def test_two_rank_cli_from_the_seed_alone_follows_the_references_noisy_run(tag, tmp_path):
    """Data parallel AND weight noise AND no saved state: `torchrun ... -m bayeslms_amd.train --seed 1111 --noise-source torch`
    with two ranks (every rank seeds torch's generator the same way, builds the same model and draws the same eps; each trains its
    columns of the global batch) against the reference's single-process `train.py --seed 1111` run of the Bayesian LSTM
    (--L_bayes_pos 3, eight weight draws per forward) WITH --dropout 0.2 (every rank draws the dropout mask of the global batch,
    as the reference's one process would have, and keeps its columns): the same valid / test losses (1e-4), LR-halving epochs
    and final checkpoint (1e-3) as the reference and as the single-process CLI.  Second case: the headline family (Bayesian
    Transformer-FFN) with all its dropout sites on -- the attention-probability mask of the global batch's heads included."""
    z, args, init, snaps = load_traj(tag)
    assert not init and int(args["batch_size"]) % 2 == 0
    d = str(tmp_path)
    write_corpus(z, d)
    base = [str(a) for a in z["argv"]] + ["--data", d, "--cuda", "--noise-source", "torch"]
    runs = {}
    for world, extra in ((1, ()), (2, ("--dist-backend", "gloo"))):
        save = os.path.join(d, "model_w%d.pt" % world)
        hist, out = _cli(world, base + ["--save", save], os.path.join(d, "hist_w%d.json" % world), extra)
        runs[world] = (hist, torch.load(save, map_location="cpu"))
    for world, (hist, final) in runs.items():
        assert list(hist["halved_epochs"]) == list(z["halved_epochs"]), (world, hist["valid_loss"], list(z["valid_loss"]))
        assert np.allclose(hist["valid_loss"], z["valid_loss"], rtol=1e-4), (world, hist["valid_loss"], list(z["valid_loss"]))
        assert abs(hist["test_loss"] - float(z["test_loss"])) <= 1e-4 * float(z["test_loss"]), world
        for k, v in snaps[-1].items():
            assert float((final[k] - v).abs().max()) <= 1e-3 * (float(v.abs().max()) + 1e-12), (world, k)


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_at_the_headline_size_follow_the_references_single_process_run(world, tmp_path):
    """BASELINE.json configs[3] in small: the headline configuration at its REAL size (Bayesian Transformer-FFN 6L d512 ff4096, 33,000
    words, dropout 0.2, global batch 64 x 128), data-parallel over two ranks of 32 (four of 16) columns each, started with `--seed 1111
    --noise-source torch` -- against the reference's SINGLE-process train.py run from the same seed
    (tests/golden/train_headline_from_seed.npz): every rank draws the global batch's eps and masks and keeps its columns / heads,
    gradients meet in the bucketed all-reduce and the compact embedding-row exchange; interval, valid and test losses 1e-4."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "train_headline_from_seed.npz"), allow_pickle=False)
    d = str(tmp_path)
    V = int(z["words_n"])
    with open(os.path.join(d, "words.txt"), "w") as f:
        for i, w in enumerate(["<s>", "<unk>"] + ["w%d" % i for i in range(V - 2)]):
            f.write("%s %d\n" % (w, i))
    for split in ("train", "valid", "test"):
        with open(os.path.join(d, split + ".txt"), "w") as f:
            f.write("\n".join(str(ln) for ln in z[split + "_txt"]) + "\n")
    hist, _ = _cli(world, [str(a) for a in z["argv"]] + ["--data", d, "--cuda", "--noise-source", "torch", "--save", os.path.join(d, "m.pt")],
                   os.path.join(d, "hist.json"), ("--dist-backend", "gloo"))
    assert np.allclose(hist["interval_loss"], z["interval_loss"], rtol=1e-4), (hist["interval_loss"], list(z["interval_loss"]))
    assert np.allclose(hist["valid_loss"], z["valid_loss"], rtol=1e-4), (hist["valid_loss"], list(z["valid_loss"]))
    assert abs(hist["test_loss"] - float(z["test_loss"])) <= 1e-4 * float(z["test_loss"])
