"""CPU: the oracle's restatement of the reference training loop (oracle/train_oracle.py) against full-precision
trajectories of the reference's OWN train.py (tests/golden/train_traj_*.npz): per-step loss,
per-interval log means, valid loss of every epoch, which epochs halved the LR, parameters after every epoch,
test loss."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

TAGS = ["lstm_none", "tlm_none", "lstm_bayes5", "tlm_gauss3", "lstm_gauss33", "lstm_var00"]
# the same script started with `--seed 1111` ALONE (no saved initial state, make_golden.py traj_seed): the run begins from what
# train.py's own model construction draws under that seed
SEED_TAGS = ["seed_lstm_none", "seed_tlm_gauss3", "seed_lstm_gauss33"]
# ... and WITH weight noise (dropout 0): the Bayesian / Variational families sample every training step, eps from torch's CPU
# generator -- seeded by train.py, advanced by the constructors, then one ``new_zeros(shape).normal_()`` per tensor and step
# (--T_bayes_pos FFN / MHA are out of reach: their layer 0 has a hard-coded dropout of 0.2, model.py:1202,1207 -- masks from the
# same generator, which no fused kernel reproduces)
NOISY_SEED_TAGS = ["seed_noisy_tlm_bayes_emb", "seed_noisy_lstm_bayes3", "seed_noisy_lstm_var11",
                   # the headline family and its MHA sibling, recorded with the HARNESS building every nn.Dropout of the reference run
                   # with p = 0 (fixture field zero_dropout; make_golden.py PROBE_ZERO_DROPOUT): the weight noise alone
                   "seed_noisy_tlm_bayes_ffn_nodrop", "seed_noisy_tlm_bayes_mha_nodrop"]
# ... and with --dropout 0.2 on top: the LSTM language models' three dropout sites (embedding, nn.LSTM's inter-layer, output) draw
# their masks through torch's CPU dropout from the same generator; the Gaussian cell's per-forward sample_parameters() draws
# (dropped, train.py never raises GPNN.sample) move it as well
DROP_SEED_TAGS = ["seed_noisy_drop_lstm_none", "seed_noisy_drop_lstm_bayes3", "seed_noisy_drop_lstm_gauss33", "seed_noisy_drop_lstm_var11",
                  # the Transformers: positional-encoding, attention-probability (B * h, T, T), dropout1, feed-forward and dropout2 masks,
                  # layer 0's hard-coded 0.2 of --T_bayes_pos FFN / MHA included
                  "seed_noisy_drop_tlm_none", "seed_noisy_drop_tlm_bayes_ffn", "seed_noisy_drop_tlm_bayes_mha", "seed_noisy_drop_tlm_bayes_emb",
                  "seed_noisy_drop_tlm_gauss3",
                  # GPNN2 random features: fresh frequencies at every call -- once per forward in the Transformer layer, once per
                  # time step inside the GP-LSTM cell (gate 3: on the cell gate; gate 7: the input projection)
                  "seed_noisy_drop_tlm_gauss4", "seed_noisy_drop_lstm_gauss34", "seed_noisy_drop_lstm_gauss74"]


def load_traj(tag):
    z = np.load(os.path.join(GOLDEN, "train_traj_%s.npz" % tag), allow_pickle=False)
    argv = [str(a) for a in z["argv"]]
    args = {}
    i = 0
    while i < len(argv):
        k = argv[i].lstrip("-").replace("-", "_")
        if i + 1 < len(argv) and not argv[i + 1].startswith("--"):
            args[k] = argv[i + 1]
            i += 2
        else:
            args[k] = True
            i += 1
    init = {k[5:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith("init/")}
    nsnap = 1 + max(int(k[4:].split("/")[0]) for k in z.files if k.startswith("snap"))
    snaps = [{k.split("/", 1)[1]: torch.from_numpy(z[k]) for k in z.files if k.startswith("snap%d/" % s)}
             for s in range(nsnap)]
    return z, args, init, snaps


def write_corpus(z, d):
    with open(os.path.join(d, "words.txt"), "w") as f:
        for i, w in enumerate(z["words"]):
            f.write("%s %d\n" % (w, i))
    for split in ("train", "valid", "test"):
        with open(os.path.join(d, split + ".txt"), "w") as f:
            f.write(str(z[split + "_txt"]))


def cli_namespace(z):
    """A fixture's argv as bayeslms_amd.train's parser sees it (defaults filled in)."""
    from bayeslms_amd import train as T
    return T.build_parser().parse_args([str(a) for a in z["argv"]])


def noisy_family(args):
    """The oracle's forward for the runs WITH weight noise: in a training step (grad mode; evaluate() runs under no_grad) the eps of
    every variational tensor is drawn from torch's CPU generator with the reference's own call, in its order -- one draw for the
    Bayesian embedding projection (model.py:1243-1248), eight per Bayes2LSTM forward (:668-703), one (1, H) row of N(0, 0.1) per
    time step and noisy cell, cell 0's T rows before cell 1's (:2555-2561, :2503-2507)."""
    from oracle import bayes_oracle as O
    if args["uncertainty"] == "none":  # the plain LSTM with its three dropout sites (masks from torch's generator, as the reference's)
        p = float(args["dropout"])
        return (lambda sd, x, h: O.rnn_lm_train(x, h, sd, p) if torch.is_grad_enabled() else O.rnn_lm(x, h, sd)), None, True
    if args["model"] == "Transformer":
        pos, nhead = args["T_bayes_pos"], int(args["nhead"])
        lg = {"EMB": "embed_lgstd", "FFN": "transformerlayers.0.linear2.weight_lgstd", "MHA": "transformerlayers.0.self_attn.o_net.weight_lgstd"}[pos]

        def fwd(sd, x, hidden):
            eps = torch.zeros(*sd[lg].shape).normal_() if torch.is_grad_enabled() else None
            return O.transformer_lm(x, sd, nhead, eps), None
        return fwd, (lambda sd: O.kl_transformer(sd, pos)), False
    if args["uncertainty"] == "Bayesian":
        pos = int(args["L_bayes_pos"])

        def fwd(sd, x, hidden):
            eps8 = [torch.zeros(*sd["rnn." + k].shape).normal_() for k in O.LSTM_EPS_ORDER] if torch.is_grad_enabled() else None
            return O.bayes_rnn_lm(x, hidden, sd, pos, eps8)
        return fwd, (lambda sd: O.kl_bayes2lstm(sd, "rnn.", pos)), True
    v, held = args["L_v_pos"], {}

    def fwd(sd, x, hidden):
        eps = None
        if torch.is_grad_enabled():
            H = sd["rnn.rnn.0.vnn.hidden_lgstd"].shape[1]
            eps = {c: torch.cat([torch.zeros(1, H).normal_(0, 0.1) for _ in range(x.shape[0])], 0) for c in (0, 1) if int(v[c]) == 1}
        logits, hidden, held["kl"] = O.variational_rnn_lm(x, hidden, sd, v, eps)
        return logits, hidden
    return fwd, (lambda sd: held["kl"]), True  # train.py:379-382: the KL of the forward that has just run


@pytest.mark.parametrize("tag", TAGS + SEED_TAGS + NOISY_SEED_TAGS + DROP_SEED_TAGS[:1])
def test_oracle_training_loop_matches_reference_train_py(tag, tmp_path):
    from bayeslms_amd import data as D
    from oracle import bayes_oracle as O, train_oracle as TO
    z, args, init, snaps = load_traj(tag)
    if tag.startswith("seed_"):
        # no saved state: OUR constructors under train.py's seed, through the CLI's model dispatch (which repeats the
        # reference's throw-away first construction for --uncertainty none), are where the reference's run started
        from bayeslms_amd import train as T
        assert not init and args["seed"] == "1111"
        torch.manual_seed(int(args["seed"]))
        init = {k: v.detach().clone() for k, v in T.build_model(cli_namespace(z), len(z["words"])).state_dict().items()
                if not k.endswith("pos_encoder.pe")}
    write_corpus(z, str(tmp_path))
    corpus = D.Corpus(str(tmp_path))
    bsz, seq_len = int(args["batch_size"]), int(args["seq_len"])
    train = TO.batchify(corpus.train, bsz)
    assert len(train) == int(z["rows"])
    valid, test = TO.batchify(corpus.valid, 20), TO.batchify(corpus.test, 20)  # eval_batch_size, train.py:182
    sd = dict(init)
    sd["decoder.weight"] = sd["encoder.weight"]  # --tied
    if args["model"] == "Transformer":
        sd["pos_encoder.pe"] = O.positional_table(5000, int(args["emsize"]))
    fwd, kl, is_rnn = noisy_family(args) if "noisy" in tag else TO.family(args)
    torch.set_num_threads(1)
    r = TO.train_run(sd, fwd, kl, is_rnn, train, valid, test, seq_len=seq_len, lr=float(args["lr"]),
                     clip=float(args["clip"]), epochs=int(args["epochs"]), log_interval=int(args["log_interval"]),
                     nlayers=int(args["nlayers"]), nhid=int(args["nhid"]))
    ref_step = z["step_loss"]
    got = np.array(r["step_loss"])
    assert got.shape == ref_step.shape
    assert np.abs(got[:20] - ref_step[:20]).max() <= 2e-5 * np.abs(ref_step[:20]).max()
    assert np.abs(got - ref_step).max() <= 1e-3 * np.abs(ref_step).max()
    assert np.allclose(r["interval_loss"], z["interval_loss"], rtol=1e-3)
    assert list(r["halved_epochs"]) == list(z["halved_epochs"])
    assert np.allclose(r["sgd_lr"], z["sgd_lr"]) and r["final_lr"] == float(z["final_lr"])
    assert np.allclose(r["valid_loss"], z["valid_loss"], rtol=1e-4)
    assert abs(r["test_loss"] - float(z["test_loss"])) <= 1e-4 * float(z["test_loss"])
    assert len(r["snapshots"]) == len(snaps)
    for mine, ref in zip(r["snapshots"], snaps):
        for k, v in mine.items():
            scale = float(ref[k].abs().max()) + 1e-12
            assert float((v - ref[k]).abs().max()) <= 1e-3 * scale, k
