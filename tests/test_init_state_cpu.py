"""CPU: the same torch seed gives the same INITIAL model as the reference, bit for bit.

train.py seeds torch's generator and then builds the model (train.py:122, :186-221); a user who moves a recipe here
with its seed should start from the very weights the reference would have started from, in the same state_dict key
order, with the generator left at the same point of its stream (whatever draws next -- a second model, a shuffled
batch order -- continues identically).  That pins down the ORDER of every draw a constructor makes, including the
ones the reference throws away (GPNN / Bayes ``sample_parameters()`` at the end of their constructors,
model.py:1812, model_search_bayes.py:813; the ``torch.rand`` births of Bayes2LSTM's log-sigmas, model.py:615-633)
and nn.TransformerEncoder's habit of cloning ONE constructed layer (model.py:134-136).

Fixture: tests/golden/init_state.json, written by ``make_golden.py init`` from the reference's own constructors
(56 of them, plus four model builds through train.py's own dispatch, which constructs the model twice for
``--uncertainty none``: every family of train.py and train_search_bayes.py, tied and untied, every position / type string the
parity fixtures use): per tensor its shape, the SHA-256 of its bytes and its float64 sum.  Constructors only create
parameters -- no kernel is launched, so this runs without a GPU."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from bayeslms_amd import model as M
from bayeslms_amd import model_search_bayes as S

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "init_state.json")) as f:
    GOLD = json.load(f)
MODS = {"model": M, "model_search_bayes": S}


def digest(t):
    a = np.ascontiguousarray(t.detach().cpu().numpy())
    return [list(a.shape), hashlib.sha256(a.tobytes()).hexdigest(), float(a.astype(np.float64).sum())]


def test_fixture_covers_every_constructor_the_entry_points_reach():
    built = {(c["module"], c["cls"]) for c in GOLD["cases"].values()}
    for cls in ("RNNModel", "TransformerModel", "BayesRNNModel", "BayesTransformerModel", "GaussRNNModel",
                "GaussTransformerModel", "VariationalRNNModel", "VTransformerModel"):  # train.py:186-221
        assert ("model", cls) in built, cls
    for cls in ("BayesTransModelSearch", "GaussTransModelSearch", "BayesLSTMModelSearch"):  # train_search_bayes.py:158-163
        assert ("model_search_bayes", cls) in built, cls
    assert len(GOLD["cases"]) >= 50


@pytest.mark.parametrize("name", sorted(GOLD["cases"]))
def test_same_seed_same_initial_state_as_the_reference(name):
    case = GOLD["cases"][name]
    torch.manual_seed(GOLD["seed"])
    if "cli" in case:  # through the CLI's dispatch: `--uncertainty none` builds the model twice and keeps the second (train.py:196-199)
        from types import SimpleNamespace
        from bayeslms_amd import train as TR
        m = TR.build_model(SimpleNamespace(**case["cli"]), GOLD["ntokens"])
        assert type(m).__name__ == case["cls"]
    else:
        m = getattr(MODS[case["module"]], case["cls"])(*case["args"])
    after = torch.rand(4, dtype=torch.float64).tolist()
    sd = m.state_dict()
    assert list(sd.keys()) == [row[0] for row in case["state"]], "state_dict keys or their order"
    for key, shape, sha, total in case["state"]:
        got = digest(sd[key])
        assert got[0] == shape, (key, got[0], shape)
        assert got[1] == sha, "%s: sum %.9g here, %.9g in the reference" % (key, got[2], total)
    if "arch" in case:
        arch = m.arch_parameters()
        assert len(arch) == len(case["arch"])
        for a, (shape, sha, total) in zip(arch, case["arch"]):
            got = digest(a)
            assert got[0] == shape and got[1] == sha, ("architecture logits", got[2], total)
    assert after == case["generator_after"], "torch's generator left the constructor at another point of its stream"


def test_baseline_transformer_layers_start_as_copies_of_layer_0():
    """nn.TransformerEncoder deep-copies the layer it is given: independent storage, equal values."""
    torch.manual_seed(3)
    m = M.TransformerModel(50, 16, 4, 32, 3, 0.2, "gelu", True)
    l0, l2 = m.transformerlayers.layers[0], m.transformerlayers.layers[2]
    for (k, a), (_, b) in zip(l0.named_parameters(), l2.named_parameters()):
        assert torch.equal(a, b) and a.data_ptr() != b.data_ptr(), k
    assert len({id(p) for p in m.parameters()}) == len(list(m.parameters()))
    assert l0._site_base != l2._site_base  # each copy still has its own dropout / noise stream ids


def test_noise_source_torch_draws_eps_as_the_reference_does(monkeypatch):
    """``set_noise_source("torch")`` (train --noise-source torch): the eps of a variational tensor is ONE
    ``zeros(shape).normal_()`` from torch's CPU generator at the moment the forward asks for it -- the reference's own call
    (model.py:1087, :671) -- so the value equals what the reference would have drawn from the same generator state; the default
    stays the Philox stream (no eps tensor on the host).  No kernel is launched here."""
    from bayeslms_amd import train as TR
    monkeypatch.setenv("BLM_NOISE_SOURCE", "torch")  # the switch an unchanged reference script is given from outside
    assert M.RNNModel("LSTM", 50, 12, 12, 2, 0.0, True).noise_state.source == "torch"
    monkeypatch.setenv("BLM_NOISE_SOURCE", "numpy")
    with pytest.raises(ValueError):
        M.RNNModel("LSTM", 50, 12, 12, 2, 0.0, True)
    monkeypatch.delenv("BLM_NOISE_SOURCE")
    torch.manual_seed(5)
    m = M.BayesTransformerModel(50, 16, 4, 32, 2, 0.0, True, "FFN")
    lin2 = m.transformerlayers[0].linear2
    m.train()
    assert m.noise_state.source == "philox" and lin2.noise().eps is None
    with pytest.raises(ValueError):
        m.set_noise_source("numpy")
    m.set_noise_source("torch")
    torch.manual_seed(9)
    want = torch.zeros(*lin2.weight_lgstd.shape).normal_()
    after = torch.rand(2)
    torch.manual_seed(9)
    got = lin2.noise().eps
    assert torch.equal(got, want) and torch.equal(torch.rand(2), after)  # same values, generator advanced by exactly that draw
    m.eval()
    assert lin2.noise() is None  # mean weights: nothing drawn
    # one (1, H) row of N(0, 0.1) per time step for the variational cells
    v = M.VariationalRNNModel("LSTM", 50, 12, 12, 2, 0.0, True, "11")
    v.set_noise_source("torch")
    torch.manual_seed(3)
    want = torch.cat([torch.zeros(1, 12).normal_(0, 0.1) for _ in range(4)], 0) * torch.exp(v.rnn.rnn[0].vnn.hidden_lgstd)
    torch.manual_seed(3)
    assert torch.equal(v.rnn.rnn[0].vnn.noise_rows(4), want)
    args = TR.build_parser().parse_args(["--noise-source", "torch"])
    assert args.noise_source == "torch" and TR.build_parser().parse_args([]).noise_source is None  # None: BLM_NOISE_SOURCE or philox
