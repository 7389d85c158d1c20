#!/usr/bin/env python3
"""Measures launch plans (block tile x K slices) of the fp32 MFMA GEMM on one MI355X and writes what
csrc/gemm_plan.hip consumes.

  gemm_tune.py --workloads cfg3,cfg2,...   IN SITU: per distinct (layout, M, N, K, epilogue, accumulate) key of a
        workload's step, every candidate plan is installed for THAT key alone (blm_gemm_plan_set; all other launches keep
        the library's current plan, so the candidate runs among its real neighbours), each launch bracketed by HIP events;
        the fastest candidate wins (a key shared by several workloads: smallest summed time).  Then the winners are
        installed together and the step is timed under three planners in the same process: cost model only / the
        built-in table / the fresh winners.  --write-inc rewrites csrc/gemm_plans.inc.
  gemm_tune.py --grid                      STAND-ALONE sweep of a log-spaced M x N x K grid, all tiles x slice counts ->
        gpurun_out/gemm_grid.jsonl (the data the cost model's constants are fitted to, tools/gemm_fit.py) and a
        regression verdict: the planner's choice must reach >= --min-frac of the best candidate's rate on every shape.
"""
import argparse
import ctypes as C
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bayeslms_amd import _lib as L, engine, model as M, ops  # noqa: E402
from bayeslms_amd.data import batchify, get_batch, synthetic_corpus  # noqa: E402

OPN = ("NT", "NN", "TN")
TILES = (11, 12, 21, 22, 28)  # 28 = 128x128 on eight waves
SPLITS = (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20)
TAIL_SPLITS = (-2, -4, -8, -16, -32)  # only the tiles beyond the last whole round of workgroup slots are sliced (in-situ search only)
V33, V10 = 33000, 10000
TRAINER_KW = {}  # tools/gemm_tune_comm.py: collective= stand-in for RCCL's channel workgroups beside the backward GEMMs


def override(tile, splits):
    L.check(L.lib().blm_gemm_plan_override(int(tile), int(splits)), "blm_gemm_plan_override")


def query(op, m, n, k, epi=0, acc=False):
    a = L.GemmArgs()
    a.abi_version = L.ABI_VERSION
    a.op, a.M, a.N, a.K = op, m, n, k
    a.lda = m if op == L.GEMM_TN else k
    a.ldb = k if op == L.GEMM_NT else n
    a.ldc = n
    a.epilogue = epi
    a.flags = L.GEMM_ACCUMULATE if acc else 0
    out = L.GemmPlan()
    L.check(L.lib().blm_gemm_plan_query(C.byref(a), C.byref(out)), "blm_gemm_plan_query")
    return out.tile, out.splits, out.source, out.model_us


# ------------------------------------------------------------------------------------------------ workloads
def _lstm_kl(mm):
    return mm.rnn.kl_divergence()


_lstm_kl.fusable = False


def _ffn_kl(mm):
    return mm.transformerlayers[0].linear2.kl_divergence()


_ffn_kl.fusable = True


def _gauss_kl(mm):
    return mm.transformerlayers[0].gpnn.kl_divergence()


_gauss_kl.fusable = False


_LAST_MODEL = [None]


def _last_model_params():
    """parameters of the model the last build() created (tools/soak.py checks them for non-finite values)"""
    return [p.detach() for p in _LAST_MODEL[0].parameters()] if _LAST_MODEL[0] is not None else []


def build(name, dev):
    """-> step() closure running ONE step of the workload."""
    torch.manual_seed(1111)
    if name in ("cfg3", "recipe_tlm", "gauss", "gauss_sample", "eval_tlm", "eval_tlm100"):
        T, B = {"cfg3": (128, 64), "recipe_tlm": (100, 32), "gauss": (128, 64), "gauss_sample": (128, 64), "eval_tlm": (128, 20),
                "eval_tlm100": (100, 20)}[name]
        if name.startswith("gauss"):
            m, kl = M.GaussTransformerModel(V33, 512, 8, 4096, 6, 0.2, True, 3).to(dev), _gauss_kl
            m.transformerlayers[0].gpnn.sample = name == "gauss_sample"  # train --gp-sample 1: the GP layer re-sampled every step
        else:
            m, kl = M.BayesTransformerModel(V33, 512, 8, 4096, 6, 0.2, True, "FFN").to(dev), _ffn_kl
        is_rnn, Vv, lr = False, V33, 0.1
    elif name in ("cfg2", "recipe_lstm", "eval_lstm", "eval_lstm100"):
        T, B = {"cfg2": (35, 64), "recipe_lstm": (100, 32), "eval_lstm": (35, 20), "eval_lstm100": (100, 20)}[name]
        m, kl = M.BayesRNNModel("LSTM", V33, 1024, 1024, 2, 0.2, True, 3).to(dev), _lstm_kl
        is_rnn, Vv, lr = True, V33, 1.0
    elif name == "cfg1":
        T, B = 35, 20
        m, kl = M.RNNModel("LSTM", V10, 1024, 1024, 2, 0.2, True).to(dev), None
        is_rnn, Vv, lr = True, V10, 1.0
    elif name in ("search_lstm", "search_tlm"):
        # the architecture-search windows of bench.search_leg (Architect.step on a validation window + the network step), one window per step()
        import types
        from bayeslms_amd import model_search_bayes as S, train_search_bayes as TS
        from bayeslms_amd.architect import Architect
        torch.manual_seed(11)
        if name == "search_tlm":
            Ts = 128
            m = S.GaussTransModelSearch(V33, 512, 8, 4096, 6, 0.2, True).to(dev)
            a = types.SimpleNamespace(model="Transformer", T_bayes_pos="FFN", uncertainty="none", L_bayes_pos=0)
        else:
            Ts = 35
            m = S.BayesLSTMModelSearch("LSTM", V33, 1024, 1024, 2, 0.2, True).to(dev)
            a = types.SimpleNamespace(model="LSTM", T_bayes_pos="none", uncertainty="none", L_bayes_pos=1)
        TS.freeze_unused(a, m)
        klf = TS.kl_selector(a)
        arch = Architect(m, V33, types.SimpleNamespace(wdecay=5e-7, clip=1.0, arch_lr=3e-3, arch_wdecay=1e-3))
        trs = engine.Trainer(m, lr=0.1, clip=1.0, kl_scale=Ts / 65536.0, weight_decay=TS.SGD_WEIGHT_DECAY)
        d = torch.randint(0, V33, (Ts + 1, 64), device=dev)
        x, y = d[:Ts], d[1:].reshape(-1)
        sst = {"i": 0, "hidden": m.init_hidden(64) if name == "search_lstm" else None, "hv": m.init_hidden(64) if name == "search_lstm" else None}
        _LAST_MODEL[0] = m

        def sstep():
            s_ = sst["i"]
            sst["i"] += 1
            m.train()
            m.set_step(2 * s_ + 1)
            arch.step(x, y, x, y, None, False, sst["hv"])
            if name == "search_tlm":
                for layer in m.transformerlayers:
                    layer.gpnn.sample = True
            else:
                sst["hidden"] = M.repackage_hidden(sst["hidden"])
            _, _, sst["hidden"] = trs.step(x, y, sst["hidden"], klf, philox_step=2 * s_)
            if name == "search_tlm":
                for layer in m.transformerlayers:
                    layer.gpnn.sample = False
        return sstep, Ts * 64
    else:
        raise SystemExit("unknown workload " + name)
    _LAST_MODEL[0] = m
    nwin = 8
    stream = synthetic_corpus(Vv, B * (nwin * T + 1) + 17, seed=1111)
    data = batchify(stream, B, dev)
    st = {"i": 0, "hidden": m.init_hidden(B) if is_rnn else None}
    if name.startswith("eval"):
        m.eval()

        def step():
            with torch.no_grad():
                d, t = get_batch(data, (st["i"] % nwin) * T, T)
                st["i"] += 1
                if is_rnn:
                    out, h = m(d, st["hidden"])
                    st["hidden"] = M.repackage_hidden(h)
                else:
                    out = m(d)
                ops.cross_entropy(out.view(-1, out.shape[-1]), t)
        return step, T * B
    tr = engine.Trainer(m, lr=lr, clip=1.0, kl_scale=float(T) / data.size(0), seed=1111, **TRAINER_KW)

    def step():
        d, t = get_batch(data, (st["i"] % nwin) * T, T)
        st["i"] += 1
        if is_rnn:
            st["hidden"] = M.repackage_hidden(st["hidden"])
        _, _, st["hidden"] = tr.step(d, t, hidden=st["hidden"], kl_fn=kl)
    return step, T * B


def timed_steps(step, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def per_key_us(step, n):
    """-> {key: [us per launch, launches per step]} of n steps with every GEMM bracketed."""
    timer = ops.KernelTimer(all_gemms=True)
    ops.set_kernel_timer(timer)
    for _ in range(n):
        step()
    ops.set_kernel_timer(None)
    out = {}
    for tag, r in timer.summary().items():
        mt = re.match(r"(\w\w) (\d+)x(\d+)x(\d+) epi(\d+)( acc)?", tag)
        if mt is None:
            continue
        key = (OPN.index(mt.group(1)), int(mt.group(2)), int(mt.group(3)), int(mt.group(4)), int(mt.group(5)), 1 if mt.group(6) else 0)
        tot, cnt = out.get(key, (0.0, 0))
        out[key] = (tot + 1e3 * r["avg_ms"] * r["n"], cnt + r["n"])
    return {k: (tot / cnt, cnt / n) for k, (tot, cnt) in out.items()}


def plan_set(k, t, s):
    L.check(L.lib().blm_gemm_plan_set(k[0], k[1], k[2], k[3], k[4], k[5], int(t), int(s)), "blm_gemm_plan_set")


def tune_workload(name, dev, reps, passes=2, min_share=0.003):
    """Every candidate of ONE key at a time, all other launches on the plan the library has now: a launch's time depends
    on its neighbours in the stream (what they left in the L2s, how their tails overlap its ramp), so a candidate forced
    on every GEMM of the step at once -- rounds 2 and early 3 -- was measured in a context it will never run in (the
    same 128x128 launch: 245 us among its real neighbours, 256 us with 128x128 forced on all of them)."""
    step, tokens = build(name, dev)
    override(0, 0)
    L.check(L.lib().blm_gemm_plan_clear(1), "clear")
    for _ in range(3):
        step()
    base = per_key_us(step, reps)                       # the planner as built
    plans = {k: query(*k[:4], k[4], bool(k[5])) for k in base}
    total = sum(us * n for us, n in base.values())
    cand = {k: {(plans[k][0], plans[k][1]): base[k][0]} for k in base}
    for k in sorted(base, key=lambda k: -base[k][0] * base[k][1]):
        if base[k][0] * base[k][1] < min_share * total:
            continue
        for _ in range(passes):  # the fastest sample counts (clock / neighbour noise is one-sided)
            seen = set()
            for t in TILES:
                for s in SPLITS + TAIL_SPLITS:
                    plan_set(k, t, s)
                    lab = query(*k[:4], k[4], bool(k[5]))[:2]  # what will run: the planner clamps to what is legal
                    if lab in seen:
                        continue
                    seen.add(lab)
                    step()
                    us = per_key_us(step, reps)[k][0]
                    if lab not in cand[k] or us < cand[k][lab]:
                        cand[k][lab] = us
        L.check(L.lib().blm_gemm_plan_clear(1), "clear")
        b = min(cand[k], key=cand[k].get)
        print("   %s %s %dx%dx%d epi%d%s: %d plans, built %d/%d %.1f us, best %d/%d %.1f us" % (
            name, OPN[k[0]], k[1], k[2], k[3], k[4], " acc" if k[5] else "", len(cand[k]), plans[k][0], plans[k][1],
            cand[k][(plans[k][0], plans[k][1])], b[0], b[1], cand[k][b]), flush=True)
    return {"name": name, "tokens": tokens, "base": base, "plans": plans, "cand": cand, "step": step}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="cfg3,gauss,cfg2,recipe_tlm,recipe_lstm,cfg1,eval_tlm,eval_tlm100,eval_lstm,eval_lstm100")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--grid", action="store_true")
    ap.add_argument("--coarse", action="store_true", help="--grid on ~300 shapes (the regression check)")
    ap.add_argument("--grid-out", default=os.path.join(ROOT, "gpurun_out", "gemm_grid.jsonl"))
    ap.add_argument("--grid-cap-gflop", type=float, default=400.0)
    ap.add_argument("--min-frac", type=float, default=0.6)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "gemm_tune.json"))
    ap.add_argument("--write-inc", action="store_true", help="rewrite bayeslms_amd/csrc/gemm_plans.inc from the winners")
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--gain", type=float, default=0.007, help="a table entry must beat the cost model's plan by this fraction")
    ap.add_argument("--small-tile-gain", type=float, default=0.03,
                    help="... by this fraction when it moves to a tile with more co-resident workgroups (more operand re-reads)")
    ap.add_argument("--from-json", default="", help="re-derive the table from stored report(s), comma separated: mean of the runs (no GPU)")
    args = ap.parse_args()
    if args.from_json:
        return write_inc(merge_reports([json.load(open(f)) for f in args.from_json.split(",")]), args)
    dev = torch.device("cuda:0")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    if args.grid:
        return grid(args, dev)
    results = []
    for name in args.workloads.split(","):
        t0 = time.time()
        r = tune_workload(name, dev, args.reps, args.passes)
        results.append(r)
        print("== %s: %d GEMM keys, tuned in %.1f s" % (name, len(r["base"]), time.time() - t0), flush=True)
    # merge: per key the candidate with the smallest time summed over the workloads (weighted by launches per step)
    tot = {}
    for r in results:
        for k, c in r["cand"].items():
            w = r["base"][k][1]
            if len(c) < 2:  # below --min-share of this workload's GEMM time: not searched here
                continue
            for lab, us in c.items():
                tot.setdefault(k, {}).setdefault(lab, 0.0)
                tot[k][lab] += w * us
    winners = {}
    for k, c in tot.items():
        full = {lab: v for lab, v in c.items() if all(lab in r["cand"][k] for r in results if len(r["cand"].get(k, ())) > 1)}
        lab = min(full or c, key=(full or c).get)
        winners[k] = lab
    # model-only plans for the same keys (built-in table and run-time entries off)
    L.check(L.lib().blm_gemm_plan_clear(0), "clear")
    model_plans = {k: query(*k[:4], k[4], bool(k[5]))[:2] for k in winners}
    report = []
    for r in results:
        L.check(L.lib().blm_gemm_plan_clear(0), "clear")
        step = r["step"]
        step()
        ms_model = min(timed_steps(step, 5) for _ in range(2))
        L.check(L.lib().blm_gemm_plan_clear(1), "clear")
        step()
        ms_builtin = min(timed_steps(step, 5) for _ in range(2))
        for k, (t, s) in winners.items():
            if k in r["cand"]:
                L.check(L.lib().blm_gemm_plan_set(k[0], k[1], k[2], k[3], k[4], k[5], t, s), "set")
        step()
        ms_win = min(timed_steps(step, 5) for _ in range(2))
        L.check(L.lib().blm_gemm_plan_clear(1), "clear")
        print("## %-12s step: cost model only %.3f ms | built-in table %.3f ms | fresh winners %.3f ms  (%d tokens)"
              % (r["name"], ms_model, ms_builtin, ms_win, r["tokens"]), flush=True)
        rows = []
        for k, (us0, n) in sorted(r["base"].items(), key=lambda kv: -kv[1][0] * kv[1][1]):
            if k not in winners:
                continue
            c = r["cand"][k]
            best = min(c, key=c.get)
            mp = model_plans[k]
            fl = 2.0 * k[1] * k[2] * k[3]
            rows.append({"key": list(k), "launches_per_step": n, "built_us": us0, "built_plan": list(r["plans"][k][:3]),
                         "winner": list(winners[k]), "winner_us": c.get(winners[k]), "best_here": list(best), "best_us": c[best],
                         "model_plan": list(mp), "model_us_measured": c.get(tuple(mp)),
                         "cands": {"%d/%d" % lab: round(v, 2) for lab, v in sorted(c.items())}})
            print("  %s %5dx%5dx%5d epi%d%s x%4.1f  built %s %7.1f us %5.1f TF | winner %s %7.1f | model %s %s"
                  % (OPN[k[0]], k[1], k[2], k[3], k[4], " acc" if k[5] else "    ", n, "%d/%d" % r["plans"][k][:2], us0,
                     fl / us0 / 1e6, "%d/%d" % winners[k], c.get(winners[k], float("nan")), "%d/%d" % mp,
                     ("%7.1f" % c[tuple(mp)]) if tuple(mp) in c else "   n/a"), flush=True)
        report.append({"workload": r["name"], "ms_model_only": ms_model, "ms_builtin_table": ms_builtin, "ms_winners": ms_win, "rows": rows})
    json.dump(report, open(args.out, "w"), indent=1)
    write_inc(report, args)


def merge_reports(reports):
    """Several stored reports (separate runs / boxes) as one: per workload, key and candidate the MEAN of the runs that
    measured it -- a candidate only some runs could launch (a legality the library gained since) keeps those runs' value."""
    by = {}
    for rep in reports:
        for r in rep:
            w = by.setdefault(r["workload"], {})
            for row in r["rows"]:
                e = w.setdefault(tuple(row["key"]), {"n": row["launches_per_step"], "model_plan": row["model_plan"], "c": {}})
                e["n"], e["model_plan"] = row["launches_per_step"], row["model_plan"]
                for lab, us in row["cands"].items():
                    if int(lab.split("/")[0]) in TILES:  # a report may hold a tile the library no longer has
                        e["c"].setdefault(lab, []).append(us)
    return [{"workload": w, "rows": [{"key": list(k), "launches_per_step": e["n"], "model_plan": e["model_plan"],
                                      "cands": {lab: sum(v) / len(v) for lab, v in e["c"].items()}} for k, e in keys.items()]}
            for w, keys in by.items()]


def write_inc(report, args):
    """Plan-table entries from a tuning report (the JSON this tool writes; --from-json re-derives the table from a stored
    report without a GPU): per key the candidate with the smallest time summed over the workloads (weighted by launches
    per step), listed only where it beats the cost model's own plan by --gain."""
    tot, model = {}, {}
    if args.from_json:  # the stored report may predate the current cost model: ask the library (host-only call) again
        L.check(L.lib().blm_gemm_plan_clear(0), "clear")
    for r in report:
        for row in r["rows"]:
            k = tuple(row["key"])
            model[k] = query(*k[:4], k[4], bool(k[5]))[:2] if args.from_json else tuple(row["model_plan"])
            for lab, us in row["cands"].items():
                t, s_ = lab.split("/")
                tot.setdefault(k, {}).setdefault((int(t), int(s_)), [0.0, 0])
                tot[k][(int(t), int(s_))][0] += row["launches_per_step"] * us
                tot[k][(int(t), int(s_))][1] += 1
    lines = []
    for k in sorted(tot):
        nw = max(n for _, n in tot[k].values())
        c = {lab: v for lab, (v, n) in tot[k].items() if n == nw}  # candidates measured in every workload that has the key
        t, s_ = min(c, key=c.get)
        mp = model[k]
        # a tile with MORE co-resident workgroups per CU than the model's (64x64: 5, 64x128 / 128x64: 3, 128x128: 2) re-reads
        # more of its operands from the memory side (tools/traffic_probe.sh: co-resident workgroups drift apart along K and
        # fetch their shared panels again; 8192 x 512 x 4096: 370 / 273 / 218 MB on 64x64 / 128x64 / 128x128 tiles): it has
        # to pay for those bytes with at least --small-tile-gain
        area = {11: 1, 12: 2, 21: 2, 22: 4, 28: 4}
        need = args.small_tile_gain if area[t] < area[mp[0]] else args.gain
        if mp in c and c[mp] <= c[(t, s_)] * (1.0 + need):
            alt = {lab: v for lab, v in c.items() if area[lab[0]] >= area[mp[0]]}
            t, s_ = min(alt, key=alt.get)
            if c[mp] <= c[(t, s_)] * (1.0 + args.gain):
                continue
        lines.append("    {%d, %d, %d, %d, %d, %d, %d, %d},  // %s: %.1f us per step in situ; cost model's plan %d/%d: %s"
                     % (k[0], k[1], k[2], k[3], k[4], k[5], t, s_, OPN[k[0]], c[(t, s_)], mp[0], mp[1],
                        ("%.1f us" % c[mp]) if mp in c else "not measured"))
    if args.from_json:
        L.check(L.lib().blm_gemm_plan_clear(1), "clear")
    names = ",".join(r["workload"] for r in report)
    inc = ("// Plan table of the fp32 MFMA GEMM: {layout, M, N, K, epilogue, accumulate, tile, K slices}.  GENERATED by\n"
           "// tools/gemm_tune.py --write-inc from IN-SITU measurements on one MI355X (every candidate tile x slice count forced\n"
           "// inside the training / evaluation step of: %s).\n"
           "// Listed: shapes whose measured winner beats the cost model's plan (gemm_plan.hip) by more than %.1f %%, summed over the\n"
           "// workloads that launch them; every other shape -- and every shape not in these workloads -- is planned by the model.\n"
           % (names, 100 * args.gain)) + "\n".join(lines) + ("\n" if lines else "")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    open(os.path.join(os.path.dirname(args.out), "gemm_plans.inc"), "w").write(inc)
    if args.write_inc:
        open(os.path.join(ROOT, "bayeslms_amd", "csrc", "gemm_plans.inc"), "w").write(inc)
    print("%d keys, %d table entries -> %s" % (len(tot), len(lines), os.path.join(os.path.dirname(args.out), "gemm_plans.inc")))


def grid(args, dev):
    dims = [64, 128, 256, 512, 1024, 2048, 3200, 4096, 8192, 16384, 33000]
    ks = [128, 512, 1024, 2048, 4096, 8192, 33000]
    if getattr(args, "coarse", False):  # the regression check of tools/gemm_fuzz.py perf: ~300 shapes, under a minute
        dims, ks = [64, 256, 1024, 3200, 8192, 33000], [512, 2048, 8192]
    cap = 2.9e8
    pool_a = torch.randn(int(cap), device=dev)
    pool_b = torch.randn(int(cap), device=dev)
    pool_c = torch.zeros(int(cap), device=dev)
    worst = (1.0, None)
    worst_big = (1.0, None)  # among shapes whose best candidate takes >= 30 us (below that the launch overhead is the time)
    nshape = 0
    f = open(args.grid_out, "w")
    t_start = time.time()
    for op in (L.GEMM_NT, L.GEMM_NN, L.GEMM_TN):
        for m in dims:
            for n in dims:
                for k in ks:
                    if 2.0 * m * n * k > args.grid_cap_gflop * 1e9 or m * n > cap or m * k > cap or n * k > cap:
                        continue
                    acc = op == L.GEMM_TN
                    if op == L.GEMM_NT:
                        A, B, lda, ldb = pool_a[:m * k].view(m, k), pool_b[:n * k].view(n, k), k, k
                    elif op == L.GEMM_NN:
                        A, B, lda, ldb = pool_a[:m * k].view(m, k), pool_b[:k * n].view(k, n), k, n
                    else:
                        A, B, lda, ldb = pool_a[:k * m].view(k, m), pool_b[:k * n].view(k, n), m, n
                    Cm = pool_c[:m * n].view(m, n)
                    override(0, 0)
                    L.check(L.lib().blm_gemm_plan_clear(0), "clear")
                    ct, cs, _, _ = query(op, m, n, k, 0, acc)
                    res = {}
                    for t in TILES:
                        for s in SPLITS:
                            if s > 1 and k // s < 128:
                                continue
                            override(t, s)
                            for _ in range(2):
                                ops.gemm(op, A, B, Cm, m, n, k, lda, ldb, n, accumulate=acc)
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            reps = 4
                            e0.record()
                            for _ in range(reps):
                                ops.gemm(op, A, B, Cm, m, n, k, lda, ldb, n, accumulate=acc)
                            e1.record()
                            e1.synchronize()
                            res[(t, s)] = 1e3 * e0.elapsed_time(e1) / reps
                    if acc:
                        pool_c[:m * n].zero_()
                    if (ct, cs) not in res:  # a plan outside the candidate set: measure it too
                        override(ct, cs)
                        for _ in range(2):
                            ops.gemm(op, A, B, Cm, m, n, k, lda, ldb, n, accumulate=acc)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(4):
                            ops.gemm(op, A, B, Cm, m, n, k, lda, ldb, n, accumulate=acc)
                        e1.record()
                        e1.synchronize()
                        res[(ct, cs)] = 1e3 * e0.elapsed_time(e1) / 4
                    best = min(res, key=res.get)
                    chosen = res.get((ct, cs))
                    frac = res[best] / chosen if chosen else 0.0
                    nshape += 1
                    if frac < worst[0]:
                        worst = (frac, (OPN[op], m, n, k, (ct, cs), best))
                    if res[best] >= 30.0 and frac < worst_big[0]:
                        worst_big = (frac, (OPN[op], m, n, k, (ct, cs), best))
                    f.write(json.dumps({"op": op, "M": m, "N": n, "K": k, "acc": int(acc), "chosen": [ct, cs],
                                        "us": {"%d/%d" % kk: round(v, 2) for kk, v in res.items()}}) + "\n")
                    f.flush()
                    if nshape % 50 == 0:
                        print("grid: %d shapes, %.0f s, worst chosen/best so far %.3f %s" % (nshape, time.time() - t_start, worst[0], worst[1]), flush=True)
    override(0, 0)
    L.check(L.lib().blm_gemm_plan_clear(1), "clear")
    f.close()
    print("grid: %d shapes; the planner's choice reaches >= %.3f of the best candidate's rate on every shape (worst: %s), >= %.3f on "
          "every shape of 30 us and more (worst: %s)" % (nshape, worst[0], worst[1], worst_big[0], worst_big[1]))
    if worst_big[0] < args.min_frac:
        print("PERF REGRESSION: a shape runs below %.2f of its best tile / slice count" % args.min_frac)
        sys.exit(1)


if __name__ == "__main__":
    main()
