#!/usr/bin/env python3
"""Train and evaluate a neural LM -- same command line as the reference's
steps/pytorchnn/train.py (argument names, types, defaults: train.py:28-103; log line formats
:426-430,476-478,544-545; best-checkpoint / LR-halving / early-stop loop :464-519), running on the
HIP engine.  New optional behaviour: launched under ``torchrun`` it trains data-parallel, each rank
owning a slice of the global batch columns (``--batch-size`` stays the GLOBAL batch).

    python -m bayeslms_amd.train --data DIR --model Transformer --emsize 512 --nhid 4096 --nlayers 6 \
        --nhead 8 --uncertainty Bayesian --T_bayes_pos FFN --tied --cuda --save model.pt
"""
import argparse
import math
import os
import random
import time

import torch
import torch.distributed as dist


def build_parser():
    p = argparse.ArgumentParser(description="Train and evaluate a neural language model (MI355X engine).")
    p.add_argument('--data', type=str, default='./data/pytorchnn', help='location of the data corpus')
    p.add_argument('--model', type=str, default='LSTM', help='LSTM or Transformer')
    p.add_argument('--emsize', type=int, default=200)
    p.add_argument('--nhid', type=int, default=200)
    p.add_argument('--nlayers', type=int, default=2)
    p.add_argument('--nhead', type=int, default=2)
    p.add_argument('--uncertainty', type=str, default='none', help='[none | Bayesian | Gaussian | Variational]')
    p.add_argument('--T_bayes_pos', type=str, default='none', help='[none | FFN | MHA | EMB]')
    p.add_argument('--L_bayes_pos', type=int, default=0, help='0 none, 1 input, 2 forget, 3 cell, 4 output gate')
    p.add_argument('--L_gauss_pos', type=str, default='00')
    p.add_argument('--L_v_pos', type=str, default='11')
    p.add_argument('--T_gauss_pos', type=int, default=3)
    p.add_argument('--T_v_pos', type=int, default=0)
    p.add_argument('--mark', type=str, default='none')
    p.add_argument('--lr', type=float, default=0.1)
    p.add_argument('--batch-size', type=int, default=20, metavar='N')
    p.add_argument('--epochs', type=int, default=20)
    p.add_argument('--seq_len', type=int, default=35)
    p.add_argument('--clip', type=float, default=0.25)
    p.add_argument('--dropout', type=float, default=0.2)
    p.add_argument('--tied', action='store_true')
    p.add_argument('--optimizer', type=str, default='SGD')
    p.add_argument('--log-interval', type=int, default=200, metavar='N')
    p.add_argument('--cuda', action='store_true', help='required: the engine has no CPU path')
    p.add_argument('--save', type=str, default='model.pt')
    p.add_argument('--seed', type=int, default=1111)
    p.add_argument('--resume', type=str, default='')
    p.add_argument('--debug', action='store_true')
    p.add_argument('--work_dir', default='TFM', type=str)
    p.add_argument('--prior', default="False", type=str)
    p.add_argument('--prior_path', default='steps/pytorchnn/prior', type=str)
    p.add_argument('--prior2_path', default='steps/pytorchnn/prior/transformer2/', type=str)
    # new, optional
    p.add_argument('--fused-sampling', type=int, default=0, help='1: eps generated inside the GEMM tile loader')
    p.add_argument('--gp-sample', type=int, default=0,
                   help='new, optional: 1 raises GPNN.sample (reference model.py:1799 leaves it False and train.py never sets '
                        'it): the GP coefficients / weights of --uncertainty Gaussian are re-sampled every training step')
    p.add_argument('--gemm-mode', type=str, default='f32', choices=['f32', 'bf16x6', 'bf16x3'],
                   help='new, optional: opt-in split-bf16 arithmetic of the GEMM family (DESIGN.md section 7); default fp32 MFMA')
    p.add_argument('--noise-source', type=str, default=None, choices=['philox', 'torch'],
                   help='new, optional: where the noise of a training run comes from (default: BLM_NOISE_SOURCE or philox).  philox: counter-based '
                        'streams keyed by (seed, tensor / site, step), generated inside the kernels.  torch: a parity mode -- every draw the '
                        'reference makes from torch\'s CPU generator (variational eps, GP draws, every dropout mask) is made by the same call in '
                        'the same order and uploaded, the blocks run unfused: a run from the same --seed follows the reference\'s CPU run, '
                        'noise and dropout on')
    p.add_argument('--deterministic', type=int, default=0,
                   help='new, optional: 1 = every reduction of the engine in a fixed order (ops.set_deterministic / BLM_DETERMINISTIC=1): '
                        'two runs from one seed give bit-identical losses and parameters, also under torchrun at the same world size; '
                        'costs about a quarter of the step')
    p.add_argument('--dist-backend', type=str, default='nccl',
                   help='new, optional (under torchrun): torch.distributed backend; nccl = RCCL over xGMI, one GPU per rank; '
                        'gloo lets several ranks rehearse on one GPU')
    p.add_argument('--dist-timeout-s', type=float, default=None,
                   help='new, optional (under torchrun): rendezvous and collective timeout in seconds (default 180, BLM_DIST_TIMEOUT_S): '
                        'a rank that never arrives, or a collective one rank never enters, ends the job instead of hanging it')
    p.add_argument('--dp-overlap', type=int, default=1,
                   help='new, optional: 1 = bucket all-reduces start while backward is still running, 0 = all after backward')
    p.add_argument('--dp-late-rows', type=int, default=1,
                   help='new, optional: 1 = the embedding half of the encoder gradient travels as the compact matrix of the '
                        'rows the global batch touched (engine.LateRows), 0 = dense')
    p.add_argument('--history', type=str, default='',
                   help='new, optional: rank 0 writes what the log lines print with two decimals (interval / valid / test '
                        'loss, LR-halving epochs, ms per batch) at full precision to this JSON file')
    return p


def build_model(args, ntokens):
    """Model dispatch of train.py:193-223.  With ``--uncertainty none`` the reference builds the model TWICE and trains the
    second one (``model_2`` first, train.py:196-199 / :211-214): the first construction is repeated here and dropped, so that
    the same ``--seed`` starts from the same weights (the constructors draw from torch's generator in the reference's order,
    tests/test_init_state_cpu.py)."""
    from . import model as M
    if args.model == 'Transformer':
        if args.uncertainty == 'none':
            M.TransformerModel(ntokens, args.emsize, args.nhead, args.nhid, args.nlayers, args.dropout, "gelu", args.tied)  # model_2
            return M.TransformerModel(ntokens, args.emsize, args.nhead, args.nhid, args.nlayers, args.dropout, "gelu", args.tied)
        if args.uncertainty == 'Bayesian':
            return M.BayesTransformerModel(ntokens, args.emsize, args.nhead, args.nhid, args.nlayers, args.dropout,
                                           args.tied, args.T_bayes_pos)
        if args.uncertainty == 'Gaussian':
            return M.GaussTransformerModel(ntokens, args.emsize, args.nhead, args.nhid, args.nlayers, args.dropout,
                                           args.tied, args.T_gauss_pos)
        if args.uncertainty == 'Variational':
            return M.VTransformerModel(ntokens, args.emsize, args.nhead, args.nhid, args.nlayers, args.dropout,
                                       args.tied, args.T_v_pos)
    else:
        if args.uncertainty == 'none':
            M.RNNModel(args.model, ntokens, args.emsize, args.nhid, args.nlayers, args.dropout, args.tied)  # model_2
            return M.RNNModel(args.model, ntokens, args.emsize, args.nhid, args.nlayers, args.dropout, args.tied)
        if args.uncertainty == 'Bayesian':
            return M.BayesRNNModel(args.model, ntokens, args.emsize, args.nhid, args.nlayers, args.dropout, args.tied,
                                   args.L_bayes_pos)
        if args.uncertainty == 'Gaussian':
            return M.GaussRNNModel(args.model, ntokens, args.emsize, args.nhid, args.nlayers, args.dropout, args.tied,
                                   args.L_gauss_pos)
        if args.uncertainty == 'Variational':
            return M.VariationalRNNModel(args.model, ntokens, args.emsize, args.nhid, args.nlayers, args.dropout,
                                         args.tied, args.L_v_pos)
    raise SystemExit("--model %s --uncertainty %s is not built by this engine yet" % (args.model, args.uncertainty))


def kl_selector(args):
    """Which module's KL train.py adds to the loss for this flag combination (train.py:335-399).
    Returns None or fn(model) -> KL tensor; ``fn.fusable`` marks KLs whose gradient the Bayesian
    wgrad epilogue can add itself."""
    fn = None
    if args.uncertainty == 'Bayesian':
        if args.model == 'LSTM' and 1 <= args.L_bayes_pos <= 5:  # train.py:337
            fn = lambda m: m.rnn.kl_divergence()  # noqa: E731
            fn.fusable = False
        elif args.model == 'Transformer':
            if args.T_bayes_pos == 'FFN':
                fn = lambda m: m.transformerlayers[0].linear2.kl_divergence()  # noqa: E731
                fn.fusable = True
            elif args.T_bayes_pos == 'MHA':
                fn = lambda m: m.transformerlayers[0].self_attn.o_net.kl_divergence()  # noqa: E731
                fn.fusable = True
            elif args.T_bayes_pos == 'EMB':
                fn = lambda m: m.embed_kl_divergence()  # noqa: E731
                fn.fusable = False
    elif args.uncertainty == 'Gaussian' and args.model == 'Transformer' and 1 <= args.T_gauss_pos <= 3:
        fn = lambda m: m.transformerlayers[0].gpnn.kl_divergence()  # noqa: E731
        fn.fusable = False
    elif args.uncertainty == 'Gaussian' and args.model == 'LSTM':
        g = args.L_gauss_pos
        if int(g[0]) > 0 and 0 < int(g[1]) <= 3:  # train.py:367-375
            cells = [0] if len(g) < 3 else ([1] if len(g) == 3 else [0, 1])
            fn = lambda m: sum(m.rnn.rnn[c].gpnn.kl_divergence() for c in cells)  # noqa: E731
            fn.fusable = False
    elif args.uncertainty == 'Variational' and args.model == 'LSTM':
        cells = [c for c in (0, 1) if int(args.L_v_pos[c]) == 1]  # train.py:379-382
        if cells:
            fn = lambda m: sum(m.rnn.rnn[c].vnn.kl_divergence() for c in cells)  # noqa: E731
            fn.fusable = False
    elif args.uncertainty == 'Variational' and args.model == 'Transformer' and int(args.T_v_pos) in (1, 2, 3):
        layers = {1: [0], 2: [1], 3: [0, 1]}[int(args.T_v_pos)]  # train.py:387-396 (raises like the reference)
        fn = lambda m: sum(m.transformerlayers[i].kl_divergence() for i in layers)  # noqa: E731
        fn.fusable = False
    return fn


def _print_coef_mean(args, model, say):
    """train.py:483-494 / :529-540: the per-epoch print of the GP activation-mixture coefficients, averaged
    over units (one value per basis activation)."""
    sd = model.state_dict()
    if args.model == 'Transformer' and args.uncertainty == 'Gaussian' and args.T_gauss_pos <= 3:
        say(sd['transformerlayers.0.gpnn.coef_mean'].mean(dim=1))
    elif args.model == 'LSTM' and args.uncertainty == 'Gaussian' and int(args.L_gauss_pos[1]) <= 3:
        g = args.L_gauss_pos
        cells = [0] if len(g) < 3 else ([1] if len(g) == 3 else [0, 1])
        for c in cells:
            say(sd['rnn.rnn.%d.gpnn.coef_mean' % c].mean(dim=1))


class ValidationSchedule:
    """The end-of-epoch decision of train.py:496-512 -- keep the checkpoint when the validation loss improved, else
    halve the learning rate (fresh SGD, reload the best checkpoint), stop after 8 halvings -- taken ONCE for the job.
    Every rank evaluates the same validation stream, but an under-filled evaluation GEMM sums its K slices with float
    atomics, so two ranks may see losses that differ in the last bits; a rank-local ``val_loss < best_val`` could then
    halve the LR on one rank only, or leave the ranks in different epochs with the next all-reduce hanging.  ``update``
    therefore replaces every rank's value by rank 0's (one 8-byte broadcast per epoch) before it is compared, logged
    or stored, so all ranks walk the same branch by construction."""

    def __init__(self, lr, world=1, device=None, group=None, patience=8):
        self.lr, self.world, self.device, self.group = lr, world, device, group
        self.best_val = None
        self.counter = 0
        self.patience = patience

    def agree(self, value):
        """rank 0's ``value`` on every rank (identity for a single process)."""
        if self.world <= 1:
            return float(value)
        on_dev = dist.get_backend(self.group) == "nccl"
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if on_dev else "cpu")
        dist.broadcast(t, src=0, group=self.group)
        return float(t.item())

    def update(self, val_loss):
        """-> (the job's validation loss, improved, stop).  ``improved`` False means: lr has been halved."""
        val_loss = self.agree(val_loss)
        improved = (not self.best_val) or val_loss < self.best_val  # train.py:497
        if improved:
            self.best_val = val_loss
        else:
            self.lr /= 2.
            self.counter += 1
        return val_loss, improved, self.counter == self.patience


def main(argv=None, history=None):
    """``history`` (optional dict) receives what the log lines print with two decimals at full precision:
    interval_loss, valid_loss, halved_epochs, test_loss."""
    args = build_parser().parse_args(argv)
    if history is None:
        history = {}
    history.update({"interval_loss": [], "valid_loss": [], "halved_epochs": [], "test_loss": None, "ms_per_batch": []})
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    is_main = rank == 0
    if world > 1:  # before the first torch.cuda call: the HSA runtime reads its environment when it is initialised
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    def say(*a):
        if is_main:
            print(*a, flush=True)

    random.seed(args.seed)
    torch.manual_seed(args.seed)
    if not args.cuda or not torch.cuda.is_available():
        raise SystemExit("bayeslms_amd.train needs --cuda and an MI355X: there is no CPU path")
    if args.dist_backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)  # rehearsal: several ranks share a GPU (RCCL refuses that)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        # bounded rendezvous / collective timeout (--dist-timeout-s, BLM_DIST_TIMEOUT_S, 180 s), RCCL's channel count pinned before
        # RCCL reads its environment, `rendezvous ok` / `first all-reduce ok` heartbeats on stderr (engine.init_distributed)
        from .engine import init_distributed
        rccl_env = init_distributed(args.dist_backend, device if args.dist_backend == "nccl" else None, args.dist_timeout_s)
        if rccl_env is not None:
            say("RCCL channels:", rccl_env)

    from . import data as D, engine, ops
    from .model import repackage_hidden
    if args.gemm_mode != 'f32':
        ops.set_gemm_mode(args.gemm_mode)
    if args.deterministic:
        ops.set_deterministic(True)

    say('Configurations')
    for k, v in vars(args).items():
        say(k, v)
    corpus = D.Corpus(args.data)
    say("train set:", len(corpus.train))
    say("valid set:", len(corpus.valid))
    say("test set:", len(corpus.test))
    say("num tokens:", len(corpus.dictionary))
    frac = {"base-0.5set": 2, "base-0.25set": 4, "base-0.1set": 10, "base-0.05set": 20}.get(args.mark, 1)  # train.py:151-165
    pruning_train = int(len(corpus.train) / frac)
    eval_batch_size = 20
    train_data = D.batchify(corpus.train[:pruning_train], args.batch_size, device, rank, world)
    val_data = D.batchify(corpus.valid, eval_batch_size, device)
    test_data = D.batchify(corpus.test, eval_batch_size, device)
    ntokens = len(corpus.dictionary)

    model = build_model(args, ntokens)
    if args.prior == "True":  # partial state-dict load, keys filtered by name (train.py:239-258)
        prior = torch.load(os.path.join(args.prior_path, 'model.pt'), map_location='cpu')
        own = model.state_dict()
        own.update({k: v for k, v in prior.items() if k in own})
        model.load_state_dict(own)
    model = model.to(device)
    model.set_fused_sampling(bool(args.fused_sampling))
    if args.noise_source is not None:
        model.set_noise_source(args.noise_source)
    if args.gp_sample:
        from .model import GPNN
        gps = [m for m in model.modules() if isinstance(m, GPNN) and m.draws_noise()]
        if not gps:
            raise SystemExit("--gp-sample 1: this model has no GPNN with Bayesian coefficients or weights")
        for m in gps:
            m.sample = True
    total_params = sum(x.data.nelement() for x in model.parameters())
    say('Args: {}'.format(args))
    say('Model total parameters: {}'.format(total_params))
    say(str(model.transformerlayers if args.model == 'Transformer' else model.rnn))

    kl_fn = kl_selector(args)
    kl_scale = float(args.seq_len) / float(len(train_data))  # KL / len(train_data) * seq_len (train.py:338)
    trainer = engine.Trainer(model, lr=args.lr, clip=args.clip, momentum=0.9, kl_scale=kl_scale, seed=args.seed,
                             rank=rank, world=world, overlap=bool(args.dp_overlap), late_rows=bool(args.dp_late_rows))
    is_rnn = args.model != 'Transformer'

    def train_epoch(epoch, lr):
        total_loss = 0.
        start = time.time()
        hidden = model.init_hidden(train_data.size(1)) if is_rnn else None
        for batch, i in enumerate(range(0, train_data.size(0) - 1, args.seq_len)):
            data, targets = D.get_batch(train_data, i, args.seq_len)
            if is_rnn:
                hidden = repackage_hidden(hidden)
            loss, kl, hidden = trainer.step(data, targets, hidden, kl_fn)
            total_loss = total_loss + loss  # stays on the device: one host sync per log interval (train.py:422 syncs per step)
            if batch % args.log_interval == 0 and batch > 0:
                if world > 1:  # the mean over the GLOBAL batch, as the single-process log line prints it
                    total_loss = total_loss.detach().clone()
                    if args.dist_backend != "nccl":
                        total_loss = total_loss.cpu()  # a host-staged transport is handed host tensors (engine.LateRows.begin)
                    dist.all_reduce(total_loss)
                    total_loss = total_loss / world
                cur = float(total_loss) / args.log_interval
                history["interval_loss"].append(cur)
                elapsed = time.time() - start
                history["ms_per_batch"].append(elapsed * 1000 / args.log_interval)
                say('| epoch {:3d} | {:5d}/{:5d} batches | lr {:02.3f} | ms/batch {:5.2f} | loss {:5.2f} | '
                    'kl_loss {:5.4} | ppl {:8.2f}'.format(epoch, batch, len(train_data) // args.seq_len, lr,
                                                          elapsed * 1000 / args.log_interval, cur,
                                                          float(kl) if kl is not None else 0., math.exp(min(cur, 80.0))))
                total_loss = 0.
                start = time.time()

    sched = ValidationSchedule(args.lr, world, device)
    say("Start training")
    try:
        for epoch in range(1, args.epochs + 1):
            t0 = time.time()
            train_epoch(epoch, sched.lr)
            if world > 1:
                engine.heartbeat("epoch %d trained" % epoch, rank)
            val_loss, improved, stop = sched.update(engine.evaluate(model, val_data, args.seq_len, rank=rank, world=world))
            say('-' * 89)
            say('| end of epoch {:3d} | time: {:5.2f}s | valid loss {:5.2f} | valid ppl {:8.2f}'.format(
                epoch, time.time() - t0, val_loss, math.exp(val_loss)))
            say('-' * 89)
            history["valid_loss"].append(val_loss)
            _print_coef_mean(args, model, say)
            if improved:
                if is_main:
                    with open(args.save, 'wb') as f:
                        torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, f)
            else:  # train.py:503-508: halve LR, fresh SGD (momentum reset), reload best
                history["halved_epochs"].append(epoch)
                trainer.reset_optimizer(sched.lr)
                if world > 1:
                    dist.barrier()  # rank 0's checkpoint of an earlier epoch is on disk before anyone reads it
                with torch.no_grad():
                    sd = torch.load(args.save, map_location='cpu')
                    own = model.state_dict()
                    for k, v in sd.items():
                        own[k].copy_(v)
            if stop:
                break
    except KeyboardInterrupt:
        say('-' * 89)
        say('Exiting from training early')

    if world > 1:
        dist.barrier()
    if os.path.exists(args.save):
        with torch.no_grad():
            sd = torch.load(args.save, map_location='cpu')
            own = model.state_dict()
            for k, v in sd.items():
                own[k].copy_(v)
    _print_coef_mean(args, model, say)
    test_loss = sched.agree(engine.evaluate(model, test_data, args.seq_len, rank=rank, world=world))
    history["test_loss"] = test_loss
    say('=' * 89)
    say('| End of training | test loss {:5.2f} | test ppl {:8.2f}'.format(test_loss, math.exp(test_loss)))
    say('=' * 89)
    if args.history and is_main:
        import json
        with open(args.history, 'w') as f:
            json.dump(history, f)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return test_loss


if __name__ == "__main__":
    main()
