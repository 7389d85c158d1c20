#!/usr/bin/env python3
"""Architecture search over Bayesian / GP components -- same command line as the reference's
steps/pytorchnn/train_search_bayes.py (argument names, types, defaults :29-103; log formats :335-341,
:397-401; best-checkpoint / LR-halving / early-stop loop :389-428), running on the HIP engine.

Per window (train_search_bayes.py:203-290): (1) ``Architect.step`` on a validation window -- Adam on the
architecture logits; (2) network step on the training window -- CE (+ KL), clip, SGD(momentum 0.9,
weight_decay 1e-5).  ``--model Transformer`` searches GELU-vs-GPNN feed-forwards
(GaussTransModelSearch), anything else standard-vs-Bayes LSTM gates (BayesLSTMModelSearch).

Deliberate differences from the reference script:
 * its end-of-epoch table (:404-410) indexes the logits as (2,4,2) and raises IndexError for
   ``--model Transformer`` before the first checkpoint is written; here the swapped table is printed
   for the LSTM shape and the plain softmax for the Transformer shape, and the run continues;
 * ``--T_bayes_pos MHA|EMB`` (:300-305) dereference attributes the search models do not have
   (AttributeError in the reference); rejected up front here;
 * parameters that never receive a gradient there (``bias_hh``; the ``*_lgstd`` tensors when no KL is
   requested) are skipped by torch.optim.SGD, weight decay included; they are frozen here, same effect;
 * single GPU (the search is not on the data-parallel path; run independent replicas).

    python -m bayeslms_amd.train_search_bayes --data DIR --model Transformer --emsize 512 --nhid 4096 \
        --nlayers 6 --nhead 8 --T_bayes_pos FFN --tied --cuda --save search.pt
"""
import argparse
import math
import os
import random
import time

import torch


def build_parser():
    p = argparse.ArgumentParser(description="Architecture search for Bayesian / GP language models (MI355X engine).")
    p.add_argument('--data', type=str, default='./data/pytorchnn', help='location of the data corpus')
    p.add_argument('--model', type=str, default='Transformer', help='Transformer or LSTM')
    p.add_argument('--emsize', type=int, default=200)
    p.add_argument('--nhid', type=int, default=200)
    p.add_argument('--nlayers', type=int, default=6)
    p.add_argument('--nhead', type=int, default=2)
    p.add_argument('--uncertainty', type=str, default='none')
    p.add_argument('--T_bayes_pos', type=str, default='none', help='[none | FFN]: FFN adds the GPNN KL terms')
    p.add_argument('--L_bayes_pos', type=int, default=0, help='> 0 adds the KL of the Bayes gates')
    p.add_argument('--L_gauss_pos', type=str, default='00')
    p.add_argument('--T_gauss_pos', type=int, default=3)
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
    p.add_argument('--seed', type=int, default=11)
    p.add_argument('--resume', type=str, default='')
    p.add_argument('--debug', action='store_true')
    p.add_argument('--work_dir', default='TFM', type=str)
    p.add_argument('--prior', default="False", type=str)
    p.add_argument('--prior_path', default='steps/pytorchnn/prior', type=str)
    p.add_argument('--unrolled', action='store_true', default=False, help='use one-step unrolled validation loss')
    p.add_argument('--wdecay', type=float, default=5e-7)
    p.add_argument('--arch_wdecay', type=float, default=1e-3)
    p.add_argument('--arch_lr', type=float, default=3e-3)
    p.add_argument('--gemm-mode', type=str, default='f32', choices=['f32', 'bf16x6', 'bf16x3'],
                   help='new, optional: opt-in split-bf16 arithmetic of the GEMM family (DESIGN.md section 7); default fp32 MFMA')
    return p


GATES = ("bayes_ingate", "bayes_forgate", "bayes_cellgate", "bayes_outgate")
SGD_WEIGHT_DECAY = 1e-5  # train_search_bayes.py:391-392


def build_model(args, ntokens):
    from . import model_search_bayes as S
    if args.model == 'Transformer':
        return S.GaussTransModelSearch(ntokens, args.emsize, args.nhead, args.nhid, args.nlayers, args.dropout, args.tied)
    return S.BayesLSTMModelSearch('LSTM', ntokens, args.emsize, args.nhid, args.nlayers, args.dropout, args.tied)


def freeze_unused(args, model):
    """Parameters the reference's loop never gives a gradient to (torch.optim.SGD skips them, weight decay
    included): bias_hh of the search cells (model_search_bayes.py:690-691), the Bayes gates' *_lgstd unless
    --L_bayes_pos > 0 adds their KL (:312-318), the GPNN *_lgstd never (sampled every network step)."""
    if args.model == 'Transformer':
        return
    for cell in model.rnn.rnn:
        cell.bias_hh.requires_grad_(False)
        if not (args.uncertainty != 'Gaussian' and args.L_bayes_pos > 0):
            for g in GATES:
                getattr(cell, g).weights_lgstd.requires_grad_(False)
                getattr(cell, g).bias_lgstd.requires_grad_(False)


def kl_selector(args):
    """-> kl_fn(model) or None: which KL train_search_bayes.py:296-330 adds (before the /len(train_data)*seq_len)."""
    if args.model == 'Transformer':
        if args.T_bayes_pos == 'FFN':
            return lambda m: sum(layer.gpnn.kl_divergence() for layer in m.transformerlayers)
        if args.T_bayes_pos in ('MHA', 'EMB'):
            raise SystemExit("--T_bayes_pos %s: the search models have no such KL term (AttributeError in the "
                             "reference, train_search_bayes.py:300-305)" % args.T_bayes_pos)
        return None
    if args.uncertainty == 'Gaussian':
        raise SystemExit("--uncertainty Gaussian with --model LSTM dereferences gpnn_cellgate on the Bayes cells "
                         "(AttributeError in the reference, train_search_bayes.py:309-311)")
    if args.L_bayes_pos > 0:
        def kl(m):  # the sample flags are raised only around the KL (:284-290,319-324): the forward stays deterministic
            total = 0
            for cell in m.rnn.rnn:
                for g in GATES:
                    b = getattr(cell, g)
                    b.sample = True
                    total = total + b.kl_divergence()
                    b.sample = False
            return total
        return kl
    return None


def arch_table(model):
    """The end-of-epoch print of train_search_bayes.py:403-411 (cell 0's rows shown swapped)."""
    probs = torch.softmax(model.arch_parameters()[0].detach(), dim=-1)
    if probs.dim() == 3 and probs.shape[0] >= 2 and probs.shape[1] == 4:
        out = torch.zeros_like(probs)
        out[1] = probs[1]
        out[0, :, 0] = probs[0, :, 1]
        out[0, :, 1] = probs[0, :, 0]
        return out
    return probs


def main(argv=None):
    args = build_parser().parse_args(argv)
    random.seed(args.seed)
    torch.manual_seed(args.seed)
    if not args.cuda or not torch.cuda.is_available():
        raise SystemExit("bayeslms_amd.train_search_bayes needs --cuda and an MI355X: there is no CPU path")
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)

    from . import data as D, engine, ops
    from .architect import Architect
    if args.gemm_mode != 'f32':
        ops.set_gemm_mode(args.gemm_mode)
    from .model import repackage_hidden

    print('Configurations')
    for k, v in vars(args).items():
        print(k, v)
    corpus = D.Corpus(args.data)
    eval_batch_size = 20
    train_data = D.batchify(corpus.train, args.batch_size, device)
    search_data = D.batchify(corpus.valid, args.batch_size, device)
    val_data = D.batchify(corpus.valid, eval_batch_size, device)
    test_data = D.batchify(corpus.test, eval_batch_size, device)
    ntokens = len(corpus.dictionary)

    model = build_model(args, ntokens)
    if args.prior == "True":
        prior = torch.load(os.path.join(args.prior_path, 'model.pt'), map_location='cpu')
        own = model.state_dict()
        own.update({k: v for k, v in prior.items() if k in own})
        model.load_state_dict(own)
    model = model.to(device)
    freeze_unused(args, model)
    architect = Architect(model, ntokens, args)
    print('Args: {}'.format(args))
    print('Model total parameters: {}'.format(sum(x.data.nelement() for x in model.parameters())))
    print(str(model.transformerlayers if args.model == 'Transformer' else model.rnn))

    is_rnn = args.model != 'Transformer'
    kl_fn = kl_selector(args)
    kl_scale = float(args.seq_len) / float(len(train_data))
    trainer = engine.Trainer(model, lr=args.lr, clip=args.clip, momentum=0.9, kl_scale=kl_scale, seed=args.seed,
                             weight_decay=SGD_WEIGHT_DECAY)

    def set_gp_sample(on):
        if not is_rnn:
            for layer in model.transformerlayers:
                layer.gpnn.sample = on

    def train_epoch(epoch, lr):
        total_loss = 0.
        start = time.time()
        hidden = model.init_hidden(args.batch_size) if is_rnn else None
        hiddens_valid = model.init_hidden(args.batch_size) if is_rnn else None
        for batch, i in enumerate(range(0, train_data.size(0) - 1, args.seq_len)):
            data, targets = D.get_batch(train_data, i, args.seq_len)
            data_valid, targets_valid = D.get_batch(search_data, i % (search_data.size(0) - 1), args.seq_len)
            # architecture step: dropout / noise streams of an odd Philox step, the network step uses the even one
            model.train()
            model.set_step(2 * trainer.step_no + 1)
            architect.step(data, targets, data_valid, targets_valid, None, args.unrolled, hiddens_valid)
            set_gp_sample(True)
            if is_rnn:
                hidden = repackage_hidden(hidden)
            loss, kl, hidden = trainer.step(data, targets, hidden, kl_fn, philox_step=2 * trainer.step_no)
            set_gp_sample(False)
            total_loss = total_loss + loss
            if batch % args.log_interval == 0 and batch > 0:
                cur = float(total_loss) / args.log_interval
                elapsed = time.time() - start
                print('| epoch {:3d} | {:5d}/{:5d} batches | lr {:02.3f} | ms/batch {:5.2f} | loss {:5.2f} | '
                      'kl_loss {:5.4} | ppl {:8.2f}'.format(epoch, batch, len(train_data) // args.seq_len, lr,
                                                            elapsed * 1000 / args.log_interval, cur,
                                                            float(kl) if kl is not None else 0., math.exp(min(cur, 80.0))),
                      flush=True)
                print(torch.softmax(model.arch_parameters()[0].detach(), dim=-1))
                total_loss = 0.
                start = time.time()

    lr = args.lr
    best_val = None
    counter = 0
    print("Start training")
    try:
        for epoch in range(1, args.epochs + 1):
            t0 = time.time()
            train_epoch(epoch, lr)
            val_loss = engine.evaluate(model, val_data, args.seq_len)
            print('-' * 89)
            print('| end of epoch {:3d} | time: {:5.2f}s | valid loss {:5.2f} | valid ppl {:8.2f}'.format(
                epoch, time.time() - t0, val_loss, math.exp(val_loss)))
            print('-' * 89)
            print(arch_table(model), flush=True)
            if not best_val or val_loss < best_val:
                with open(args.save, 'wb') as f:
                    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, f)
                torch.save(model.arch_parameters()[0].detach().cpu(), args.save + ".arch")  # new: the logits themselves
                best_val = val_loss
            else:  # :420-424: halve LR, fresh SGD (momentum reset); no reload of the best checkpoint
                lr /= 2.
                trainer.reset_optimizer(lr)
                counter += 1
            if counter == 8:
                break
    except KeyboardInterrupt:
        print('-' * 89)
        print('Exiting from training early')

    if os.path.exists(args.save):
        with torch.no_grad():
            sd = torch.load(args.save, map_location='cpu')
            own = model.state_dict()
            for k, v in sd.items():
                own[k].copy_(v)
    test_loss = engine.evaluate(model, test_data, args.seq_len)
    print('=' * 89)
    print('| End of training | test loss {:5.2f} | test ppl {:8.2f}'.format(test_loss, math.exp(test_loss)))
    print('=' * 89)
    return test_loss


if __name__ == "__main__":
    main()
