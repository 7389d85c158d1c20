"""Corpus / vocabulary loading and the batchify layout (host side).

Same file contract and API as the reference's data.py (steps/pytorchnn/data.py:9-52) and the
batchify/get_batch helpers of train.py:167-179,299-303: ``words.txt`` holds ``word id`` per line,
text files one sentence per line, ``<s>`` is appended to every line, OOV maps to ``<unk>``.
Tokenisation is vectorised over the whole file instead of a tensor per line.
"""
import os

import numpy as np
import torch


class Dictionary(object):
    def __init__(self):
        self.word2idx = {}
        self.idx2word = []

    def read_vocab(self, path):
        with open(path, "r", encoding="utf-8") as f:
            for line in f:
                fields = line.split()
                assert len(fields) == 2
                w = fields[0]
                if w not in self.word2idx:
                    self.word2idx[w] = len(self.idx2word)
                    self.idx2word.append(w)

    def __len__(self):
        return len(self.idx2word)


class Corpus(object):
    def __init__(self, path):
        self.dictionary = Dictionary()
        self.dictionary.read_vocab(os.path.join(path, "words.txt"))
        self.train = self.tokenize(os.path.join(path, "train.txt"))
        self.valid = self.tokenize(os.path.join(path, "valid.txt"))
        self.test = self.tokenize(os.path.join(path, "test.txt"))

    def tokenize(self, path):
        """-> 1-D int64 tensor of token ids with '<s>' closing every line."""
        assert os.path.exists(path)
        w2i = self.dictionary.word2idx
        unk = w2i.get("<unk>")
        eos = w2i["<s>"]
        ids = []
        with open(path, "r", encoding="utf-8") as f:
            for line in f:
                for w in line.split():
                    i = w2i.get(w, unk)
                    if i is None:
                        raise KeyError("<unk>")
                    ids.append(i)
                ids.append(eos)
        return torch.from_numpy(np.asarray(ids, dtype=np.int64))


def batchify(data, bsz, device=None, rank=0, world=1):
    """(N,) stream -> (N // bsz, bsz) columns of contiguous text (train.py:167-179).

    Data parallel: ``bsz`` is the GLOBAL batch; rank r keeps columns [r*bsz/world, (r+1)*bsz/world)
    of exactly the single-process layout, so W ranks together see the same
    global batch as one process with batch size bsz."""
    nbatch = data.size(0) // bsz
    data = data.narrow(0, 0, nbatch * bsz).view(bsz, -1).t().contiguous()
    if world > 1:
        if bsz % world:
            raise ValueError("global batch %d is not divisible by world size %d" % (bsz, world))
        per = bsz // world
        data = data[:, rank * per:(rank + 1) * per].contiguous()
    return data.to(device) if device is not None else data


def get_batch(source, i, seq_len):
    """Window i of length <= seq_len and its next-token targets (train.py:299-303)."""
    n = min(seq_len, len(source) - 1 - i)
    return source[i:i + n], source[i + 1:i + 1 + n].reshape(-1)


def synthetic_corpus(vocab, n_tokens, seed=1111):
    """AMI-shaped synthetic stream: ids ~ Zipf(1.0) over [2,V), sentence length
    1 + Poisson(7) clipped to [1,60], '<s>' (=0) closing every sentence."""
    rng = np.random.RandomState(seed)
    ranks = np.arange(1, vocab - 1, dtype=np.float64)
    p = 1.0 / ranks
    p /= p.sum()
    out = np.empty(n_tokens, dtype=np.int64)
    words = rng.choice(vocab - 2, size=n_tokens, p=p) + 2
    pos = 0
    while pos < n_tokens:
        ln = int(np.clip(1 + rng.poisson(7), 1, 60))
        end = min(n_tokens, pos + ln)
        out[pos:end] = words[pos:end]
        if end < n_tokens:
            out[end] = 0
        pos = end + 1
    return torch.from_numpy(out)
