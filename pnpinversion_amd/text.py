"""Tokenizer / text-encoder plumbing.

The reference uses transformers' CLIPTokenizer + CLIPTextModel of the SD-1.4 checkpoint (models/p2p/inversion.py:291-306).
Neither the CLIP vocabulary nor any checkpoint exists on the build / GPU boxes, so this module provides seeded stand-ins with
the same interface (what models/p2p/* and utils/utils.py actually touch: `encode`, `decode`, `__call__`, `model_max_length`;
`text_encoder(ids)[0]`).  The text TRANSFORMER itself is native (pipeline.NativeTextEncoder -> pnpi_text_encode: transformers'
CLIPTextModel on the device, weights in its state-dict layout); what has no offline source is the BPE vocabulary, so the
tokenizer stays this word-level stand-in (with a real checkpoint directory pass transformers' CLIPTokenizer as `tokenizer`).
SyntheticTextEncoder is the weight-free stand-in used by the golden fixtures."""
import zlib

import numpy as np
import torch

BOS, EOS = 49406, 49407


class WordTokenizer:
    """CLIP conventions (BOS 49406, EOS/pad 49407, 77 positions) over a deterministic word-piece vocabulary: words longer
    than 6 characters are split into two pieces so that the multi-token paths of get_word_inds / seq_aligner are exercised."""
    model_max_length = 77

    def __init__(self):
        self._id2piece = {}

    def _piece_id(self, piece):
        i = 1000 + zlib.crc32(piece.encode()) % 48000
        while i in self._id2piece and self._id2piece[i] != piece:
            i = 1000 + (i - 999) % 48000
        self._id2piece[i] = piece
        return i

    def _pieces(self, word):
        return [word[:4], word[4:]] if len(word) > 6 else [word]

    def encode(self, text):
        ids = [BOS]
        for w in text.split(" "):
            if w:
                ids += [self._piece_id(p) for p in self._pieces(w)]
        return (ids + [EOS])[: self.model_max_length]

    def decode(self, ids):
        out = []
        for i in ids:
            i = int(i)
            if i == BOS:
                out.append("<|startoftext|>")
            elif i == EOS:
                out.append("<|endoftext|>")
            else:
                out.append(self._id2piece.get(i, "?"))
        return " ".join(out) if len(out) != 1 else out[0]

    class _Out:
        def __init__(self, ids):
            self.input_ids = ids

    def __call__(self, texts, padding="max_length", max_length=None, truncation=True, return_tensors="pt"):
        if isinstance(texts, str):
            texts = [texts]
        L = max_length or self.model_max_length
        rows = []
        for t in texts:
            ids = self.encode(t)[:L]
            rows.append(ids + [EOS] * (L - len(ids)))
        return WordTokenizer._Out(torch.tensor(rows, dtype=torch.int64))


class SyntheticTextEncoder:
    """Seeded stand-in for CLIPTextModel: y[b, p] = normalise(E[id] + P[p] + 0.5 * mean_{q<=p} E[id_q]) (causal, like CLIP).
    Returns a tuple so that `text_encoder(ids)[0]` works; values are fp16-representable."""

    def __init__(self, dim=768, seed=0, device="cpu"):
        self.dim, self.seed, self.device = dim, seed, device
        self._cache = {}

    def _emb(self, key):
        if key not in self._cache:
            g = np.random.Generator(np.random.Philox(key=[self.seed & 0xFFFFFFFF, key & 0xFFFFFFFF]))
            self._cache[key] = g.standard_normal(self.dim).astype(np.float32)
        return self._cache[key]

    def to(self, device):
        self.device = device
        return self

    def __call__(self, input_ids):
        ids = input_ids.cpu().numpy()
        out = np.zeros(ids.shape + (self.dim,), dtype=np.float32)
        for b in range(ids.shape[0]):
            run = np.zeros(self.dim, dtype=np.float32)
            for p in range(ids.shape[1]):
                e = self._emb(int(ids[b, p]))
                run = run + e
                v = e + self._emb(100000 + p) + 0.5 * run / (p + 1)
                v = (v - v.mean()) / (v.std() + 1e-5)
                out[b, p] = v
        out = out.astype(np.float16).astype(np.float32)
        return (torch.from_numpy(out).to(self.device),)
