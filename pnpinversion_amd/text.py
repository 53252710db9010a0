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
        body = []
        for w in text.split(" "):
            if w:
                body += [self._piece_id(p) for p in self._pieces(w)]
        # CLIP truncation keeps the EOS token (body cut to max_length - 2), as ClipBPETokenizer.encode does
        return [BOS] + body[: self.model_max_length - 2] + [EOS]

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
            ids = self.encode(t)
            if truncation and len(ids) > L:
                ids = ids[: L - 1] + [EOS]
            rows.append(ids + [EOS] * (L - len(ids)) if padding else ids)
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


# ------------------------------------------------------------------------------------------------------------------ CLIP BPE
def _bytes_to_unicode():
    """The byte <-> printable-unicode table of GPT-2 / CLIP byte-level BPE."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


class ClipBPETokenizer:
    """CLIP's byte-level BPE tokenizer (what the reference gets from transformers' CLIPTokenizer: models/p2p/inversion.py:291-301,
    utils/utils.py:84-102 use `__call__`, `encode`, `decode`, `model_max_length`), from a checkpoint's vocab.json / merges.txt.
    Pipeline as in transformers 5.x' tokenizers-backed class: NFC + whitespace collapse + lowercase; split by CLIP's regex;
    bytes -> unicode symbols; BPE merges by rank with the end-of-word suffix "</w>"; <|startoftext|> ... <|endoftext|>, padded
    with <|endoftext|>.  Pure Python (host integer work, ~10 tokens per prompt); pinned against transformers' implementation on
    a synthetic vocabulary in tests/test_host_tables.py (no real vocabulary exists offline)."""
    model_max_length = 77

    def __init__(self, vocab, merges, unk_token="<|endoftext|>", bos_token="<|startoftext|>", eos_token="<|endoftext|>"):
        import json
        import regex
        if isinstance(vocab, str):
            vocab = json.load(open(vocab, encoding="utf-8"))
        if isinstance(merges, str):
            lines = open(merges, encoding="utf-8").read().split("\n")
            merges = [tuple(l.split()) for l in lines if l and not l.startswith("#version")]
        self.encoder = dict(vocab)
        self.decoder = {i: t for t, i in self.encoder.items()}
        self.ranks = {tuple(m) if not isinstance(m, str) else tuple(m.split()): i for i, m in enumerate(merges)}
        self.b2u = _bytes_to_unicode()
        self.u2b = {u: b for b, u in self.b2u.items()}
        self.bos_token_id, self.eos_token_id = self.encoder[bos_token], self.encoder[eos_token]
        self.unk_token_id = self.encoder[unk_token]
        self.pad_token_id = self.eos_token_id
        self._pat = regex.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""")
        self._ws = regex.compile(r"\s+")
        self._cache = {}

    def _bpe(self, piece):
        if piece in self._cache:
            return self._cache[piece]
        word = list(piece[:-1]) + [piece[-1] + "</w>"]
        while len(word) > 1:
            pairs = [(self.ranks.get((word[i], word[i + 1]), 1 << 60), i) for i in range(len(word) - 1)]
            rank, at = min(pairs)
            if rank == 1 << 60:
                break
            first, second = word[at], word[at + 1]
            out, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == first and word[i + 1] == second:
                    out.append(first + second)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = out
        self._cache[piece] = word
        return word

    def _tokenize(self, text):
        import unicodedata
        text = self._ws.sub(" ", unicodedata.normalize("NFC", text)).lower()
        ids = []
        for piece in self._pat.findall(text):
            if piece in ("<|startoftext|>", "<|endoftext|>"):
                ids.append(self.encoder[piece])
                continue
            sym = "".join(self.b2u[b] for b in piece.encode("utf-8"))
            ids += [self.encoder.get(tok, self.unk_token_id) for tok in self._bpe(sym)]
        return ids

    def encode(self, text, max_length=None):
        ids = self._tokenize(text)
        if max_length is not None:
            ids = ids[: max_length - 2]
        return [self.bos_token_id] + ids + [self.eos_token_id]

    def decode(self, ids):
        toks = [self.decoder.get(int(i), "") for i in (ids.tolist() if hasattr(ids, "tolist") else ids)]
        text = "".join(toks)
        data = bytearray()
        for ch in text.replace("</w>", " "):
            data += bytes([self.u2b[ch]]) if ch in self.u2b else ch.encode("utf-8")
        return data.decode("utf-8", errors="replace").strip()

    def __call__(self, texts, padding="max_length", max_length=None, truncation=True, return_tensors="pt"):
        if isinstance(texts, str):
            texts = [texts]
        L = max_length or self.model_max_length
        rows = []
        for t in texts:
            ids = self.encode(t, max_length=L if truncation else None)
            rows.append(ids + [self.pad_token_id] * (L - len(ids)) if padding == "max_length" else ids)
        return WordTokenizer._Out(torch.tensor(rows, dtype=torch.int64))
