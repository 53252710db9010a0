"""Host-side token alignment tables for the Prompt-to-Prompt controllers (integer work, stays on the CPU).

Same names, arguments and results as the reference's models/p2p/seq_aligner.py:
  get_word_inds (:131-149, also utils/utils.py:84-102), get_replacement_mapper (:152-195), get_refinement_mapper (:107-128).
Independent implementation; the tables are checked bit-for-bit against the reference in tests/test_host_tables.py."""
import numpy as np
import torch


def get_word_inds(text, word_place, tokenizer):
    """Token positions (+1 for BOS) of the words selected by `word_place` (a word string -> every equal word, or a word index)."""
    words = text.split(" ")
    if type(word_place) is str:
        wanted = {i for i, w in enumerate(words) if w == word_place}
    elif type(word_place) is int:
        wanted = {word_place}
    else:
        wanted = set(word_place)
    if not wanted:
        return np.array([], dtype=np.int64)
    pieces = [tokenizer.decode([tok]).strip("#") for tok in tokenizer.encode(text)][1:-1]
    out, word_idx, filled = [], 0, 0
    for pos, piece in enumerate(pieces):
        filled += len(piece)
        if word_idx in wanted:
            out.append(pos + 1)
        if filled >= len(words[word_idx]):
            word_idx, filled = word_idx + 1, 0
    return np.array(out)


def _global_alignment(x, y, gap=0, match=1, mismatch=-1):
    """Needleman-Wunsch with the reference's tie-breaking (left, then up, then diagonal; seq_aligner.py:61-76)."""
    nx, ny = len(x), len(y)
    score = np.zeros((nx + 1, ny + 1), dtype=np.int32)
    score[0, 1:] = (np.arange(ny) + 1) * gap
    score[1:, 0] = (np.arange(nx) + 1) * gap
    trace = np.zeros((nx + 1, ny + 1), dtype=np.int32)
    trace[0, 1:] = 1
    trace[1:, 0] = 2
    trace[0, 0] = 4
    for i in range(1, nx + 1):
        for j in range(1, ny + 1):
            left = score[i, j - 1] + gap
            up = score[i - 1, j] + gap
            diag = score[i - 1, j - 1] + (match if x[i - 1] == y[j - 1] else mismatch)
            best = max(left, up, diag)
            score[i, j] = best
            trace[i, j] = 1 if best == left else (2 if best == up else 3)
    return trace


def _target_to_source(x, y, trace):
    """[(target position j, aligned source position i or -1)] in target order (seq_aligner.py:79-104)."""
    i, j = len(x), len(y)
    pairs = []
    while i > 0 or j > 0:
        step = trace[i, j]
        if step == 3:
            i, j = i - 1, j - 1
            pairs.append((j, i))
        elif step == 1:
            j -= 1
            pairs.append((j, -1))
        elif step == 2:
            i -= 1
        else:
            break
    pairs.reverse()
    return torch.tensor(pairs, dtype=torch.int64)


def get_mapper(x, y, tokenizer, max_len=77):
    x_seq, y_seq = tokenizer.encode(x), tokenizer.encode(y)
    base = _target_to_source(x_seq, y_seq, _global_alignment(x_seq, y_seq))
    alphas = torch.ones(max_len)
    alphas[: base.shape[0]] = base[:, 1].ne(-1).float()
    mapper = torch.zeros(max_len, dtype=torch.int64)
    mapper[: base.shape[0]] = base[:, 1]
    mapper[base.shape[0]:] = len(y_seq) + torch.arange(max_len - len(y_seq))
    return mapper, alphas


def get_refinement_mapper(prompts, tokenizer, max_len=77):
    mappers, alphas = zip(*[get_mapper(prompts[0], p, tokenizer, max_len) for p in prompts[1:]])
    return torch.stack(mappers), torch.stack(alphas)


def get_replacement_mapper_(x, y, tokenizer, max_len=77):
    words_x, words_y = x.split(" "), y.split(" ")
    if len(words_x) != len(words_y):
        raise ValueError(f"attention replacement edit can only be applied on prompts with the same length"
                         f" but prompt A has {len(words_x)} words and prompt B has {len(words_y)} words.")
    changed = [i for i in range(len(words_y)) if words_y[i] != words_x[i]]
    src = [get_word_inds(x, i, tokenizer) for i in changed]
    tgt = [get_word_inds(y, i, tokenizer) for i in changed]
    mapper = np.zeros((max_len, max_len))
    i = j = k = 0
    while i < max_len and j < max_len:
        if k < len(src) and src[k][0] == i:
            s, t = src[k], tgt[k]
            if len(s) == len(t):
                mapper[s, t] = 1
            else:
                for col in t:
                    mapper[s, col] = 1 / len(t)
            k += 1
            i += len(s)
            j += len(t)
        elif k < len(src):
            mapper[i, j] = 1
            i += 1
            j += 1
        else:
            mapper[j, j] = 1
            i += 1
            j += 1
    return torch.from_numpy(mapper).float()


def get_replacement_mapper(prompts, tokenizer, max_len=77):
    return torch.stack([get_replacement_mapper_(prompts[0], p, tokenizer, max_len) for p in prompts[1:]])
