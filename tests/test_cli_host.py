"""Host logic of the sweep driver (no GPU): RLE mask decode semantics of run_editing_p2p.py:11-27."""
import numpy as np

from run_editing_p2p import mask_decode


def _reference_semantics(encoded, shape=(512, 512)):
    n = shape[0] * shape[1]
    a = np.zeros(n)
    for i in range(0, len(encoded), 2):
        for j in range(min(encoded[i + 1], n - encoded[i])):
            a[encoded[i] + j] = 1
    a = a.reshape(shape)
    a[0, :] = 1; a[-1, :] = 1; a[:, 0] = 1; a[:, -1] = 1
    return a


def test_mask_decode():
    rng = np.random.default_rng(0)
    enc = []
    pos = 0
    while pos < 512 * 512 - 5000:
        pos += int(rng.integers(1, 4000))
        ln = int(rng.integers(1, 3000))
        enc += [pos, ln]
        pos += ln
    enc += [512 * 512 - 10, 500]           # run past the end is clipped
    assert np.array_equal(mask_decode(enc), _reference_semantics(enc))
    assert mask_decode([]).sum() == 4 * 512 - 4
