"""CTC prefix beam search (host): exhaustive check on tiny problems, agreement with greedy on peaked logits."""
import itertools

import numpy as np

from ocr_amd.utils.ctc_beam import ctc_beam_search, ctc_beam_search_one
from oracle.ctc import ctc_greedy


def _brute_force(logits, blank=0):
    T, V = logits.shape
    logp = logits - np.log(np.exp(logits).sum(axis=1, keepdims=True))
    probs = {}
    for path in itertools.product(range(V), repeat=T):
        seq, prev = [], None
        for c in path:
            if c != prev and c != blank:
                seq.append(c)
            prev = c
        p = float(np.exp(sum(logp[t, c] for t, c in enumerate(path))))
        probs[tuple(seq)] = probs.get(tuple(seq), 0.0) + p
    best = max(probs.items(), key=lambda kv: kv[1])
    return list(best[0]), float(np.log(best[1]))


def test_beam_search_finds_the_most_probable_labelling():
    rng = np.random.default_rng(0)
    for _ in range(25):
        T, V = int(rng.integers(1, 6)), int(rng.integers(2, 5))
        logits = rng.normal(0, 1.5, (T, V))
        seq, lp = ctc_beam_search_one(logits, num_beams=64)
        ref_seq, ref_lp = _brute_force(logits)
        assert seq == ref_seq
        assert abs(lp - ref_lp) < 1e-9


def test_beam_search_equals_greedy_on_peaked_logits():
    rng = np.random.default_rng(1)
    B, T, V = 6, 30, 97
    logits = rng.normal(0, 0.1, (B, T, V)).astype(np.float32)
    idx = rng.integers(0, V, (B, T))
    idx[rng.random((B, T)) < 0.5] = 0
    np.put_along_axis(logits, idx[..., None], 12.0, axis=2)
    lengths = np.array([30, 30, 17, 1, 0, 29], dtype=np.int32)
    bt, bs = ctc_beam_search(logits, lengths, num_beams=10)
    gt, _ = ctc_greedy(logits, lengths, 0)
    np.testing.assert_array_equal(bt, gt)
    assert np.all(bs <= 0) and bs[4] == 0


def _plain_prefix_search(logits, num_beams, blank=0):
    """Textbook prefix beam search: every label extends every kept prefix, dictionaries all the way."""
    logp = logits - np.log(np.exp(logits).sum(axis=1, keepdims=True))
    lae = np.logaddexp
    beams = {(): (0.0, -np.inf)}
    for row in logp:
        nxt = {}
        for p, (pb, pnb) in beams.items():
            tot = lae(pb, pnb)
            b0, nb0 = nxt.get(p, (-np.inf, -np.inf))
            nxt[p] = (lae(b0, tot + row[blank]), lae(nb0, pnb + row[p[-1]]) if p else nb0)
            for c in range(len(row)):
                if c == blank:
                    continue
                q = p + (c,)
                b1, nb1 = nxt.get(q, (-np.inf, -np.inf))
                nxt[q] = (b1, lae(nb1, (pb if p and c == p[-1] else tot) + row[c]))
        beams = dict(sorted(nxt.items(), key=lambda kv: -lae(*kv[1]))[:num_beams])
    p, v = max(beams.items(), key=lambda kv: lae(*kv[1]))
    return list(p), float(lae(*v))


def test_no_label_of_a_step_is_pruned():
    """A narrow beam over flat distributions: the vectorised search keeps exactly what the textbook search keeps."""
    rng = np.random.default_rng(2)
    for _ in range(12):
        T, V, beams = int(rng.integers(3, 14)), int(rng.integers(5, 40)), int(rng.integers(1, 6))
        logits = rng.normal(0, 0.7, (T, V))
        seq, lp = ctc_beam_search_one(logits, num_beams=beams)
        ref_seq, ref_lp = _plain_prefix_search(logits, beams)
        assert seq == ref_seq and abs(lp - ref_lp) < 1e-9
