"""CTC prefix beam search on the host (NumPy).

Stands in for `ctc_decode(method='beam_search')` of the reference (utils/text/ctc_decoder.py:14-35 ->
`keras.ops.ctc_decode(strategy='beam_search')`, which on TensorFlow is `tf.nn.ctc_beam_search_decoder`
over the log-softmax of the logits).  Keras is not available to compare against, so this follows
the published algorithm (Graves' prefix search with blank / non-blank path probabilities in log
space); two deliberate simplifications: only the `num_beams` most probable labels of a time step
(plus the blank) are expanded, and only the best sequence is returned.  The score is the log
probability of the sequence divided by the input length, like the reference's wrapper returns.

This is the compatibility path for `BaseOCR.infer(method='beam_search')` (the reference's default);
the fast path of this package is greedy decoding on the GPU.
"""
import numpy as np

_NEG = -np.inf


def _logaddexp(a, b):
    if a == _NEG:
        return b
    if b == _NEG:
        return a
    m = a if a > b else b
    return m + np.log1p(np.exp(-abs(a - b)))


def ctc_beam_search_one(logits, num_beams=10, blank=0):
    """logits (T,V) -> (best label sequence as list of int, log probability)."""
    logits = np.asarray(logits, dtype=np.float64)
    T, V = logits.shape
    m = logits.max(axis=1, keepdims=True)
    logp = logits - (m + np.log(np.exp(logits - m).sum(axis=1, keepdims=True)))
    beams = {(): (0.0, _NEG)}                     # prefix -> (log p ending in blank, log p ending in non-blank)
    k_sym = min(V, num_beams + 1)
    for t in range(T):
        row = logp[t]
        cand = np.argpartition(-row, k_sym - 1)[:k_sym] if k_sym < V else np.arange(V)
        cand = [int(c) for c in cand if c != blank]
        nxt = {}
        for prefix, (pb, pnb) in beams.items():
            total = _logaddexp(pb, pnb)
            # stay on the prefix: a blank, or a repeat of its last label
            b0, nb0 = nxt.get(prefix, (_NEG, _NEG))
            b0 = _logaddexp(b0, total + row[blank])
            if prefix:
                nb0 = _logaddexp(nb0, pnb + row[prefix[-1]])
            nxt[prefix] = (b0, nb0)
            for c in cand:
                ext = prefix + (c,)
                b1, nb1 = nxt.get(ext, (_NEG, _NEG))
                if prefix and c == prefix[-1]:
                    nb1 = _logaddexp(nb1, pb + row[c])        # a repeat needs a blank in between
                else:
                    nb1 = _logaddexp(nb1, total + row[c])
                nxt[ext] = (b1, nb1)
        best = sorted(nxt.items(), key=lambda kv: -_logaddexp(kv[1][0], kv[1][1]))[:num_beams]
        beams = dict(best)
    prefix, (pb, pnb) = max(beams.items(), key=lambda kv: _logaddexp(kv[1][0], kv[1][1]))
    return list(prefix), float(_logaddexp(pb, pnb))


def ctc_beam_search(logits, lengths=None, num_beams=10, blank=0):
    """logits (B,T,V) padded, lengths (B,) -> (tokens (B,T) int32 padded with -1, scores (B,) float32)."""
    logits = np.asarray(logits)
    B, T, _ = logits.shape
    tokens = np.full((B, T), -1, dtype=np.int32)
    scores = np.zeros((B,), dtype=np.float32)
    for b in range(B):
        L = int(lengths[b]) if lengths is not None else T
        if L == 0:
            continue
        seq, lp = ctc_beam_search_one(logits[b, :L], num_beams, blank)
        tokens[b, :len(seq)] = seq
        scores[b] = lp / L
    return tokens, scores
