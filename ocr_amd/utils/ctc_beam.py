"""CTC prefix beam search on the host (NumPy).

Stands in for `ctc_decode(method='beam_search')` of the reference (utils/text/ctc_decoder.py:14-35 ->
`keras.ops.ctc_decode(strategy='beam_search')`, which on TensorFlow is `tf.nn.ctc_beam_search_decoder`
over the log-softmax of the logits).  Keras is not available to compare against, so this follows
the published algorithm (Graves' prefix search with blank / non-blank path probabilities in log
space): at every step EVERY label extends every kept prefix and the `num_beams` most probable prefixes
survive (no pruning of the labels of a step; the expansion is vectorised over the vocabulary and only the
extensions that can reach the beam, or that merge into a kept prefix, are materialised).  Only the best
sequence is returned.  The score is the log probability of the sequence divided by the input length, like
the reference's wrapper returns.

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
    for t in range(T):
        row = logp[t]
        prefixes = list(beams)
        pb = np.array([beams[p][0] for p in prefixes])
        pnb = np.array([beams[p][1] for p in prefixes])
        total = np.logaddexp(pb, pnb)
        # log p of every extension (prefix n, label c): any path into the prefix, except that a repeat of the prefix's
        # last label needs a blank in between
        ext = total[:, None] + row[None, :]
        for n, p in enumerate(prefixes):
            if p:
                ext[n, p[-1]] = pb[n] + row[p[-1]]
        ext[:, blank] = _NEG
        # the extensions worth a dictionary entry: the num_beams most probable ones, and those that land on a kept prefix
        # (their probability adds to what that prefix keeps by staying)
        # Ties are broken the way the GPU search breaks them (csrc/ctc_beam.hip), so that both give the same answer on
        # any input: extensions in the order (value descending, prefix ascending, label rank ascending), the rank of a
        # label being its place in the step's order (logit descending, label ascending).
        flat = ext.ravel()
        k = min(num_beams, flat.size)
        label_rank = np.empty(V, dtype=np.int64)
        label_rank[np.lexsort((np.arange(V), -logits[t]))] = np.arange(V)
        near = np.argpartition(-flat, k - 1)[:k]
        cut = flat[near].min()
        cand = np.nonzero(flat >= cut)[0]                    # everything that ties with the k-th value takes part
        cand = cand[np.lexsort((label_rank[cand % V], cand // V, -flat[cand]))][:k]
        picked = [int(i) for i in cand]
        index = {p: n for n, p in enumerate(prefixes)}
        seen = set(picked)
        for p in prefixes:
            if p and p[:-1] in index:
                i = index[p[:-1]] * V + p[-1]
                if i not in seen:
                    picked.append(i)
                    seen.add(i)
        nxt = {}
        for n, p in enumerate(prefixes):
            # stay on the prefix: a blank, or a repeat of its last label
            nxt[p] = (total[n] + row[blank], pnb[n] + row[p[-1]] if p else _NEG)
        for i in picked:
            n, c = divmod(i, V)
            if flat[i] == _NEG:
                continue
            q = prefixes[n] + (c,)
            b1, nb1 = nxt.get(q, (_NEG, _NEG))
            nxt[q] = (b1, _logaddexp(nb1, flat[i]))
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
