"""CTC greedy decoding and id -> text, restated in NumPy (test infrastructure).

Follows `utils/text/ctc_decoder.py:14-35` (wrapper around `keras.ops.ctc_decode(strategy='greedy',
mask_index=0)`) and `utils/text/tokenizer.py:189-219,283-285,599-669` (char-level vocabulary built
with `setdefault`, blank index 0, ids -> string).

Keras itself is not available here.  The token output is fully determined by argmax + collapse and
needs no pinning; the score formula (`-sum_t max_v logit[t] / length`, what the TensorFlow greedy
decoder returns as `neg_sum_logits`, divided by the length in the wrapper) is restated from
knowledge of Keras 3 and is **parity unpinned**.
"""
import numpy as np


def ctc_greedy(logits, lengths=None, blank_index=0):
    """(B,T,V) float32 -> (tokens (B,T) int32 padded with -1, scores (B,) float32).

    argmax over the raw logits (first maximum wins), merge repeats, drop blank, left-pack.
    """
    logits = np.asarray(logits, dtype=np.float32)
    B, T, V = logits.shape
    if lengths is None:
        lengths = np.full((B,), T, dtype=np.int32)
    tokens = np.full((B, T), -1, dtype=np.int32)
    scores = np.zeros((B,), dtype=np.float32)
    for b in range(B):
        L = int(lengths[b])
        if L == 0:
            continue
        idx = np.argmax(logits[b, :L], axis=-1)
        mx = logits[b, :L][np.arange(L), idx]
        acc = np.float32(0)
        for v in mx:                                    # sequential float32 sum over time
            acc = np.float32(acc + v)
        scores[b] = np.float32(-acc) / np.float32(L)
        prev, k = -1, 0
        for t in range(L):
            tok = int(idx[t])
            if tok != prev and tok != blank_index:
                tokens[b, k] = tok
                k += 1
            prev = tok
    return tokens, scores


class CharTokenizer:
    """Char-level vocabulary as the reference tokenizer builds it (tokenizer.py:189-219).

    Ids are assigned with `dict.setdefault`, so a character that appears twice in the vocabulary
    keeps its first id and every later character moves up by one; `vocab_size` shrinks accordingly.
    """

    def __init__(self, characters, blank='<blank>'):
        self.symbol_to_id = {}
        for s in [blank] + list(characters):
            self.symbol_to_id.setdefault(s, len(self.symbol_to_id))
        self.id_to_symbol = {i: s for s, i in self.symbol_to_id.items()}
        self.blank_token_idx = self.symbol_to_id[blank]

    @property
    def vocab_size(self):
        return len(self.id_to_symbol)

    def decode(self, tokens):
        """tokenizer.py:599-656 for one row of ids (skip_padding=True)."""
        tokens = [int(t) for t in tokens]
        symbols = []
        for i, tok in enumerate(tokens):
            if tok != self.blank_token_idx:
                symbols.append(tok)
            if tok == self.blank_token_idx and i < len(tokens) - 1 and tokens[i + 1] == self.blank_token_idx:
                break
        return ''.join(self.id_to_symbol.get(t, '') for t in symbols)

    def ctc_decode(self, logits, lengths=None):
        tokens, scores = ctc_greedy(logits, lengths, self.blank_token_idx)
        return [self.decode(t) for t in tokens], scores


def top2(logits):
    """(T,V) logits -> (classes (T,2) int, logits (T,2) float32 of the two largest per step (first maximum first),
    largest magnitude): all a tie check needs."""
    logits = np.asarray(logits, dtype=np.float32)
    order = np.argsort(-logits, axis=1, kind='stable')[:, :2]
    return order, np.take_along_axis(logits, order, axis=1), np.abs(logits).max()


def tie_flips(ref_logits, got_text, tokenizer, rel_tol=1e-4, max_ties=14):
    """Is `got_text` a greedy decoding of `ref_logits` apart from argmax ties?  `ref_logits`: (T,V) logits, or the
    `top2(logits)` triple.

    A step is a tie when the two largest logits of the step differ by less than `rel_tol` times the
    largest magnitude in `ref_logits` (the tolerance the network outputs are held to).  Returns the
    smallest number of tie steps at which the runner-up has to be taken to produce `got_text`
    (0 = the plain greedy decoding), or None when no choice at the tie steps produces it.
    """
    order, top, absmax = ref_logits if isinstance(ref_logits, tuple) else top2(ref_logits)
    tol = np.float32(rel_tol) * np.float32(absmax)
    ties = np.nonzero((top[:, 0] - top[:, 1]) < tol)[0]
    if len(ties) > max_ties:
        ties = ties[np.argsort((top[:, 0] - top[:, 1])[ties])[:max_ties]]
    blank = tokenizer.blank_token_idx
    best = None
    for choice in range(1 << len(ties)):
        n_flips = bin(choice).count('1')
        if best is not None and n_flips >= best:
            continue
        ids = order[:, 0].copy()
        for j, t in enumerate(ties):
            if (choice >> j) & 1:
                ids[t] = order[t, 1]
        prev, toks = -1, []
        for t in ids:
            if t != prev and t != blank:
                toks.append(int(t))
            prev = t
        if tokenizer.decode(toks + [-1]) == got_text:
            best = n_flips
    return best


def assert_texts_match_or_tie(got_texts, ref_texts, ref_logits, tokenizer, rel_tol=1e-4):
    """Every string equals the oracle's, or differs from it only through argmax ties of the oracle's own
    logits (`tie_flips`).  `ref_logits[k]` are the (T_k,V) logits the oracle decoded string k from (or their `top2`).
    Returns the number of strings that needed a tie flip."""
    assert len(got_texts) == len(ref_texts), (len(got_texts), len(ref_texts))
    flipped = 0
    for k, (g, r) in enumerate(zip(got_texts, ref_texts)):
        if g == r:
            continue
        n = tie_flips(ref_logits[k], g, tokenizer, rel_tol)
        assert n is not None and n > 0, 'string %d: %r differs from the oracle\'s %r and no argmax tie explains it' % (k, g, r)
        flipped += 1
    return flipped
