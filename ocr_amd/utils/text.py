"""Char-level vocabulary and id -> text decoding, as the reference's tokenizer does it for the
CRNN models (utils/text/tokenizer.py:189-219,283-285,599-656; models/ocr/crnn.py:23,33-38)."""

# character table of EasyOCR's english_g2 recogniser (architectures/crnn_arch.py, EASYOCR_MODELS)
ENGLISH_G2_CHARACTERS = ('0123456789!"#$%&\'()*+,-./:;<=>?@[\\]^_`{|}~ €'
                         'ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz')


_CHARSETS = None


def easyocr_models():
    """name -> dict(generation, filename, script, characters) of every EasyOCR recogniser the reference lists
    (architectures/crnn_arch.py:45-203): `characters` is the class order of the checkpoint after the blank.
    Read from the data file charsets.json next to this module (made by tools/make_charsets.py)."""
    global _CHARSETS
    if _CHARSETS is None:
        import json
        import os
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'charsets.json'), encoding='utf-8') as fh:
            _CHARSETS = json.load(fh)
    return _CHARSETS


def get_characters(model):
    """Characters of an EasyOCR recogniser by name ('english_g2', 'zh_sim_g2', ...): ValueError when unknown
    (architectures/crnn_arch.py:785-793)."""
    table = easyocr_models()
    if model not in table:
        raise ValueError('No model found for {} ; known : {}'.format(model, tuple(sorted(table))))
    return table[model]['characters']


class CharTokenizer:
    """vocab = ['<blank>'] + list(characters); ids assigned with `dict.setdefault`, so a repeated
    character keeps its first id and the ids after it shift down (the reference's behaviour)."""

    def __init__(self, characters=ENGLISH_G2_CHARACTERS, blank='<blank>'):
        self._symbol_to_id = {}
        for s in [blank] + list(characters):
            self._symbol_to_id.setdefault(s, len(self._symbol_to_id))
        self._id_to_symbol = {i: s for s, i in self._symbol_to_id.items()}
        self.blank_token_idx = self._symbol_to_id[blank]

    @property
    def vocab_size(self):
        return len(self._id_to_symbol)

    def decode(self, tokens):
        """One row of token ids (padded with -1) -> str."""
        tokens = [int(t) for t in tokens]
        blank = self.blank_token_idx
        out = []
        for i, tok in enumerate(tokens):
            if tok != blank:
                out.append(self._id_to_symbol.get(tok, ''))
            if tok == blank and i < len(tokens) - 1 and tokens[i + 1] == blank:
                break
        return ''.join(out)

    def decode_batch(self, tokens):
        """Rows of token ids (n, T) -> list of n str, the same strings as `decode` row by row.  Vectorised over the
        batch: a page batch of the pipeline has thousands of rows, and the per-row Python loop over all T positions
        costs more host time than the GPU needs for the batch."""
        import numpy as np
        tokens = np.asarray(tokens)
        if tokens.ndim != 2 or tokens.shape[0] == 0:
            return [self.decode(row) for row in tokens]
        n, T = tokens.shape
        blank = self.blank_token_idx
        # position of the first blank that is followed by a blank: decoding stops there
        is_blank = tokens == blank
        cut = np.full(n, T, dtype=np.int64)
        if T > 1:
            stop = is_blank[:, :-1] & is_blank[:, 1:]
            cut = np.where(stop.any(axis=1), stop.argmax(axis=1), T)
        table = self._table()
        keep = ~is_blank & (np.arange(T)[None, :] < cut[:, None]) & (tokens >= 0) & (tokens < len(table))
        counts = keep.sum(axis=1)
        picked = tokens[keep]                              # row-major: row after row, positions ascending
        ends = np.cumsum(counts)
        if self._single_chars:
            text = ''.join(table[picked].tolist())
            starts = ends - counts
            return [text[a:b] for a, b in zip(starts.tolist(), ends.tolist())]
        symbols = table[picked].tolist()
        return [''.join(symbols[a:b]) for a, b in zip((ends - counts).tolist(), ends.tolist())]

    def _table(self):
        """Symbols by id as a NumPy array ('' for ids without a symbol), built on first use."""
        if getattr(self, '_table_cache', None) is None:
            import numpy as np
            size = max(self._id_to_symbol) + 1
            table = np.array([self._id_to_symbol.get(i, '') for i in range(size)], dtype=object)
            self._single_chars = all(len(sym) == 1 for i, sym in self._id_to_symbol.items() if i != self.blank_token_idx)
            if self._single_chars:
                table[self.blank_token_idx] = ' '          # never picked: blanks are dropped before the lookup
                table = table.astype('<U1')
            self._table_cache = table
        return self._table_cache
