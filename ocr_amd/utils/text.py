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
        return [self.decode(row) for row in tokens]
