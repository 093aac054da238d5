"""Extracts the character tables of the EasyOCR recognisers from the reference checkout into a data file.

    python tools/make_charsets.py [/root/reference]  ->  ocr_amd/utils/charsets.json

The tables are the constant `EASYOCR_MODELS` literal of architectures/crnn_arch.py:45-203 (the class order of
every EasyOCR checkpoint).  The file is read as text and the two assignments are folded from their syntax trees
(string / list / dict constants, `+`, `''.join(list)`); nothing of the reference is imported or executed.  Only the model name, generation, checkpoint file name and
characters are kept (no download locations).
"""
import ast
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fold(node, names):
    """Constant folding of the few node kinds the two tables consist of."""
    if isinstance(node, ast.Constant):
        return node.value
    if isinstance(node, ast.Name):
        return names[node.id]
    if isinstance(node, ast.List):
        return [fold(e, names) for e in node.elts]
    if isinstance(node, ast.Dict):
        return {fold(k, names): fold(v, names) for k, v in zip(node.keys, node.values)}
    if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Add):
        return fold(node.left, names) + fold(node.right, names)
    if (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == 'join'
            and len(node.args) == 1 and not node.keywords):
        return fold(node.func.value, names).join(fold(node.args[0], names))
    raise ValueError('unexpected syntax at line %d: %s' % (node.lineno, ast.dump(node)[:80]))


def main(ref='/root/reference'):
    with open(os.path.join(ref, 'architectures', 'crnn_arch.py'), encoding='utf-8') as fh:
        tree = ast.parse(fh.read())
    names = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], 'id', None) in ('separator_char', 'EASYOCR_MODELS'):
            names[node.targets[0].id] = fold(node.value, names)
    models = names['EASYOCR_MODELS']
    out = {}
    for gen, table in models.items():
        for name, info in table.items():
            out[name] = {'generation': int(gen[-1]), 'filename': info['filename'], 'script': info.get('model_script'),
                         'characters': info['characters']}
    path = os.path.join(ROOT, 'ocr_amd', 'utils', 'charsets.json')
    with open(path, 'w', encoding='utf-8') as fh:
        json.dump(out, fh, ensure_ascii=False, indent=0, sort_keys=True)
    for name, info in sorted(out.items()):
        chars = info['characters']
        print('%-16s gen%d %5d characters, %5d unique' % (name, info['generation'], len(chars), len(set(chars))))


if __name__ == '__main__':
    main(*sys.argv[1:])
