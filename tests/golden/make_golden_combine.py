"""Generates tests/golden/combine_vectors.npz: inputs and outputs of the reference's word -> line ->
paragraph grouping (utils/image/bounding_box/combination.py).

Inputs: the four box layouts the reference's own tests use (tests/test_utils_boxes.py:380-557; read
as data with `ast`, together with the expected groups / boxes those tests assert) and kept boxes of
synthetic pages.  Outputs: what the reference's functions return for them in this container.
"""
import ast
import json
import os
import sys

import numpy as np

REF = os.environ.get('OCR_REFERENCE', '/root/reference')
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))


def literal(node):
    """np.array([...]) or a plain literal -> Python object."""
    if isinstance(node, ast.Call) and getattr(node.func, 'attr', '') == 'array':
        return ast.literal_eval(node.args[0])
    return ast.literal_eval(node)


def reference_test_cases():
    src = open(os.path.join(REF, 'tests', 'test_utils_boxes.py')).read()
    tree = ast.parse(src)
    cases = {}
    for cls in tree.body:
        if isinstance(cls, ast.ClassDef) and cls.name == 'TestCombination':
            for fn in cls.body:
                if isinstance(fn, ast.FunctionDef) and fn.name.startswith('test_'):
                    vals = {}
                    for st in fn.body:
                        if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name):
                            try:
                                vals[st.targets[0].id] = literal(st.value)
                            except Exception:
                                pass
                    if 'boxes' in vals:
                        cases[fn.name] = vals
    return cases


def jsonable(groups):
    return json.dumps(groups, default=lambda o: int(o) if isinstance(o, np.integer) else list(o))


def main():
    sys.path.insert(0, ROOT)
    from ocr_amd import synth
    from oracle import pipeline as opipe
    cases = reference_test_cases()
    os.chdir(REF)
    sys.path.insert(0, REF)
    from utils.image.bounding_box import combine_boxes, combine_boxes_horizontal, combine_boxes_vertical

    gold = {}
    inputs = {name: np.array(v['boxes'], dtype=np.float64) for name, v in cases.items()}
    for k in range(3):
        images, maps, _ = synth.make_pages(300 + k, 1, 512, 640, 60)
        _, boxes, _ = opipe.detect_page(images[0], None, maps_override=maps[0], run_network=False)
        inputs['synthetic_page_%d' % k] = boxes
    names = sorted(inputs)
    gold['names'] = np.array(names)
    for name in names:
        b = inputs[name]
        gold[name + '/boxes'] = b
        hb, hg, hrows = combine_boxes_horizontal(b.copy(), source='xyxy')
        vb, vg, vrows = combine_boxes_vertical(np.array(hb).copy(), indices=hg, source='xyxy')
        cb, cg, crows = combine_boxes(b.copy(), source='xyxy')
        gold[name + '/h_boxes'] = np.array(hb)
        gold[name + '/h_groups'] = np.array(jsonable(hg))
        gold[name + '/hv_boxes'] = np.array(vb)
        gold[name + '/hv_groups'] = np.array(jsonable(vg))
        gold[name + '/c_boxes'] = np.array(cb)
        gold[name + '/c_groups'] = np.array(jsonable(cg))
        gold[name + '/c_rows'] = np.array(jsonable([np.asarray(r).tolist() for r in crows]))
        if name in cases:        # what the reference's tests assert for this layout
            for key in ('target_groups_h', 'target_boxes_h', 'target_groups_hv', 'target_boxes_hv'):
                if key in cases[name]:
                    gold[name + '/' + key] = np.array(jsonable(cases[name][key]))
        print(name, len(b), '->', len(hb), 'lines ->', len(cb), 'paragraphs')
    np.savez_compressed(os.path.join(OUT, 'combine_vectors.npz'), **gold)


if __name__ == '__main__':
    main()
