"""Host logic of the OCR stream: the repetition filter against what the reference's own filter kept on seeded
sequences (tests/golden/stream_golden.npz, made by tests/golden/make_stream_golden.py), text filtering and the
edit distance against hand-checked answers (the reference's text-metrics module does not import here: it needs
`unidecode`; its `edit_distance` with default costs is the textbook Levenshtein distance)."""
import importlib.util
import os

import numpy as np

from ocr_amd.models.ocr.streaming import StreamState, edit_distance, ioa_matrix, keep_text_result
from ocr_amd.utils.filters import RepetitionFilter, filter_boxes

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _generator():
    spec = importlib.util.spec_from_file_location('make_stream_golden', os.path.join(GOLD, 'make_stream_golden.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_repetition_filter_keeps_what_the_reference_keeps():
    gen = _generator()
    gold = np.load(os.path.join(GOLD, 'stream_golden.npz'))
    for s, kw in enumerate(gen.SCENARIOS):
        filt = RepetitionFilter(**kw)
        kept, waiting = [], []
        for boxes, rows, idx in gen.make_sequence(100 + s):
            kept.append(np.asarray(filt(boxes=boxes, indices=idx, rows=rows), dtype=np.int64) if len(boxes)
                        else np.zeros((0,), np.int64))
            waiting.append(len(filt))
        np.testing.assert_array_equal(np.array([len(k) for k in kept]), gold['s%d_kept_count' % s], err_msg=str(kw))
        np.testing.assert_array_equal(np.concatenate(kept), gold['s%d_kept' % s], err_msg=str(kw))
        np.testing.assert_array_equal(np.array(waiting), gold['s%d_waiting' % s], err_msg=str(kw))
    assert gold['s0_kept_count'].sum() > 10 and gold['s2_kept_count'].sum() < gold['s0_kept_count'].sum()


def test_repetition_filter_in_a_chain():
    """Second sighting passes, third does not (the track left the waiting list and starts again)."""
    filt = RepetitionFilter(n_repeat=2)
    boxes = np.array([[0.1, 0.1, 0.4, 0.2], [0.5, 0.5, 0.9, 0.6]], np.float32)
    rows = [boxes[:1], boxes[1:]]
    idx = [[[0, 1]], [[2]]]
    out = [len(filter_boxes([filt], boxes, idx, rows)[0]) for _ in range(4)]
    assert out == [0, 2, 0, 2]
    # a different number of words in the row is a different paragraph
    filt = RepetitionFilter(n_repeat=2)
    assert len(filt(boxes=boxes[:1], indices=[[[0, 1]]], rows=rows[:1])) == 0
    assert len(filt(boxes=boxes[:1], indices=[[[0, 1, 2]]], rows=rows[:1])) == 0
    assert len(filt(boxes=boxes[:1], indices=[[[0, 1]]], rows=rows[:1])) == 1


def test_edit_distance_known_answers():
    assert edit_distance('kitten', 'sitting', normalize=False) == 3
    assert abs(edit_distance('kitten', 'sitting') - 3 / 7) < 1e-12
    assert edit_distance('', 'abc', normalize=False) == 3 and edit_distance('abc', '', normalize=False) == 3
    assert edit_distance('flaw', 'lawn', normalize=False) == 2
    assert edit_distance('same text', 'same text') == 0
    assert edit_distance('intention', 'execution', normalize=False) == 5


def test_text_results_are_filtered_like_the_reference():
    def res(text, scores=(0.,)):
        return {'text': text, 'scores': np.array(scores, np.float32)}

    reject = set()
    assert not keep_text_result(res('   '), reject)
    assert not keep_text_result(res('word'), reject)                                  # a single word
    assert keep_text_result(res('word'), reject, skip_single_word=False)
    assert not keep_text_result(res('12 34'), reject)                                 # no letter
    assert keep_text_result(res('two words</s>'), reject)
    assert not keep_text_result(res('two words'), {'two words'})                      # already emitted
    r = res('good line \nbad line', (-0.1, -0.5))
    assert keep_text_result(r, reject, threshold=-0.2) and r['text'] == 'good line'   # low-score rows are blanked
    assert not keep_text_result(res('bad line \nworse line', (-0.3, -0.5)), reject, threshold=-0.2)
    assert not keep_text_result(res('hello big world'), reject, ['hello big w0rld'])  # 1 / 15 < 0.2
    assert 'hello big world' in reject
    assert keep_text_result(res('something else here'), reject, ['hello big world'])
    assert keep_text_result(res('hello big world!'), reject, ['x y'] * 5 + ['hello big world'][:0] + ['a b'] * 5, k=5)


def test_stream_state_and_ioa():
    st = StreamState()
    st.last_texts.append('a b')
    st.n_duplicates = 3
    st.last_boxes = np.zeros((1, 4))
    st.set_new_frame('map')
    assert st.prev_output == 'map' and st.last_texts == [] and st.last_boxes is None and st.n_duplicates == 0
    a = np.array([[0, 0, 2, 2], [0, 0, 0, 0]], np.float32)
    b = np.array([[1, 1, 3, 3], [0, 0, 2, 2]], np.float32)
    np.testing.assert_allclose(ioa_matrix(a, b), [[0.25, 1.0], [0.0, 0.0]])
