#!/opt/conda/bin/python3.9
"""Run the reference's OWN segmentation bookkeeping lines and record what they produce.

/root/reference/inaSpeechSegmenter/segmenter.py cannot be imported (tensorflow), but the functions that
decide slot/segment bookkeeping need only numpy + skimage, which /opt/conda/bin/python3.9 has.  This
script parses the reference file with `ast`, compiles exactly these definitions, unmodified,

    _media2feats (53-67), _energy_activity (69-73), _get_patches (76-88), _binidx2seglist (91-108),
    class DnnSegmenter (111-179) + SpeechMusic / SpeechMusicNoise / Gender (182-204),
    Segmenter.segment_feats (250-276)

into a namespace whose other names are the reference's own importable modules (pyannote_viterbi,
viterbi_utils, sidekit_mfcc) and skimage's view_as_windows, replaces only `self.nn.predict` by
tests/golden/fake_predict.py (the Keras models are un-vendored release assets) and the ffmpeg decode by
"return this array", and writes inputs + outputs to an .npz.  Called by make_golden.py; needs
/root/reference, so it runs in the build container only.  Nothing is copied from the reference: the
source is read, compiled and executed in memory.

    /opt/conda/bin/python3.9 tests/golden/ref_segmenter_pin.py <sidekit_feats.npz> <out.npz>
"""
import ast
import gc
import importlib.util
import os
import sys
import warnings

import numpy as np
from skimage.util import view_as_windows as vaw

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/inaSpeechSegmenter'
sys.path.insert(0, HERE)
from fake_predict import make_predict  # noqa: E402


def ref_module(name):
    spec = importlib.util.spec_from_file_location('ref_' + name, f'{REF}/{name}.py')
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def compile_reference_defs():
    src = open(f'{REF}/segmenter.py').read()
    tree = ast.parse(src)
    want_funcs = {'_media2feats', '_energy_activity', '_get_patches', '_binidx2seglist'}
    want_classes = {'DnnSegmenter', 'SpeechMusic', 'SpeechMusicNoise', 'Gender'}
    body = []
    seg_feats = None
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in want_funcs:
            body.append(node)
        elif isinstance(node, ast.ClassDef) and node.name in want_classes:
            body.append(node)
        elif isinstance(node, ast.ClassDef) and node.name == 'Segmenter':
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == 'segment_feats':
                    seg_feats = sub
    assert len(body) == len(want_funcs) + len(want_classes) and seg_feats is not None
    body.append(seg_feats)                                 # as a plain function taking `self`
    mod = ast.Module(body=body, type_ignores=[])
    vit = ref_module('pyannote_viterbi')
    vu = ref_module('viterbi_utils')
    sk = ref_module('sidekit_mfcc')
    ns = {'np': np, 'warnings': warnings, 'gc': gc, 'vaw': vaw,
          'viterbi_decoding': vit.viterbi_decoding, 'pred2logemission': vu.pred2logemission,
          'diag_trans_exp': vu.diag_trans_exp, 'log_trans_exp': vu.log_trans_exp, 'mfcc': sk.mfcc,
          'media2sig16kmono': lambda medianame, start_sec, stop_sec, ffmpeg, dtype: medianame}    # 'decode' = the array itself
    exec(compile(mod, f'{REF}/segmenter.py', 'exec'), ns)
    return ns


class _FakeNN:
    def __init__(self, fn):
        self.fn = fn
        self.calls = []

    def predict(self, batch, batch_size=32, verbose=0):
        self.calls.append(batch.shape)
        return self.fn(batch)


def main(feats_npz, out_npz):
    ns = compile_reference_defs()
    feats = np.load(feats_npz)
    out = {}

    def make(cls_name, nclass, salt):
        o = object.__new__(ns[cls_name])                   # __init__ would download a Keras model
        o.nn = _FakeNN(make_predict(nclass, salt))
        o.batch_size = 32
        return o

    class Seg:
        pass

    for engine, vadcls, nvad in (('smn', 'SpeechMusicNoise', 3), ('sm', 'SpeechMusic', 2)):
        for tag in ('musanmix', 'silence', 'synth'):
            mspec, loge = feats[tag + '_mspec'], feats[tag + '_loge']
            s = Seg()
            s.energy_ratio = 0.03
            s.vad = make(vadcls, nvad, 1)
            s.detect_gender = True
            s.gender = make('Gender', 2, 2)
            with np.errstate(divide='ignore', invalid='ignore'):
                lseg = ns['segment_feats'](s, mspec.copy(), loge.copy(), 0, 0)
            out[f'{engine}_{tag}_labels'] = np.array([l for l, _, _ in lseg])
            out[f'{engine}_{tag}_bounds'] = np.array([[a, b] for _, a, b in lseg], dtype=np.float64)
            out[f'{engine}_{tag}_ncalls'] = np.array([len(s.vad.nn.calls), len(s.gender.nn.calls)])
    # short media: the reference's own _media2feats padding (float64 promotion, segmenter.py:61-65) and difflen path
    for tag in ('short',):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            m2, loge, difflen = ns['_media2feats'](feats['short_sig'], None, None, None)     # segmenter.py:53-67, unmodified
        assert difflen == 68 - len(feats['short_loge']) > 0 and np.array_equal(loge, feats['short_loge'])
        s = Seg()
        s.energy_ratio = 0.03
        s.vad = make('SpeechMusicNoise', 3, 1)
        s.detect_gender = True
        s.gender = make('Gender', 2, 2)
        with np.errstate(divide='ignore', invalid='ignore'):
            lseg = ns['segment_feats'](s, m2, loge.copy(), difflen, 0)
        out['smn_short_labels'] = np.array([l for l, _, _ in lseg])
        out['smn_short_bounds'] = np.array([[a, b] for _, a, b in lseg], dtype=np.float64)
        out['short_padded_mspec'] = m2
        out['short_difflen'] = np.array(difflen)

    # _get_patches itself: full output on the small inputs, sampled rows + finite mask on musanmix
    for tag in ('synth', 'silence', 'short'):
        m = out['short_padded_mspec'] if tag == 'short' else feats[tag + '_mspec']
        for h in (21, 24):
            with np.errstate(divide='ignore', invalid='ignore'):
                p, f = ns['_get_patches'](m[:, :h].copy(), 68, 2)
            out[f'patches_{tag}_{h}'] = p                       # f32; f64 for the padded short file (promotion at segmenter.py:65)
            out[f'finite_{tag}_{h}'] = f
    m = feats['musanmix_mspec']
    for h in (21, 24):
        with np.errstate(divide='ignore', invalid='ignore'):
            p, f = ns['_get_patches'](m[:, :h].copy(), 68, 2)
        idx = np.unique(np.concatenate((np.arange(0, 40), np.arange(len(p) - 40, len(p)), np.arange(0, len(p), 97))))
        out[f'patches_musanmix_{h}_idx'] = idx
        out[f'patches_musanmix_{h}'] = p[idx]
        out[f'finite_musanmix_{h}'] = f
        out[f'patches_musanmix_{h}_rowsum'] = p.astype(np.float64).sum(axis=(1, 2))
    # odd / even frame counts (rfill length depends on len(mspec) % 2, segmenter.py:84)
    for T in (68, 69, 70, 131):
        mm = m[100:100 + T, :21].copy()
        p, f = ns['_get_patches'](mm, 68, 2)
        out[f'patches_T{T}_shape'] = np.array(p.shape)
        out[f'patches_T{T}_first_last'] = np.stack((p[0], p[-1]))
    # _binidx2seglist on mixed-type sequences
    seqs = [[0, 0, 1, 1, 1, 0], [1], [2.0, 2.0, 0.0], ['a', 'a', 'b']]
    out['binidx_cases'] = np.array(repr([ns['_binidx2seglist'](s) for s in seqs]))
    np.savez_compressed(out_npz, **out)
    print('reference segmenter bookkeeping pinned:', len(out), 'arrays ->', out_npz)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
