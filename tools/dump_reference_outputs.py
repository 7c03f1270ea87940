#!/usr/bin/env python3
"""Dump what the REFERENCE computes with its real weights, for tests/test_real_weights.py to compare against.

Run it anywhere the reference runs (TensorFlow + onnxruntime + the model files that `get_remote` downloads,
remote_utils.py:4-27), from the reference checkout or with inaSpeechSegmenter installed:

    python tools/dump_reference_outputs.py [--media-dir <reference>/media] [--out tests/golden/real_reference_dump.npz]

then commit / copy the .npz next to the other fixtures and put the same model files under ~/.keras/inaSpeechSegmenter/ on the
GPU box (as .hdf5 with h5py, or as the .npz export of tools/convert_keras_hdf5.py; final.onnx as it is).

Contents (keys):
  <engine>_model_config            the JSON `model_config` of each Keras file (engine in smn, sm, gender, vfp, bgc): the topology
  <engine>_musanmix_{batch_slots,rawpred}
                                   for media/musanmix.wav: the 20 ms slot index of every row handed to `nn.predict`
                                   (segmenter.py:156-163) and the raw softmax output per row -- the frame logits the north star
                                   asks to match within 1e-3; gender rows come from the smn pass' speech segments
  musanmix_<engine>_segments       final (label, start, stop) rows, as strings / floats
  lamartine_xvectors, lamartine_xvector_times
                                   OnnxBackendExtractor output for media/lamartine.wav (vbx_segmenter.py:217-266), pre x10
  lamartine_vf_{vfp,bgc}           (score, speech_duration, nb_vectors) of VoiceFemininityScoring (run_test.py:177-187: 0.534884)
  versions                         tensorflow / onnxruntime / numpy versions used
Nothing in the product or the tests needs this script to have run: the consuming tests skip without the file."""
import argparse
import json
import os
import sys

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--media-dir', default='./media')
    ap.add_argument('--out', default='real_reference_dump.npz')
    args = ap.parse_args()
    from inaSpeechSegmenter import Segmenter
    from inaSpeechSegmenter import segmenter as refseg
    out = {}
    versions = {'numpy': np.__version__}
    try:
        import tensorflow as tf
        versions['tensorflow'] = tf.__version__
    except Exception as exc:                                  # noqa: BLE001
        versions['tensorflow'] = f'unavailable: {exc}'
    mus = os.path.join(args.media_dir, 'musanmix.wav')

    def record(dnn, store):
        """wrap dnn.nn.predict: keep (a copy of) every output; the slot indices are rebuilt from the segment list below"""
        orig = dnn.nn.predict

        def wrapped(batch, *a, **k):
            r = orig(batch, *a, **k)
            store.append(np.array(r, dtype=np.float32))
            return r
        dnn.nn.predict = wrapped

    for engine in ('smn', 'sm'):
        seg = Segmenter(vad_engine=engine, detect_gender=True)
        vad_out, gen_out = [], []
        record(seg.vad, vad_out)
        record(seg.gender, gen_out)
        mspec, loge, difflen = refseg._media2feats(mus, None, None, seg.ffmpeg)
        # the same calls Segmenter.segment_feats makes (segmenter.py:250-276), with the intermediate lists kept
        lseg0 = []
        for lab, start, stop in refseg._binidx2seglist(refseg._energy_activity(loge, seg.energy_ratio)[::2]):
            lseg0.append(('noEnergy' if lab == 0 else 'energy', start, stop))
        lseg1 = seg.vad(mspec, lseg0, difflen)
        lseg2 = seg.gender(mspec, lseg1, difflen)
        for name, dnn, store, lin in ((engine, seg.vad, vad_out, lseg0), ('gender' if engine == 'smn' else None, seg.gender, gen_out, lseg1)):
            if name is None:
                continue
            slots = np.concatenate([np.arange(a, b) for lab, a, b in lin if lab == dnn.inlabel] or [np.zeros(0, int)])
            raw = np.concatenate(store) if store else np.zeros((0, len(dnn.outlabels)), np.float32)
            assert len(raw) == len(slots), (name, len(raw), len(slots))
            out[f'{name}_musanmix_batch_slots'] = slots.astype(np.int32)
            out[f'{name}_musanmix_rawpred'] = raw
            out[f'{name}_model_config'] = np.array(dnn.nn.to_json())
        final = [(lab, start * .02, stop * .02) for lab, start, stop in lseg2]
        out[f'musanmix_{engine}_segments_labels'] = np.array([l for l, _, _ in final])
        out[f'musanmix_{engine}_segments_times'] = np.array([[a, b] for _, a, b in final], np.float64)
    # ---- x-vectors and the voice femininity score
    try:
        from inaSpeechSegmenter.vbx_segmenter import VoiceFemininityScoring
        import onnxruntime
        versions['onnxruntime'] = onnxruntime.__version__
        lam = os.path.join(args.media_dir, 'lamartine.wav')
        for crit in ('vfp', 'bgc'):
            vfs = VoiceFemininityScoring(gd_model_criteria=crit)
            score, dur, nb = vfs(lam)
            out[f'lamartine_vf_{crit}'] = np.array([score, dur, nb], np.float64)
            out[f'{crit}_model_config'] = np.array(vfs.gender_detection_mlp_model.to_json())
            if crit == 'vfp':
                from inaSpeechSegmenter.io import media2sig16kmono
                from inaSpeechSegmenter.vbx_segmenter import get_features
                sig = media2sig16kmono(lam, dtype='float64')
                xv = vfs.xvector_model('lamartine', get_features(sig), len(sig) / 16000)
                out['lamartine_xvectors'] = np.array([x / 10 for _, _, x in xv], np.float32)
                out['lamartine_xvector_times'] = np.array([t for _, t, _ in xv], np.float64)
    except Exception as exc:                                  # noqa: BLE001
        print('x-vector / VFS part skipped:', exc, file=sys.stderr)
    out['versions'] = np.array(json.dumps(versions))
    np.savez_compressed(args.out, **out)
    print('wrote', args.out, {k: getattr(v, 'shape', None) for k, v in out.items()})


if __name__ == '__main__':
    main()
