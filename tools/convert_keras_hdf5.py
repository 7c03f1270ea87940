#!/usr/bin/env python3
"""Keras HDF5 -> flat .npz (numpy only at load time).

The reference loads `keras_speech_music_noise_cnn.hdf5`, `keras_speech_music_cnn.hdf5` and
`keras_male_female_cnn.hdf5` (remote_utils.py:7-15) with keras.models.load_model
(segmenter.py:129-131).  The MI355X runtime does not carry TensorFlow; when h5py is not importable
either (it is not in the ROCm image's default interpreter), run this once with any python that has
h5py (e.g. /opt/conda/bin/python3.9) and put the .npz next to the .hdf5:

    python tools/convert_keras_hdf5.py ~/.keras/inaSpeechSegmenter/keras_male_female_cnn.hdf5

Output keys: 'model_config' (the JSON string stored in the HDF5 attribute) and
'<layer_name>/<kernel|bias|gamma|beta|moving_mean|moving_variance>' arrays.
"""
import os
import sys

import numpy as np


def convert(path, out=None):
    try:
        import h5py
    except ImportError:                              # the package's own reader has the same interface for what is used here
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
        from inaspeechsegmenter_amd import hdf5_reader as h5py
    out = out or os.path.splitext(path)[0] + '.npz'
    arrays = {}
    with h5py.File(path, 'r') as f:
        mc = f.attrs['model_config']
        arrays['model_config'] = np.array(mc.decode('utf-8') if isinstance(mc, bytes) else str(mc))
        g = f['model_weights'] if 'model_weights' in f else f
        for lname in g:
            for wn in g[lname].attrs.get('weight_names', []):
                wn = wn.decode('utf-8') if isinstance(wn, bytes) else wn
                short = wn.split('/')[-1].split(':')[0]
                arrays[f'{lname}/{short}'] = np.asarray(g[lname][wn])
    np.savez(out, **arrays)
    return out


if __name__ == '__main__':
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    for p in sys.argv[1:]:
        print(convert(p))
