"""CPU: the package's own HDF5 reader (inaspeechsegmenter_amd/hdf5_reader.py) against files written by the real h5py / libhdf5.

The reference loads its three CNNs with keras.models.load_model from `*.hdf5` release assets (remote_utils.py:7-15,
segmenter.py:129-131); the target image has no h5py, so the product reads the format itself.  tests/golden/make_keras_hdf5.py
(run once with an interpreter that has h5py) wrote three files laid out like a Keras `model.save()` -- classic format with
fixed-length string attributes (Keras 2.x / h5py 2.x), chunked + gzip + shuffle datasets, and libver='latest' with variable-length
string attributes (version-2 object headers, link messages, global heap), and the same with twelve layers, a 66 KB model_config
and 600 attributes (dense storage: fractal heaps, version-2 B-trees two levels deep, a 'huge' heap object) plus datasets on each
version-4 chunk index -- and `keras_hdf5_expected.npz` = what h5py reads back.
The reader must return exactly that, and `keras_model.load_model_file` must lower the model from the .hdf5 directly."""
import json
import os

import numpy as np
import pytest

from inaspeechsegmenter_amd import hdf5_reader as H, keras_model as KM
from oracle import keras_cnn as ocnn
import prog_interp
from conftest import GOLDEN

FILES = ('keras2_like.hdf5', 'keras2_chunked_gzip.hdf5', 'tfkeras_latest.h5', 'tfkeras_latest_dense.h5')
DENSE = 'tfkeras_latest_dense.h5'


@pytest.mark.parametrize('fname', FILES)
def test_reader_returns_what_h5py_returns(fname):
    exp = np.load(os.path.join(GOLDEN, 'keras_hdf5_expected.npz'), allow_pickle=True)
    with H.File(os.path.join(GOLDEN, fname)) as f:
        mc = f.attrs['model_config']
        assert (mc.decode('utf8') if isinstance(mc, bytes) else mc) == str(exp[f'{fname}|model_config'])
        assert f.attrs.get('no_such_attribute') is None and 'model_weights' in f and 'nothing_here' not in f
        g = f['model_weights']
        assert list(g.attrs['layer_names']) == list(exp[f'{fname}|layer_names'])
        assert sorted(g) == sorted(n.decode() for n in exp[f'{fname}|layer_names'])
        n = 0
        for lname in g:
            names = g[lname].attrs['weight_names']
            assert list(names) == list(exp[f'{fname}|{lname}|weight_names'])
            for wn in names:
                wn = wn.decode('utf8')
                got, want = np.asarray(g[lname][wn]), exp[f'{fname}|{lname}|{wn}']
                assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want), (lname, wn)
                assert np.array_equal(np.asarray(f[f'/model_weights/{lname}/{wn}']), want)       # absolute path through the file
                n += 1
        assert n == 8
        assert int(np.asarray(f['optimizer_weights/Adam/iterations:0'])) == int(exp[f'{fname}|iterations']) == 1234
        assert np.array_equal(np.asarray(f['a_float64_matrix']), exp[f'{fname}|f64']) and f['a_float64_matrix'].shape == (3, 4)
        assert np.array_equal(np.asarray(f['a_float16_vector']), exp[f'{fname}|f16'])
        with pytest.raises(KeyError):
            f['model_weights/no_such_layer']


@pytest.mark.parametrize('fname', FILES)
def test_model_is_lowered_from_the_hdf5_file_itself(fname, monkeypatch):
    """keras_model.load_model_file on the .hdf5 (no h5py importable: the package's reader is what runs), then the lowered program
    against the oracle on the layers the same loader returned -- and those layers carry the file's own arrays."""
    import builtins
    real_import = builtins.__import__

    def no_h5py(name, *a, **k):
        if name == 'h5py':
            raise ImportError('h5py is not installed (test)')
        return real_import(name, *a, **k)
    monkeypatch.setattr(builtins, '__import__', no_h5py)
    layers, shp = KM.load_model_file(os.path.join(GOLDEN, fname))
    assert shp == (68, 21, 1) and [L['type'] for L in layers] == (['conv2d', 'batchnorm', 'activation', 'maxpool', 'flatten'] +
                                                                   ['dropout'] * (6 if fname == DENSE else 1) + ['dense'])
    exp = np.load(os.path.join(GOLDEN, 'keras_hdf5_expected.npz'), allow_pickle=True)
    assert np.array_equal(layers[0]['W'], exp[f'{fname}|conv2d_1|conv2d_1/kernel:0'])
    assert np.array_equal(layers[1]['var'], exp[f'{fname}|batch_normalization_1|batch_normalization_1/moving_variance:0'])
    comp = KM.compile_layers(layers, shp)
    x = np.random.default_rng(1).normal(0, 1, (3,) + shp).astype(np.float32)
    assert np.abs(prog_interp.run(comp, x) - ocnn.forward(layers, x)).max() < 2e-5


def test_dense_storage_and_version4_chunk_indexes():
    """libver='latest' past the compact limits: `model_weights` has 12 members (dense links), the root's model_config is a 'huge'
    fractal-heap object, `many_attributes` has 603 attributes behind a two-level B-tree v2; `chunk_indexes` holds one dataset per
    chunk-index kind of the version-4 layout message (the extensible array of an unlimited dimension is refused by name)."""
    exp = np.load(os.path.join(GOLDEN, 'keras_hdf5_expected.npz'), allow_pickle=True)
    with H.File(os.path.join(GOLDEN, DENSE)) as f:
        assert len(list(f['model_weights'])) == 12 and len(f.attrs['model_config']) > 65536
        a = f['many_attributes'].attrs
        keys = [k.split('|')[2] for k in exp.files if k.startswith(f'{DENSE}|many_attributes|')]
        assert len(keys) == 603 and sorted(a) == sorted(keys)
        for k in keys:
            want = exp[f'{DENSE}|many_attributes|{k}']
            got = np.asarray(a[k])
            assert np.array_equal(got, want) if want.dtype.kind != 'U' else str(got) == str(want), k
        n = 0
        for k in f['chunk_indexes']:
            if k == 'unlimited':
                with pytest.raises(NotImplementedError, match='extensible array'):
                    np.asarray(f['chunk_indexes'][k])
                continue
            got, want = np.asarray(f['chunk_indexes'][k]), exp[f'{DENSE}|chunk_indexes|{k}']
            assert got.dtype == want.dtype and np.array_equal(got, want), k
            n += 1
        assert n == 7 and float(np.asarray(f['chunk_indexes/fixed_array_paged_sparse']).sum()) == 10.0


def test_reader_rejects_what_it_does_not_understand(tmp_path):
    p = tmp_path / 'x.hdf5'
    p.write_bytes(b'not an hdf5 file at all')
    with pytest.raises(H.Hdf5Error):
        H.File(str(p))
    p.write_bytes(b'\x89HDF\r\n\x1a\n' + bytes([9]) + bytes(64))                 # unknown superblock version
    with pytest.raises(NotImplementedError):
        H.File(str(p))
