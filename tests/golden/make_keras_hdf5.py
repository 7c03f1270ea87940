#!/usr/bin/env python3
"""Writes the small HDF5 fixtures of tests/test_hdf5_reader.py with the real h5py / libhdf5 (run with an interpreter that has h5py,
e.g. /opt/conda/bin/python3.9 tests/golden/make_keras_hdf5.py): files laid out the way Keras 2.x / tf.keras save a Sequential model
(`model.save('x.hdf5')`: root attributes model_config / keras_version / backend, `model_weights/<layer>/<layer>/<weight>:0` datasets,
`layer_names` / `weight_names` attributes), in four flavours of the file format (the fourth: libver='latest' with so many layers,
attributes and bytes of `model_config` that libhdf5 switches to dense storage -- fractal heaps and version-2 B-trees -- and with the
version-4 layout's chunk indexes), plus what h5py itself reads back from them
(`keras_hdf5_expected.npz`) -- the inaspeechsegmenter_amd/hdf5_reader.py under test has to return exactly that."""
import json
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def model(rng, nmel=21, ncls=3):
    cfg = {'class_name': 'Sequential', 'config': {'name': 'sequential_1', 'layers': [
        {'class_name': 'Conv2D', 'config': {'name': 'conv2d_1', 'batch_input_shape': [None, 68, nmel, 1], 'filters': 8, 'kernel_size': [4, 5],
                                             'strides': [1, 1], 'padding': 'valid', 'activation': 'linear', 'use_bias': True}},
        {'class_name': 'BatchNormalization', 'config': {'name': 'batch_normalization_1', 'axis': -1, 'epsilon': 0.001, 'center': True, 'scale': True}},
        {'class_name': 'Activation', 'config': {'name': 'activation_1', 'activation': 'relu'}},
        {'class_name': 'MaxPooling2D', 'config': {'name': 'max_pooling2d_1', 'pool_size': [2, 2], 'strides': [2, 2], 'padding': 'valid'}},
        {'class_name': 'Flatten', 'config': {'name': 'flatten_1'}},
        {'class_name': 'Dropout', 'config': {'name': 'dropout_1', 'rate': 0.2}},
        {'class_name': 'Dense', 'config': {'name': 'dense_1', 'units': ncls, 'activation': 'softmax', 'use_bias': True}}]}}
    w = {'conv2d_1': {'kernel:0': rng.normal(0, 0.3, (4, 5, 1, 8)).astype(np.float32), 'bias:0': rng.normal(0, 0.1, 8).astype(np.float32)},
         'batch_normalization_1': {'gamma:0': rng.uniform(0.8, 1.2, 8).astype(np.float32), 'beta:0': rng.normal(0, 0.1, 8).astype(np.float32),
                                   'moving_mean:0': rng.normal(0, 0.1, 8).astype(np.float32),
                                   'moving_variance:0': rng.uniform(0.5, 1.5, 8).astype(np.float32)},
         'activation_1': {}, 'max_pooling2d_1': {}, 'flatten_1': {}, 'dropout_1': {},
         'dense_1': {'kernel:0': rng.normal(0, 0.05, (32 * ((nmel - 4) // 2) * 8, ncls)).astype(np.float32), 'bias:0': np.zeros(ncls, np.float32)}}
    return cfg, w


def dense_variant(cfg, w):
    """The same net with five more (parameterless) layers -- twelve members of `model_weights`, past the eight a new-style group keeps
    in its object header -- and a model_config over the 64 KB an object header message can hold."""
    cfg = json.loads(json.dumps(cfg))
    w = dict(w)
    for i in range(2, 7):
        cfg['config']['layers'].insert(-1, {'class_name': 'Dropout', 'config': {'name': f'dropout_{i}', 'rate': 0.1}})
        w[f'dropout_{i}'] = {}
    cfg['config']['a_very_long_comment'] = 'what a newer Keras might add. ' * 2200
    return cfg, w


def write_index_cases(f, rng):
    """Datasets whose chunk index is each of the version-4 kinds a fixed-shape dataset can get, and a group with 600 attributes."""
    g = f.create_group('chunk_indexes')
    g.create_dataset('single', data=rng.normal(size=(7, 5)).astype(np.float32), chunks=(7, 5))
    g.create_dataset('single_gzip', data=rng.normal(size=(7, 5)).astype(np.float32), chunks=(7, 5), compression='gzip', shuffle=True)
    g.create_dataset('fixed_array', data=rng.normal(size=(17, 9)).astype(np.float32), chunks=(4, 4))
    g.create_dataset('fixed_array_gzip', data=rng.normal(size=(17, 9)), chunks=(4, 4), compression='gzip', fletcher32=True)
    g.create_dataset('fixed_array_paged', data=np.arange(2600, dtype=np.int16), chunks=(2,))
    d = g.create_dataset('fixed_array_paged_sparse', shape=(2100,), dtype=np.float32, chunks=(2,))
    d[2096:2098] = 5.0
    dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)                  # implicit index: early allocation, no filter
    dcpl.set_chunk((3, 4))
    dcpl.set_alloc_time(h5py.h5d.ALLOC_TIME_EARLY)
    did = h5py.h5d.create(g.id, b'implicit', h5py.h5t.NATIVE_FLOAT, h5py.h5s.create_simple((10, 9)), dcpl)
    h5py.Dataset(did)[...] = rng.normal(size=(10, 9)).astype(np.float32)
    g.create_dataset('unlimited', data=np.arange(10.), maxshape=(None,), chunks=(4,))      # extensible array: the reader must refuse
    a = f.create_group('many_attributes')
    for i in range(600):                                             # a depth-2 name index
        a.attrs['a%04d' % i] = np.int32(3 * i)
    a.attrs['names'] = np.array([b'abc', b'defg'])
    a.attrs['text'] = 'variable-length text'
    a.attrs['big_array'] = np.arange(16500, dtype=np.float32)       # > 64 KB: a 'huge' heap object


def write(path, cfg, w, flavour):
    kw = {'libver': 'latest'} if flavour in ('latest', 'dense') else {}
    with h5py.File(path, 'w', **kw) as f:
        mc = json.dumps(cfg)
        if flavour == 'keras2':                                   # Keras 2.x + h5py 2.x: bytes -> fixed-length strings
            f.attrs['keras_version'] = b'2.2.4'
            f.attrs['backend'] = b'tensorflow'
            f.attrs['model_config'] = mc.encode('utf8')
            f.attrs['training_config'] = json.dumps({'optimizer': {'class_name': 'Adam'}, 'loss': 'categorical_crossentropy'}).encode('utf8')
        else:                                                     # h5py 3.x with str: variable-length UTF-8 strings (global heap)
            f.attrs['keras_version'] = '2.11.0'
            f.attrs['backend'] = 'tensorflow'
            f.attrs['model_config'] = mc
        g = f.create_group('model_weights')
        g.attrs['layer_names'] = np.array([n.encode('utf8') for n in w])
        g.attrs['backend'] = b'tensorflow' if flavour == 'keras2' else 'tensorflow'
        for lname, ws in w.items():
            lg = g.create_group(lname)
            lg.attrs['weight_names'] = np.array([f'{lname}/{k}'.encode('utf8') for k in ws]) if ws else np.zeros((0,), 'S1')
            for k, arr in ws.items():
                if flavour == 'chunked':
                    lg.create_dataset(f'{lname}/{k}', data=arr, chunks=tuple(max(1, s // 2 + 1) for s in arr.shape), compression='gzip',
                                      compression_opts=4, shuffle=True)
                else:
                    lg.create_dataset(f'{lname}/{k}', data=arr)
        og = f.create_group('optimizer_weights')
        og.attrs['weight_names'] = np.array([b'Adam/iterations:0'])
        og.create_dataset('Adam/iterations:0', data=np.int64(1234))
        f.create_dataset('a_float64_matrix', data=np.arange(12, dtype=np.float64).reshape(3, 4) / 7)
        f.create_dataset('a_float16_vector', data=np.arange(5, dtype=np.float16))
        if flavour == 'dense':
            write_index_cases(f, np.random.default_rng(20250928))


def main():
    rng = np.random.default_rng(20250926)
    cfg, w = model(rng)
    expected = {}
    for flavour, fname in (('keras2', 'keras2_like.hdf5'), ('latest', 'tfkeras_latest.h5'), ('chunked', 'keras2_chunked_gzip.hdf5'),
                           ('dense', 'tfkeras_latest_dense.h5')):
        path = os.path.join(HERE, fname)
        write(path, *(dense_variant(cfg, w) if flavour == 'dense' else (cfg, w)), flavour)
        with h5py.File(path, 'r') as f:                           # what h5py reads back
            mc = f.attrs['model_config']
            expected[f'{fname}|model_config'] = np.array(mc.decode('utf8') if isinstance(mc, bytes) else str(mc))
            expected[f'{fname}|layer_names'] = np.asarray(f['model_weights'].attrs['layer_names'])
            for lname in f['model_weights']:
                names = f['model_weights'][lname].attrs['weight_names']
                expected[f'{fname}|{lname}|weight_names'] = np.asarray(names)
                for wn in names:
                    wn = wn.decode('utf8')
                    expected[f'{fname}|{lname}|{wn}'] = np.asarray(f['model_weights'][lname][wn])
            expected[f'{fname}|iterations'] = np.asarray(f['optimizer_weights/Adam/iterations:0'])
            expected[f'{fname}|f64'] = np.asarray(f['a_float64_matrix'])
            expected[f'{fname}|f16'] = np.asarray(f['a_float16_vector'])
            if flavour == 'dense':
                for k in f['chunk_indexes']:
                    expected[f'{fname}|chunk_indexes|{k}'] = np.asarray(f['chunk_indexes'][k])
                for k, v in f['many_attributes'].attrs.items():
                    expected[f'{fname}|many_attributes|{k}'] = np.asarray(v)
        print(fname, os.path.getsize(path), 'bytes')
    np.savez_compressed(os.path.join(HERE, 'keras_hdf5_expected.npz'), **expected)
    # a 24-band / 2-class sibling, so that a model directory with BOTH of the reference's file names can be staged
    # (tests/test_gpu_segmenter.py::test_segmenter_loads_keras_hdf5_files_from_the_model_dir)
    cfg2, w2 = model(np.random.default_rng(20250927), nmel=24, ncls=2)
    write(os.path.join(HERE, 'keras2_like_gender.hdf5'), cfg2, w2, 'keras2')


if __name__ == '__main__':
    main()
