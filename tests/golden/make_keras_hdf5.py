#!/usr/bin/env python3
"""Writes the small HDF5 fixtures of tests/test_hdf5_reader.py with the real h5py / libhdf5 (run with an interpreter that has h5py,
e.g. /opt/conda/bin/python3.9 tests/golden/make_keras_hdf5.py): files laid out the way Keras 2.x / tf.keras save a Sequential model
(`model.save('x.hdf5')`: root attributes model_config / keras_version / backend, `model_weights/<layer>/<layer>/<weight>:0` datasets,
`layer_names` / `weight_names` attributes), in three flavours of the file format, plus what h5py itself reads back from them
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


def write(path, cfg, w, flavour):
    kw = {'libver': 'latest'} if flavour == 'latest' else {}
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


def main():
    rng = np.random.default_rng(20250926)
    cfg, w = model(rng)
    expected = {}
    for flavour, fname in (('keras2', 'keras2_like.hdf5'), ('latest', 'tfkeras_latest.h5'), ('chunked', 'keras2_chunked_gzip.hdf5')):
        path = os.path.join(HERE, fname)
        write(path, cfg, w, flavour)
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
        print(fname, os.path.getsize(path), 'bytes')
    np.savez_compressed(os.path.join(HERE, 'keras_hdf5_expected.npz'), **expected)
    # a 24-band / 2-class sibling, so that a model directory with BOTH of the reference's file names can be staged
    # (tests/test_gpu_segmenter.py::test_segmenter_loads_keras_hdf5_files_from_the_model_dir)
    cfg2, w2 = model(np.random.default_rng(20250927), nmel=24, ncls=2)
    write(os.path.join(HERE, 'keras2_like_gender.hdf5'), cfg2, w2, 'keras2')


if __name__ == '__main__':
    main()
