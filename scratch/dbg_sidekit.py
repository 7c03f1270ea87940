import numpy as np, sys, os
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from conftest import read_wav_int16, GOLDEN
from inaspeechsegmenter_amd import _native, tables
c = _native.Context(0)
c.sidekit_tables(tables.sidekit_window(), tables.sidekit_melbank())
g = np.load(os.path.join(GOLDEN, 'sidekit_feats.npz'))
pcm = read_wav_int16(os.path.join(GOLDEN, 'musanmix.wav'))
c.set_signal(pcm); T = c.sidekit(); m = c.get_mspec(); l = c.get_loge()
ref = g['musanmix_mspec']
d = np.abs(m - ref)
print('max', d.max(), 'mean', d.mean())
idx = np.argsort(d.ravel())[::-1][:20]
for i in idx:
    t, b = divmod(i, 24)
    print(t, b, m[t, b], ref[t, b], d[t, b], 'rowmax', ref[t].max())
print('per band max', d.max(axis=0))
print('hist', np.histogram(d.ravel(), bins=[0,1e-6,1e-5,1e-4,1e-3,1])[0])
