"""Prototype: fit the last dense layer of the seeded stand-in nets on generator ground truth + musanmix golden labels."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inaspeechsegmenter_amd import keras_model as KM
from inaspeechsegmenter_amd.io import decode_pcm
from oracle import sidekit as osk, segment as oseg, keras_cnn as ocnn

FS = 16000


def synth_plan(file_index, n_samples):
    """CPU twin of bench.synth_recording: same plan (numpy rng), noise from a CPU torch generator; returns pcm int16 and
    per-sample kind (0 silence 1 noise 2 voiced 3 music) and f0."""
    rng = np.random.default_rng(20250926 + file_index)
    gen = torch.Generator(device='cpu'); gen.manual_seed(20250926 + file_index)
    out = torch.zeros(n_samples, dtype=torch.float32)
    kinds = np.zeros(n_samples, np.int8); f0s = np.zeros(n_samples, np.float32)
    pos = 0
    while pos < n_samples:
        dur = int(rng.uniform(2.0, 20.0) * FS)
        kind = rng.choice(4, p=[0.1, 0.2, 0.4, 0.3])
        f0 = float(rng.choice([110.0, 200.0]))
        nch = int(rng.integers(3, 6))
        chord = rng.uniform(130.0, 1000.0, size=5)
        trem = float(rng.uniform(0.2, 1.0))
        n = min(dur, n_samples - pos)
        kinds[pos:pos + n] = kind; f0s[pos:pos + n] = f0
        if kind == 1:
            out[pos:pos + n] = torch.randn(n, generator=gen, dtype=torch.float32) * 10 ** (-30 / 20)
        elif kind >= 2:
            t = torch.arange(n, dtype=torch.float32) / FS
            if kind == 2:
                x = torch.zeros(n)
                for k in range(1, 31):
                    if f0 * k < 7600:
                        x += torch.sin(2 * np.pi * f0 * k * t) / k
                x *= 0.6 + 0.4 * torch.sin(2 * np.pi * 4.0 * t)
                level = 10 ** (-20 / 20)
            else:
                x = torch.zeros(n)
                for f in chord[:nch]:
                    x += torch.sin(2 * np.pi * float(f) * t)
                x *= 0.8 + 0.2 * torch.sin(2 * np.pi * trem * t)
                level = 10 ** (-18 / 20)
            x *= level / torch.sqrt(torch.mean(x * x) + 1e-20)
            out[pos:pos + n] = x
        pos += n
    pcm = torch.clamp(torch.round(out * 32768.0), -32768, 32767).to(torch.int16).numpy()
    return pcm, kinds, f0s


def trunk_features(layers, patches):
    """Output of everything before the last dense layer."""
    head = layers[-1]
    assert head['type'] == 'dense' and head['activation'] == 'softmax'
    body = list(layers[:-1]) + [dict(type='dense', W=np.eye(head['W'].shape[0], dtype=np.float32), b=None, activation='linear')]
    return ocnn.forward(body, patches, batch_size=1024)


def slot_patches(mspec, nmel):
    p, fin = oseg.get_patches(mspec[:, :nmel].copy() if nmel != 24 else mspec, 68, 2)
    return p.reshape(len(p), 68, nmel, 1), fin


def fit_head(F, y, C, ridge=1e-2, margin=8.0):
    X = np.concatenate((F, np.ones((len(F), 1))), axis=1).astype(np.float64)
    Tm = np.full((len(F), C), -margin / 2); Tm[np.arange(len(F)), y] = margin / 2
    A = X.T @ X + ridge * len(F) * np.eye(X.shape[1])
    Wb = np.linalg.solve(A, X.T @ Tm)
    return Wb[:-1].astype(np.float32), Wb[-1].astype(np.float32)


t0 = time.time()
torch.set_num_threads(8)
train = []
for fi, nsec in ((0, 300), (1, 200)):
    pcm, kinds, f0s = synth_plan(fi, nsec * FS)
    sig = (pcm / 32768.0).astype(np.float32)
    mspec, loge, difflen = osk.media2feats(sig)
    train.append((mspec, kinds, f0s))
print('feats', time.time() - t0)

# musanmix golden
mus = decode_pcm('tests/golden/musanmix.wav', ffmpeg=None)
msig = (mus / 32768.0).astype(np.float32) if mus.dtype == np.int16 else mus
mm, ml, md = osk.media2feats(msig)
gold = [l.strip().split('\t') for l in open('tests/golden/musanmix-smn-gender.csv')][1:]
print(gold)

results = {}
for name, nmel, labels, seed in (('vad', 21, ('speech', 'music', 'noise'), 1), ('gender', 24, ('female', 'male'), 2)):
    layers, in_shape = KM.synthetic_ina_like(nmel, len(labels), seed=seed)
    Fs, ys = [], []
    for mspec, kinds, f0s in train:
        P, fin = slot_patches(mspec, nmel)
        nsl = len(P)
        centre = np.clip((np.arange(nsl) * 320 + 160), 0, len(kinds) - 1)
        k = kinds[centre]; f = f0s[centre]
        if name == 'vad':
            y = np.select([k == 2, k == 3, k == 1], [0, 1, 2], -1)
        else:
            y = np.where(k == 2, np.where(f == 200.0, 0, 1), -1)
        # keep slots whose whole 68-frame window lies inside one segment (clean targets)
        lo = np.clip((np.arange(nsl) - 17) * 320, 0, len(kinds) - 1); hi = np.clip((np.arange(nsl) + 17) * 320, 0, len(kinds) - 1)
        clean = (kinds[lo] == k) & (kinds[hi] == k) & fin & (y >= 0)
        sel = np.flatnonzero(clean)[::3]
        Fs.append(trunk_features(layers, P[sel])); ys.append(y[sel])
    # musanmix with golden labels
    P, fin = slot_patches(mm, nmel)
    y = np.full(len(P), -1)
    for lab, a, b in gold:
        a, b = int(round(float(a) / .02)), int(round(float(b) / .02))
        if name == 'vad':
            c = {'music': 1, 'noise': 2, 'male': 0, 'female': 0, 'speech': 0}.get(lab, -1)
        else:
            c = {'male': 1, 'female': 0}.get(lab, -1)
        y[a + 20:b - 20] = c
    sel = np.flatnonzero((y >= 0) & fin)[::2]
    Fm = trunk_features(layers, P[sel]); ym = y[sel]
    print(name, 'train sizes', [len(v) for v in ys], len(ym), 't', time.time() - t0)
    F = np.concatenate(Fs + [Fm]); Y = np.concatenate(ys + [ym])
    W, b = fit_head(F, Y, len(labels))
    logits = F @ W + b
    acc = (logits.argmax(1) == Y).mean()
    print(name, 'train acc', acc, 'hist', np.bincount(Y), 'feature scale', np.abs(F).mean(), 'logit margin mean', np.sort(logits, 1)[:, -1].mean() - np.sort(logits, 1)[:, -2].mean())
    layers2 = list(layers[:-1]) + [dict(type='dense', W=W, b=b, activation='softmax')]
    results[name] = layers2

# evaluate the pipeline with the oracle on a held-out generator file and musanmix
def run(sig):
    mspec, loge, difflen = osk.media2feats(sig)
    l0 = oseg.energy_seglist(loge, 0.03)
    l1, rv = oseg.dnn_segment('smn', lambda b: ocnn.forward(results['vad'], b), mspec, l0, difflen, return_raw=True)
    l2, rg = oseg.dnn_segment('gender', lambda b: ocnn.forward(results['gender'], b), mspec, l1, difflen, return_raw=True)
    return l0, l1, l2, rv, rg

for title, sig in (('musanmix', msig), ('gen file 7 (120 s)', (synth_plan(7, 120 * FS)[0] / 32768.0).astype(np.float32))):
    l0, l1, l2, rv, rg = run(sig)
    print(title, 'energy segs', len(l0), 'after vad', len(l1), 'after gender', len(l2))
    print(' vad argmax hist', np.bincount(rv.argmax(1), minlength=3), 'gender argmax hist', np.bincount(rg.argmax(1), minlength=2) if rg is not None and len(rg) else None)
    print(' ', [(l, round(a * .02, 2), round(b * .02, 2)) for l, a, b in l2][:40])
np.savez('/tmp/calib_heads.npz', vad_W=results['vad'][-1]['W'], vad_b=results['vad'][-1]['b'], gen_W=results['gender'][-1]['W'], gen_b=results['gender'][-1]['b'])
print('total', time.time() - t0)
