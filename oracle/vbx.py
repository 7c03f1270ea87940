"""Oracle: VBx 64-band fbank front end + x-vector window loop + ResNet-101.
Test infrastructure only.

Follows /root/reference/inaSpeechSegmenter/features_vbx.py (framing 12-15,
preemphasis 27-28, mel_fbank_mx 31-59, fbank_htk 62-120, povey_window 123-124,
add_dither 127-128, cmvn_floating_kaldi 131-149), vbx_segmenter.py
(get_features 72-89, VBxExtractor.__call__ 217-246) and resnet.py (Bottleneck
48-75, ResNet 78-130, ResNet101 133-135).  All front-end arithmetic is float64
as in the reference; the result is cast to float32 at vbx_segmenter.py:88.
"""
import numpy as np

SR = 16000
WINLEN_S = 400
HOP_S = 160
NFFT = 512
NCH = 64
STEP = 24      # vbx_segmenter.py:21
WINLEN = 144   # vbx_segmenter.py:22
EMBED = 256


def mel(x):
    return 1127. * np.log(1. + x / 700.)            # features_vbx.py:23-24


def mel_inv(x):
    return (np.exp(x / 1127.) - 1.) * 700.          # features_vbx.py:19-20


def mel_bank(nfft=NFFT, fs=SR, nch=NCH, lo=20.0, hi=7600.0):
    """(nfft/2+1, nch) float64, features_vbx.py:31-59 with htk_bug=False."""
    fbin = mel(np.arange(nfft / 2 + 1, dtype=float) * fs / nfft)
    cbin = np.linspace(mel(lo), mel(hi), nch + 2)
    cind = np.floor(mel_inv(cbin) / fs * nfft).astype(int) + 1
    m = np.zeros((len(fbin), nch))
    for i in range(nch):
        m[cind[i]:cind[i + 1], i] = (cbin[i] - fbin[cind[i]:cind[i + 1]]) / (cbin[i] - cbin[i + 1])
        m[cind[i + 1]:cind[i + 2], i] = (cbin[i + 2] - fbin[cind[i + 1]:cind[i + 2]]) / (cbin[i + 2] - cbin[i + 1])
    return m


def povey_window(n=WINLEN_S):
    return np.power(0.5 - 0.5 * np.cos(np.linspace(0, 2 * np.pi, n)), 0.85)   # :123-124


def dither_stream(n, seed=3):
    """The n uniform doubles `np.random.seed(3); np.random.rand(n)` yields
    (vbx_segmenter.py:84, features_vbx.py:127-128)."""
    return np.random.RandomState(seed).rand(n)


def prepare_signal(signal, u=None):
    """vbx_segmenter.py:84-86: int truncation, +-8 LSB dither, reflect pad 120/200."""
    signal = np.asarray(signal, dtype=np.float64)
    if u is None:
        u = dither_stream(len(signal))
    x = (signal * 2 ** 15).astype(int) + 8 * (u * 2 - 1)
    return np.r_[x[119::-1], x, x[-1:-201:-1]]


def fbank(seg, window=None, bank=None):
    """fbank_htk(seg, window, 240, bank, USEPOWER=True, ZMEANSOURCE=True), :62-120."""
    window = povey_window() if window is None else window
    bank = mel_bank() if bank is None else bank
    t = (len(seg) - WINLEN_S) // HOP_S + 1
    idx = np.arange(WINLEN_S)[None, :] + HOP_S * np.arange(t)[:, None]
    x = seg.astype("float")[idx]                                   # :99
    x -= x.mean(axis=1)[:, np.newaxis]                             # :100-101
    x = x - np.c_[x[..., :1], x[..., :-1]] * 0.97                  # :104-105, 27-28
    x *= window                                                    # :106
    x = np.fft.rfft(x, NFFT)                                       # :109
    x = x.real ** 2 + x.imag ** 2                                  # :110
    return np.log(np.maximum(1.0, np.dot(x, bank)))                # :113


def cmn_floating(x, LC=150, RC=149):
    """cmvn_floating_kaldi(x, LC, RC, norm_vars=False), features_vbx.py:131-149."""
    N, dim = x.shape
    win_len = min(len(x), LC + RC + 1)
    win_start = np.maximum(np.minimum(np.arange(-LC, N - LC), N - win_len), 0)
    f = np.r_[np.zeros((1, dim)), np.cumsum(x, 0)]
    return x - (f[win_start + win_len] - f[win_start]) / win_len


def get_features(signal, u=None):
    """vbx_segmenter.py:72-89 -> (T,64) float32."""
    return cmn_floating(fbank(prepare_signal(signal, u))).astype(np.float32)


def window_list(T):
    """(start, stop) frame windows of VBxExtractor.__call__, vbx_segmenter.py:222-243."""
    wins = []
    start = 0
    for start in range(0, T - WINLEN, STEP):
        wins.append((start, start + WINLEN))
    if T - start - STEP >= 10:
        wins.append((start + STEP, T))
    return wins


# ---------------------------------------------------------------- ResNet-101
def resnet101_param_shapes(feat_dim=64, embed_dim=EMBED, m=32):
    """Ordered (name, shape) list equal to resnet.py's state_dict layout
    (conv weights OIHW, BN weight/bias/running_mean/running_var)."""
    shapes = []

    def bn(prefix, c):
        for s in ('weight', 'bias', 'running_mean', 'running_var'):
            shapes.append((f'{prefix}.{s}', (c,)))

    shapes.append(('conv1.weight', (m, 1, 3, 3))); bn('bn1', m)
    in_planes = m
    for li, (planes, nblocks, stride) in enumerate(zip((m, 2 * m, 4 * m, 8 * m), (3, 4, 23, 3), (1, 2, 2, 2)), 1):
        for bi in range(nblocks):
            s = stride if bi == 0 else 1
            p = f'layer{li}.{bi}'
            shapes.append((f'{p}.conv1.weight', (planes, in_planes, 1, 1))); bn(f'{p}.bn1', planes)
            shapes.append((f'{p}.conv2.weight', (planes, planes, 3, 3))); bn(f'{p}.bn2', planes)
            shapes.append((f'{p}.conv3.weight', (4 * planes, planes, 1, 1))); bn(f'{p}.bn3', 4 * planes)
            if s != 1 or in_planes != 4 * planes:
                shapes.append((f'{p}.shortcut.0.weight', (4 * planes, in_planes, 1, 1))); bn(f'{p}.shortcut.1', 4 * planes)
            in_planes = 4 * planes
    shapes.append(('embedding.weight', (embed_dim, int(feat_dim / 8) * m * 16 * 4)))
    shapes.append(('embedding.bias', (embed_dim,)))
    return shapes


def resnet101_random_params(seed=0):
    """Seeded, numerically tame synthetic weights (He-scaled convs, BN stats near
    identity) keyed like resnet.py's state_dict."""
    rng = np.random.default_rng(seed)
    params = {}
    for name, shape in resnet101_param_shapes():
        if name.endswith('running_var'):
            a = rng.uniform(0.5, 1.5, shape)
        elif name.endswith('running_mean'):
            a = rng.normal(0, 0.1, shape)
        elif '.bn' in name or name.startswith('bn') or 'shortcut.1' in name:
            a = rng.uniform(0.8, 1.2, shape) if name.endswith('weight') else rng.normal(0, 0.1, shape)
        elif name.endswith('bias'):
            a = rng.normal(0, 0.05, shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            a = rng.normal(0, np.sqrt(1.0 / fan_in), shape)
        params[name] = a.astype(np.float32)
    return params


def resnet101_forward(params, fea_bft, eps=1e-5):
    """fea_bft: (B, 64, T) float32 (feature-major, vbx_segmenter.py:265) -> (B,256).
    torch-CPU functional restatement of resnet.py:115-130."""
    import torch
    import torch.nn.functional as F
    P = {k: torch.from_numpy(np.asarray(v)) for k, v in params.items()}

    def bn(x, p):
        return F.batch_norm(x, P[p + '.running_mean'], P[p + '.running_var'], P[p + '.weight'], P[p + '.bias'],
                            False, 0.0, eps)

    with torch.no_grad():
        x = torch.from_numpy(np.ascontiguousarray(fea_bft, dtype=np.float32)).unsqueeze(1)
        out = F.relu(bn(F.conv2d(x, P['conv1.weight'], padding=1), 'bn1'))
        in_planes = 32
        for li, (planes, nblocks, stride) in enumerate(zip((32, 64, 128, 256), (3, 4, 23, 3), (1, 2, 2, 2)), 1):
            for bi in range(nblocks):
                s = stride if bi == 0 else 1
                p = f'layer{li}.{bi}'
                o = F.relu(bn(F.conv2d(out, P[p + '.conv1.weight']), p + '.bn1'))
                o = F.relu(bn(F.conv2d(o, P[p + '.conv2.weight'], stride=s, padding=1), p + '.bn2'))
                o = bn(F.conv2d(o, P[p + '.conv3.weight']), p + '.bn3')
                sc = out
                if (p + '.shortcut.0.weight') in P:
                    sc = bn(F.conv2d(out, P[p + '.shortcut.0.weight'], stride=s), p + '.shortcut.1')
                out = F.relu(o + sc)
                in_planes = 4 * planes
        mean = torch.mean(out, dim=-1)
        meansq = torch.mean(out * out, dim=-1)
        std = torch.sqrt(meansq - mean ** 2 + 1e-10)
        v = torch.cat((torch.flatten(mean, start_dim=1), torch.flatten(std, start_dim=1)), 1)
        return F.linear(v, P['embedding.weight'], P['embedding.bias']).numpy()
