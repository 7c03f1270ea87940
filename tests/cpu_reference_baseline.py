#!/usr/bin/env python3
"""SURVEY.md section 8(d) "CPU baseline beside it", run in the BUILD container (the only place /root/reference exists):
the reference's own modules imported unmodified from /root/reference, timed on the synthetic recording bench.py uses.

  (1) feature path: sidekit_mfcc.mfcc + the oracle's restated _get_patches + pyannote_viterbi.viterbi_decoding,
      single process and N-process file-parallel
  (2) CNN forward: torch-CPU Keras-semantics oracle at batch_size 32 and 1024 (TensorFlow is not installable here;
      seeded stand-in weights of the reference's I/O contract)
  (3) vbx: features_vbx functions glued as vbx_segmenter.get_features:72-89 does + resnet.py ResNet101 on torch-CPU,
      batch 1 (reference behaviour, vbx_segmenter.py:217-246) and batch 64

Writes profiles/<tag>_cpu_reference_baseline.json.  Nothing here is used by the product, the tests or bench.py."""
import importlib.util
import json
import multiprocessing as mp
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # tests/ -> repo root
sys.path.insert(0, ROOT)
REF = '/root/reference/inaSpeechSegmenter'
FS = 16000


def ref_module(name):
    spec = importlib.util.spec_from_file_location('ref_' + name, os.path.join(REF, name + '.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def recording(rank, nsec):
    import torch
    import bench
    return bench.synth_recording(rank, nsec * FS, torch.device('cpu')).numpy()


_BARRIER = None


def _init(barrier):
    global _BARRIER
    _BARRIER = barrier


def feature_path(args):
    """One file through the reference feature path + bookkeeping (no CNN): returns (seconds of audio, wall seconds, legs).
    In the pool the workers synthesise their file first and start the timed part together (barrier)."""
    rank, nsec = args
    from oracle import segment as oseg
    sk = ref_module('sidekit_mfcc')
    vit = ref_module('pyannote_viterbi')
    vu = ref_module('viterbi_utils')
    pcm = recording(rank, nsec)
    sig = (pcm / 32768.0).astype(np.float32)
    legs = {}
    if _BARRIER is not None:
        _BARRIER.wait()
    t0 = time.perf_counter()
    with np.errstate(divide='ignore'), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        _, loge, _, mspec = sk.mfcc(sig, get_mspec=True)
        t1 = time.perf_counter()
        legs['mfcc_s'] = t1 - t0
        for h in (21, 24):
            oseg.get_patches(mspec[:, :h].copy(), 68, 2)
        t2 = time.perf_counter()
        legs['get_patches_21_24_s'] = t2 - t1
        # the three smoothing passes at the sizes the pipeline runs them (energy: T frames x 2; smn: T/2 x 3; gender: T/2 x 2)
        T = len(loge)
        thr = np.mean(loge[np.isfinite(loge)]) + np.log(0.03)
        vit.viterbi_decoding(vu.pred2logemission(loge > thr), vu.log_trans_exp(150, cost0=-5))
        rng = np.random.default_rng(0)
        for k, arg in ((3, 80), (2, 80)):
            p = rng.dirichlet(np.ones(k), T // 2)
            vit.viterbi_decoding(np.log(p), vu.diag_trans_exp(arg, k))
        legs['viterbi_x3_s'] = time.perf_counter() - t2
    return nsec, time.perf_counter() - t0, legs


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
    nsec = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    ncpu = os.cpu_count() or 1
    out = {"host": {"nproc": ncpu, "where": "build container (no GPU); the GPU box has no /root/reference",
                    "cpu": next((l.split(':')[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')), '?')},
           "input": f"bench.synth_recording(rank, {nsec} s): the generator of SURVEY 8(d), PCM16 -> float32 / 32768"}
    # (1) feature path
    a, t, legs = feature_path((0, nsec))
    out["feature_path_single_process"] = {"audio_s": a, "wall_s": t, "x_realtime": a / t, "hours_per_s": a / 3600 / t, "legs_s": legs,
                                          "what": "unmodified sidekit_mfcc.mfcc(get_mspec=True) + oracle get_patches (21 and 24 bands) + "
                                                  "unmodified pyannote_viterbi.viterbi_decoding x3 (energy, smn, gender sizes)"}
    cx = mp.get_context('spawn')
    with cx.Pool(ncpu, initializer=_init, initargs=(cx.Barrier(ncpu),)) as pool:
        res = pool.map(feature_path, [(r, nsec) for r in range(ncpu)], chunksize=1)
    t = max(r[1] for r in res)
    out["feature_path_file_parallel"] = {"processes": ncpu, "audio_s": sum(r[0] for r in res), "wall_s": t,
                                         "x_realtime": sum(r[0] for r in res) / t, "hours_per_s": sum(r[0] for r in res) / 3600 / t,
                                         "note": "one file per process, started together after synthesis (barrier); wall = the slowest "
                                                 "process; numpy's own threads are left at their default in every process"}
    # (2) CNN forward, torch-CPU stand-in
    import torch
    torch.set_num_threads(ncpu)
    from oracle import keras_cnn as ocnn
    from inaspeechsegmenter_amd import keras_model as KM
    cnn = {}
    for name, nmel, ncls in (('smn', 21, 3), ('gender', 24, 2)):
        layers, _ = KM.synthetic_ina_like(nmel, ncls, seed=1)
        layers = KM.layers_for_oracle(layers) if hasattr(KM, 'layers_for_oracle') else layers
        x = np.random.default_rng(0).normal(0, 1, (2048, 68, nmel, 1)).astype(np.float32)
        for bs in (32, 1024):
            ocnn.forward(layers, x[:bs], batch_size=bs)
            t0 = time.perf_counter()
            ocnn.forward(layers, x, batch_size=bs)
            dt = time.perf_counter() - t0
            cnn[f'{name}_batch{bs}'] = {"slots_per_s": len(x) / dt, "x_realtime_if_every_slot": len(x) / dt * 0.02}
    out["cnn_forward_torch_cpu"] = {"threads": ncpu, "label": "TensorFlow unavailable; torch-CPU stand-in for the reference TF/CPU path "
                                    "(oracle/keras_cnn.py, seeded stand-in weights, (68,21,1)->3 and (68,24,1)->2)", **cnn}
    # (3) vbx
    fv = ref_module('features_vbx')
    rn = ref_module('resnet')
    vsec = min(nsec, 120)
    sig = recording(0, vsec).astype(np.float64) / 32768.0

    def get_features(signal, LC=150, RC=149):            # the glue of vbx_segmenter.get_features:72-89 around the reference's functions
        noverlap, winlen = 240, 400
        window = fv.povey_window(winlen)
        fbank_mx = fv.mel_fbank_mx(winlen, FS, NUMCHANS=64, LOFREQ=20.0, HIFREQ=7600, htk_bug=False)
        np.random.seed(3)
        signal = fv.add_dither((signal * 2 ** 15).astype(int))
        seg = np.r_[signal[noverlap // 2 - 1::-1], signal, signal[-1:-winlen // 2 - 1:-1]]
        fea = fv.fbank_htk(seg, window, noverlap, fbank_mx, USEPOWER=True, ZMEANSOURCE=True)
        return fv.cmvn_floating_kaldi(fea, LC, RC, norm_vars=False).astype(np.float32)

    t0 = time.perf_counter()
    fea = get_features(sig)
    tf = time.perf_counter() - t0
    model = rn.ResNet101(feat_dim=64, embed_dim=256).eval()
    starts = list(range(0, len(fea) - 144, 24))
    vb = {"get_features": {"audio_s": vsec, "wall_s": tf, "x_realtime": vsec / tf}}
    with torch.no_grad():
        for bs, nwin in ((1, 16), (64, 128)):
            xs = np.stack([fea[s:s + 144].T for s in starts[:nwin]]).astype(np.float32)
            model(torch.from_numpy(xs[:bs]))
            t0 = time.perf_counter()
            for i in range(0, nwin, bs):
                model(torch.from_numpy(xs[i:i + bs]))
            dt = time.perf_counter() - t0
            per = dt / nwin
            vb[f'resnet101_batch{bs}'] = {"windows": nwin, "ms_per_window": per * 1e3, "x_realtime": 0.24 / per,
                                          "x_realtime_with_features": 1.0 / (per / 0.24 + tf / vsec)}
    out["vbx"] = {"threads": ncpu, "what": "reference features_vbx functions (glued as vbx_segmenter.get_features does) + reference "
                  "resnet.py ResNet101(64, 256), random init, torch-CPU; one window per 0.24 s of audio", **vb}
    path = os.path.join(ROOT, 'profiles', f'{tag}_cpu_reference_baseline.json')
    json.dump(out, open(path, 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
