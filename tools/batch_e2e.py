#!/usr/bin/env python3
"""End-to-end throughput of the drop-in path on real files (BASELINE.json configs[2] shape, scaled):
WAV files on a local filesystem -> Segmenter.batch_process -> CSV files.  Everything is inside the timed
region: RIFF parse, PCM16 host->device copy, device features + CNNs, host Viterbi, pandas CSV export.

    python tools/batch_e2e.py [--files 32] [--minutes 5] [--dir /dev/shm/iss_e2e]
"""
import argparse
import json
import os
import struct
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_wav(path, pcm):
    with open(path, 'wb') as f:
        f.write(b'RIFF' + struct.pack('<I', 36 + pcm.nbytes) + b'WAVEfmt ' + struct.pack('<IHHIIHH', 16, 1, 1, 16000, 32000, 2, 16)
                + b'data' + struct.pack('<I', pcm.nbytes))
        f.write(pcm.tobytes())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--files', type=int, default=32)
    ap.add_argument('--minutes', type=float, default=5.0)
    ap.add_argument('--dir', default='/dev/shm/iss_e2e')
    args = ap.parse_args()
    import torch
    import bench
    from inaspeechsegmenter_amd import Segmenter
    os.makedirs(args.dir, exist_ok=True)
    n = int(args.minutes * 60 * 16000)
    dev = torch.device('cuda', 0)
    lin, lout = [], []
    for i in range(args.files):
        p = os.path.join(args.dir, f'f{i:05d}.wav')
        if not os.path.exists(p):
            write_wav(p, bench.synth_recording(i, n, dev).cpu().numpy())
        lin.append(p)
        lout.append(os.path.join(args.dir, 'out', f'f{i:05d}.csv'))
    seg = Segmenter(vad_engine='smn', detect_gender=True, ffmpeg=None, models='synthetic')
    seg.batch_process(lin[:2], [o + '.warm' for o in lout[:2]])
    for o in lout:
        if os.path.exists(o):
            os.remove(o)
    t0 = time.perf_counter()
    t, nb, avg, lmsg = seg.batch_process(lin, lout)
    dt = time.perf_counter() - t0
    assert nb == args.files, [m for m in lmsg if m[1] != 0][:3]
    hours = args.files * args.minutes / 60.0
    print(json.dumps({"what": "batch_process on WAV files (decode + H2D + device + Viterbi + CSV)", "files": args.files,
                      "minutes_each": args.minutes, "wall_s": dt, "hours_of_audio_per_s": hours / dt,
                      "x_realtime": hours * 3600 / dt, "ms_per_file": dt / args.files * 1e3}))


if __name__ == '__main__':
    main()
