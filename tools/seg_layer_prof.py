#!/usr/bin/env python3
"""Per-layer HIP-event times of the two segmenter networks on one dense step of bench.py's recording
(iss_prof_get_row; rows of the two nets share the index space, both nets have the same program shape):
    python tools/seg_layer_prof.py [--minutes 60]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench                                                                   # noqa: E402
from inaspeechsegmenter_amd import Segmenter, _native as N                     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--minutes', type=float, default=60.0)
    args = ap.parse_args()
    import torch
    dev = torch.device('cuda', 0)
    seg = Segmenter(vad_engine='smn', detect_gender=True, ffmpeg=None, models='synthetic', device=0)
    n = int(args.minutes * 60 * 16000)
    pcm = bench.synth_recording(0, n, dev)
    torch.cuda.synchronize()
    seg.segment_device_pcm(pcm.data_ptr(), n, dense=True)
    seg.ctx.prof_enable(True)
    seg.ctx.prof_reset()
    seg.segment_device_pcm(pcm.data_ptr(), n, dense=True)
    for name, net in (('vad', seg.vad), ('gender', seg.gender)):
        prog = np.asarray(net.compiled.prog).reshape(-1, N.PROG_COLS) if hasattr(net, 'compiled') else None
        if prog is None:
            continue
        print(f"## {name}: program rows")
        for i, r in enumerate(prog):
            if r[N.C_OP] != N.OP_CONV:
                continue
            print(f"row {i}: {r[N.C_KH]}x{r[N.C_KW]} {r[N.C_CIN]}->{r[N.C_COUT]} {r[N.C_H]}x{r[N.C_W]} -> {r[N.C_HO]}x{r[N.C_WO]}")
    print("## per-row time of one dense step (both nets summed)")
    for i in range(16):
        ms, nl = seg.ctx.prof_get_row(i)
        if nl:
            print(f"row {i}: {ms:8.3f} ms in {nl} launches ({ms / nl * 1e3:8.1f} us per launch)")


if __name__ == '__main__':
    main()
