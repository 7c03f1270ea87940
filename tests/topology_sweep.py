#!/usr/bin/env python3
"""Parity AND throughput of the CNN stage over the topology sweep (tests/topologies.py), on one MI355X.

For every topology: both nets (smn VAD 21 mel / 3 classes, gender 24 mel / 2 classes) are evaluated on every 20 ms
slot of one hour of log-mel rows (179 999 overlapping windows, the dense mode of bench.py) -- time -> audio-hours/s of the
CNN stage and algorithmic TFLOP/s -- and 512 sampled slots are compared with the Keras-semantics oracle.

    python tests/topology_sweep.py --out profiles/r02_topology_sweep.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=None)
    ap.add_argument('--minutes', type=float, default=60.0)
    ap.add_argument('--only', nargs='*', default=None)
    ap.add_argument('--reps', type=int, default=3, help='timed runs per net; the fastest counts')
    args = ap.parse_args()
    from inaspeechsegmenter_amd import _native, keras_model as KM, segmenter as S, tables
    import topologies as TP
    from test_gpu_topologies import _mspec, _oracle_probs
    ctx = _native.Context(0)
    rng = np.random.default_rng(7)
    T = int(args.minutes * 6000) - 2
    mspec = _mspec(rng, T)
    ctx.set_mspec(mspec)
    rows = S._window_rows(T)
    hours = args.minutes / 60.0
    res = []
    for name in (args.only or sorted(TP.SPECS)):
        entry = {'topology': name}
        t_total, fl_total, worst = 0.0, 0.0, 0.0
        for net, (layers, shp) in sorted(TP.nets(name).items()):
            comp = KM.compile_layers(layers, shp)
            ctx.cnn_load(5, comp)
            ctx.cnn_probs(5, rows)                                      # warm up at full size (code objects, workspace growth)
            ctx.synchronize()
            dt = None
            for _ in range(args.reps):                                  # the fastest of `reps` runs: one run per net caught clock /
                t0 = time.perf_counter()                                # host hiccups of up to 15 % on single rows
                probs, fin = ctx.cnn_probs(5, rows)
                d = time.perf_counter() - t0
                dt = d if dt is None else min(dt, d)
            idx = np.sort(rng.integers(0, len(rows), 512))
            ref, rfin = _oracle_probs(layers, mspec, shp[1], rows[idx])
            err = float(np.abs(probs[idx] - ref).max())
            assert np.array_equal(fin[idx], rfin)
            nparams = int(sum(np.asarray(L[k]).size for L in layers for k in ('W', 'b', 'gamma', 'beta') if L.get(k) is not None))
            entry[net] = {'ms': dt * 1e3, 'mflop_per_slot': comp.flops_per_sample / 1e6, 'params': nparams,
                          'tflops_algorithmic': comp.flops_per_sample * len(rows) / dt / 1e12, 'max_abs_dprob': err}
            t_total += dt
            fl_total += comp.flops_per_sample * len(rows)
            worst = max(worst, err)
        entry['cnn_stage_hours_per_s'] = hours / t_total
        entry['x_realtime'] = hours * 3600 / t_total
        entry['tflops_algorithmic'] = fl_total / t_total / 1e12
        entry['max_abs_dprob'] = worst
        entry['ok'] = bool(worst < 1e-4 and entry['x_realtime'] >= 2000)
        print(json.dumps(entry), flush=True)
        res.append(entry)
    if args.out:
        with open(args.out, 'w') as f:
            json.dump({'workload': f'{args.minutes:g} min of log-mel rows, both nets dense on every 20 ms slot, CNN stage only '
                                   '(wall time of iss_cnn_probs incl. result copy; the fastest of ' + str(args.reps) + ' runs per net)', 'results': res}, f, indent=1)
    bad = [e['topology'] for e in res if not e['ok']]
    print('all topologies ok' if not bad else f'FAILED: {bad}')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
