#!/usr/bin/env python3
"""bench.py -- hours-of-audio segmented per second (smn + gender, 16 kHz mono) on N MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched as
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU,
RCCL).  Rank 0 prints ONE JSON line.

Workloads
  segmenter (default at EVERY --gpus N)  the configuration the metric is quoted on, below: one recording per rank (weak scaling,
                                     "replicas" of configs[1]'s input: a single file is never split over GPUs), one all-gather
                                     of the segment tables per step at N > 1 -- the same code path at N = 1 and N = 8
  batch                              BASELINE.json configs[2]: 128 x 5 min WAV files in /dev/shm through
                                     Segmenter.batch_process -- decode, H2D, device, Viterbi and CSV export inside the timed region
  archive                            BASELINE.json configs[3] shape: 3-minute WAV files, file-parallel over the ranks through
                                     archive.segment_archive, ONE RCCL all-gather of the segment tables per step (C-ABI
                                     iss_allgather_segments); --files-per-gpu of them per rank (weak scaling)
  vbx                                BASELINE.json configs[4]
The default line carries the other single-GPU configurations as driver-timed `companions`: `archive` at every N (so that the
file-based configs[3] path has a 1 -> N curve of its own), `batch` and `vbx` at N = 1 (--no-companions skips them).

segmenter workload (BASELINE.json configs[1] input, with the metric's smn + gender nets): every rank holds
ONE 1 h synthetic 16 kHz mono PCM16 recording, already resident in HBM when the timed region
starts (SURVEY.md section 8(d) generator: silence / -30 dBFS noise / harmonic "voiced" source /
sustained chords, seeded with 20250926 + file index).  One step = one pass of the whole hot path
over that recording:

    PCM16 in HBM -> SIDEKIT log-mel kernel -> log-energy to host -> energy Viterbi (compiled host)
                 -> VAD CNN on every 20 ms slot -> per-segment Viterbi
                 -> gender CNN on every 20 ms slot -> per-segment Viterbi
                 -> slot-unit segment table -> (N > 1) one RCCL all-gather of the tables

"dense" mode: both networks are evaluated on 100 % of the slots, which is the most work the
reference semantics can ever require (it runs the VAD net on `energy` slots and the gender net on
`speech` slots only, segmenter.py:157-159) and makes the device work independent of what the
seeded stand-in weights decide.  The reference-semantics rate is reported next to it in `config`.
Weights are seeded stand-ins of the reference's I/O contract ((68,nmel,1) -> softmax, ~1.25 M
parameters, Dockerfile:18): the real Keras files are release assets that cannot be fetched here.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FS = 16000
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # dense f32-input MFMA peak (v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16); AMD's 5 PF figure is 2:1 sparse


# ------------------------------------------------------------------------------ synthetic audio
def synth_plan(file_index, n_samples):
    """SURVEY.md 8(d) plan of one recording: [(kind, start, n, f0, chord[:nch], trem)] with kind 0 silence / 1 noise /
    2 voiced / 3 music, from numpy's default_rng(20250926 + file_index) (durations U(2,20) s, kinds 10/20/40/30 %)."""
    rng = np.random.default_rng(20250926 + file_index)
    plan, pos = [], 0
    while pos < n_samples:
        dur = int(rng.uniform(2.0, 20.0) * FS)
        kind = int(rng.choice(4, p=[0.1, 0.2, 0.4, 0.3]))
        f0 = float(rng.choice([110.0, 200.0]))
        nch = int(rng.integers(3, 6))
        chord = rng.uniform(130.0, 1000.0, size=5)
        trem = float(rng.uniform(0.2, 1.0))
        n = min(dur, n_samples - pos)
        plan.append((kind, pos, n, f0, [float(f) for f in chord[:nch]], trem))
        pos += n
    return plan


def synth_recording(file_index, n_samples, device):
    """concatenation of U(2,20) s segments; kinds: exact silence 10 %, Gaussian
    noise -30 dBFS 20 %, voiced (<= 30 harmonics of f0 in {110,200} Hz, 1/k roll-off, 4 Hz AM,
    -20 dBFS) 40 %, music (3-5 sustained sines with slow tremolo, -18 dBFS) 30 %.  Returns a torch
    int16 tensor on `device`.  The plan (kinds, durations, frequencies) comes from `synth_plan`; the noise samples from a
    torch generator on `device` with the same seed (CPU and GPU generators give different, equally distributed samples)."""
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(20250926 + file_index)
    out = torch.zeros(n_samples, dtype=torch.float32, device=device)
    for kind, pos, n, f0, chord, trem in synth_plan(file_index, n_samples):
        if kind == 1:
            out[pos:pos + n] = torch.randn(n, generator=gen, device=device, dtype=torch.float32) * 10 ** (-30 / 20)
        elif kind >= 2:
            t = torch.arange(n, device=device, dtype=torch.float32) / FS
            if kind == 2:
                x = torch.zeros(n, device=device)
                for k in range(1, 31):
                    if f0 * k < 7600:
                        x += torch.sin(2 * np.pi * f0 * k * t) / k
                x *= 0.6 + 0.4 * torch.sin(2 * np.pi * 4.0 * t)
                level = 10 ** (-20 / 20)
            else:
                x = torch.zeros(n, device=device)
                for f in chord:
                    x += torch.sin(2 * np.pi * float(f) * t)
                x *= 0.8 + 0.2 * torch.sin(2 * np.pi * trem * t)
                level = 10 ** (-18 / 20)
            x *= level / torch.sqrt(torch.mean(x * x) + 1e-20)
            out[pos:pos + n] = x
    return torch.clamp(torch.round(out * 32768.0), -32768, 32767).to(torch.int16)


# ------------------------------------------------------------------------------ per-kernel roofline
def _pmc_static():
    """profiles/pmc_latest.json (tools/pmc_report.py): per-kernel FETCH_SIZE / WRITE_SIZE of the latest committed rocprofv3
    --pmc passes.  PMC counters need rocprofv3 passes of their own, so HBM traffic cannot be measured inside a bench run:
    every figure taken from here is labelled with the run it came from."""
    try:
        return json.load(open(os.path.join(ROOT, 'profiles', 'pmc_latest.json')))
    except Exception:                                       # noqa: BLE001
        return {}


def _pmc_kernel(pj, inst_name):
    """per_kernel entry of pmc_latest.json for an instantiation name as the library spells it (no blanks, no namespace)"""
    want = inst_name.replace(' ', '')
    for k, v in (pj.get('per_kernel') or {}).items():
        if k.replace(' ', '').replace('issk::', '') == want:
            return v
    return None


def gemm_kernel_table(ctxs, peak_tf, pj=None):
    """HIP-event time, launches and algorithmic flops of every GEMM kernel INSTANTIATION (iss_prof_get_instance: template
    arguments spelled out) summed over the given contexts -> ([{kernel, ms_per_step, launches, flops, avg_launch_ms,
    flops_per_launch, achieved, frac, traffic...}] sorted by time, the entry with the most time)."""
    acc = {}
    for c in ctxs:
        for e in c.prof_instances():
            a = acc.setdefault(e['kernel'], [0.0, 0, 0.0])
            a[0] += e['ms']; a[1] += e['launches']; a[2] += e['flops']
    rows = []
    for name, (ms, nl, fl) in acc.items():
        if not nl or fl <= 0:
            continue
        tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        row = {"kernel": name, "ms_per_step": ms, "launches": nl, "flops": fl, "avg_launch_ms": ms / nl,
               "flops_per_launch": fl / nl, "achieved": tf, "frac": tf / peak_tf}
        pk = _pmc_kernel(pj or {}, name)
        if pk and 'FETCH_SIZE_bytes' in pk and 'WRITE_SIZE_bytes' in pk:
            raw = 2 * pk['FETCH_SIZE_bytes'] + pk['WRITE_SIZE_bytes']            # mean launch of the PMC run
            # the PMC run is a shorter recording (rocprofv3 --pmc crashes on the hour, profiles/r05_pmc_1h_attempt.txt): its mean launch
            # mixes full passes and a remainder in other proportions than this run's.  A launch's bytes scale with its rows like its
            # flops do, so the PMC run's bytes per algorithmic flop x THIS run's flops per launch is this run's mean launch
            fpl_pmc = ((pj or {}).get('pmc_run_flops_per_launch') or {}).get(name.replace(' ', ''))
            scale = (fl / nl) / fpl_pmc if fpl_pmc else None
            row["traffic_per_launch_pmc_run"] = raw
            row["traffic_scale_to_this_run"] = scale
            row["traffic_per_launch_static"] = raw * scale if scale else raw
            row["mfma_busy_frac_static"] = pk.get('mfma_busy_frac')
        rows.append(row)
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows, (rows[0] if rows else None)


# ------------------------------------------------------------------------------ CPU baseline
def _cpu_feature_worker(job):
    """One process of the file-parallel CPU leg (leg 1b): oracle feature path + energy detector +
    per-segment Viterbi on a stand-in emission array for one file's worth of samples.  Runs as `bench.py --cpu-worker`."""
    file_index, nsec = job
    import time as _t
    from oracle import sidekit as osk, segment as oseg
    from oracle.viterbi import viterbi_decoding, diag_trans_exp
    pcm = synth_recording_numpy(file_index, nsec * FS)
    sig = (pcm / 32768.0).astype(np.float32)
    t0 = _t.perf_counter()
    mspec, loge, difflen = osk.media2feats(sig)
    lseg0 = oseg.energy_seglist(loge, 0.03)
    oseg.get_patches(mspec[:, :21].copy(), 68, 2)
    oseg.get_patches(mspec, 68, 2)
    # the two per-segment smoothing passes on emissions of the right shape (the CNN itself is a leg of its own)
    rng = np.random.default_rng(file_index)
    for k, arg in ((3, 80), (2, 80)):
        for lab, a, b in lseg0:
            if lab == 'energy' and b > a:
                viterbi_decoding(np.log(rng.dirichlet(np.ones(k), b - a).astype(np.float32)), diag_trans_exp(arg, k))
    return _t.perf_counter() - t0


def synth_recording_numpy(file_index, n_samples):
    """numpy twin of synth_recording (same plan; numpy noise): for CPU-only legs that must not touch torch / the GPU."""
    rng = np.random.default_rng(20250926 + file_index)
    out = np.zeros(n_samples, np.float32)
    for kind, pos, n, f0, chord, trem in synth_plan(file_index, n_samples):
        if kind == 1:
            out[pos:pos + n] = rng.standard_normal(n).astype(np.float32) * 10 ** (-30 / 20)
        elif kind >= 2:
            t = np.arange(n, dtype=np.float32) / FS
            x = np.zeros(n, np.float32)
            if kind == 2:
                for k in range(1, 31):
                    if f0 * k < 7600:
                        x += np.sin(2 * np.pi * f0 * k * t) / k
                x *= 0.6 + 0.4 * np.sin(2 * np.pi * 4.0 * t)
                level = 10 ** (-20 / 20)
            else:
                for f in chord:
                    x += np.sin(2 * np.pi * float(f) * t)
                x *= 0.8 + 0.2 * np.sin(2 * np.pi * trem * t)
                level = 10 ** (-18 / 20)
            x *= level / np.sqrt(np.mean(x * x) + 1e-20)
            out[pos:pos + n] = x
    return np.clip(np.round(out * 32768.0), -32768, 32767).astype(np.int16)


def _cpu_full_worker(job):
    """One process of the full-path all-cores CPU leg (1c): the WHOLE hot path on the CPU for one file -- oracle feature path,
    energy detector, torch-CPU forward of both stand-in networks (reference semantics: VAD net on energy slots, gender net on
    speech slots, batch 32 as segmenter.py:208), reference-order Viterbi -- with `threads` torch threads.
    Runs as `bench.py --cpu-full-worker index nsec threads`; prints the seconds the pipeline took (imports excluded)."""
    file_index, nsec, threads = job
    import time as _t
    import torch
    torch.set_num_threads(threads)
    from oracle import sidekit as osk, segment as oseg, keras_cnn as ocnn
    from inaspeechsegmenter_amd import keras_model as KM
    vad_layers, _ = KM.synthetic_ina_like(21, 3, seed=1)          # what Segmenter(models='synthetic') loads (net ids 0 / 1 -> seeds 1 / 2)
    gen_layers, _ = KM.synthetic_ina_like(24, 2, seed=2)
    pcm = synth_recording_numpy(file_index, nsec * FS)
    sig = (pcm / 32768.0).astype(np.float32)
    ocnn.forward(vad_layers, np.zeros((32, 68, 21, 1), np.float32), batch_size=32)         # thread pool / allocator warm-up
    t0 = _t.perf_counter()
    mspec, loge, difflen = osk.media2feats(sig)
    lseg0 = oseg.energy_seglist(loge, 0.03)
    lseg1 = oseg.dnn_segment('smn', lambda b: ocnn.forward(vad_layers, b, batch_size=32), mspec, lseg0, difflen)
    oseg.dnn_segment('gender', lambda b: ocnn.forward(gen_layers, b, batch_size=32), mspec, lseg1, difflen)
    return _t.perf_counter() - t0


def host_cpu_info():
    """What the host really gives this process: os.cpu_count() (hardware threads the kernel shows), the affinity mask, and the
    cgroup CPU quota of the container -- a 256-thread node behind a 32-core quota is a 32-core host for every CPU leg."""
    info = {"cpu_count": os.cpu_count() or 1}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        info["affinity"] = info["cpu_count"]
    quota = None
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt and txt[0] != 'max':
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    info["cgroup_quota_cores"] = quota
    eff = min(info["cpu_count"], info["affinity"])
    if quota:
        eff = max(1, min(eff, int(quota + 0.5)))
    quota_known = bool(quota)
    # ... and what it measurably gives: k copies of a fixed interpreter loop at once against one copy alone (a quota, SMT siblings
    # or neighbours on the node show up here whether or not the cgroup files are readable)
    try:
        import subprocess
        k = min(info["cpu_count"], 256)
        cmd = [sys.executable, '-S', '-c', 'import time\nt=time.perf_counter()\nx=0\nfor i in range(6000000): x+=i\nprint(time.perf_counter()-t)']
        t1 = float(subprocess.run(cmd, capture_output=True, text=True, timeout=60).stdout.strip())
        t0 = time.perf_counter()
        ps = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for _ in range(k)]
        tk = [float(p_.communicate(timeout=120)[0].decode().strip()) for p_ in ps]
        wall = time.perf_counter() - t0
        info["measured_parallelism"] = {"copies": k, "one_copy_s": t1, "mean_copy_s": sum(tk) / k, "wall_s": wall,
                                        "effective_cores": k * t1 / (sum(tk) / k)}
        if not quota_known:                                  # (a stated quota wins: the interpreter loop under-reports SMT pairs)
            eff = max(1, min(eff, int(info["measured_parallelism"]["effective_cores"] + 0.5)))
    except Exception as exc:                                # noqa: BLE001
        info["measured_parallelism"] = {"error": repr(exc)}
    info["effective_cores"] = eff
    return info


def cpu_full_path_all_cores_leg(cores_host, threads=8, nsec=90, budget_s=120.0):
    """Leg 1c: cores_host / threads processes x `threads` torch threads, every process the whole CPU pipeline incl. both CNNs on
    its own `nsec`-second synthetic file: what the host ALONE could do on this path with all of its cores."""
    import subprocess
    nproc = max(1, cores_host // threads)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), OPENBLAS_NUM_THREADS='1', HIP_VISIBLE_DEVICES='',
               ROCR_VISIBLE_DEVICES='')
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-full-worker', str(2000 + i), str(nsec), str(threads)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env) for i in range(nproc)]
    per, failed = [], 0
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=max(1.0, budget_s - (time.perf_counter() - t0)))
            per.append(float(out.decode().strip().splitlines()[-1]))
        except Exception:                                   # noqa: BLE001
            failed += 1
            pr.kill()
    wall = time.perf_counter() - t0
    if not per:
        return {"processes": nproc, "threads_per_process": threads, "error": "no worker finished"}
    return {"processes": nproc, "threads_per_process": threads, "cores": nproc * threads, "finished": len(per), "failed": failed,
            "seconds_per_file": nsec, "wall_s": wall, "x_realtime_aggregate_incl_startup": len(per) * nsec / wall,
            "x_realtime_aggregate_steady": len(per) * nsec / max(per), "x_realtime_one_process_mean": nsec / (sum(per) / len(per)),
            "what": "the whole CPU path per process (oracle numpy features, energy detector, torch-CPU forward of both stand-in nets "
                    "at batch 32 on reference-semantics slots, reference-order Python Viterbi), one synthetic file each, all processes "
                    "at once on all host cores; `steady` = files x seconds / the slowest process's own pipeline time (imports and "
                    "interpreter start-up excluded), `incl_startup` = / the wall time of the whole leg"}


def cpu_file_parallel_leg(nproc, nsec=120, budget_s=90.0):
    """Leg 1b: `nproc` processes (`python bench.py --cpu-worker i nsec`, numpy / BLAS pinned to one thread each, no GPU), each
    running the oracle feature path + Viterbi on its own `nsec`-second synthetic file.  -> dict (x real time aggregate)."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS='1', MKL_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', HIP_VISIBLE_DEVICES='',
               ROCR_VISIBLE_DEVICES='')
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-worker', str(1000 + i), str(nsec)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env) for i in range(nproc)]
    per, failed = [], 0
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=max(1.0, budget_s - (time.perf_counter() - t0)))
            per.append(float(out.decode().strip().splitlines()[-1]))
        except Exception:                                   # noqa: BLE001  (timeout, crash, unparsable output)
            failed += 1
            pr.kill()
    wall = time.perf_counter() - t0
    if not per:
        return {"processes": nproc, "error": "no worker finished"}
    return {"processes": nproc, "finished": len(per), "failed": failed, "seconds_per_file": nsec, "wall_s": wall,
            "x_realtime_aggregate": len(per) * nsec / wall, "x_realtime_one_process_mean": nsec / (sum(per) / len(per)),
            "what": "oracle/ numpy feature path (sidekit mfcc restatement) + energy detector + both patch builders + the "
                    "per-segment reference-order Python Viterbi, one synthetic file per process, one BLAS thread each; the wall "
                    "time includes interpreter start-up and module imports of every worker (as a file-parallel CPU job would pay them)"}


def cpu_baseline(seg, pcm_host, target_s=25.0):
    """The oracle (numpy restatement of the reference feature path + torch-CPU Keras-semantics
    forward + the reference-order Viterbi) on a bounded sample of the same recording.  Also returns what the parity check
    needs: the oracle's segments and raw network outputs on that sample.  Legs as BASELINE.md section 3 plans them:
    (1a) one process, (1b) file-parallel over all host cores, (2) the CNN forward at batch 32 / 1024 over a thread sweep."""
    import torch
    from oracle import sidekit as osk, segment as oseg, keras_cnn as ocnn
    cores_host = os.cpu_count() or 1
    vad_layers, gen_layers = seg.vad.layers, seg.gender.layers
    legs = {}

    # ---- leg 2: CNN forward alone (segmenter.py:222-224: default batch 32, 1024 recommended for GPUs), thread sweep
    x = np.random.default_rng(0).normal(0, 1, (1024, 68, 21, 1)).astype(np.float32)
    sweep = []
    best = None
    for th in [t for t in (8, 16, 32, 64, 128) if t <= cores_host] or [cores_host]:
        torch.set_num_threads(th)
        for bs in (32, 1024):
            ocnn.forward(vad_layers, x[:bs], batch_size=bs)                    # warm the thread pool / allocator
            t0 = time.perf_counter()
            ocnn.forward(vad_layers, x, batch_size=bs)
            rate = len(x) / (time.perf_counter() - t0)
            sweep.append({"threads": th, "batch_size": bs, "slots_per_s": rate})
            if best is None or rate > best["slots_per_s"]:
                best = sweep[-1]
    legs["vad_cnn_forward_sweep"] = sweep
    legs["vad_cnn_forward_best"] = best
    threads, bs_best = best["threads"], best["batch_size"]
    torch.set_num_threads(threads)

    # ---- leg 1a: the whole pipeline in one process on a bounded sample (also the parity sample)
    def run(nsec, keep=False):
        sig = (pcm_host[:nsec * FS] / 32768.0).astype(np.float32)
        t0 = time.perf_counter()
        mspec, loge, difflen = osk.media2feats(sig)
        t_feat = time.perf_counter() - t0
        lseg0 = oseg.energy_seglist(loge, 0.03)
        lseg1, raw_vad = oseg.dnn_segment('smn', lambda b: ocnn.forward(vad_layers, b, batch_size=bs_best), mspec, lseg0, difflen, return_raw=True)
        lseg2, raw_gen = oseg.dnn_segment('gender', lambda b: ocnn.forward(gen_layers, b, batch_size=bs_best), mspec, lseg1, difflen, return_raw=True)
        dt = time.perf_counter() - t0
        det = dict(lseg0=lseg0, lseg1=lseg1, lseg2=lseg2, raw_vad=raw_vad, raw_gen=raw_gen, nframes=len(loge), t_feat=t_feat) if keep else None
        return dt, det

    probe = 20
    t_probe, _ = run(probe)
    nsec = int(max(probe, min(len(pcm_host) // FS, probe * target_s / max(t_probe, 1e-3))))
    # at least 600 s when that stays under ~60 s of CPU work: the parity check on this sample should see >= 20 boundaries that
    # the networks decided (the generator changes segment every ~11 s, one in ten is silence)
    if probe * 60.0 / max(t_probe, 1e-3) >= 600:
        nsec = max(nsec, 600)
    nsec = min(nsec, 660, len(pcm_host) // FS)
    t, det = run(nsec, keep=True)
    legs["one_process"] = {"x_realtime": nsec / t, "wall_s": t, "sample_s": nsec, "features_x_realtime_1thread": nsec / det['t_feat'],
                           "cnn_threads": threads, "cnn_batch_size": bs_best}
    # ---- leg 1b: file-parallel feature path + Viterbi on every host core
    legs["file_parallel_features_viterbi"] = cpu_file_parallel_leg(min(cores_host, 256))
    # ---- leg 1c: the FULL path (both CNNs included) on every host core: cores / 8 processes x 8 threads
    hinfo = host_cpu_info()
    legs["host_cpu"] = hinfo
    eff = min(hinfo["effective_cores"], 256)
    legs["full_path_all_cores"] = cpu_full_path_all_cores_leg(eff, threads=8 if eff >= 8 else max(1, eff))
    if os.path.isdir('/root/reference/inaSpeechSegmenter'):      # build container only: the unmodified reference front end
        import importlib.util
        spec = importlib.util.spec_from_file_location('ref_sidekit_mfcc', '/root/reference/inaSpeechSegmenter/sidekit_mfcc.py')
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        sig = (pcm_host[:nsec * FS] / 32768.0).astype(np.float32)
        t0 = time.perf_counter()
        with np.errstate(divide='ignore'):
            m.mfcc(sig, get_mspec=True)
        legs['reference_sidekit_mfcc_x_realtime_1thread'] = nsec / (time.perf_counter() - t0)
    legs["unmodified_reference_legs"] = ("profiles/r04_cpu_reference_baseline.json (tests/cpu_reference_baseline.py, run in the build "
                                         "container where /root/reference exists: sidekit_mfcc.mfcc + _get_patches + viterbi_decoding, "
                                         "features_vbx, resnet.py on torch-CPU); /root/reference does not travel to the GPU box")
    out = {"value": (nsec / 3600.0) / t, "unit": "hours-of-audio/s", "cores": threads, "threads_used": threads,
           "cores_host": cores_host, "kind": "port",
           "sample": f"first {nsec} s of the rank-0 recording, reference semantics (VAD on energy slots, gender on "
                     f"speech slots), ONE process: oracle/ numpy feature path (1 thread) + torch-CPU Keras-semantics CNN forward at "
                     f"the fastest (threads, batch_size) of the sweep in legs ({threads} threads, batch {bs_best}) + "
                     f"reference-order Python Viterbi; {t:.2f} s wall; "
                     f"stand-in for the TensorFlow/CPU path (TensorFlow is not installable here)",
           "x_realtime": nsec / t, "legs": legs}
    return out, nsec, det


def cnn_driven_boundaries(lseg):
    """Boundaries of a final segmentation that a NETWORK decided: adjacent segments neither of which is `noEnergy` (every
    boundary of the energy detector has `noEnergy` on one side; the VAD / gender Viterbi passes run inside `energy` /
    `speech` segments, segmenter.py:156-178)."""
    return sum(1 for x, y in zip(lseg, lseg[1:]) if x[0] != 'noEnergy' and y[0] != 'noEnergy' and x[2] == y[1])


def viterbi_min_margin(logp, trans):
    """Score of the best path minus the score of the best path that passes through a DIFFERENT state at some slot,
    minimised over the slots (max-sum forward / backward, uniform initial log(1/K) like pyannote_viterbi.py:166-167):
    how far the emissions are from changing any label of the segment."""
    T, K = logp.shape
    if T == 0 or K < 2:
        return float('inf')
    e = np.asarray(logp, np.float64)
    a = np.empty((T, K))
    b = np.zeros((T, K))
    a[0] = e[0] + np.log(1.0 / K)
    for t in range(1, T):
        a[t] = e[t] + (a[t - 1][:, None] + trans).max(0)
    for t in range(T - 2, -1, -1):
        b[t] = (trans + (e[t + 1] + b[t + 1])[None, :]).max(1)
    through = a + b                                     # best path constrained to state s at slot t
    best = through.max(1)
    srt = np.sort(through, axis=1)
    return float((srt[:, -1] - srt[:, -2]).min()) if np.isfinite(best).all() else 0.0


def _own_flops_share(seg, kernel_name):
    """Share of a first-layer-fused launch's credited flops that its own convolution performs: the library credits the fused first
    layer as the reference computes it (once per window, segmenter.py:82-84) although it runs once per log-mel row.  1.0 for any
    other kernel; both networks' second convolutions weighted by their flops (they alternate, the same slots each)."""
    from inaspeechsegmenter_amd import _native as N
    if 'wq_kernel' not in kernel_name and 'FUSED' not in kernel_name and ',true,1' not in kernel_name:
        return 1.0
    own = cred = 0.0
    for net in (seg.vad, seg.gender if seg.detect_gender else None):
        if net is None:
            continue
        rows = [r for r in net.compiled.prog if r[N.C_OP] == N.OP_CONV]
        if len(rows) < 2:
            return 1.0
        f = [2.0 * r[N.C_KH] * r[N.C_KW] * r[N.C_CIN] * r[N.C_COUT] * r[N.C_HO] * r[N.C_WO] for r in rows[:2]]
        own += f[1]; cred += f[0] + f[1]
    return own / cred if cred else 1.0


def parity_check(seg, pcm_host, nsec, det):
    """GPU vs oracle on the cpu_baseline sample: identical segments; max |p_gpu - p_oracle| and the number of slots whose
    arg-max class differs, over every slot the oracle evaluated (VAD net on its energy slots, gender net on its speech
    slots); and what makes `segments_equal` mean something: how many boundaries the networks decided, how the slots spread
    over the classes, and the smallest Viterbi path margin of an evaluated segment."""
    from inaspeechsegmenter_amd import segmenter as S
    sig = np.ascontiguousarray(pcm_host[:nsec * FS])
    got = seg.segment_signal(sig)
    want = [(lab, a * .02, b * .02) for lab, a, b in det['lseg2']]
    rows = S._window_rows(det['nframes'])
    worst, nslots, mism, margin = 0.0, 0, 0, float('inf')
    classes = {}
    for net, raw, lseg_in in ((seg.vad, det['raw_vad'], det['lseg0']), (seg.gender, det['raw_gen'], det['lseg1'])):
        spans = [(a, b) for lab, a, b in lseg_in if lab == net.inlabel]
        if not spans or raw is None:
            continue
        idx = np.concatenate([np.arange(a, b) for a, b in spans])
        p, fin = seg.ctx.cnn_probs(net.net_id, rows[idx])
        ok = fin & np.all(np.isfinite(raw), axis=1)
        worst = max(worst, float(np.abs(p[ok] - raw[ok]).max()) if ok.any() else 0.0)
        mism += int((p[ok].argmax(1) != raw[ok].argmax(1)).sum())
        nslots += int(ok.sum())
        hist = np.bincount(raw[ok].argmax(1), minlength=len(net.outlabels)) / max(int(ok.sum()), 1)
        classes.update({lab: round(float(h), 4) for lab, h in zip(net.outlabels, hist)})
        trans = S.diag_trans_exp(net.viterbi_arg, len(net.outlabels))
        pos = 0
        for a, b in spans:
            with np.errstate(divide='ignore'):
                margin = min(margin, viterbi_min_margin(np.log(p[pos:pos + b - a]), trans))
            pos += b - a
    return {"segments_equal": got == want, "segments": len(want), "cnn_driven_boundaries": cnn_driven_boundaries(det['lseg2']),
            "classes_present": classes, "argmax_mismatch_slots": mism, "max_abs_dprob": worst, "slots": nslots,
            "min_viterbi_path_margin_nats": margin, "sample_s": nsec,
            "what": "Segmenter.segment_signal on the cpu_baseline sample vs the oracle pipeline (labels and boundaries); "
                    "iss_cnn_probs vs the oracle's network outputs on every slot it evaluated (max |dp|, slots whose arg-max "
                    "differs); cnn_driven_boundaries = boundaries between two non-noEnergy segments (decided by a network, not by "
                    "the energy detector); classes_present = share of the evaluated slots each class wins (oracle); margin = best "
                    "Viterbi path score minus the best path that differs at some slot, smallest over the evaluated segments"}


# ------------------------------------------------------------------------------ file workloads
def write_wav(path, pcm):
    import struct
    with open(path, 'wb') as f:
        f.write(b'RIFF' + struct.pack('<I', 36 + pcm.nbytes) + b'WAVEfmt ' + struct.pack('<IHHIIHH', 16, 1, 1, 16000, 32000, 2, 16)
                + b'data' + struct.pack('<I', pcm.nbytes))
        f.write(pcm.tobytes())


def make_files(indices, minutes, dev, root):
    """Synthetic WAV files (SURVEY 8(d) generator, seed 20250926 + file index) under `root`; returns their paths by index."""
    os.makedirs(root, exist_ok=True)
    n = int(minutes * 60 * FS)
    paths = {}
    for i in indices:
        p = os.path.join(root, f'f{i:06d}.wav')
        if not (os.path.exists(p) and os.path.getsize(p) == 44 + 2 * n):
            write_wav(p, synth_recording(i, n, dev).cpu().numpy())
        paths[i] = p
    return paths, n


def make_comm(ctx, rank, world, dev, kind='rccl'):
    """The exchange step of an N > 1 run.  -> (comm, description).  `kind` is what the command line asked for -- there is no
    silent fallback from one backend to another (north_star: no dual backend):
      rccl   the C-ABI's ncclAllGather (RcclComm on `ctx`, librccl dlopen'ed): the only data path of a multi-GPU run; a rendezvous
             that cannot be set up is an error on every rank;
      torch  a torch.distributed "nccl" process group on the same RCCL (callers that already run one; asked for explicitly);
      gloo   a torch.distributed "gloo" group over host memory: the single-GPU REHEARSAL of the N-rank path (RCCL refuses two
             ranks on one device) and the CPU tests -- never a multi-GPU result, and the line says so."""
    if world == 1:
        return None, None
    from inaspeechsegmenter_amd import sharding
    if kind == 'rccl':
        try:
            comm = sharding.rccl_rendezvous(ctx, rank, world)
        except Exception as exc:                            # noqa: BLE001
            raise SystemExit(f"[bench] rank {rank}: the C-ABI RCCL rendezvous failed ({exc}); no fallback is taken -- "
                             "run with --comm torch (torch.distributed on the same RCCL) or --comm gloo (single-GPU rehearsal) "
                             "if that is what you want") from exc
        return comm, ("iss_allgather_segments: ncclAllGather of int32 segment tables through the C-ABI (librccl dlopen'ed, no torch "
                      "in the data path)")
    import torch.distributed as dist
    if kind == 'torch':
        if not dist.is_initialized():
            dist.init_process_group('nccl', device_id=dev)
        return sharding.TorchComm(device=dev), "torch.distributed nccl all_gather_into_tensor (--comm torch)"
    if kind == 'gloo':
        if not dist.is_initialized():
            # gloo's transport prints "[Gloo] Rank r is connected to ..." on STDOUT when the mesh is built: keep the one-JSON-line
            # contract by pointing fd 1 at stderr while the group comes up (first collective included)
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group('gloo')
                dist.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
        return sharding.TorchComm(device=None), ("torch.distributed gloo all_gather_into_tensor over host memory (--comm gloo: "
                                                 "rehearsal of the N-rank path, NOT an RCCL / xGMI exchange)")
    raise SystemExit(f"--comm {kind}: unknown")


def self_spawn(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the driver's own command line does
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`), and
    hand back the launcher's exit status.  Rank 0 of the child job prints the JSON line on our stdout."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC only on these hosts (RCCL across processes)
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    print(f"[bench] --gpus {n} without a launcher: spawning {' '.join(cmd)}", file=sys.stderr)
    return subprocess.call(cmd, env=env)


def bench_files(args, torch, dev, local_rank, rank, world, kind, seg=None, comm=None, comm_kind=None, steps=None, warmup=None,
                per_gpu=None, dense=None):
    """`batch` (configs[2], one GPU) and `archive` (configs[3] shape, file-parallel + one all-gather) workloads.
    -> the JSON line as a dict on rank 0 (None elsewhere).  seg / comm: reuse the caller's (companion runs)."""
    from inaspeechsegmenter_amd import Segmenter, _native, sharding
    from inaspeechsegmenter_amd.archive import segment_archive
    steps = steps or args.steps
    warmup = max(args.warmup if warmup is None else warmup, 1)
    own_seg = seg is None
    if own_seg:
        seg = Segmenter(vad_engine='smn', detect_gender=True, ffmpeg=None, models='synthetic', device=local_rank)
    x3 = args.precision != 'f32'
    seg.ctx.set_precision({'bf16x3': _native.PREC_BF16X3, 'f16x3': _native.PREC_F16X3, 'f32': _native.PREC_F32}[args.precision])
    dense = bool(args.dense_files if dense is None else dense)
    seg.dense_batches = dense                        # both networks on every slot of every file (comparable with the resident-path figure)
    if comm is None and world > 1:
        comm, comm_kind = make_comm(seg.ctx, rank, world, dev, args.comm)
    minutes = args.file_minutes or (5.0 if kind == 'batch' else 3.0)
    per_gpu = per_gpu or args.files_per_gpu or 128
    nfiles = per_gpu * world
    try:
        os.makedirs(args.dir, exist_ok=True)
        base = args.dir if os.access(args.dir, os.W_OK) else None
    except OSError:
        base = None
    if base is None:                                 # no writable /dev/shm: any local directory will do (page cache)
        import tempfile
        base = os.path.join(tempfile.gettempdir(), 'iss_bench')
    root = os.path.join(base, f'{kind}_{minutes:g}min')
    mine = [i for i in range(nfiles) if i % world == rank]           # = sharding.shard_files for equal sizes
    try:
        paths, n = make_files(mine, minutes, dev, root)
    except OSError as exc:                           # e.g. a /dev/shm too small for this rank's share: use the local disk
        import shutil
        import tempfile
        shutil.rmtree(root, ignore_errors=True) if world == 1 else None
        for i in mine:
            try:
                os.remove(os.path.join(root, f'f{i:06d}.wav'))
            except OSError:
                pass
        print(f'[bench] {root}: {exc}; falling back to {tempfile.gettempdir()}', file=sys.stderr)
        root = os.path.join(tempfile.gettempdir(), 'iss_bench', f'{kind}_{minutes:g}min')
        paths, n = make_files(mine, minutes, dev, root)
    if comm:
        comm.barrier()                               # every rank's files exist before anybody's first step
    lin = [os.path.join(root, f'f{i:06d}.wav') for i in range(nfiles)]
    lout = [os.path.join(root, f'out_r{rank}', f'f{i:06d}.csv') for i in range(nfiles)]
    sizes = [44 + 2 * n] * nfiles
    hours = nfiles * minutes / 60.0

    def step():
        if kind == 'batch':
            t, nb, avg, lmsg = seg.batch_process(lin, lout, batch_files=args.batch_files or None, workers=args.workers or None,
                                                 batch_seconds=args.batch_seconds or None)
            assert nb == nfiles, [m for m in lmsg if m[1] != 0][:3]
            return nb
        table, lmsg = segment_archive(seg, lin, lout, sizes=sizes, comm=comm, batch_files=args.batch_files or None,
                                      workers=args.workers or None, batch_seconds=args.batch_seconds or None)
        assert len(table) == nfiles and all(m[1] == 0 for m in lmsg), (len(table), [m for m in lmsg if m[1] != 0][:3])
        return sum(len(v) for v in table.values())

    for _ in range(warmup):
        nseg = step()

    def barrier():
        if comm:
            comm.barrier()
    def host_mem():
        """resident set of this process (MB) and bytes in use under the files' mount (MB): must stay flat across steps"""
        try:
            import psutil
            import shutil
            return round(psutil.Process().memory_info().rss / 2 ** 20, 1), round(shutil.disk_usage(root).used / 2 ** 20, 1)
        except Exception:                                   # noqa: BLE001
            return None, None
    barrier()
    torch.cuda.synchronize()
    mem_trace = [host_mem()]
    for w in seg.__dict__.get('_pipeline_workers', []):
        w.stats = {k: 0.0 for k in w.stats}
    t0 = time.perf_counter()
    for _ in range(steps):
        nseg = step()
        mem_trace.append(host_mem())
    seg.ctx.synchronize()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0                  # this rank's own steps, before it waits for the others
    barrier()
    dt = time.perf_counter() - t0
    pipe = [{k: (round(v * 1e3 / steps, 1) if k not in ('batches', 'files') else v / steps) for k, v in w.stats.items()}
            for w in seg.__dict__.get('_pipeline_workers', [])]
    if comm:
        dt = comm.max_over_ranks(dt)
    # roofline of the conv/dense GEMM launches, live: HIP events on the library's streams around every launch of ONE extra
    # step (all ranks run it -- the step holds the collective; rank 0's device contexts are read)
    ctxs = [w.ctx for w in seg.__dict__.get('_pipeline_workers', [])] or [seg.ctx]
    for c in ctxs:
        c.prof_enable(True)
        c.prof_reset()
    step()
    conv_ms = conv_fl = 0.0
    conv_n = 0
    for c in ctxs:
        ms, nl, fl = c.prof_get(0)
        conv_ms += ms; conv_n += nl; conv_fl += fl
    ktab, kdom = gemm_kernel_table(ctxs, MFMA_BF16_PEAK_TF if x3 else MFMA_F32_PEAK_TF, _pmc_static())
    for c in ctxs:
        c.prof_enable(False)
    # audit trail of the multi-GPU run: what RCCL reports for the communicator, and the spread of the ranks' own step times
    rccl = None
    dt_min = dt_max = dt_local
    if comm:
        dt_max = comm.max_over_ranks(dt_local)
        dt_min = -comm.max_over_ranks(-dt_local)
        rccl = comm_audit(comm, comm_kind, rank, world)
    line = None
    if rank == 0:
        value = steps * hours / dt
        peak_tf = MFMA_BF16_PEAK_TF if x3 else MFMA_F32_PEAK_TF
        step_s = dt / steps
        ach = conv_fl / step_s / 1e12                # whole-step figure: the two device contexts' launches overlap on the GPU,
        ach_ev = conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0     # so per-launch event durations are stretched
        roofline = {"bound": "mfma", "kernel": "conv/dense implicit-GEMM launches of rank 0's device contexts (see the segmenter workload's line)",
                    "achieved": ach, "peak": peak_tf, "unit": "TFLOP/s", "frac": ach / peak_tf, "traffic": None,
                    "launches_per_step": conv_n, "flops_per_launch": conv_fl / max(conv_n, 1),
                    "event_based": {"achieved": ach_ev, "frac": ach_ev / peak_tf, "kernel_ms_per_step": conv_ms,
                                    "avg_launch_ms": conv_ms / max(conv_n, 1)},
                    "dominant": kdom, "kernels": ktab,
                    "note": "achieved = algorithmic flops of rank 0's GEMM launches in one step / the step's WALL time (decode, copies, "
                            "host Viterbi and export included): launches of the two device contexts run concurrently, so the sum of "
                            "their HIP-event durations (event_based) counts shared time twice; the segmenter workload's line has the "
                            "kernel-only figure.  Reference semantics: VAD net on energy slots, gender net on speech slots"}
        line = {
            "metric": "hours-of-audio segmented/sec (smn+gender, 16 kHz mono)",
            "value": value, "unit": "hours-of-audio/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": (f"f32 ({args.precision} split-operand MFMA, f32 accumulate; f64 FFT)" if x3 else "f32 (f32 MFMA; f64 FFT)"),
            "data": "synthetic", "x_realtime_per_gpu": value * 3600.0 / world,
            "config": {"workload": (f"BASELINE.json configs[2]: {nfiles} x {minutes:g} min synthetic 16 kHz mono PCM16 WAV files in {args.dir} "
                                    "through Segmenter.batch_process" if kind == 'batch' else
                                    f"BASELINE.json configs[3] shape: {nfiles} x {minutes:g} min synthetic WAV files ({per_gpu} per GPU, weak scaling), "
                                    "file-parallel through archive.segment_archive, one all-gather of the segment tables per step") +
                                   ("; DENSE: both nets on 100% of the slots of every file (same work per audio-hour as the segmenter workload's "
                                    "dense figure)" if dense else "; reference semantics (VAD net on energy slots, gender net on speech slots)") +
                                   "; RIFF parse, H2D copy, device "
                                   "features + CNNs, compiled Viterbi and CSV export are all inside the timed region",
                       "files": nfiles, "files_per_gpu": per_gpu, "minutes_per_file": minutes, "audio_hours_per_step": hours,
                       "ms_per_file_per_gpu": dt / steps / per_gpu * 1e3, "segments_per_step": nseg,
                       "weights": "seeded stand-ins, (68,21,1)->3 and (68,24,1)->2, ~1.25 M params each, last layer calibrated "
                                  "(tests/golden/make_standin_heads.py); the real Keras files are un-vendored release assets",
                       "parallelism": (f"file-parallel x{world}: files dealt by size (LPT), no data-path collective, ONE all-gather of int32 segment "
                                       f"tables per step: {comm_kind}") if world > 1 else "single GPU"},
            "roofline": roofline,
            "pipeline_workers_ms_per_step": {"workers": pipe,
                                             "what": "wall ms per step each device worker thread spent packing PCM into its page-locked buffer, in "
                                                     "H2D + sidekit + log-energy read-back, in the host energy detector, blocked in iss_cnn_probs, "
                                                     "and in Viterbi smoothing / bookkeeping (Python threads: the host phases share the GIL)"},
            "host_memory": {"rss_mb_after_warmup_then_each_step": [m[0] for m in mem_trace],
                            "files_mount_used_mb": [m[1] for m in mem_trace],
                            "what": "rank 0's resident set and the space in use under the WAV / CSV directory, sampled after the "
                                    "warm-up and after every timed step (a leak in the pinned buffers, queues or exporters would show)"},
            "ranks": {"ms_per_step_min": dt_min / steps * 1e3, "ms_per_step_max": dt_max / steps * 1e3,
                      "what": "each rank's own wall time for the timed steps (barrier to barrier), min / max over the ranks"},
        }
        if rccl is not None:
            line["rccl"] = rccl
    barrier()
    if own_seg:
        seg.close()
    return line


def comm_audit(comm, comm_kind, rank, world):
    """What the communicator itself reports (every rank calls this: it holds two collectives)."""
    out = {"collective": comm_kind}
    if hasattr(comm, 'info'):
        info = comm.info()
        ranks_ok = int(round(-comm.max_over_ranks(-float(info['world'] == world and info['rank'] == rank))))     # min over ranks
        out.update({"world": info['world'], "ranks_seen": info['world'], "rank0_user_rank": info['rank'], "version": info['version'],
                    "lib": info['lib'], "every_rank_agrees": bool(ranks_ok),
                    "what": "ncclCommCount / ncclCommUserRank / ncclGetVersion of rank 0's communicator (iss_comm_info) and the path of "
                            "the librccl it was bound from; the collective of every step is ncclAllGather through iss_allgather_segments"})
    else:
        out.update({"world": comm.world})
    return out


def bench_vbx(args, torch, dev, local_rank, rank, world, steps=None, warmup=None, cpu_leg=True):
    """BASELINE.json configs[4]: 1 h of 16 kHz audio through get_features (VBx 64-band fbank + CMN) and the
    ResNet-101 x-vector network (144-frame windows, hop 24) -- not the headline metric.  -> the JSON line as a dict."""
    from inaspeechsegmenter_amd import _native, vbx as V
    from inaspeechsegmenter_amd import keras_model as KM
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    ctx = _native.Context(local_rank)
    ctx.set_precision({'bf16x3': _native.PREC_BF16X3, 'f16x3': _native.PREC_F16X3, 'f32': _native.PREC_F32}[args.precision])
    if args.workspace_mb:
        ctx.set_workspace_limit(args.workspace_mb << 20)
    fe = V.FeatureExtractor(ctx)
    ex = V.VBxExtractor(ctx, KM.synthetic_resnet101(0), batch_windows=512)
    n = int(args.minutes * 60 * FS)
    pcm_t = synth_recording(rank, n, dev).cpu()
    pcm = ctx.pinned_empty((n,), np.int16)           # what a decoder writing into page-locked memory hands over
    pcm[:] = pcm_t.numpy()
    hours = n / FS / 3600.0

    def step():
        t0 = time.perf_counter()
        fea = fe(pcm, to_host=False)                 # PCM16 in, the (T, 64) features stay in HBM
        ctx.synchronize()
        t1 = time.perf_counter()
        xv = ex('utt', fea, n / FS)
        return t1 - t0, time.perf_counter() - t1, len(xv)

    for _ in range(warmup):
        step()
    ctx.synchronize()
    t0 = time.perf_counter()
    tf = tx = 0.0
    for _ in range(steps):
        a, b_, nwin = step()
        tf += a
        tx += b_
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ctx.prof_enable(True)
    ctx.prof_reset()
    step()
    conv_ms, conv_launches, conv_flops = ctx.prof_get(0)
    fb_ms, _, _ = ctx.prof_get(1)                    # vbx_fbank_kernel alone (HIP events on the library's stream)
    ktab, kdom = gemm_kernel_table([ctx], MFMA_BF16_PEAK_TF if args.precision != 'f32' else MFMA_F32_PEAK_TF, _pmc_static())
    ctx.prof_enable(False)
    # feature stage: algorithmic bytes = PCM16 in + the cached dither stream (f64) + (T, 64) f32 out
    T = n // 160
    fea_bytes = n * 2 + n * 8 + T * 64 * 4
    cpu = None
    if cpu_leg and not args.no_cpu_baseline:
        import torch as _t
        from oracle import vbx as ovbx
        threads = min(os.cpu_count() or 1, 64)
        _t.set_num_threads(threads)
        nsec = 40
        sig = pcm[:nsec * FS].astype(np.float64) / 32768.0
        c0 = time.perf_counter()
        fea_o = ovbx.get_features(sig)
        c1 = time.perf_counter()
        starts = list(range(0, len(fea_o) - 144, 24))[:48]
        x = np.stack([fea_o[s:s + 144].T for s in starts]).astype(np.float32)
        params = KM.synthetic_resnet101(0)
        c2 = time.perf_counter()
        emb = ovbx.resnet101_forward(params, x)
        c3 = time.perf_counter()
        per_win = (c3 - c2) / len(starts)
        nwin_h = 14995
        t_hour = (c1 - c0) * (3600 / nsec) + per_win * nwin_h
        got = ex.get_embeddings(fe(pcm[:nsec * FS]), starts, 144)
        cpu = {"value": 1.0 / t_hour, "unit": "hours-of-audio/s", "x_realtime": 3600.0 / t_hour, "cores": threads, "threads_used": threads,
               "cores_host": os.cpu_count(), "kind": "port",
               "sample": f"oracle get_features on {nsec} s ({c1 - c0:.2f} s, numpy, 1 thread) + torch-CPU ResNet-101 on {len(starts)} windows "
                         f"in one batch ({per_win * 1e3:.1f} ms per window, {threads} threads), extrapolated to 1 h = 14 995 windows",
               "parity_max_rel_err_vs_gpu": float(np.abs(got - emb).max() / max(np.abs(emb).max(), 1e-9))}
    line = {"metric": "hours-of-audio through the vbx x-vector path per second", "value": steps * hours / dt,
            "unit": "hours-of-audio/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "x_realtime": steps * hours * 3600 / dt,
            "higher_is_better": True, "dtype": f"f32 ({args.precision} split-operand MFMA; f64 fbank front end)", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[4]: {args.minutes:g} min synthetic audio, get_features + ResNet-101 on "
                                   f"{nwin} windows (seeded stand-in weights); PCM16 in page-locked host memory -> device, features stay in HBM",
                       "features_ms_per_step": tf / steps * 1e3, "xvectors_ms_per_step": tx / steps * 1e3},
            "roofline": {"bound": "mfma", "achieved": conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms else 0.0, "peak": MFMA_BF16_PEAK_TF,
                         "frac": (conv_flops / (conv_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF) if conv_ms else 0.0,
                         "unit": "TFLOP/s", "kernel_ms_per_step": conv_ms, "launches_per_step": conv_launches, "flops_per_step": conv_flops,
                         "dominant": kdom, "kernels": ktab,
                         "secondary": {"kernel": "vbx_fbank_kernel + vbx_cumsum_kernel + vbx_cmn_kernel (PCM16 + dither -> 64-band fbank, CMN)",
                                       "bound": "hbm", "algorithmic_bytes": fea_bytes, "stage_ms_per_step": tf / steps * 1e3,
                                       "fbank_kernel_ms": fb_ms,
                                       "achieved": fea_bytes / (fb_ms * 1e-3) / 1e9 if fb_ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": (fea_bytes / (fb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if fb_ms > 0 else 0.0,
                                       "note": "achieved = algorithmic bytes of the stage / vbx_fbank_kernel time (f64 FFT per frame: "
                                               "latency- and f64-VALU-bound, not HBM-bound); stage_ms is wall time incl. the PCIe copy of "
                                               "the PCM16 (2 B/sample), the sequential CMN cumsum and launch overheads"}}}
    if cpu:
        line["cpu_baseline"] = cpu
    ctx.close()
    return line


def make_step(segment, rank, comm):
    """One step of the segmenter workload on this rank: `segment(dense)` -> slot-unit segments of the rank's recording, packed as
    int32 rows and (N > 1) all-gathered so that every rank holds every rank's table."""
    from inaspeechsegmenter_amd.sharding import pack_segments

    def step(dense=True):
        lseg = segment(dense)
        rows = pack_segments(rank, lseg)
        if comm:
            rows = comm.allgather(rows, 8192)
        return lseg, rows
    return step


def timed_steps(step, k, comm, sync):
    """The contract's timed region: barrier + device sync, EXACTLY k steps, device sync + barrier; the job's time is the MAX over the
    ranks.  -> (seconds for the job, the last step's output, this rank's own seconds)."""
    if comm:
        comm.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(k):
        out = step()
    sync()
    dt_own = time.perf_counter() - t0                    # this rank's k steps, before it waits for the others
    if comm:
        comm.barrier()
    dt = time.perf_counter() - t0
    dt = comm.max_over_ranks(dt) if comm else dt
    return dt, out, dt_own


def main_fake_device(args, rank, world):
    """Tests only (`--fake-device module:factory`): the launcher / rank / communicator / timing / line-assembly code of an N-rank
    run with the Segmenter replaced by `factory(rank)` (an object with segment_device_pcm(ptr, n, dense=...)), so that the N > 1
    branch of this file executes on a machine without a GPU.  The line it prints is labelled a self-test and carries no metric."""
    import importlib
    mod, fn = args.fake_device.split(':')
    fake = getattr(importlib.import_module(mod), fn)(rank)
    comm, comm_kind = make_comm(None, rank, world, None, args.comm)
    n = int(args.minutes * 60 * FS)
    step = make_step(lambda dense: fake.segment_device_pcm(0, n, dense=dense), rank, comm)
    for _ in range(args.warmup):
        step()
    dt, (lseg, rows), dt_local = timed_steps(step, args.steps, comm, lambda: None)
    dt_min = -comm.max_over_ranks(-dt_local) if comm else dt_local
    if comm:
        comm.barrier()
    if rank == 0:
        print(json.dumps({"metric": "SELF-TEST of bench.py's N-rank path on a fake device (no audio was segmented)", "value": None,
                          "unit": None, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                          "data": "fake-device", "scaling": "weak",
                          "config": {"parallelism": f"x{world}: {comm_kind}", "rows_gathered": int(len(rows)),
                                     "files_seen": sorted({int(r[0]) for r in rows})},
                          "ranks": {"ms_per_step_min": dt_min / args.steps * 1e3, "ms_per_step_max": dt / args.steps * 1e3}}))


def main():
    if len(sys.argv) == 4 and sys.argv[1] == '--cpu-worker':          # one process of cpu_file_parallel_leg: numpy only, no torch
        print(_cpu_feature_worker((int(sys.argv[2]), int(sys.argv[3]))))
        return
    if len(sys.argv) == 5 and sys.argv[1] == '--cpu-full-worker':     # one process of cpu_full_path_all_cores_leg: torch-CPU, no GPU
        print(_cpu_full_worker((int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--minutes', type=float, default=60.0, help='length of each rank\'s recording')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', choices=['segmenter', 'batch', 'archive', 'vbx'], default='segmenter',
                    help="segmenter = the metric's workload (default at every --gpus N, with the other configurations as "
                         "`companions`); archive = file-parallel configs[3] shape; batch = configs[2]; vbx = configs[4] (x-vector path)")
    ap.add_argument('--timing-only', action='store_true', help='experiment builds of the library that compute wrong results on purpose: no consistency assert')
    ap.add_argument('--no-companions', action='store_true', help='segmenter workload: skip the archive / batch / vbx companion runs')
    ap.add_argument('--companion-files-per-gpu', type=int, default=64, help='files per GPU of the archive companion (3 min each)')
    ap.add_argument('--files-per-gpu', type=int, default=0, help='batch / archive: files per GPU and step (default 128)')
    ap.add_argument('--file-minutes', type=float, default=0.0, help='batch / archive: minutes per file (default 5 / 3)')
    ap.add_argument('--dir', default='/dev/shm/iss_bench', help='batch / archive: where the synthetic WAV files live')
    ap.add_argument('--dense-files', action='store_true', help='batch / archive: both networks on every slot of every file (Segmenter.dense_batches)')
    ap.add_argument('--no-f32-companion', action='store_true', help='skip the exact-f32 (ISS_PREC_F32) companion step')
    ap.add_argument('--batch-files', type=int, default=0, help='batch / archive: files per device pass at most (0 = library default, 32)')
    ap.add_argument('--batch-seconds', type=float, default=0, help='batch / archive: audio per device pass at most (0 = library default, 2400 s)')
    ap.add_argument('--workers', type=int, default=0, help='batch / archive: device contexts taking super-batches in turn (0 = library default, 4)')
    ap.add_argument('--workspace-mb', type=int, default=0, help='activation workspace cap (0 = library default)')
    ap.add_argument('--precision', choices=['bf16x3', 'f16x3', 'f32'], default='f16x3',
                    help='conv/dense GEMM arithmetic: split-fp16 MFMA (ISS_PREC_F16X3, the library default), split-bf16 MFMA or exact-f32 MFMA')
    ap.add_argument('--comm', choices=['rccl', 'torch', 'gloo'], default='rccl',
                    help="N > 1 exchange step: rccl = iss_allgather_segments (ncclAllGather through the C-ABI; the default and the only "
                         "multi-GPU data path -- a failed rendezvous is an error, there is no fallback); torch = a torch.distributed nccl "
                         "group, asked for explicitly; gloo = host-memory group, which also lets the N ranks SHARE the visible GPUs "
                         "(single-GPU rehearsal of the N-rank code path; the line is labelled, it is not a scaling result)")
    ap.add_argument('--fake-device', default='', help=argparse.SUPPRESS)      # tests only: 'module:factory' replacing the Segmenter
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:             # no launcher: become one (the driver's own command line)
        raise SystemExit(self_spawn(args.gpus, sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if args.fake_device:
        return main_fake_device(args, rank, world)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    ndev = torch.cuda.device_count()
    shared = None
    if local_rank >= ndev:
        if args.comm != 'gloo':
            raise SystemExit(f"--gpus {args.gpus}: local rank {local_rank} has no device of its own ({ndev} visible) and RCCL refuses two "
                             "ranks on one device; --comm gloo rehearses the N-rank path on the visible device(s)")
    if args.comm == 'gloo' and world > ndev:
        shared = f"{world} ranks on {ndev} visible device(s): rank r runs on device r % {ndev}"
        local_rank = local_rank % ndev                               # from here on: the device index of this rank
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    args.shared_devices = shared
    if args.workload in ('batch', 'archive'):
        if args.workload == 'batch' and world > 1:
            raise SystemExit("--workload batch is the single-GPU configs[2]; use --workload archive for N > 1")
        line = bench_files(args, torch, dev, local_rank, rank, world, args.workload)
        if rank == 0:
            print(json.dumps(line))
        return
    if args.workload == 'vbx':
        if world > 1:
            raise SystemExit("--workload vbx is the single-GPU configs[4] (a single recording is not split over GPUs)")
        print(json.dumps(bench_vbx(args, torch, dev, local_rank, rank, world)))
        return

    from inaspeechsegmenter_amd import Segmenter, _native

    seg = Segmenter(vad_engine='smn', detect_gender=True, ffmpeg=None, models='synthetic', device=local_rank)
    x3 = args.precision != 'f32'
    seg.ctx.set_precision({'bf16x3': _native.PREC_BF16X3, 'f16x3': _native.PREC_F16X3, 'f32': _native.PREC_F32}[args.precision])
    if args.workspace_mb:
        seg.ctx.set_workspace_limit(args.workspace_mb << 20)
    comm, comm_kind = make_comm(seg.ctx, rank, world, dev, args.comm)
    n = int(args.minutes * 60 * FS)
    pcm = synth_recording(rank, n, dev)
    torch.cuda.synchronize()
    hours = n / FS / 3600.0

    step = make_step(lambda dense: seg.segment_device_pcm(pcm.data_ptr(), n, dense=dense), rank, comm)

    def sync():
        seg.ctx.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()

    def timed(k, dense):
        return timed_steps(lambda: step(dense), k, comm, sync)

    dt, (lseg, rows), dt_local = timed(args.steps, True)
    value = world * args.steps * hours / dt
    dt_min = -comm.max_over_ranks(-dt_local) if comm else dt_local
    rccl = comm_audit(comm, comm_kind, rank, world) if comm else None

    # reference-semantics rate (VAD on energy slots, gender on speech slots), one step, for the record
    step(False)
    dt_ref, (lseg_ref, _), _ = timed(1, False)
    assert lseg_ref == lseg or args.timing_only, "dense and reference-semantics passes disagree"
    P = (seg.ctx.T + 1) // 2
    slots = {lab: 0 for lab in ('noEnergy', 'music', 'noise', 'female', 'male')}
    for lab, a, b in lseg:
        slots[lab] = slots.get(lab, 0) + (b - a)

    # ---- roofline, measured live with HIP events on the library's own stream around every launch of one extra dense step
    # (every rank runs it: the step holds the collective; rank 0's context is read)
    pj = _pmc_static()
    seg.ctx.prof_enable(True)
    seg.ctx.prof_reset()
    step(True)
    conv_ms, conv_launches, conv_flops = seg.ctx.prof_get(0)
    sk_ms, sk_launches, _ = seg.ctx.prof_get(1)
    other_ms, other_launches, _ = seg.ctx.prof_get(2)
    peak_tf = MFMA_BF16_PEAK_TF if x3 else MFMA_F32_PEAK_TF
    ktab, kdom = gemm_kernel_table([seg.ctx], peak_tf, pj)
    seg.ctx.prof_enable(False)
    all_tf = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    sk_bytes = n * 2 + seg.ctx.T * 25 * 4                # PCM16 in + (24 mel + 1 loge) f32 out
    kd = kdom or {"kernel": None, "achieved": 0.0, "frac": 0.0, "flops_per_launch": 0.0, "avg_launch_ms": 0.0, "launches": 0, "ms_per_step": 0.0}
    traffic_src = ("static_from_profiles: profiles/pmc_latest.json = rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate runs) of "
                   + str(pj.get('source', 'bench.py --minutes 20 --steps 1 --warmup 0 of the round-3 build'))
                   + "; per launch of THIS instantiation, 2 x FETCH_SIZE + WRITE_SIZE (the guide's gfx950 correction for 16-byte-per-lane "
                     "reads); PMC counters cannot be collected inside a bench run, and rocprofv3 --pmc crashes on the 1 h recording "
                     "(profiles/r05_pmc_1h_attempt.txt): the PMC run's bytes per launch are scaled by this run's / the PMC run's algorithmic "
                     "flops per launch of the same instantiation (traffic_scale_to_this_run; same pass geometry, other mix of full and "
                     "remainder passes)")
    roofline = {"bound": "mfma",
                "kernel": kd["kernel"],
                "achieved": kd["achieved"], "peak": peak_tf, "unit": "TFLOP/s", "frac": kd["frac"],
                "traffic": kd.get("traffic_per_launch_static"), "traffic_source": traffic_src,
                "traffic_pmc_run_mean_launch": kd.get("traffic_per_launch_pmc_run"), "traffic_scale_to_this_run": kd.get("traffic_scale_to_this_run"),
                "frac_own_flops": kd["frac"] * _own_flops_share(seg, kd["kernel"]),
                "flops_per_launch": kd["flops_per_launch"], "avg_launch_ms": kd["avg_launch_ms"],
                "launches_per_step": kd["launches"], "kernel_ms_per_step": kd["ms_per_step"],
                "mfma_executed_tflops": kd["achieved"] * (3 if x3 else 1),
                # what the matrix pipe SUSTAINS under the chip's power cap on register-resident operands with every CU busy
                # (tools/microbench/mfma_power_cap.hip, profiles/r06_mfma_power_cap.txt; static: measured once per round, on one box)
                "sustained_mfma_ceiling": ({"f16_relu_data_tflops": 1620.0, "bf16_relu_data_tflops": 1715.0, "zeros_tflops": 2440.0,
                                            "frac_of_ceiling": kd["achieved"] * 3 / (1620.0 if args.precision == 'f16x3' else 1715.0),
                                            "what": "v_mfma_f32_32x32x16 with nothing else in the loop, 0.4-1.3 s per case: random-normal "
                                                    "operands with relu'd A 1.62 PF (f16) / 1.72 PF (bf16) of the 2.5 PF dense peak, zeros "
                                                    "2.44 PF -- the clock follows the power the operands' bit activity costs; frac_of_ceiling = "
                                                    "this kernel's EXECUTED MFMA rate over that figure (profiles/r06_mfma_power_cap.txt)"}
                                           if x3 else None),
                "what": ("the kernel INSTANTIATION with the most HIP-event time in one dense step (template arguments spelled out: "
                         "<KH,KW,PADDED,TR,FUSED,NH,EPI>); achieved = its algorithmic flops per launch / its average launch duration. "
                         "split modes (f16x3 / bf16x3): three v_mfma_f32_32x32x16_{f16,bf16} per k-step on 16-bit hi/lo operand halves -- the "
                         "matrix pipe executes 3 x the algorithmic flops, so frac <= 1/3 by construction; the first layer of a fused launch "
                         "(2.5 % of its flops) is counted as the reference computes it (once per window) although it runs once per log-mel "
                         "row: frac_own_flops leaves that credit out")
                        if x3 else "conv_igemm_kernel (conv2d/dense implicit GEMM on v_mfma_f32_32x32x2_f32)",
                "kernels": ktab,
                "all_gemm_launches": {"achieved": all_tf, "frac": all_tf / peak_tf, "kernel_ms_per_step": conv_ms, "launches_per_step": conv_launches,
                                      "flops_per_step": conv_flops,
                                      "traffic_per_launch_static": pj.get('conv_hbm_bytes_per_launch_corrected_' + args.precision),
                                      "traffic_per_launch_static_raw": pj.get('conv_hbm_bytes_per_launch_' + args.precision)},
                "secondary": {"kernel": "sidekit_kernel (PCM16 -> log-energy + 24-band log-mel)", "bound": "hbm",
                              "achieved": sk_bytes / (sk_ms * 1e-3) / 1e9 if sk_ms > 0 else 0.0, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": (sk_bytes / (sk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if sk_ms > 0 else 0.0,
                              "kernel_ms_per_step": sk_ms, "algorithmic_bytes": sk_bytes},
                "other_kernels_ms_per_step": other_ms}

    # ---- exact-f32 companion (ISS_PREC_F32: v_mfma_f32_32x32x2_f32, bit-wise an fmaf chain), one timed dense step
    f32c = None
    if x3 and not args.no_f32_companion:
        seg.ctx.set_precision(_native.PREC_F32)
        step(True)
        dt32, (lseg32, _), _ = timed(1, True)
        seg.ctx.prof_enable(True)
        seg.ctx.prof_reset()
        step(True)
        c_ms, c_n, c_fl = seg.ctx.prof_get(0)
        seg.ctx.prof_enable(False)
        seg.ctx.set_precision({'bf16x3': _native.PREC_BF16X3, 'f16x3': _native.PREC_F16X3, 'f32': _native.PREC_F32}[args.precision])
        tf32 = c_fl / (c_ms * 1e-3) / 1e12 if c_ms > 0 else 0.0
        f32c = {"value": world * hours / dt32, "unit": "hours-of-audio/s", "ms_per_step": dt32 * 1e3, "achieved": tf32,
                "peak": MFMA_F32_PEAK_TF, "frac": tf32 / MFMA_F32_PEAK_TF, "segments_equal_split_mode": lseg32 == lseg,
                "dtype": "f32 (v_mfma_f32_32x32x2_f32, exact f32 products and accumulation)"}

    cpu = par = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        host = pcm[:min(n, 660 * FS)].cpu().numpy()
        cpu, nsec, det = cpu_baseline(seg, host)
        par = parity_check(seg, host, nsec, det)

    # ---- companions: the other configurations, driver-timed in the same line.  archive (configs[3] shape) at every N -- the
    # file-based path gets its own 1 -> N curve --, batch (configs[2]) and vbx (configs[4]) on one GPU.
    companions = None
    if not args.no_companions:
        del pcm
        torch.cuda.empty_cache()
        companions = {}
        arch = bench_files(args, torch, dev, local_rank, rank, world, 'archive', seg=seg, comm=comm, comm_kind=comm_kind,
                           steps=2, warmup=1, per_gpu=args.companion_files_per_gpu)
        if rank == 0:
            companions["archive"] = _companion_view(arch)
        if world == 1:
            companions["batch"] = _companion_view(bench_files(args, torch, dev, local_rank, rank, world, 'batch', seg=seg, steps=2, warmup=1, dense=False))
            # the same files with both nets on every slot: the file path's counterpart of the headline (dense, resident) figure
            companions["batch_dense"] = _companion_view(bench_files(args, torch, dev, local_rank, rank, world, 'batch', seg=seg, steps=2, warmup=1,
                                                                    dense=True))
            seg.dense_batches = False
            vb = bench_vbx(args, torch, dev, local_rank, rank, world, steps=2, warmup=1, cpu_leg=False)
            companions["vbx"] = _companion_view(vb)

    if comm:
        comm.barrier()
    if rank == 0:
        vad_f = seg.ctx.cnn_flops(0)
        gen_f = seg.ctx.cnn_flops(1)
        line = {
            "metric": "hours-of-audio segmented/sec (smn+gender, 16 kHz mono)",
            "value": value, "unit": "hours-of-audio/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": (f"f32 ({args.precision} split-operand MFMA, f32 accumulate; f64 FFT)" if x3 else "f32 (f32 MFMA; f64 FFT)"),
            "data": "synthetic",
            "x_realtime_per_gpu": value * 3600.0 / world,
            "config": {"workload": f"BASELINE.json configs[1] input ({args.minutes:g} min synthetic 16 kHz mono PCM16 per GPU, "
                                   "resident in HBM) through smn VAD + gender (the metric's nets; configs[1] itself lists smn only), "
                                   "dense mode: both CNNs on 100% of the 20 ms slots; the same workload and code path at every --gpus N "
                                   "(one recording per rank, one all-gather of the segment tables per step when N > 1)",
                       "audio_hours_per_step_per_gpu": hours, "slots_per_step_per_gpu": P,
                       "weights": "seeded stand-ins, (68,21,1)->3 and (68,24,1)->2, ~1.25 M params each, last layer calibrated on the generator's "
                                  "ground truth and the reference's musanmix goldens (tests/golden/make_standin_heads.py); the real "
                                  "Keras files are un-vendored release assets",
                       "vad_flops_per_slot": vad_f, "gender_flops_per_slot": gen_f,
                       "parallelism": (f"file-parallel x{world} (one recording per rank, no data-path collective), one all-gather of int32 "
                                       f"segment tables per step: {comm_kind}") if world > 1 else "single GPU",
                       "segments": len(lseg), "label_slots": slots, "cnn_driven_boundaries": cnn_driven_boundaries(lseg),
                       "reference_semantics": {"value": world * hours / dt_ref, "unit": "hours-of-audio/s",
                                               "ms_per_step": dt_ref * 1e3,
                                               "vad_slot_frac": 1.0 - slots['noEnergy'] / P,
                                               "gender_slot_frac": (slots['female'] + slots['male']) / P}},
            "roofline": roofline,
            "ranks": {"ms_per_step_min": dt_min / args.steps * 1e3, "ms_per_step_max": dt / args.steps * 1e3,
                      "what": "each rank's own wall time for the timed steps (barrier to barrier), min / max over the ranks"},
        }
        if rccl is not None:
            line["rccl"] = rccl
        if args.comm == 'gloo' and world > 1:
            line["rehearsal"] = {"comm": "gloo", "shared_devices": args.shared_devices,
                                 "what": "the N-rank code path of this file (launcher, ranks, barriers, max-over-ranks timing, one all-gather "
                                         "of the segment tables per step, archive companion at N) executed with a host-memory exchange; when "
                                         "shared_devices is set the ranks time-share the GPU(s), so `value` is NOT a scaling figure"}
        if f32c is not None:
            line["precision_f32"] = f32c
        if cpu is not None:
            line["cpu_baseline"] = cpu
            line["parity_check"] = par
        # precision guard (include/iss.h iss_set_precision_guard): what the library measured on the weights in use at their first call
        line["precision_guard"] = {"threshold_dlogp": getattr(seg.ctx, 'guard_threshold', None) or 5e-4,
                                   "vad": seg.ctx.cnn_precision_info(seg.vad.net_id),
                                   "gender": seg.ctx.cnn_precision_info(seg.gender.net_id) if seg.detect_gender else None,
                                   "what": "max |log p(split 16-bit operands) - log p(exact f32)| over up to 256 windows of each network's "
                                           "first call, measured by the library; above the threshold it switches that network to the other "
                                           "split mode or, failing that, to exact f32 (north star: within 1e-3)"}
        if companions:
            line["companions"] = companions
        print(json.dumps(line))
    seg.close()


def _companion_view(line):
    """The fields of a companion run's own line that the headline line carries (the full line is what
    `--workload <name>` prints on its own)."""
    if line is None:
        return None
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "x_realtime", "x_realtime_per_gpu", "scaling", "dtype")
    out = {k: line[k] for k in keep if k in line}
    cfg = line.get("config", {})
    out["config"] = {k: cfg[k] for k in ("workload", "files", "files_per_gpu", "minutes_per_file", "audio_hours_per_step", "ms_per_file_per_gpu",
                                          "features_ms_per_step", "xvectors_ms_per_step", "parallelism") if k in cfg}
    r = line.get("roofline", {})
    out["roofline"] = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel_ms_per_step") if k in r}
    if r.get("dominant"):
        out["roofline"]["dominant"] = {k: r["dominant"][k] for k in ("kernel", "ms_per_step", "launches", "achieved", "frac") if k in r["dominant"]}
    for k in ("ranks", "rccl"):
        if k in line:
            out[k] = line[k]
    return out


if __name__ == '__main__':
    main()
