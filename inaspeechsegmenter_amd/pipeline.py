"""Multi-file engine behind `Segmenter.batch_process` and the archive driver.

The reference processes a file list one file at a time: features of file i+1 are extracted on a helper thread while the
main thread runs the two Keras `predict` calls, the Viterbi smoothing and the export of file i
(segmenter.py:297-335, medialist2feats :338-374).  On an MI355X a 5-minute file is ~14 ms of device work, so per-file
launches, copies and Python bookkeeping become the limit.  Here files are processed in SUPER-BATCHES:

  decode threads   N files  ->  int16 PCM (RIFF parse, or the ffmpeg pipe)
  packer           the PCM of a super-batch (default <= 32 files / ~40 min of audio) is laid end to end in ONE page-locked
                   buffer, every file starting on a multiple of 160 samples: frame t of file f is then frame
                   off_f / 160 + t of the concatenation, and the frames that straddle two files are simply never used
  device worker    ONE H2D copy, ONE sidekit launch, one log-energy read-back; per-file energy Viterbi (compiled);
                   ONE iss_cnn_probs call for the VAD windows of all files, per-segment Viterbi; ONE call for the
                   gender windows, Viterbi; hand the segment lists to the exporter
  exporter         CSV / TextGrid writers

Four device workers with a context each (own stream, own workspace) take super-batches in turn, so while some are in their
host phases (packing, Viterbi, bookkeeping) the others' kernels run.  Results are identical to per-file processing: every frame and
every 20 ms slot is computed from the same samples by the same kernels (tests/test_gpu_segmenter.py).
"""
import queue
import sys
import threading
import time
import warnings

import numpy as np

from . import _native
from .io import decode_pcm

FRAME_HOP = 160
MIN_SAMPLES = 400 + FRAME_HOP * 67                  # 68 frames: shorter media take the single-file path (mspec padding)


class _Batch:
    __slots__ = ('idx', 'sigs', 'names')

    def __init__(self):
        self.idx, self.sigs, self.names = [], [], []

    def samples(self):
        return sum(-(-s.size // FRAME_HOP) * FRAME_HOP for s in self.sigs)


class _AudioBudget:
    """Bounds the decoded-but-not-yet-packed audio by DURATION (the item count of the queue says nothing about memory:
    a 2 h file is 230 MB of PCM16).  A decode thread that holds a decoded signal waits until the signals queued in front
    of it fit the budget again; a single signal larger than the budget passes when nothing else is queued."""

    def __init__(self, max_samples):
        self.max, self.held, self.cv = int(max_samples), 0, threading.Condition()

    def acquire(self, n):
        with self.cv:
            while self.held > 0 and self.held + n > self.max:
                self.cv.wait(0.5)
            self.held += n

    def release(self, n):
        with self.cv:
            self.held -= n
            self.cv.notify_all()


def _decode_stage(items, ffmpeg, nbtry, trydelay, out_q, nthreads, budget=None):
    """items: [(index, src)].  Puts (index, src, sig | None, errtext | None) on out_q in completion order, then None."""
    import random
    in_q = queue.Queue()
    for it in items:
        in_q.put(it)

    def work():
        while True:
            try:
                i, src = in_q.get_nowait()
            except queue.Empty:
                return
            sig, err, itry = None, None, 0
            while sig is None and itry < nbtry:
                try:
                    sig = decode_pcm(src, None, None, ffmpeg)
                except:                                            # noqa: E722  (reference semantics, segmenter.py:364-370)
                    itry += 1
                    err = 'error: ' + str(sys.exc_info()[0])
                    if itry != nbtry:
                        time.sleep(random.random() * trydelay)
            if budget is not None and sig is not None:
                budget.acquire(sig.size)
            out_q.put((i, src, sig, err))

    ths = [threading.Thread(target=work, daemon=True) for _ in range(max(1, nthreads))]
    for t in ths:
        t.start()

    def closer():
        for t in ths:
            t.join()
        out_q.put(None)
    threading.Thread(target=closer, daemon=True).start()


class _Worker:
    """One device context + the networks of `seg` loaded on it."""

    def __init__(self, seg, ctx=None):
        from . import tables
        self.seg = seg
        if ctx is None:
            ctx = _native.Context(seg.ctx.device)
            ctx.sidekit_tables(tables.sidekit_window(), tables.sidekit_melbank())
            ctx.cnn_load(seg.vad.net_id, seg.vad.compiled)
            if seg.detect_gender:
                ctx.cnn_load(seg.gender.net_id, seg.gender.compiled)
            self.owned = True
        else:
            self.owned = False
        self.ctx = ctx
        self.pin = None
        # wall seconds this worker spent per phase since the last reset (bench.py reports them: where a step's host time goes)
        self.stats = {k: 0.0 for k in ('pack', 'features', 'energy_host', 'cnn_device', 'smooth_host', 'batches', 'files')}

    def close(self):
        if self.owned:
            self.ctx.close()

    def pinned(self, nsamples):
        if self.pin is None or self.pin.size < nsamples:
            if self.pin is not None:
                self.ctx.pinned_free(self.pin)
            self.pin = self.ctx.pinned_empty((int(nsamples * 1.25) + 4096,), np.int16)
        return self.pin

    def sync_settings(self):
        """Arithmetic mode and workspace cap follow the Segmenter's own context (they may have changed since this
        worker was created: the workers are cached between calls)."""
        if self.owned:
            self.ctx.mirror_settings(self.seg.ctx)

    def run(self, batch):
        """-> [ [(label, start_slot, stop_slot)] per file of the batch ]"""
        from . import segmenter as S
        seg, ctx = self.seg, self.ctx
        st = self.stats
        t_ = time.perf_counter()
        offs, pos = [], 0
        for s in batch.sigs:
            offs.append(pos)
            pos += -(-s.size // FRAME_HOP) * FRAME_HOP
        buf = self.pinned(pos)
        for s, o in zip(batch.sigs, offs):
            if s.dtype == np.int16:
                buf[o:o + s.size] = s
            else:                                    # float sources: what libsndfile's float32 read holds, re-quantised is NOT exact
                raise TypeError('float media take the single-file path')
            buf[o + s.size:o + -(-s.size // FRAME_HOP) * FRAME_HOP] = 0
        st['pack'] += time.perf_counter() - t_; t_ = time.perf_counter()
        ctx.set_signal(buf[:pos])
        ctx.sidekit()
        loge = ctx.get_loge()
        st['features'] += time.perf_counter() - t_; t_ = time.perf_counter()
        g0 = [o // FRAME_HOP for o in offs]
        nfr = [(s.size - 400) // FRAME_HOP + 1 for s in batch.sigs]
        # energy segmentation per file (segmenter.py:261-267)
        lsegs = []
        for f in range(len(batch.sigs)):
            le = loge[g0[f]:g0[f] + nfr[f]]
            lseg = []
            for lab, start, stop in S._binidx2seglist(S._energy_activity(le, seg.energy_ratio)[::2]):
                lseg.append(('noEnergy' if lab == 0 else 'energy', start, stop))
            lsegs.append(lseg)
        st['energy_host'] += time.perf_counter() - t_
        wrs = [S._window_rows(nfr[f]) + np.int32(g0[f]) for f in range(len(lsegs))]
        # Segmenter.dense_batches (an extension, default False): every network on EVERY slot of every file -- the most work
        # the reference semantics can require, independent of what the networks decide (bench.py's "dense" figures); the rows
        # of the `inlabel` segments are then picked from the full result, so the segments are the same
        dense = bool(getattr(seg, 'dense_batches', False))
        if dense:
            wbase = np.concatenate(([0], np.cumsum([len(w) for w in wrs])))
            allw = np.concatenate(wrs)
        for net in ([seg.vad, seg.gender] if seg.detect_gender else [seg.vad]):
            t_ = time.perf_counter()
            # slots of every `inlabel` segment of every file, in order (segmenter.py:156-159), in ONE device call; the
            # per-segment smoothing (:168-178) in ONE host call on np.log of the whole probability array (elementwise: the
            # same values as per-segment np.log calls); run-length encoding vectorised over the pass
            rows, seglen, sel = [], [], []
            for f, lseg in enumerate(lsegs):
                wr = wrs[f]
                for lab, start, stop in lseg:
                    if lab == net.inlabel:
                        rows.append(wr[start:stop])
                        seglen.append(stop - start)
                        if dense:
                            sel.append(np.arange(wbase[f] + start, wbase[f] + stop))
            if not rows:
                st['smooth_host'] += time.perf_counter() - t_
                continue
            allrows = np.concatenate(rows)
            st['smooth_host'] += time.perf_counter() - t_; t_ = time.perf_counter()
            if dense:
                probs, _fin = net.probs(ctx, allw)
                probs = probs[np.concatenate(sel)]
            else:
                probs, _fin = net.probs(ctx, allrows)
            st['cnn_device'] += time.perf_counter() - t_; t_ = time.perf_counter()
            with np.errstate(divide='ignore'):
                logp = np.log(probs)
            states = _native.viterbi_segments(logp, seglen, S.diag_trans_exp(net.viterbi_arg, len(net.outlabels)))
            seg_off = np.concatenate(([0], np.cumsum(seglen)))
            change = np.empty(len(states), dtype=bool)
            change[0] = True
            np.not_equal(states[1:], states[:-1], out=change[1:])
            change[seg_off[:-1]] = True
            run_start = np.flatnonzero(change)
            run_stop = np.concatenate((run_start[1:], [len(states)]))
            run_seg = np.searchsorted(seg_off, run_start, side='right') - 1
            run_lab = states[run_start].tolist()
            rs = (run_start - seg_off[run_seg]).tolist()
            re_ = (run_stop - seg_off[run_seg]).tolist()
            run_seg = run_seg.tolist()
            out, k, sidx = [], 0, 0
            for lseg in lsegs:
                ret = []
                for lab, start, stop in lseg:
                    if lab != net.inlabel:
                        ret.append((lab, start, stop))
                        continue
                    while k < len(run_seg) and run_seg[k] == sidx:
                        ret.append((net.outlabels[run_lab[k]], start + rs[k], start + re_[k]))
                        k += 1
                    sidx += 1
                out.append(ret)
            lsegs = out
            st['smooth_host'] += time.perf_counter() - t_
        st['batches'] += 1
        st['files'] += len(batch.sigs)
        return lsegs


DEFAULT_BATCH_FILES = 32          # files per device pass at most ...
DEFAULT_BATCH_SECONDS = 40 * 60   # ... and ~40 minutes of audio (120 k slots: four full-size passes of each network).  Smaller
RAMP_SECONDS = 300                # first pass of a call; doubles per pass up to batch_seconds
DEFAULT_WORKERS = 4               # passes with more contexts in flight overlap the host phases better than 32-file passes on
                                  # two contexts did (same box, 128 x 5 min files: 7.2 -> 8.2 audio-hours/s)


def process_files(seg, linput, on_result, skip=None, nbtry=1, trydelay=2., batch_files=None, batch_seconds=None,
                  workers=None, decode_threads=4):
    """Segment `linput` with the networks of `seg`; on_result(index, src, lseg | None, errtext | None, secs) is called from
    the worker threads (serialised by a lock) as results become available, lseg = [(label, start_sec, stop_sec)], secs =
    this file's share of the processing time of its device pass (the pass's wall time / its files: what the reference's
    per-file 'ok <secs>' message reports, segmenter.py:322-327, when files are processed one by one).
    skip: set of indices not to process.  Device failures (NativeError) and exceptions raised by on_result (unwritable
    outputs) propagate to the caller: the first one is re-raised here once every stage has drained."""
    from . import segmenter as S
    batch_files = batch_files or DEFAULT_BATCH_FILES
    batch_seconds = batch_seconds or DEFAULT_BATCH_SECONDS
    workers = workers or DEFAULT_WORKERS
    items = [(i, src) for i, src in enumerate(linput) if not (skip and i in skip)]
    dec_q = queue.Queue(maxsize=4 * batch_files)
    budget = _AudioBudget(2 * batch_seconds * 16000)        # decoded audio waiting for the packer: <= 2 super-batches
    _decode_stage(items, seg.ffmpeg, nbtry, trydelay, dec_q, decode_threads, budget)
    batch_q = queue.Queue(maxsize=max(2, workers))
    lock = threading.Lock()
    failure = []

    def packer():
        cur = _Batch()
        nbatch = 0
        ended = False                                              # the decoders' end marker has been read
        try:
            while True:
                it = dec_q.get()
                if it is None:
                    ended = True
                    break
                i, src, sig, err = it
                if sig is not None:
                    budget.release(sig.size)
                if failure:                                        # a worker failed: drain the decoders, pack nothing more
                    continue
                if sig is None:
                    try:
                        with lock:
                            on_result(i, src, None, err, 0.0)
                    except BaseException as exc:                   # noqa: B902
                        failure.append(exc)
                    continue
                if sig.dtype != np.int16 or sig.size < MIN_SAMPLES:
                    batch_q.put(('single', i, src, sig))
                    continue
                cur.idx.append(i); cur.sigs.append(sig); cur.names.append(src)
                # the first passes of a call are short (5, 10, 20 ... minutes of audio) so that the device starts as soon as
                # a few files are decoded instead of after a whole 40-minute pass per worker
                lim_s = min(batch_seconds, RAMP_SECONDS * (1 << min(nbatch, 20)))
                if len(cur.idx) >= batch_files or cur.samples() >= lim_s * 16000:
                    batch_q.put(('batch', cur))
                    cur = _Batch()
                    nbatch += 1
            if cur.idx and not failure:
                batch_q.put(('batch', cur))
        except BaseException as exc:                               # noqa: B902  (MemoryError while batching, a bad array, ...)
            # record it and keep draining the decoders -- they block on the bounded queue / the audio budget while holding
            # their PCM -- until their end marker, exactly like the `if failure: continue` path above
            failure.append(exc)
            while not ended:                                       # (an exception AFTER the marker was read: nothing left to drain --
                it = dec_q.get()                                   #  a second get() would block forever)
                if it is None:
                    break
                if it[2] is not None:
                    budget.release(it[2].size)
        finally:
            for _ in range(workers):
                batch_q.put(None)

    def work(w):
        # A failure (device error, or an exception out of on_result) is recorded and the loop KEEPS consuming batch_q until
        # the packer's end marker: a worker that simply returned would leave the packer blocked on the bounded queue (and
        # the decode threads behind it) with nobody left to drain it, and process_files would hang instead of raising.
        while True:
            job = batch_q.get()
            if job is None:
                return
            if failure:
                continue
            try:
                t0 = time.time()
                if job[0] == 'single':
                    _, i, src, sig = job
                    try:
                        with warnings.catch_warnings():
                            warnings.simplefilter('ignore')
                            mspec, loge, difflen = S._sig2feats(w.ctx, sig, src)
                        lseg = [(lab, a * .02, b * .02) for lab, a, b in _slots(seg, w.ctx, mspec, loge, difflen)]
                        res = (lseg, None)
                    except ValueError as exc:                      # too short to analyse: a per-file error
                        res = (None, 'error: %s %s' % (type(exc), exc))
                    with lock:
                        on_result(i, src, res[0], res[1], time.time() - t0)
                    continue
                b = job[1]
                lsegs = w.run(b)
                share = (time.time() - t0) / max(len(b.idx), 1)
                with lock:
                    for i, src, lseg in zip(b.idx, b.names, lsegs):
                        on_result(i, src, [(lab, a * .02, c * .02) for lab, a, c in lseg], None, share)
            except BaseException as exc:                           # noqa: B902  propagate to the caller's thread
                failure.append(exc)

    # device workers are kept with the Segmenter between calls (context creation, network upload and the first
    # workspace allocation cost more than a 5-minute file)
    cache = seg.__dict__.setdefault('_pipeline_workers', [])
    while len(cache) < max(1, workers):
        cache.append(_Worker(seg, seg.ctx if not cache else None))
    ws = cache[:max(1, workers)]
    for w in ws:
        w.sync_settings()
    threads = [threading.Thread(target=work, args=(w,), daemon=True) for w in ws]
    pk = threading.Thread(target=packer, daemon=True)
    pk.start()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    pk.join()
    if failure:
        raise failure[0]


def close_workers(seg):
    for w in seg.__dict__.pop('_pipeline_workers', []):
        w.close()


def _slots(seg, ctx, mspec, loge, difflen):
    """segment_slots of `seg` on another context (the networks are loaded there under the same ids)."""
    from . import segmenter as S
    lseg = []
    for lab, start, stop in S._binidx2seglist(S._energy_activity(loge, seg.energy_ratio)[::2]):
        lseg.append(('noEnergy' if lab == 0 else 'energy', start, stop))
    for net in ([seg.vad, seg.gender] if seg.detect_gender else [seg.vad]):
        lseg = net(mspec, lseg, difflen, ctx=ctx)
    return lseg
