"""Long-archive driver: file-parallel over the GPUs of one node (BASELINE.json configs[3]).

The reference scales out with a Pyro4 pull queue handing (src, dst) pairs to independent
`batch_process` workers (scripts/ina_speech_segmenter_pyro_server.py:34-68, ..._client.py:64-74).
Here the same independence is used inside one node: one process per GPU (launch with
`python -m torch.distributed.run --nproc-per-node N ...` or any launcher that sets RANK / WORLD_SIZE), files dealt to ranks by size
(sharding.shard_files), every rank runs its own files through its own Segmenter and writes their
outputs, and ONE all-gather of int32 segment tables (`iss_allgather_segments`: ncclAllGather from librccl, no torch)
leaves the complete {file: segments} table on every rank.  Without a communicator it degrades to a plain loop.
"""
import os
import sys
import time

import numpy as np

from . import sharding
from ._native import NativeError
from .export_funcs import seg2csv, seg2textgrid


def _default_comm(device):
    """A running torch.distributed process group, if the caller set one up (CPU tests: gloo); else single process.
    torch is only looked at when the caller has already imported it."""
    dist = getattr(sys.modules.get('torch'), 'distributed', None)
    if dist is not None and dist.is_available() and dist.is_initialized():
        return sharding.TorchComm(device=device)
    return None


def segment_archive(segment_file, linput, loutput=None, output_format='csv', sizes=None, skipifexist=False,
                    capacity=None, device=None, comm=None, batch_files=None, workers=None, batch_seconds=None):
    """segment_file: a `Segmenter` (its share of the files runs through pipeline.process_files), or any callable
    path -> [(label, start_sec, stop_sec)] with times on the 20 ms grid (what Segmenter.__call__ returns).  linput / loutput: all files, identical on every rank.
    Returns (table, lmsg): table = {file_index: [(label, start_sec, stop_sec)]} for ALL files (gathered),
    lmsg = this rank's [(dst, code, text)] in the reference's batch_process convention
    (0 ok / 1 already exists / 2 error, segmenter.py:352,370,372).
    comm: sharding.RcclComm (GPU box: the C-ABI's ncclAllGather) or sharding.TorchComm; None = a running
    torch.distributed group if there is one, else a single process."""
    if comm is None:
        comm = _default_comm(device)
    world = comm.world if comm else 1
    rank = comm.rank if comm else 0
    if output_format not in ('csv', 'textgrid'):
        raise NotImplementedError()
    fexport = seg2csv if output_format == 'csv' else seg2textgrid
    if sizes is None:
        sizes = [os.path.getsize(f) if os.path.exists(f) else 0 for f in linput]
    mine = sharding.shard_files(sizes, world)[rank]
    # A collective cannot lose a participant: if this rank's local work fails (device error, unwritable output) it still
    # takes part in the all-gather, flagged, so that every other rank raises `sharding.RankFailure` instead of waiting
    # for it; then the original exception is re-raised here.  (A rank that is gone altogether is caught by the
    # communicator's timeout: ISS_COMM_TIMEOUT_S on RCCL, the process group's timeout on torch.distributed.)
    cap = capacity or max(1024, 64 * (len(linput) // world + 1))      # the same on every rank, failing or not
    try:
        local, lmsg = _segment_share(segment_file, linput, loutput, fexport, mine, skipifexist,
                                     dict(batch_files=batch_files, workers=workers, batch_seconds=batch_seconds))
    except BaseException:
        if comm:
            try:
                comm.allgather(None, cap)
            except Exception:                                  # noqa: BLE001  the original failure is the one to report
                pass
        raise
    if comm:
        allrows = comm.allgather(local, cap)
    else:
        allrows = local
    return sharding.unpack_segments(allrows), lmsg


def _segment_share(segment_file, linput, loutput, fexport, mine, skipifexist, pipe_kw):
    """This rank's files -> ((k,4) int32 segment rows, lmsg)."""
    rows, lmsg = [], []
    if hasattr(segment_file, 'batch_process') and hasattr(segment_file, 'ctx'):
        # a Segmenter: this rank's share runs through the multi-file pipeline (super-batches, two device contexts)
        from . import pipeline
        seg = segment_file
        mine = sorted(mine)
        msgs, skip = {}, set()
        for k, i in enumerate(mine):
            dst = loutput[i] if loutput is not None else None
            if skipifexist and dst is not None and os.path.exists(dst):
                msgs[k] = (dst, 1, 'already exists')
                skip.add(k)
            elif dst is not None:
                d = os.path.dirname(dst)
                if d and not os.path.isdir(d):
                    os.makedirs(d, exist_ok=True)
        got = {}

        def on_result(k, src, lseg, err, secs=0.0):
            dst = loutput[mine[k]] if loutput is not None else None
            if lseg is None:
                msgs[k] = (dst, 2, err)
                return
            b = time.time()
            if dst is not None:
                fexport(lseg, dst)
            got[k] = sharding.pack_segments(mine[k], [(lab, int(round(s / .02)), int(round(e / .02))) for lab, s, e in lseg])
            msgs[k] = (dst, 0, 'ok ' + str(secs + time.time() - b))      # per-file time, segmenter.py:322-327

        pipeline.process_files(seg, [linput[i] for i in mine], on_result, skip=skip, **pipe_kw)
        rows = [got[k] for k in sorted(got)]
        lmsg = [msgs[k] for k in range(len(mine))]
        mine = []
    for i in sorted(mine):
        dst = loutput[i] if loutput is not None else None
        if skipifexist and dst is not None and os.path.exists(dst):
            lmsg.append((dst, 1, 'already exists'))
            continue
        b = time.time()
        try:
            lseg = segment_file(linput[i])
        except NativeError:                                   # a broken device context is not a per-file problem
            raise
        except Exception as exc:                              # undecodable / missing / too short media do not stop the
            lmsg.append((dst, 2, 'error: %s %s' % (type(exc), exc)))   # archive (segmenter.py:364-370; ffmpeg failures are
            continue                                          # plain Exception(stderr), io.py:72-75)
        if dst is not None:                                   # export errors propagate, as in the reference (:322)
            d = os.path.dirname(dst)
            if d and not os.path.isdir(d):
                os.makedirs(d, exist_ok=True)
            fexport(lseg, dst)
        slots = [(lab, int(round(s / .02)), int(round(e / .02))) for lab, s, e in lseg]
        rows.append(sharding.pack_segments(i, slots))
        lmsg.append((dst, 0, 'ok ' + str(time.time() - b)))
    local = np.concatenate(rows, axis=0) if rows else np.zeros((0, 4), np.int32)
    return local, lmsg


class JobList:
    """The job-table half of the reference's Pyro job server without the RPC layer -- deliberately a line-for-line mirror of
    `GenderJobServer` (scripts/ina_speech_segmenter_pyro_server.py:33-66) minus its prints and the Pyro4 decorators, because
    its behaviour IS the compatibility contract (pinned by the reference's own run_test.py:166-172 on media/pyroserver_test.csv): a CSV with columns `source_path, dest_path` is
    stripped, de-duplicated and shuffled, and handed out in chunks of `nbjobs` (20 by default).  On one node the
    chunks feed `segment_archive`; the Pyro daemon itself (network job dispatch) is out of scope."""

    def __init__(self, csvjobs, shuffle=True, seed=None):
        self.set_jobs(csvjobs, shuffle, seed)

    def set_jobs(self, csvjobs, shuffle=True, seed=None):
        import pandas as pd
        df = pd.read_csv(csvjobs)
        df.source_path = df.source_path.str.strip()
        df.dest_path = df.dest_path.str.strip()
        df = df.drop_duplicates()
        if shuffle:
            df = df.sample(frac=1, random_state=seed)
        df = df.reset_index(drop=True)
        self.lsource = list(df.source_path)
        self.ldest = list(df.dest_path)
        self.i = 0
        return '%s jobs have been set' % csvjobs

    def get_job(self, msg=''):
        self.i += 1
        return (self.lsource.pop(0), self.ldest.pop(0))

    def get_njobs(self, msg='', nbjobs=20):
        ret = (self.lsource[:nbjobs], self.ldest[:nbjobs])
        self.lsource = self.lsource[nbjobs:]
        self.ldest = self.ldest[nbjobs:]
        self.i += nbjobs
        return ret

    def has_more_jobs(self):
        return len(self.lsource) > 0
