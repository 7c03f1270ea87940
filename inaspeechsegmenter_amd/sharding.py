"""File-parallel sharding of a long archive over the GPUs of one node + the single exchange step.

The reference has no multi-GPU code; its only scale-out facility is a Pyro4 pull queue that
hands (src, dst) pairs to independent workers (scripts/ina_speech_segmenter_pyro_server.py:34-68)
because files are independent units (segmenter.py:314-327 loops over them with no shared state).
Here: one process per GPU, files are dealt to ranks up front, every rank segments its own files,
and ONE all-gather of a fixed-capacity int32 table collects all segment boundaries on every rank.
On the GPU box the collective is `iss_allgather_segments` of the C-ABI (include/iss.h): ncclAllGather
from librccl on the context's own stream -- no torch in the data path (`RcclComm`; the 128-byte
communicator id travels through the launcher's key-value store or a plain TCP socket, `rccl_rendezvous`).  `TorchComm` (torch.distributed,
"gloo") exists for the world_size-2 CPU tests and for callers that already run a process group.

Segment table row = (file_id, label_id, start_slot, stop_slot) int32; times are slot * 0.02 s
(segmenter.py:276) and are materialised on the host after the gather.  Row 0 of each rank's
buffer is a header (n_rows, capacity, rank, 0) so that the common case really is a single
collective; only if some rank overflowed the agreed capacity is a second, larger gather issued.
"""
import numpy as np

LABELS = ('noEnergy', 'energy', 'speech', 'music', 'noise', 'female', 'male')
LABEL_ID = {l: i for i, l in enumerate(LABELS)}


def shard_files(sizes, world):
    """Longest-processing-time-first assignment of files to `world` ranks.
    sizes: per-file cost (e.g. sample count).  Returns a list of index lists, one per rank;
    equal sizes degenerate to round-robin (i mod world)."""
    sizes = np.asarray(sizes, dtype=np.int64)
    order = np.argsort(-sizes, kind='stable')
    load = np.zeros(world, dtype=np.int64)
    out = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(load))          # first minimum -> deterministic
        out[r].append(int(i))
        load[r] += int(sizes[i])
    return out


def pack_segments(file_id, lseg):
    """[(label, start_slot, stop_slot)] -> (k,4) int32 rows."""
    rows = np.empty((len(lseg), 4), dtype=np.int32)
    for i, (lab, a, b) in enumerate(lseg):
        rows[i] = (file_id, LABEL_ID[lab], a, b)
    return rows


def unpack_segments(rows, start_sec=0):
    """(k,4) int32 rows -> {file_id: [(label, start_sec + a*.02, start_sec + b*.02)]} in row order."""
    out = {}
    for fid, lid, a, b in np.asarray(rows).tolist():
        out.setdefault(fid, []).append((LABELS[lid], start_sec + a * .02, start_sec + b * .02))
    return out


class RankFailure(RuntimeError):
    """Some rank reported that its local work failed (header count -1): every rank raises, none keeps a partial table."""


def _check_counts(counts):
    bad = [r for r, n in enumerate(counts) if int(n) < 0]
    if bad:
        raise RankFailure('rank(s) %s failed before the all-gather of the segment tables; no table was assembled' % bad)


def _merge_gathered(parts, counts, capacity, regather):
    """parts[r]: (capacity,4) rows of rank r, counts[r] = rows it has; one larger gather if someone overflowed."""
    _check_counts(counts)
    need = int(max(counts))
    if need > capacity:                      # rare: some rank had more rows than agreed -> one bigger gather
        parts, counts = regather(need)
    out = [np.asarray(parts[r])[:int(counts[r])] for r in range(len(counts))]
    return np.concatenate(out, axis=0) if out else np.zeros((0, 4), np.int32)


class RcclComm:
    """The exchange step on the GPU box: a native context whose communicator was set up by `rccl_rendezvous`."""

    def __init__(self, ctx):
        self.ctx, self.rank, self.world = ctx, ctx.comm_rank, ctx.comm_world

    def allgather(self, rows, capacity):
        """rows = None: this rank's local work failed -- it still takes part so that the others learn about it."""
        if rows is not None:
            rows = np.ascontiguousarray(rows, dtype=np.int32).reshape(-1, 4)
        first = self.ctx.allgather_segments(rows, int(capacity))
        return _merge_gathered(first[0], first[1], int(capacity), lambda cap: self.ctx.allgather_segments(rows, cap))

    def info(self):
        return self.ctx.comm_info()

    def max_over_ranks(self, value):
        return self.ctx.comm_allreduce_max(value)

    def barrier(self):
        self.ctx.comm_allreduce_max(0.0)


class TorchComm:
    """Same interface on a torch.distributed process group ("gloo" in the CPU tests)."""

    def __init__(self, group=None, device=None):
        import torch.distributed as dist
        self.group, self.device = group, device
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def allgather(self, rows, capacity):
        return allgather_segment_tables(rows, capacity=capacity, device=self.device, group=self.group)

    def max_over_ranks(self, value):
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device or 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def barrier(self):
        import torch.distributed as dist
        dist.barrier(self.group)


_uid_round = [0]


def exchange_unique_id(uid, rank, world, addr=None, port=None, timeout=120.0):
    """Rank 0 hands `uid` (bytes) to the other ranks; returns the id on every rank.

    Under `python -m torch.distributed.run` (TORCHELASTIC_USE_AGENT_STORE=True) the launcher's own key-value store at
    MASTER_ADDR:MASTER_PORT carries it -- the channel the launcher provides for exactly this, no extra port.  Otherwise
    (any other launcher that sets RANK / WORLD_SIZE) a plain TCP socket: addr / port default to MASTER_ADDR (127.0.0.1)
    and ISS_RDV_PORT or MASTER_PORT + 1."""
    import os
    import socket
    import time
    if world == 1:
        return uid
    if port is None and addr is None and os.environ.get('TORCHELASTIC_USE_AGENT_STORE', '').lower() == 'true':
        from datetime import timedelta
        from torch.distributed import TCPStore
        store = TCPStore(os.environ.get('MASTER_ADDR', '127.0.0.1'), int(os.environ['MASTER_PORT']), world, False,
                         timedelta(seconds=timeout))
        # the agent's store outlives worker restarts: key the id by the restart count as well, or a restarted rank > 0
        # could read the previous incarnation's id before rank 0 has published the new one
        key = 'iss_rccl_unique_id_r%s_%d' % (os.environ.get('TORCHELASTIC_RESTART_COUNT', '0'), _uid_round[0])
        _uid_round[0] += 1
        if rank == 0:
            store.set(key, bytes(uid))
            return uid
        return bytes(store.get(key))
    addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = int(port or os.environ.get('ISS_RDV_PORT') or int(os.environ.get('MASTER_PORT', '29500')) + 1)
    if rank == 0:
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(world)
        srv.settimeout(timeout)
        seen = set()
        try:
            while len(seen) < world - 1:
                conn, _ = srv.accept()
                with conn:
                    conn.settimeout(timeout)
                    hdr = b''
                    while len(hdr) < 4:                            # a 4-byte read may come back short
                        chunk = conn.recv(4 - len(hdr))
                        if not chunk:
                            break
                        hdr += chunk
                    r = int.from_bytes(hdr, 'little') if len(hdr) == 4 else -1
                    if not 0 < r < world:                          # not one of ours: no id for it
                        continue
                    conn.sendall(uid)
                    seen.add(r)
        finally:
            srv.close()
        return uid
    deadline = time.time() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as conn:
                conn.sendall(int(rank).to_bytes(4, 'little'))
                buf = b''
                while len(buf) < 128:
                    chunk = conn.recv(128 - len(buf))
                    if not chunk:
                        raise ConnectionError('rendezvous peer closed the connection')
                    buf += chunk
                return buf
        except (ConnectionRefusedError, ConnectionResetError, socket.timeout, ConnectionError):
            if time.time() > deadline:
                raise
            time.sleep(0.05)


def rccl_rendezvous(ctx, rank, world, addr=None, port=None):
    """Create the RCCL communicator of `ctx` (one context = one GPU = one rank) and return an `RcclComm`."""
    uid = ctx.comm_unique_id() if rank == 0 else None
    uid = exchange_unique_id(uid, rank, world, addr, port)
    ctx.comm_init(uid, rank, world)
    return RcclComm(ctx)


def allgather_segment_tables(rows, capacity=4096, device=None, group=None):
    """All ranks contribute their (k,4) int32 rows; every rank gets the concatenation (ordered by
    rank) back as one (K,4) int32 array.  One collective when every k <= capacity."""
    import torch
    import torch.distributed as dist
    failed = rows is None                    # this rank's local work failed: header count -1, no rows
    rows = np.zeros((0, 4), np.int32) if failed else np.ascontiguousarray(rows, dtype=np.int32).reshape(-1, 4)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    backend = dist.get_backend(group)
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')

    def gather(cap):
        buf = np.zeros((cap + 1, 4), dtype=np.int32)
        buf[0] = (-1 if failed else len(rows), cap, rank, 0)
        k = min(len(rows), cap)
        buf[1:1 + k] = rows[:k]
        send = torch.from_numpy(buf).to(device)
        recv = torch.empty((world * (cap + 1), 4), dtype=torch.int32, device=device)
        dist.all_gather_into_tensor(recv, send, group=group)
        return recv.cpu().numpy().reshape(world, cap + 1, 4)

    got = gather(int(capacity))
    _check_counts(got[:, 0, 0])
    need = int(got[:, 0, 0].max())
    if need > capacity:                      # rare: some rank had more rows than agreed -> one bigger gather
        got = gather(need)
    parts = [got[r, 1:1 + got[r, 0, 0]] for r in range(world)]
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, 4), np.int32)
