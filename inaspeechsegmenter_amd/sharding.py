"""File-parallel sharding of a long archive over the GPUs of one node + the single exchange step.

The reference has no multi-GPU code; its only scale-out facility is a Pyro4 pull queue that
hands (src, dst) pairs to independent workers (scripts/ina_speech_segmenter_pyro_server.py:34-68)
because files are independent units (segmenter.py:314-327 loops over them with no shared state).
Here: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests), files are dealt to ranks up front, every rank segments its own files,
and ONE all-gather of a fixed-capacity int32 table collects all segment boundaries on every rank.

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


def allgather_segment_tables(rows, capacity=4096, device=None, group=None):
    """All ranks contribute their (k,4) int32 rows; every rank gets the concatenation (ordered by
    rank) back as one (K,4) int32 array.  One collective when every k <= capacity."""
    import torch
    import torch.distributed as dist
    rows = np.ascontiguousarray(rows, dtype=np.int32).reshape(-1, 4)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    backend = dist.get_backend(group)
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')

    def gather(cap):
        buf = np.zeros((cap + 1, 4), dtype=np.int32)
        buf[0] = (len(rows), cap, rank, 0)
        k = min(len(rows), cap)
        buf[1:1 + k] = rows[:k]
        send = torch.from_numpy(buf).to(device)
        recv = torch.empty((world * (cap + 1), 4), dtype=torch.int32, device=device)
        dist.all_gather_into_tensor(recv, send, group=group)
        return recv.cpu().numpy().reshape(world, cap + 1, 4)

    got = gather(int(capacity))
    need = int(got[:, 0, 0].max())
    if need > capacity:                      # rare: some rank had more rows than agreed -> one bigger gather
        got = gather(need)
    parts = [got[r, 1:1 + got[r, 0, 0]] for r in range(world)]
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, 4), np.int32)
