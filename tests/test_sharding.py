"""CPU: the N > 1 path -- file dealing and the single all-gather of segment tables -- with two
gloo ranks (the GPU box runs the same code over backend "nccl" = RCCL)."""
import os
import socket

import numpy as np
import pytest

from inaspeechsegmenter_amd import sharding as sh


def test_shard_files_lpt_and_round_robin():
    assert sh.shard_files([5] * 8, 4) == [[0, 4], [1, 5], [2, 6], [3, 7]]          # equal sizes -> i mod R
    parts = sh.shard_files([10, 1, 1, 1, 9, 8, 2], 3)
    assert sorted(i for p in parts for i in p) == list(range(7))
    loads = [sum([10, 1, 1, 1, 9, 8, 2][i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= 2
    assert sh.shard_files([], 2) == [[], []]
    assert sh.shard_files([3], 4) == [[0], [], [], []]


def test_pack_unpack_roundtrip_times_are_reference_floats():
    lseg = [('noEnergy', 0, 1124), ('music', 1124, 1454), ('female', 1454, 3725)]
    rows = sh.pack_segments(7, lseg)
    assert rows.dtype == np.int32 and rows.shape == (3, 4)
    back = sh.unpack_segments(rows)[7]
    assert back == [(l, 0 + a * .02, 0 + b * .02) for l, a, b in lseg]
    assert repr(back[1][2]) == '29.080000000000002'                                  # media/musanmix-smn-gender.csv:3


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nrows, capacity, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(100 + rank)
        rows = np.column_stack([np.full(nrows[rank], rank), rng.integers(0, 7, nrows[rank]),
                                np.arange(nrows[rank]), np.arange(nrows[rank]) + 1]).astype(np.int32).reshape(-1, 4)
        got = sh.allgather_segment_tables(rows, capacity=capacity)
        q.put((rank, got.tobytes(), got.shape, rows.tobytes()))
    except Exception as e:                                   # surface the failure instead of a queue timeout
        q.put((rank, repr(e), None, None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('nrows,capacity', [((5, 9), 16), ((0, 3), 4), ((40, 2), 8)])
def test_allgather_two_gloo_ranks(nrows, capacity):
    """Every rank ends up with rank 0's rows followed by rank 1's; (40,2) with capacity 8 takes the
    overflow branch (second, larger gather); (0,3) covers a rank with no segments at all."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nrows, capacity, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert r[2] is not None, r[1]
    res.sort()
    own = [np.frombuffer(r[3], dtype=np.int32).reshape(-1, 4) for r in res]
    want = np.concatenate(own, axis=0)
    for rank, blob, shape, _ in res:
        got = np.frombuffer(blob, dtype=np.int32).reshape(shape)
        assert np.array_equal(got, want), rank
