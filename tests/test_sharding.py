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


def test_unique_id_exchange_over_tcp():
    """The out-of-band step of the RCCL rendezvous (sharding.exchange_unique_id): rank 0 serves the 128-byte id, the
    other ranks (started first here, so they must retry) fetch it."""
    import threading
    port = _free_port()
    uid = bytes(range(128))
    got = {}

    def run(rank):
        got[rank] = sh.exchange_unique_id(uid if rank == 0 else None, rank, 3, '127.0.0.1', port, timeout=30)

    th = [threading.Thread(target=run, args=(r,)) for r in (1, 2)]
    for t in th:
        t.start()
    import time
    time.sleep(0.3)
    run(0)
    for t in th:
        t.join()
    assert got == {0: uid, 1: uid, 2: uid}


def test_merge_gathered_overflow_path():
    rows = [np.arange(20, dtype=np.int32).reshape(5, 4), np.arange(100, 136, dtype=np.int32).reshape(9, 4)]

    def regather(cap):
        parts = np.zeros((2, cap, 4), np.int32)
        for r in range(2):
            parts[r, :len(rows[r])] = rows[r]
        return parts, np.array([5, 9])

    parts = np.stack([rows[0][:4], rows[1][:4]])     # capacity 4 < 5, 9: truncated first pass
    out = sh._merge_gathered(parts, np.array([5, 9]), 4, regather)
    assert np.array_equal(out, np.concatenate(rows))
    with pytest.raises(sh.RankFailure, match=r'\[1\]'):
        sh._merge_gathered(parts, np.array([5, -1]), 4, regather)


@pytest.mark.gpu
def test_rccl_allgather_single_rank():
    """iss_comm_* / iss_allgather_segments of the C-ABI on the one GPU of the test box: world size 1 exercises the
    librccl binding, communicator set-up, staging and the header-row protocol (the 8-GPU run is the driver's)."""
    from inaspeechsegmenter_amd import _native
    ctx = _native.Context(0)
    comm = sh.rccl_rendezvous(ctx, 0, 1)
    rows = np.array([[0, 3, 0, 10], [0, 5, 10, 25], [2, 0, 0, 7]], dtype=np.int32)
    assert np.array_equal(comm.allgather(rows, 8), rows)
    assert np.array_equal(comm.allgather(rows, 2), rows)          # overflow -> second, larger gather
    assert np.array_equal(comm.allgather(np.zeros((0, 4), np.int32), 4), np.zeros((0, 4), np.int32))
    assert comm.max_over_ranks(3.25) == 3.25
    comm.barrier()
    info = comm.info()                                            # what RCCL itself says about the communicator
    assert info['world'] == 1 and info['rank'] == 0 and info['version'] > 0 and 'rccl' in info['lib'].lower()
    with pytest.raises(sh.RankFailure):                           # a rank that reports failure: nobody keeps a table
        comm.allgather(None, 8)
    ctx.comm_destroy()
    ctx.close()


def test_unique_id_exchange_through_the_launcher_store(tmp_path):
    """Under `python -m torch.distributed.run` (how the driver starts the N-GPU bench) the communicator id travels through
    the launcher's own key-value store: two ranks, two rounds."""
    import subprocess
    import sys
    script = tmp_path / 'rdv.py'
    script.write_text(f'''
import os, sys
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
from inaspeechsegmenter_amd import sharding as sh
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
assert os.environ.get('TORCHELASTIC_USE_AGENT_STORE') == 'True'
a = sh.exchange_unique_id(bytes(range(128)) if rank == 0 else None, rank, world)
b = sh.exchange_unique_id(bytes(reversed(range(128))) if rank == 0 else None, rank, world)
assert a == bytes(range(128)) and b == bytes(reversed(range(128)))
print('rank', rank, 'ok')
''')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(_free_port()), str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'rank 0 ok' in r.stdout and 'rank 1 ok' in r.stdout, r.stdout + r.stderr


def _run_bench(argv, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env['PYTHONPATH'] = root + os.pathsep + env.get('PYTHONPATH', '')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + argv, capture_output=True, text=True, timeout=timeout, env=env,
                       cwd=root)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    return r, [json.loads(ln) for ln in lines]


def test_bench_gpus_2_spawns_its_own_ranks_on_a_fake_device():
    """`python bench.py --gpus 2` with no launcher in the environment (how a driver calls it at N = 1) must not exit: it starts
    its own two ranks under torch.distributed.run, and the N > 1 branch of the file -- communicator, barriers, max-over-ranks
    timing, one all-gather per step, line assembly on rank 0 only -- runs (here with a fake device and the gloo exchange)."""
    r, lines = _run_bench(['--gpus', '2', '--steps', '3', '--warmup', '1', '--minutes', '1', '--comm', 'gloo',
                           '--fake-device', 'tests.fake_bench_device:make'])
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'spawning' in r.stderr and 'torch.distributed.run' in r.stderr
    assert len(lines) == 1, r.stdout                                   # rank 0 only
    line = lines[0]
    assert line['n_gpus'] == 2 and line['steps'] == 3 and line['data'] == 'fake-device' and line['value'] is None
    assert line['config']['files_seen'] == [0, 1] and line['config']['rows_gathered'] == 3 + 4      # both ranks' tables on rank 0
    assert 'gloo' in line['config']['parallelism']
    assert line['ranks']['ms_per_step_min'] < line['ranks']['ms_per_step_max']                      # rank 1 sleeps twice as long
    assert 18.0 <= line['ranks']['ms_per_step_max'] and line['ms_per_step'] >= line['ranks']['ms_per_step_max'] - 1e-6


def test_bench_has_no_silent_comm_fallback():
    """north_star: no dual backend.  The default exchange is the C-ABI's RCCL all-gather; when its rendezvous cannot be set up
    the run fails and names the explicit alternatives, it does not quietly continue on torch.distributed."""
    import bench

    class NoRccl:
        def comm_unique_id(self):
            raise RuntimeError('librccl.so: cannot open shared object file')

    with pytest.raises(SystemExit) as ei:
        bench.make_comm(NoRccl(), 0, 2, None, 'rccl')
    assert 'no fallback' in str(ei.value) and '--comm torch' in str(ei.value) and '--comm gloo' in str(ei.value)
    assert bench.make_comm(NoRccl(), 0, 1, None, 'rccl') == (None, None)
    r, lines = _run_bench(['--gpus', '2', '--steps', '1', '--warmup', '0', '--minutes', '1', '--fake-device', 'tests.fake_bench_device:make'])
    assert r.returncode != 0 and not lines and 'no fallback' in r.stderr, r.stdout + r.stderr       # default --comm rccl, no device
