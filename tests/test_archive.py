"""CPU: the long-archive driver (file sharding + per-file outputs + gathered table) with a stand-in
`segment_file` and two gloo ranks; the CLI's argument surface against the reference's flags."""
import os
import socket
import sys

import numpy as np
import pytest

from inaspeechsegmenter_amd import archive


def _fake_segment(path):
    n = int(os.path.basename(path).split('_')[1].split('.')[0])
    if n == 3:
        raise IOError('decode failed')
    return [('noEnergy', 0.0, n * .02), ('female' if n % 2 else 'music', n * .02, (2 * n + 5) * .02)]


def _files(tmp_path, k=7):
    lin = []
    for i in range(k):
        p = tmp_path / f'f_{i}.wav'
        p.write_bytes(b'x' * (100 * (i + 1)))
        lin.append(str(p))
    return lin, [str(tmp_path / 'out' / f'f_{i}.csv') for i in range(k)]


def test_single_process(tmp_path):
    lin, lout = _files(tmp_path)
    table, lmsg = archive.segment_archive(_fake_segment, lin, lout)
    assert sorted(table) == [0, 1, 2, 4, 5, 6]                       # file 3 failed
    assert [m[1] for m in lmsg] == [0, 0, 0, 2, 0, 0, 0] and lmsg[3][2].startswith('error: ')
    assert table[5] == _fake_segment(lin[5])
    assert open(lout[5]).read().splitlines()[0] == 'labels\tstart\tstop'
    table2, lmsg2 = archive.segment_archive(_fake_segment, lin, lout, skipifexist=True)
    assert [m[1] for m in lmsg2] == [1, 1, 1, 2, 1, 1, 1] and sorted(table2) == []
    with pytest.raises(NotImplementedError):
        archive.segment_archive(_fake_segment, lin, lout, output_format='json')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, lin, lout, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=2)
    try:
        table, lmsg = archive.segment_archive(_fake_segment, lin, lout, capacity=4)   # tiny capacity: overflow path too
        q.put((rank, table, lmsg))
    except Exception as e:
        q.put((rank, repr(e), None))
    finally:
        dist.destroy_process_group()


def test_two_gloo_ranks(tmp_path):
    import torch.multiprocessing as mp
    lin, lout = _files(tmp_path)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, lin, lout, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    want, _ = archive.segment_archive(_fake_segment, lin, None)
    for rank, table, lmsg in res:
        assert lmsg is not None, table
        assert table == want                                         # every rank holds the complete table
    done = sorted(m[0] for r in res for m in r[2] if m[1] == 0)
    assert done == sorted(lout[i] for i in want)                     # each file processed by exactly one rank
    assert all(os.path.exists(f) for f in done)


def test_cli_flags_match_reference():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'scripts'))
    import ina_speech_segmenter_amd as cli
    a = cli.build_parser().parse_args(['-i', 'a.wav', 'b*.wav', '-o', '/tmp', '-d', 'sm', '-g', 'false', '-b', 'None',
                                       '-e', 'textgrid', '-r', '0.05', '-s', '1024'])
    assert a.input == ['a.wav', 'b*.wav'] and a.vad_engine == 'sm' and a.detect_gender is False
    assert a.ffmpeg_binary == 'None' and a.export_format == 'textgrid' and a.energy_ratio == 0.05 and a.batch_size == 1024
    d = cli.build_parser().parse_args(['-i', 'x', '-o', 'y'])
    assert (d.vad_engine, d.detect_gender, d.ffmpeg_binary, d.export_format, d.energy_ratio, d.batch_size) == \
        ('smn', True, 'ffmpeg', 'csv', 0.03, 32)


def test_joblist_matches_reference_server_semantics():
    """run_test.py:166-172 test_pyroserver on the reference's own fixture (media/pyroserver_test.csv)."""
    from conftest import GOLDEN
    gs = archive.JobList(os.path.join(GOLDEN, 'pyroserver_test.csv'))
    assert gs.has_more_jobs()
    lsrc, ldst = gs.get_njobs('')
    assert len(lsrc) == 7 and len(ldst) == 7
    assert sorted(lsrc) == ['/my_/source_4', 'my_source_1', 'my_source_2', 'my_source_3', 'my_source_5', 'my_source_6', 'my_source_7']
    assert sorted(ldst) == ['my_dest_1', 'my_dest_2', 'my_dest_3', 'my_dest_4', 'my_dest_5', 'my_dest_6', 'my_dest_7@@@!!']
    assert not gs.has_more_jobs() and gs.get_njobs('') == ([], [])
    a = archive.JobList(os.path.join(GOLDEN, 'pyroserver_test.csv'), seed=1)
    b = archive.JobList(os.path.join(GOLDEN, 'pyroserver_test.csv'), seed=1)
    assert a.lsource == b.lsource and a.get_njobs('', 3)[0] == b.lsource[:3] and a.has_more_jobs()
    pairs = dict(zip(*archive.JobList(os.path.join(GOLDEN, 'pyroserver_test.csv'), shuffle=False).get_njobs('')))
    assert pairs['my_source_1'] == 'my_dest_1' and pairs['/my_/source_4'] == 'my_dest_4'


def _failing_worker(rank, port, lin, lout, q, mode):
    """Rank 1 fails before the all-gather: 'raise' = its local work throws (it still takes part, flagged);
    'die' = the process is gone (the survivors hit the process group's timeout)."""
    from datetime import timedelta
    import torch.distributed as dist
    from inaspeechsegmenter_amd import _native, sharding
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=2, timeout=timedelta(seconds=8))

    def seg(path):
        if rank == 1:
            if mode == 'die':
                os._exit(7)
            raise _native.NativeError('device fault on rank 1 (simulated)')     # not a per-file error: propagates
        return _fake_segment(path)
    try:
        archive.segment_archive(seg, lin, lout)
        q.put((rank, 'no error'))
    except sharding.RankFailure as e:
        q.put((rank, 'RankFailure: %s' % e))
    except _native.NativeError as e:
        q.put((rank, 'NativeError: %s' % e))
    except Exception as e:                                     # gloo: connection closed by peer / timed out
        q.put((rank, 'comm error: %s' % type(e).__name__))
    q.close()
    q.join_thread()                                            # flush the result before leaving without teardown
    os._exit(0)                                                # no orderly process-group teardown with a dead peer


@pytest.mark.parametrize('mode', ['raise', 'die'])
def test_rank_failure_is_loud_on_every_rank(tmp_path, mode):
    """SURVEY 8(e) hardening: a rank that fails (or disappears) before the all-gather must not leave the others waiting
    (the reference's Pyro workers fail independently, ina_speech_segmenter_pyro_client.py:64-74; a collective cannot)."""
    import time
    import torch.multiprocessing as mp
    lin, lout = _files(tmp_path)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, port, lin, lout, q, mode)) for r in range(2)]
    t0 = time.time()
    for p in procs:
        p.start()
    want = 2 if mode == 'raise' else 1
    res = dict(q.get(timeout=90) for _ in range(want))
    for p in procs:
        p.join(timeout=30)
    assert time.time() - t0 < 90
    if mode == 'raise':
        assert res[1].startswith('NativeError') and 'simulated' in res[1]      # the failing rank reports ITS error
        assert res[0].startswith('RankFailure') and '[1]' in res[0]            # the other one learns who failed
    else:
        assert procs[1].exitcode == 7
        assert res[0].startswith('comm error') or res[0].startswith('RankFailure'), res
