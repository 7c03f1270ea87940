"""A stand-in for the Segmenter behind `bench.py --fake-device tests.fake_bench_device:make` (tests/test_sharding.py): the N-rank
branch of bench.py -- self-spawn, ranks, communicator, barriers, max-over-ranks timing, line assembly -- runs without a GPU."""
import time


class _Fake:
    def __init__(self, rank):
        self.rank = rank

    def segment_device_pcm(self, ptr, n, dense=True):
        time.sleep(0.01 * (1 + self.rank))              # uneven ranks: min / max of the line must differ
        slots = n // 320
        k = 3 + self.rank
        edges = [slots * i // k for i in range(k + 1)]
        labs = ('noEnergy', 'music', 'male', 'noise', 'female')
        return [(labs[(i + self.rank) % len(labs)], edges[i], edges[i + 1]) for i in range(k)]


def make(rank):
    return _Fake(rank)
