import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from inaspeechsegmenter_amd import Segmenter, segmenter as S, export_funcs
n = 5 * 60 * 16000
pcm = bench.synth_recording(0, n, torch.device('cuda', 0)).cpu().numpy()
seg = Segmenter(vad_engine='smn', detect_gender=True, ffmpeg=None, models='synthetic')
import cProfile, pstats
def one():
    mspec, loge, difflen = S._sig2feats(seg.ctx, pcm, None)
    lseg = seg.segment_feats(mspec, loge, difflen, 0)
    export_funcs.seg2csv(lseg, '/dev/shm/x.csv')
for _ in range(3): one()
t0 = time.perf_counter()
for _ in range(10): one()
print('ms per file', (time.perf_counter() - t0) * 100)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): one()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
