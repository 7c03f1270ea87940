"""A stand-in for `keras predict` that is EXACTLY reproducible across numpy versions / BLAS builds.

Used to pin the segmentation bookkeeping (`_get_patches`, `DnnSegmenter.__call__`,
`Segmenter.segment_feats`, segmenter.py:76-88,135-179,250-276) against the reference's own lines:
tests/golden/ref_segmenter_pin.py runs those lines (extracted with `ast`, under the interpreter that has
skimage) with this function in place of the un-vendored Keras models, and tests/test_oracle_golden.py
runs oracle/segment.py with the same function.  Every operation is either integer arithmetic or a
single correctly-rounded IEEE operation, so the probabilities are bit-identical everywhere.
"""
import numpy as np


def make_predict(nclass, salt, run=150, big=2000):
    """-> predict(batch (N,68,h,1) float32) -> (N,nclass) float32 rows.

    Row i prefers class ((i // run) + (content hash // big)) % nclass with probability 0.998: the row-index term makes the
    decoded labels change every few seconds of gathered slots (so a mis-ordered gather / scatter shows up as different
    segments), the content term (integer sum of the quantised middle rows of the patch) ties them to the patch values."""
    def predict(batch, batch_size=32, verbose=0):
        x = np.asarray(batch)
        n = x.shape[0]
        ok = np.isfinite(x)
        # quantise: patches are z-normalised, |x| rarely exceeds 8; NaN/inf -> 0 (those rows are overridden by 0.5)
        q = np.floor(np.where(ok, x, 0).astype(np.float64) * 64.0).astype(np.int64)
        h = q[:, 20:44, :, 0].sum(axis=(1, 2)) + salt * 1000                               # exact int64
        pref = ((np.arange(n, dtype=np.int64) // run) + (h // big)) % nclass
        out = np.full((n, nclass), 0.002 / (nclass - 1))
        out[np.arange(n), pref] = 0.998
        return out.astype(np.float32)
    return predict
