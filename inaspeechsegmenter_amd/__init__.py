"""MI355X-native implementation of inaSpeechSegmenter's per-frame feature + CNN hot path.

Drop-in surface (inaSpeechSegmenter/__init__.py:26): `Segmenter`, `seg2csv`, `seg2textgrid`.
Everything numeric runs in hand-written HIP kernels (csrc/, built into libiss_hip.so) behind
the C ABI of include/iss.h; importing this package does not require a GPU, constructing a
`Segmenter` does.
"""
from .segmenter import Segmenter
from .export_funcs import seg2csv, seg2textgrid

__all__ = ['Segmenter', 'seg2csv', 'seg2textgrid']
__version__ = '0.1.0'
