"""CPU oracle for the inaSpeechSegmenter per-frame feature + CNN hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``inaspeechsegmenter_amd`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the timed CPU baseline.

Every function here restates, in plain numpy (torch-CPU for the two network
forwards), the arithmetic of one reference function and cites it as
``file:line`` into ``/root/reference``.

Pinning status (see DESIGN.md "Oracle"):
  * sidekit front end, Viterbi, energy activity, patch builder bookkeeping,
    VBx fbank/CMN and the ResNet-101 topology are PINNED: bit-checked in this
    container against the reference modules imported with importlib
    (``tests/golden/make_golden.py``) and against the reference's own golden
    files (``media/*.csv``, ``media/test.h5``), committed under tests/golden/.
  * the Keras small-CNN forward (smn / sm / gender nets) is **parity unpinned**:
    the topology and weights live in un-vendored release assets
    (remote_utils.py:4-15) and no TensorFlow runtime exists here; the oracle
    restates documented Keras layer semantics and is checked against torch-CPU
    functional ops only.
"""
