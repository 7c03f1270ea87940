"""Segmentation writers, byte-compatible with the reference's exporters.

seg2csv      <-> export_funcs.py:29-31  pandas `DataFrame.to_csv(sep='\\t', index=False)`:
                 header `labels\\tstart\\tstop`, one row per segment, floats written with
                 Python's shortest round-trip repr (what pandas emits for float64 columns).
seg2textgrid <-> export_funcs.py:33-39  pytextgrid `PraatTextGrid.save`: long ("ooTextFile")
                 format, one IntervalTier named inaSpeechSegmenter, `%f` times.
Both are written natively (no pandas / pytextgrid import): per-file export cost matters when
a node segments thousands of short files per second.  tests/test_host.py::test_exporters_byte_identical_to_reference_goldens checks the bytes
against the reference's golden files (media/musanmix-smn-gender.{csv,TextGrid}) and against
pandas.
"""
import sys


def _open(fout):
    if fout is None:
        return sys.stdout, False
    if hasattr(fout, 'write'):
        return fout, False
    return open(fout, 'w', newline=''), True


def seg2csv(lseg, fout=None):
    rows = ['labels\tstart\tstop\n']
    for lab, start, stop in lseg:
        rows.append('%s\t%s\t%s\n' % (lab, repr(float(start)), repr(float(stop))))
    f, close = _open(fout)
    try:
        f.write(''.join(rows))
    finally:
        if close:
            f.close()


def seg2textgrid(lseg, fout=None):
    xmin, xmax = lseg[0][1], lseg[-1][2]
    out = ['File type = "ooTextFile"\n', 'Object class = "TextGrid"\n', '\n',
           'xmin = %f\n' % xmin, 'xmax = %f\n' % xmax, 'tiers? <exists> \n', 'size = 1\n', 'item []:\n',
           '\titem [1]:\n', '\t\tclass = "IntervalTier"\n', '\t\tname = "inaSpeechSegmenter"\n',
           '\t\txmin = %f\n' % xmin, '\t\txmax = %f\n' % xmax, '\t\tintervals: size = %d\n' % len(lseg)]
    for i, (label, start, stop) in enumerate(lseg, 1):
        out.append('\t\tintervals[%d]:\n' % i)
        out.append('\t\t\t xmin = %f\n' % start)
        out.append('\t\t\t xmax = %f\n' % stop)
        out.append('\t\t\t text = "%s"\n' % label)
    f, close = _open(fout)
    try:
        f.write(''.join(out))
    finally:
        if close:
            f.close()
