"""TEST INFRASTRUCTURE -- a CPU restatement of the slice of `pyannote.core` that the reference's voice-femininity tail
calls (vbx_segmenter.py:28-69,129-145,186-197).  Only tests/ may import this file.

`pyannote.core` is a third-party dependency of the reference (setup.py:143, UNPINNED; absent from /root/reference and from
this image, no network), so its published algorithm is restated here from the package's documented behaviour
(pyannote/core/segment.py, timeline.py, annotation.py of the 4.x / 5.x line -- the classes have not changed semantics
between them) and parity is anchored on the reference's own call sites:

    Segment(start, end)                       vbx_segmenter.py:58,68,138,142       (NB `:50` reads `s.stop`, which Segment does not
                                                                                   have -- kept: it raises AttributeError here too)
    Annotation()[Segment, '_'] = label        :56-58, :65-68
    Annotation.itertracks(yield_label=True)   :36
    Annotation.get_timeline()                 :138
    Annotation.label_timeline(label)          :61
    Annotation.label_duration(label)          :163
    len(Annotation)                           :61
    Timeline([Segment]).crop(Timeline)        :138        (mode='intersection', the default)
    Timeline.duration()                       :140,142

PARITY UNPINNED against the package itself (it cannot be imported here); the rules that matter for numbers are spelled
out next to the code so that a reader with the package at hand can check them line by line:
  * a segment is EMPTY when end - start <= SEGMENT_PRECISION (1e-6 s); empty segments are dropped by Timeline() and ignored
    by Annotation.__setitem__;
  * Timeline is a SET of segments kept in (start, end) order; duplicates collapse;
  * crop(support: Timeline) first replaces the support by its own support() (overlapping / touching segments merged), then
    yields `segment & other` for every pair that `intersects` (strictly more than SEGMENT_PRECISION of overlap, or equal starts);
  * duration() is the duration of the timeline's SUPPORT (overlaps are not counted twice);
  * Annotation keeps {segment: {track: label}}: writing the same (segment, track) again REPLACES the label; len() counts segments.
"""
import bisect

SEGMENT_PRECISION = 1e-6


class Segment:
    __slots__ = ('start', 'end')

    def __init__(self, start=0.0, end=0.0):
        object.__setattr__(self, 'start', start)
        object.__setattr__(self, 'end', end)

    def __setattr__(self, k, v):
        raise AttributeError('Segment is frozen')

    def _key(self):
        return (self.start, self.end)

    def __eq__(self, o):
        return isinstance(o, Segment) and self._key() == o._key()

    def __lt__(self, o):
        return self._key() < o._key()

    def __le__(self, o):
        return self._key() <= o._key()

    def __hash__(self):
        return hash(self._key())

    def __iter__(self):
        yield self.start
        yield self.end

    def __bool__(self):
        return bool((self.end - self.start) > SEGMENT_PRECISION)

    @property
    def duration(self):
        return self.end - self.start if self else 0.

    def __and__(self, other):
        return Segment(max(self.start, other.start), min(self.end, other.end))

    def intersects(self, other):
        return ((self.start < other.start and other.start < self.end - SEGMENT_PRECISION) or
                (self.start > other.start and self.start < other.end - SEGMENT_PRECISION) or
                (self.start == other.start))

    def __or__(self, other):
        if not self:
            return other
        if not other:
            return self
        return Segment(min(self.start, other.start), max(self.end, other.end))

    def __xor__(self, other):
        if (not self) or (not other):
            raise ValueError('The gap between a segment and an empty segment is not defined.')
        return Segment(min(self.end, other.end), max(self.start, other.start))

    def __repr__(self):
        return f'<Segment({self.start:g}, {self.end:g})>'


class Timeline:
    def __init__(self, segments=None):
        self.segments_set_ = set(s for s in (segments or ()) if s)
        self.segments_list_ = sorted(self.segments_set_)

    def __len__(self):
        return len(self.segments_set_)

    def __bool__(self):
        return len(self.segments_set_) > 0

    def __iter__(self):
        return iter(self.segments_list_)

    def co_iter(self, other):
        for segment in self.segments_list_:
            # the other timeline's segments that sort at or before Segment(segment.end, segment.end)
            hi = bisect.bisect_right(other.segments_list_, Segment(segment.end, segment.end))
            for other_segment in other.segments_list_[:hi]:
                if segment.intersects(other_segment):
                    yield segment, other_segment

    def support_iter(self, collar=0.):
        if not self:
            return
        new_segment = self.segments_list_[0]
        for segment in self:
            possible_gap = segment ^ new_segment
            if not possible_gap or possible_gap.duration < collar:
                new_segment = new_segment | segment
            else:
                yield new_segment
                new_segment = segment
        yield new_segment

    def support(self, collar=0.):
        return Timeline(self.support_iter(collar))

    def crop_iter(self, support):
        if isinstance(support, Segment):
            support = Timeline([support] if support else [])
        support = support.support()
        for segment, other_segment in self.co_iter(support):
            mapped_to = segment & other_segment
            if not mapped_to:
                continue
            yield mapped_to

    def crop(self, support, mode='intersection'):
        assert mode == 'intersection'
        return Timeline(self.crop_iter(support))

    def duration(self):
        return sum(s.duration for s in self.support_iter())


class Annotation:
    def __init__(self):
        self._tracks = {}

    def __setitem__(self, key, label):
        if isinstance(key, Segment):
            key = (key, '_')
        segment, track = key
        if not segment:
            return
        self._tracks.setdefault(segment, {})[track] = label

    def __len__(self):
        return len(self._tracks)

    def itertracks(self, yield_label=False):
        for segment in sorted(self._tracks):
            for track, lbl in sorted(self._tracks[segment].items(), key=lambda tl: (str(tl[0]), str(tl[1]))):
                yield (segment, track, lbl) if yield_label else (segment, track)

    def get_timeline(self):
        return Timeline(self._tracks)

    def labels(self):
        return {lbl for tr in self._tracks.values() for lbl in tr.values()}

    def label_timeline(self, label, copy=True):
        return Timeline(s for s, tr in self._tracks.items() if any(lbl == label for lbl in tr.values()))

    def label_duration(self, label):
        return self.label_timeline(label).duration()


# ------------------------------------------------------------------------------ the reference's tail on these classes
# (vbx_segmenter.py:28-69,129-145, restated statement by statement; numpy only where the reference uses it)
import numpy as np     # noqa: E402


def is_mid_speech(start, stop, a_vad):                                      # :28-37
    m = (start + stop) / 2
    is_speech = [True if seg.start < m < seg.end else False for seg, _, _ in a_vad.itertracks(yield_label=True)]
    return np.any(is_speech)


def add_needed_vectors(xvectors, t_mid):                                    # :40-52
    min_pred = round(0.5 * len(t_mid))
    if len(xvectors) < min_pred:
        t_mid = np.asarray(t_mid, dtype=object)
        t_mid = t_mid[t_mid[:, 0].astype(np.float64).argsort()][::-1]
        diff = min_pred - len(xvectors)
        for _, k, s, x in t_mid[len(xvectors):len(xvectors) + diff]:
            xvectors.append((k, (s.start, s.stop), x))                      # AttributeError in the reference as well
    return xvectors


def get_femininity_score(g_preds):                                          # :55-61
    a_temp = Annotation()
    for start, stop, p in g_preds:
        a_temp[Segment(start, stop), '_'] = (p >= 0.5)
    return len(a_temp.label_timeline(True)) / len(a_temp)


def get_annot_VAD(vad_tuples):                                              # :64-69
    annot_vad = Annotation()
    for lab, start, end in vad_tuples:
        if lab == "speech":
            annot_vad[Segment(start, end), '_'] = lab
    return annot_vad


def apply_vad(xvectors, a_vad, vad_thresh, fixed_stop_attribute=False):     # :129-145
    """fixed_stop_attribute: read `s.end` where the reference reads the non-existent `s.stop` (the evident intent)."""
    midpoint_seg = []
    n_xvectors = []
    for key, (start, stop), x in xvectors:
        if is_mid_speech(start, stop, a_vad):
            seg_total_duration = stop - start
            seg_cropped = Timeline([Segment(start, stop)]).crop(a_vad.get_timeline())
            if seg_cropped.duration() / seg_total_duration >= vad_thresh:
                n_xvectors.append((key, (start, stop), x))
            midpoint_seg.append(((seg_cropped.duration() / seg_total_duration), key, Segment(start, stop), x))
    if not fixed_stop_attribute:
        return add_needed_vectors(n_xvectors, midpoint_seg)
    min_pred = round(0.5 * len(midpoint_seg))
    if len(n_xvectors) < min_pred:
        t_mid = np.asarray(midpoint_seg, dtype=object)
        t_mid = t_mid[t_mid[:, 0].astype(np.float64).argsort()][::-1]
        diff = min_pred - len(n_xvectors)
        for _, k, s, x in t_mid[len(n_xvectors):len(n_xvectors) + diff]:
            n_xvectors.append((k, (s.start, s.end), x))
    return n_xvectors
