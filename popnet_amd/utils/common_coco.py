"""The two result classes ``paf_to_pose_cpp`` returns (tpm/lib/utils/common_coco.py:27-134): plain data holders with the reference's
attribute names, so that code written against the reference's ``Human`` / ``BodyPart`` objects (evaluate/coco_eval.py:270-290 reads
``human.body_parts[i].x / .y / .score`` and ``human.score``) runs unchanged.  The pair-based assembly methods (add_pair / is_connected / merge) are data-only; no drawing, no
face / upper-body boxes (visualisation is out of scope)."""


class BodyPart:
    """part_idx: COCO part index; x, y: position as a FRACTION of the up-sampled map (paf_to_pose.py:405-406); score: the peak's score."""
    __slots__ = ('uidx', 'part_idx', 'x', 'y', 'score')

    def __init__(self, uidx, part_idx, x, y, score):
        self.uidx = uidx
        self.part_idx = part_idx
        self.x, self.y = x, y
        self.score = score

    def get_part_name(self):
        return self.part_idx

    def __repr__(self):
        return 'BodyPart:%d-(%.2f, %.2f) score=%.2f' % (self.part_idx, self.x, self.y, self.score)


class Human:
    """body_parts: {part_idx: BodyPart}; score: the C++ side's human score (pafprocess.cpp:216-225).

    ``pairs`` (tpm/lib/utils/common_coco.py:27-60): limb connections, each an object with ``part_idx1 / idx1 / coord1`` and
    ``part_idx2 / idx2 / coord2`` (part type, peak index within that type, (x, y)) and a ``score``.  ``paf_to_pose_cpp`` always passes
    ``[]``; the pair form is the reference's pure-Python assembly interface, kept as data-only methods: every pair contributes both of
    its end points (a later pair overwrites an earlier one's part of the same type), ``uidx_list`` holds the ``"<type>-<index>"`` keys of
    every end point seen, two humans are connected when they share one, and ``merge`` replays the other human's pairs."""
    __slots__ = ('body_parts', 'pairs', 'uidx_list', 'score')

    def __init__(self, pairs):
        self.pairs = []
        self.uidx_list = set()
        self.body_parts = {}
        for pr in pairs or ():
            self.add_pair(pr)
        self.score = 0.0

    @staticmethod
    def _get_uidx(part_idx, idx):
        return '%d-%d' % (part_idx, idx)

    def add_pair(self, pair):
        self.pairs.append(pair)
        for ptype, pidx, xy in ((pair.part_idx1, pair.idx1, pair.coord1), (pair.part_idx2, pair.idx2, pair.coord2)):
            key = self._get_uidx(ptype, pidx)
            self.body_parts[ptype] = BodyPart(key, ptype, xy[0], xy[1], pair.score)
            self.uidx_list.add(key)

    def is_connected(self, other):
        return not self.uidx_list.isdisjoint(other.uidx_list)

    def merge(self, other):
        for pr in other.pairs:
            self.add_pair(pr)

    def part_count(self):
        return len(self.body_parts)

    def get_max_score(self):
        return max(x.score for x in self.body_parts.values())

    def __repr__(self):
        return ' '.join(repr(x) for x in self.body_parts.values())
