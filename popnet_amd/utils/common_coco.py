"""The two result classes ``paf_to_pose_cpp`` returns (tpm/lib/utils/common_coco.py:27-134): plain data holders with the reference's
attribute names, so that code written against the reference's ``Human`` / ``BodyPart`` objects (evaluate/coco_eval.py:270-290 reads
``human.body_parts[i].x / .y / .score`` and ``human.score``) runs unchanged.  Only what that caller reads is provided: no drawing, no
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
    """body_parts: {part_idx: BodyPart}; score: the C++ side's human score (pafprocess.cpp:216-225)."""
    __slots__ = ('body_parts', 'pairs', 'uidx_list', 'score')

    def __init__(self, pairs):
        if pairs:
            raise NotImplementedError("Human(pairs): only the empty form paf_to_pose_cpp builds is provided")
        self.pairs = []
        self.uidx_list = set()
        self.body_parts = {}
        self.score = 0.0

    def part_count(self):
        return len(self.body_parts)

    def get_max_score(self):
        return max(x.score for x in self.body_parts.values())

    def __repr__(self):
        return ' '.join(repr(x) for x in self.body_parts.values())
