"""popnet_amd.utils.common_coco.Human: the pair-based assembly surface of tpm/lib/utils/common_coco.py:27-60 (CPU; data only)."""
import os
import sys
import types

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from popnet_amd.utils.common_coco import BodyPart, Human  # noqa: E402


class Pair:
    def __init__(self, p1, i1, c1, p2, i2, c2, score):
        self.part_idx1, self.idx1, self.coord1 = p1, i1, c1
        self.part_idx2, self.idx2, self.coord2 = p2, i2, c2
        self.score = score


PAIRS = [Pair(1, 0, (0.5, 0.25), 2, 3, (0.4, 0.3), 0.9), Pair(2, 3, (0.4, 0.3), 3, 1, (0.35, 0.45), 0.7),
         Pair(1, 1, (0.8, 0.2), 5, 0, (0.9, 0.3), 0.6), Pair(5, 0, (0.9, 0.3), 6, 2, (0.95, 0.5), 0.8), Pair(3, 1, (0.35, 0.45), 4, 4, (0.3, 0.6), 0.5)]


def _state(h):
    return (sorted(h.uidx_list), {k: (v.uidx, v.part_idx, v.x, v.y, v.score) for k, v in h.body_parts.items()}, len(h.pairs), h.score)


def test_pair_constructor_and_methods():
    a = Human(PAIRS[:2])
    assert a.part_count() == 3 and sorted(a.uidx_list) == ["1-0", "2-3", "3-1"] and a.score == 0.0
    assert a.body_parts[2].score == 0.7 and a.body_parts[2].uidx == "2-3"          # the later pair's end point replaces the earlier one's
    assert a.body_parts[1].x == 0.5 and a.body_parts[3].y == 0.45 and a.get_max_score() == 0.9
    b = Human(PAIRS[2:4])
    assert not a.is_connected(b) and not b.is_connected(a)
    c = Human([PAIRS[4]])
    assert a.is_connected(c) and c.is_connected(a) and not b.is_connected(c)
    a.merge(c)
    assert a.part_count() == 4 and len(a.pairs) == 3 and "4-4" in a.uidx_list and a.body_parts[3].score == 0.5
    e = Human([])
    assert e.part_count() == 0 and e.pairs == [] and e.uidx_list == set()
    e.body_parts[0] = BodyPart("0-0", 0, 0.1, 0.2, 0.3)                           # how paf_to_pose_cpp fills it
    assert e.get_max_score() == 0.3


@pytest.mark.skipif(not os.path.isdir("/root/reference/third_party_methods"), reason="the reference tree is not on this box")
def test_same_states_as_the_reference_class():
    """Every constructor / add_pair / merge / is_connected outcome against the reference's own class (imported with a stub cv2: the module
    only draws with it)."""
    saved = sys.modules.get("cv2")
    sys.modules["cv2"] = types.ModuleType("cv2")
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("_ref_common_coco", "/root/reference/third_party_methods/lib/utils/common_coco.py")
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
    finally:
        if saved is None:
            del sys.modules["cv2"]
        else:
            sys.modules["cv2"] = saved
    for split in range(len(PAIRS) + 1):
        mine, theirs = Human(PAIRS[:split]), ref.Human(PAIRS[:split])
        assert _state(mine) == _state(theirs)
        m2, t2 = Human(PAIRS[split:]), ref.Human(PAIRS[split:])
        assert mine.is_connected(m2) == theirs.is_connected(t2)
        mine.merge(m2)
        theirs.merge(t2)
        assert _state(mine) == _state(theirs) and mine.part_count() == theirs.part_count()
        if mine.body_parts:
            assert mine.get_max_score() == theirs.get_max_score()
