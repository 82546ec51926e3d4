"""Oracle (oracle/targets.py) == the reference's training-target code, on the vectors tests/golden/make_golden.py produced
by running get_ground_truth and KDH3D_Keypoints.__getitem__ themselves."""
import os

import numpy as np

from oracle import targets as ot

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "targets.npz"))


def test_ground_truth_restatement_equals_the_reference():
    for ci in range(int(G["n_gt"])):
        h, p, z, f = ot.ground_truth(G["gt%d_kp2d" % ci], G["gt%d_kp3d" % ci], G["gt%d_depth" % ci])
        for got, key in ((h, "heat"), (p, "paf"), (z, "z"), (f, "fg")):
            assert np.array_equal(got, G["gt%d_%s" % (ci, key)]), (ci, key, np.abs(got - G["gt%d_%s" % (ci, key)]).max())
    # the cases are not trivial: overlapping persons saturate the heat map, limbs overlap, joints fall outside
    assert (G["gt3_heat"][:, :, :15] == 1.0).sum() > 0 and np.abs(G["gt3_paf"]).max() > 0.9 and G["gt2_fg"].sum() > 0


def test_dataset_item_restatement_equals_the_reference():
    for i in range(int(G["n_items"])):
        img, (h, p, z, f) = ot.mpaug_item(G["it%d_fg_depth" % i], G["it%d_fg_mask" % i], G["it%d_bg" % i], G["it%d_kp2d_org" % i], G["it%d_kp3d" % i])
        assert np.array_equal(img, G["it%d_image" % i])
        for got, key in ((h, "heat"), (p, "paf"), (z, "z"), (f, "fg")):
            assert np.array_equal(got, G["it%d_%s" % (i, key)]), (i, key)


def test_mpaug_sampler_follows_the_reference_control_flow():
    """popnet_amd.targets.MPAugSampler (host logic, no GPU): sources come from ONE aug_mods entry, in its order, at least one
    source always, frame = index % len(set), background = index % n_backgrounds, and the draws are reproducible under
    random.seed like the reference's (datasets_kdh3d_rtpose_mpaug.py:231-262)."""
    import random
    from popnet_amd.targets import AUG_MODS, MPAugSampler
    sizes = [11, 7, 5, 13, 3]
    s = MPAugSampler(sizes, n_backgrounds=4)
    random.seed(5)
    picks = [s.sources(i) for i in range(400)]
    random.seed(5)
    assert picks == [s.sources(i) for i in range(400)]
    counts = {}
    for i, (src, bg) in enumerate(picks):
        assert bg == i % 4 and 1 <= len(src) <= 2
        sets = [a for a, _ in src]
        assert any(all(x in m for x in sets) and sets == [x for x in m if x in sets] for m in AUG_MODS) or len(src) == 1
        assert all(f == i % sizes[a] for a, f in src)
        counts[len(src)] = counts.get(len(src), 0) + 1
    assert counts[2] > counts[1] > 0          # 0.8 * 0.8 of the two-set draws keep both
    src, n, bg = s.batch(range(6))
    assert src.shape == (6, 2, 2) and n.dtype == np.int32 and (src[np.arange(6), n - 1, 0] >= 0).all() and bg.tolist() == [0, 1, 2, 3, 0, 1]
