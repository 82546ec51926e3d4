"""The loss of the Open-Pose+ trainer with the reference's signature (tpm/lib/network/losses.py:65-106), so that
tpm/train_rtpose_light3d_kdh3d_mpaug.py:160-180 (CR) runs with the import swapped: per stage the mean squared error of the PAF
and heat maps and the foreground-weighted (0.1 background, 1.0 foreground) mean squared error of the depth maps, summed over
the stages; the log carries the six terms and the extrema the reference prints.  Element-wise torch glue around the outputs of
the HIP primitives (network/_autograd.py); popnet_amd.train.TrainEngine computes the same six terms inside pn_head_forward."""
from collections import OrderedDict


def build_names(num_stages=2):
    """losses.py / train_rtpose_light3d_kdh3d_mpaug.py: 'l1_paf', 'l1_heat', 'l1_z', 'l2_paf', ..."""
    names = []
    for j in range(1, num_stages + 1):
        names += ["l%d_paf" % j, "l%d_heat" % j, "l%d_z" % j]
    return names


def rtpose_light3d_loss_fgweight(saved_for_loss, heat_gt, vec_temp, posedepth_temp, fg_mask, num_stages, names):
    log = OrderedDict()
    weight = fg_mask * 0.9 + 0.1
    total = 0
    for j in range(num_stages):
        paf, heat, z = saved_for_loss[3 * j], saved_for_loss[3 * j + 1], saved_for_loss[3 * j + 2]
        terms = (((paf - vec_temp) ** 2).mean(), ((heat - heat_gt) ** 2).mean(), (((z - posedepth_temp) ** 2) * weight).mean())
        for k, t in enumerate(terms):
            total = total + t
            log[names[3 * j + k]] = t.item()
    heat, paf, z = saved_for_loss[-2].detach(), saved_for_loss[-3].detach(), saved_for_loss[-1].detach()
    log["max_ht"], log["min_ht"] = heat[:, 0:-1].max().item(), heat[:, 0:-1].min().item()
    log["max_paf"], log["min_paf"] = paf.max().item(), paf.min().item()
    log["max_z"], log["min_z"] = z.max().item(), z.min().item()
    return total, log
