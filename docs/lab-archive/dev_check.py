"""Development diagnostic (not a test): runs every HIP stage against the oracle and prints error
statistics + rough timings.  Usage on a GPU box:  python scripts/dev_check.py [section ...]"""
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import popnet_amd  # noqa: E402
from popnet_amd import synth, _lib  # noqa: E402
from popnet_amd.config import default_cfg  # noqa: E402
from popnet_amd.network.rtpose_light3d import rtpose_light3d  # noqa: E402
from popnet_amd.network.yolo_posenet import YoloPoseNet  # noqa: E402
from oracle import nets as onets, parse_paf as oparse, parse_yolo as oyolo, preproc as opre  # noqa: E402

dev = torch.device("cuda", 0)


def stats(name, a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b)
    print("  %-10s max|d|=%.3e mean|d|=%.3e max|ref|=%.3e rel=%.3e nan=%d" % (
        name, d.max(), d.mean(), np.abs(b).max(), d.max() / (np.abs(b).max() + 1e-30), int(np.isnan(a).sum())))


def sec_rtpose():
    for prec in ("fp32", "bf16"):
        print("[rtpose %s]" % prec)
        m = rtpose_light3d(15, 14, 2, input_dim=1).eval()
        synth.load_synth_weights(m, seed=0)
        m.precision = prec
        x = torch.from_numpy(np.random.default_rng(0).normal(0, 1, (3, 1, 224, 224)).astype(np.float32))
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        (rp, rh, rz), inter = onets.rtpose_light3d_forward(x, sd, return_intermediate=True)
        (p, h, z), saved = m(x.to(dev))
        torch.cuda.synchronize()
        stats("feat", m.stem_features(3).cpu().numpy(), inter['feat'].numpy())
        stats("paf1", saved[0].cpu().numpy(), inter['paf1'].numpy())
        stats("heat1", saved[1].cpu().numpy(), inter['heat1'].numpy())
        stats("z1", saved[2].cpu().numpy(), inter['z1'].numpy())
        stats("paf", p.cpu().numpy(), rp.numpy())
        stats("heat", h.cpu().numpy(), rh.numpy())
        stats("z", z.cpu().numpy(), rz.numpy())
        print("  flops/frame %.4f G" % (m.flops_per_frame() / 1e9))
        for B in (32,):
            xb = torch.randn(B, 1, 224, 224, device=dev)
            for _ in range(3):
                m(xb)
            torch.cuda.synchronize()
            t0 = time.time()
            n = 10
            for _ in range(n):
                m(xb)
            torch.cuda.synchronize()
            dt = (time.time() - t0) / n
            print("  B=%d forward %.3f ms  -> %.1f frames/s, %.1f TFLOP/s" % (
                B, dt * 1e3, B / dt, B * m.flops_per_frame() / dt / 1e12))


def sec_yolo():
    for prec in ("fp32", "bf16"):
        print("[yolo %s]" % prec)
        m = YoloPoseNet(15, input_dim=1).eval()
        synth.load_synth_weights(m, seed=1)
        m.precision = prec
        x = torch.from_numpy(np.random.default_rng(1).normal(0, 1, (2, 1, 224, 224)).astype(np.float32))
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        ref, inter = onets.yolo_posenet_forward(x, sd, return_intermediate=True)
        out = m(x.to(dev))
        torch.cuda.synchronize()
        stats("feat", m.backbone_features(2).cpu().numpy(), inter['feat'].numpy())
        stats("out", out.cpu().numpy(), ref.numpy())
        xb = torch.randn(32, 1, 224, 224, device=dev)
        for _ in range(3):
            m(xb)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(10):
            m(xb)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 10
        print("  B=32 forward %.3f ms -> %.1f frames/s %.1f TFLOP/s" % (dt * 1e3, 32 / dt, 32 * m.flops_per_frame() / dt / 1e12))


def sec_parse():
    from popnet_amd.utils.paf_to_pose import parse_paf_batch, make_parse_cfg, frame_joint_list, frame_assoc
    persons = [0, 1, 2, 3, 4, 6, 8, 3]
    heat, paf, z = synth.planted_batch(11, persons)
    cfg = make_parse_cfg(default_cfg())
    frames = parse_paf_batch(torch.from_numpy(heat).to(dev), torch.from_numpy(paf).to(dev), torch.from_numpy(z).to(dev), cfg)
    for b, P in enumerate(persons):
        rec = oparse.frame_to_records(heat[b].transpose(1, 2, 0).copy(), paf[b].transpose(1, 2, 0).copy(), z[b].transpose(1, 2, 0).copy())
        fr = frames[b]
        jl, assoc = frame_joint_list(fr), frame_assoc(fr)
        ok_jl = jl.shape == rec['joint_list'].shape and (jl.size == 0 or np.array_equal(jl, rec['joint_list']))
        ok_as = assoc.shape == rec['assoc'].shape and (assoc.size == 0 or np.array_equal(assoc[:, :15], rec['assoc'][:, :15]))
        msg = "frame %d P=%d peaks %d/%d persons %d/%d status=%d joint_list_exact=%s assoc_ids_exact=%s" % (
            b, P, int(fr['n_peaks']), len(rec['joint_list']), int(fr['n_persons']), len(rec['assoc']), int(fr['status']), ok_jl, ok_as)
        if ok_as and assoc.size:
            n = len(rec['assoc'])
            msg += " score_d=%.2e 2d_d=%.2e 3d_d=%.2e conf_d=%.2e" % (
                np.abs(assoc[:, 15:] - rec['assoc'][:, 15:]).max(),
                np.abs(fr['joints_2d'][:n] - np.array(rec['humans_2d'])).max(),
                np.abs(fr['joints_3d'][:n] - np.array(rec['humans_3d'])).max(),
                np.abs(fr['part_conf'][:n] - np.array(rec['conf'])).max())
        elif not ok_jl and jl.shape == rec['joint_list'].shape:
            bad = np.argwhere(jl != rec['joint_list'])
            msg += " first diffs: %s" % [(tuple(i), jl[tuple(i)], rec['joint_list'][tuple(i)]) for i in bad[:4]]
        print(" ", msg)
    B = 32
    heat, paf, z = synth.planted_batch(12, [3] * B)
    th, tp, tz = (torch.from_numpy(a).to(dev) for a in (heat, paf, z))
    for _ in range(3):
        parse_paf_batch(th, tp, tz, cfg)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        parse_paf_batch(th, tp, tz, cfg)
    torch.cuda.synchronize()
    print("  parse B=32 P=3 (incl. D2H of records): %.3f ms" % ((time.time() - t0) / 10 * 1e3))


def sec_preproc():
    import ctypes as C
    d = synth.synth_depth(2, 640, 480)
    ref = opre.preprocess_batch(d)
    td = torch.from_numpy(d).to(dev)
    out = torch.empty((2, 1, 224, 224), device=dev)
    ctx = _lib.Context.for_device(0)
    ctx.check(_lib.lib().pn_preprocess(ctx.handle, C.c_void_p(td.data_ptr()), _lib.PN_DEPTH_F16, 2, 640, 480,
                                       C.c_void_p(out.data_ptr()), 224, 6.0, 3.0, 2.0, _lib.current_stream_ptr(dev)), "pre")
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    stats("preproc", o, ref)
    print("  bit-exact:", np.array_equal(o, ref))


def sec_yoloparse():
    from popnet_amd.utils.prior_pose_align import parse_prior_pose
    rng = np.random.default_rng(5)
    pm = rng.uniform(-1, 1, (3, 100, 14, 14)).astype(np.float32)
    pm[:, 4] = rng.uniform(0, 0.6, (3, 14, 14)); pm[:, 54] = rng.uniform(0, 0.55, (3, 14, 14))
    pm[:, 2:4] = rng.uniform(0.5, 2, (3, 2, 14, 14)); pm[:, 52:54] = rng.uniform(0.5, 2, (3, 2, 14, 14))
    rb, rh, rv = oyolo.parse_prior_pose(pm.copy(), [(6., 3.), (12., 6.)], 15, 224, 224, 3, 2, 0.5, 0.5)
    b, h, v = parse_prior_pose(torch.from_numpy(pm.copy()).to(dev), [(6., 3.), (12., 6.)], 15, 224, 224, 3, 2, 0.5, 0.5)
    for i in range(3):
        same_n = len(b[i]) == len(rb[i])
        print("  img %d: det %d/%d" % (i, len(b[i]), len(rb[i])), end="")
        if same_n and len(b[i]):
            print(" bbox_exact=%s human_exact=%s vis_exact=%s" % (
                np.array_equal(np.array(b[i]), np.array(rb[i])), np.array_equal(np.array(h[i]), np.array(rh[i])),
                np.array_equal(np.array(v[i]), np.array(rv[i]))))
        else:
            print()


SECTIONS = {"rtpose": sec_rtpose, "yolo": sec_yolo, "parse": sec_parse, "preproc": sec_preproc, "yoloparse": sec_yoloparse}

if __name__ == "__main__":
    names = sys.argv[1:] or list(SECTIONS)
    print(torch.cuda.get_device_name(0))
    for n in names:
        try:
            SECTIONS[n]()
        except Exception:
            print("SECTION %s FAILED" % n)
            traceback.print_exc()
