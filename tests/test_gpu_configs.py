"""BASELINE.json configs at their stated sizes on one GPU.

configs[2]  "test_mpreal full 4k-frame sweep, batch-sharded across 8 MI355X via RCCL all-gather": 4 484 distinct synthetic
            frames (dataset_summary test count) go through dataset.run_sweep_streaming rank by rank with
            shard_indices(.., world=8) on ONE GPU; the eight shards are put back in global order exactly as gather_records
            does after the all-gather.  Checked: coverage and order of all 4 484 records, shard sizes, every record equal
            to what a plain engine gives for that frame (a sample), and a sample against the CPU oracle end to end.
configs[3]  the >= 4 persons per frame stream: 32-frame batches of planted 4 / 6 / 8-person maps through the parse kernels
            against the oracle, every frame (the throughput side is bench.py's "mpaug_parse" leg).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_config2_4484_frame_sweep_sharded_over_8_ranks(gpu):
    from popnet_amd import _lib, dataset, synth
    from popnet_amd.pipeline import PoseEngine, StreamingEngine, deinterleave, shard_indices
    from oracle import nets as onets, parse_paf as oparse, preproc as opre
    N, WORLD, BS = 4484, 8, 32
    frames = synth.SynthSweep(N)
    se = StreamingEngine(PoseEngine, depth=3, precision="fp32", device=gpu, max_batch=BS)
    se.capture()
    per = (N + WORLD - 1) // WORLD
    item = _lib.POSE_FRAME_DTYPE.itemsize
    shards = torch.zeros((WORLD, per, item), dtype=torch.uint8, device=gpu)
    for rank in range(WORLD):                                     # what the 8 processes of the real run do, one after the other
        local = dataset.run_sweep_streaming(se, frames, BS, rank=rank, world=WORLD, gather=False)
        mine = shard_indices(N, rank, WORLD)
        assert local.shape == (len(mine), item) and len(mine) in (per, per - 1)
        shards[rank, :len(mine)] = local                          # all_gather_into_tensor delivers exactly this layout
    recs = deinterleave(shards, N).cpu().numpy().view(_lib.POSE_FRAME_DTYPE).reshape(-1)
    assert len(recs) == N and int((recs["status"] != 0).sum()) == 0
    # order / coverage: record i must be the record of frame i -- compare a spread sample (incl. both ends and the ragged
    # tail of the last rank) with a plain engine on exactly that frame
    with pytest.raises(_lib.PopnetError, match="locked"):       # the captured engines are frozen at their batch size (pn_net_lock)
        se.engines[0].predict(torch.from_numpy(frames.load(0)[None]).to(gpu))
    eng = PoseEngine(precision="fp32", device=gpu, max_batch=1)   # same seeded + calibrated weights, its own net
    sample = [0, 1, 7, 8, 63, 64, 65, 1000, 2241, 4470, 4476, 4477, 4483]
    for i in sample:
        one = eng.predict(torch.from_numpy(frames.load(i)[None]).to(gpu)).cpu().numpy().view(_lib.POSE_FRAME_DTYPE).reshape(-1)[0]
        assert recs[i].tobytes() == one.tobytes(), i
    # distinct frames give distinct records almost everywhere (a sweep that repeated or dropped frames would not)
    assert len({r.tobytes() for r in recs[::7]}) > 0.9 * len(recs[::7])
    # a sample against the CPU oracle, end to end (fp32 engine: assignment exact, 3D within 1e-3 m)
    sd = {k: v.detach().cpu() for k, v in eng.model.state_dict().items()}
    for i in (5, 4483):
        x = opre.preprocess_batch(frames.load(i)[None])
        paf, heat, z = (a.numpy().transpose(0, 2, 3, 1) for a in onets.rtpose_light3d_forward(torch.from_numpy(x), sd))
        ref = oparse.frame_to_records(heat[0].copy(), paf[0].copy(), z[0].copy())
        n = int(recs[i]["n_persons"])
        assert n == len(ref["humans_3d"])
        assert np.array_equal(recs[i]["person_joint"][:n], np.asarray(ref["assoc"]).reshape(-1, 17)[:, :15].astype(np.int32))
        if n:
            assert np.abs(recs[i]["joints_3d"][:n] - np.array(ref["humans_3d"])).max() < 1e-3


@pytest.mark.parametrize("persons", [4, 6, 8])
def test_config3_multi_person_batches_match_the_oracle(gpu, persons):
    from popnet_amd import _lib, synth
    from popnet_amd.pipeline import PoseEngine, records_to_numpy
    from popnet_amd.utils.paf_to_pose import frame_assoc, frame_joint_list
    from oracle import parse_paf as oparse
    eng = PoseEngine(precision="bf16", device=gpu, max_batch=32)
    heat, paf, z = synth.planted_batch(900 + persons, [persons] * 32, noise=0.01)
    eng.heat.copy_(torch.from_numpy(heat).to(gpu))
    eng.paf.copy_(torch.from_numpy(paf).to(gpu))
    eng.z.copy_(torch.from_numpy(z).to(gpu))
    wire = torch.zeros((32, _lib.POSE_WIRE_DTYPE.itemsize), device=gpu, dtype=torch.uint8)
    eng.parse(32, wire=wire)
    recs = records_to_numpy(eng.frames)
    wrec = wire.cpu().numpy().view(_lib.POSE_WIRE_DTYPE).reshape(-1)
    found = 0
    for b in range(32):
        ref = oparse.frame_to_records(heat[b].transpose(1, 2, 0).copy(), paf[b].transpose(1, 2, 0).copy(), z[b].transpose(1, 2, 0).copy())
        assert int(recs[b]["status"]) == 0
        jl, assoc = frame_joint_list(recs[b]), frame_assoc(recs[b])
        assert jl.shape == ref["joint_list"].shape and np.array_equal(jl, ref["joint_list"])
        ra = np.asarray(ref["assoc"]).reshape(-1, 17)
        assert assoc.shape == ra.shape and np.array_equal(assoc[:, :15], ra[:, :15])
        n = assoc.shape[0]
        found += n
        if n:
            assert np.array_equal(recs[b]["joints_3d"][:n], np.array(ref["humans_3d"]))
            assert int(wrec[b]["n_persons"]) == n and np.array_equal(wrec[b]["person_joint"][:min(n, 16)], recs[b]["person_joint"][:min(n, 16)].astype(np.int16))
    assert found >= 32 * persons * 0.9          # the planted skeletons are (nearly) all assembled


def test_locked_engine_survives_an_invalidated_module(gpu):
    """invalidate() / load_state_dict() on the module of a LOCKED engine (its hipGraphs point at the pn_net's device
    buffers and descriptors).  The locked handle must not be destroyed (it is retired: a replay stays memory-safe and reproduces
    the records of the weights it was captured with), the engine must refuse to run -- and not silently compile a second net --
    until it is unlocked, and the unlock frees the retired handle and lets the engine recompile."""
    from popnet_amd import _lib, synth
    from popnet_amd.pipeline import PoseEngine
    eng = PoseEngine(precision="bf16", device=gpu, max_batch=4)
    d = torch.from_numpy(synth.synth_depth(4, seed=3)).to(gpu)
    out = torch.zeros((4, _lib.POSE_FRAME_DTYPE.itemsize), device=gpu, dtype=torch.uint8)
    eng.predict(d, out)
    torch.cuda.synchronize()
    want = out.clone()
    eng.lock()
    handle = eng._locked_net
    side = torch.cuda.Stream(device=gpu)
    side.wait_stream(torch.cuda.current_stream(gpu))
    with torch.cuda.stream(side):
        eng.predict(d, out)
    torch.cuda.current_stream(gpu).wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        eng.predict(d, out)
    eng.model.invalidate()                                        # what load_state_dict() and a train-mode forward do as well
    assert eng.model._net is None and eng.model._retired == [handle] and eng.model._pins == {handle: 1}
    with pytest.raises(_lib.PopnetError, match="locked / captured"):
        eng.predict(d, out)
    assert eng.model._net is None                                 # ... and no fresh net was compiled behind the graph's back
    out.zero_()
    g.replay()                                                    # the retired net is still alive: same records as before
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    del g
    eng.lock(False)
    assert eng.model._retired == [] and eng.model._pins == {}
    out.zero_()
    eng.predict(d, out)                                           # unlocked: recompiles and runs
    torch.cuda.synchronize()
    assert torch.equal(out, want)
