"""Multi-process view sharding on REAL kernels (GPU): two ranks on cuda:0 (gloo transport, because RCCL refuses
two ranks on one device) each run the sharded IGGT forward on their half of the views; every rank's outputs must
match the reference fixture for its views at the same tolerance as the unsharded run.  This exercises exactly the
code path bench.py --gpus N uses (ViewShard K/V all-gather in front of the 24 global attentions -- pipelined over
head groups with per-group attention launches on side streams, or as one gather -- and the camera-token gather),
only the transport differs."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, kv_groups, graphs, overlap, ret, backend="gloo"):
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    if backend == "nccl":      # RCCL: one rank per device -- only a world of one fits a one-GPU box
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import load_golden
        from helpers import build_gpu_model, errors
        from iggt_official_amd import precision
        from iggt_official_amd.dist import ViewShard
        from oracle import weights

        precision.set_gather_overlap(overlap)

        torch.cuda.set_device(0)
        g = load_golden(case)
        m = g["meta"]
        track = "query_points" in g            # track fixtures: the query_points path, feature maps gathered over ranks
        model = build_gpu_model(m["mode"], m["weight_seed"], include_track=track)
        # kv_groups 1: one gather (default); 4: gather pipelined over head groups.  A world of one issues its collectives anyway
        shard = ViewShard(kv_groups=kv_groups, force=(world == 1))
        assert shard.kv_groups == kv_groups
        model.set_view_shard(shard)
        v0, v1 = shard.local_views(m["S"])
        images = weights.make_images(m["S"], m["H"], m["W"], seed=m["image_seed"], device="cuda")[v0:v1]
        if graphs:   # hipGraph segments with the collectives as eager steps between them (iggt_official_amd/graphs.py)
            model.enable_graphs(True)
            model(images)                       # capture
        pred = model(images, query_points=g["query_points"].cuda()) if track else model(images)   # graphs: replay
        torch.cuda.synchronize()
        if graphs:
            seg = next(iter(model._gcache._graphs.values()))[1]
            # eager steps: per global block the K/V gather (begin + finish when it overlaps the own-key attention) + 1 camera gather
            assert seg.num_segments == (48 if overlap else 24) + 1 + 1, seg.num_segments
            model.enable_graphs(False)
        res = {}
        for k in ("depth", "depth_conf", "world_points", "world_points_conf", "part_feat"):
            if k in g:
                res[k] = errors(pred[k], g[k][:, v0:v1])[1]
        if track:                              # every rank tracks over ALL views
            res["track"] = errors(pred["track"], g["coord_preds"][-1])[1]
            res["vis"] = errors(pred["vis"], g["vis"])[1] / 5       # gated at 5e-3 like the unsharded run
            res["conf"] = errors(pred["conf"], g["conf"])[1] / 5
        else:
            res["pose_enc"] = errors(torch.stack(pred["pose_enc"], 0), g["pose_enc"])[1]   # all views on every rank
        ret[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,kv_groups,graphs,overlap", [("tiny_s2_56_stress", 4, False, False),
                                                           ("tiny_s2_56_stress", 1, False, False),
                                                           ("tiny_s2_56_stress", 1, False, True),
                                                           ("tiny_s2_56_stress", 1, True, True),
                                                           ("tiny_s2_56_stress", 1, True, False),
                                                           ("track_s2_140x182_stress", 1, False, True)])
def test_two_rank_sharded_forward_matches_reference(case, kv_groups, graphs, overlap):
    """kv_groups 4: gather pipelined over head groups (online-max kernel); kv_groups 1: one gather, static-bound attention --
    either gather -> one launch, or (overlap) own keys while the gather is in flight, then the other rank's keys, one slot
    each, folded by the combine kernel; graphs: the same as hipGraph segments with the collectives as eager steps."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), case, kv_groups, graphs, overlap, ret), nprocs=world, join=True)
    assert set(ret.keys()) == {0, 1}
    for rank, res in ret.items():
        for k, l2 in res.items():
            assert l2 < 1e-3, (rank, k, l2)   # same gate as the unsharded run (fp16 operands)


@pytest.mark.parametrize("kv_groups,graphs,overlap", [(1, False, False), (1, False, True), (1, True, True), (4, False, False)])
def test_rccl_world_of_one(kv_groups, graphs, overlap):
    """The REAL RCCL calls of the sharded path on the one GPU this box has: backend "nccl", a world of one rank that issues
    every collective regardless (ViewShard(force=True)): communicator set-up with device_id, all_gather_into_tensor on the
    16-bit K/V buffers (synchronous, asynchronous with work.wait() for the overlapped variant, per head group), the fp32
    camera-token gather, and the collectives as eager steps between hipGraph segments.  What it cannot show is transport
    between devices; outputs must match the reference fixture as in the unsharded run."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(1, _free_port(), "tiny_s2_56_stress", kv_groups, graphs, overlap, ret, "nccl"), nprocs=1, join=True)
    assert set(ret.keys()) == {0}
    for k, l2 in ret[0].items():
        assert l2 < 1e-3, (k, l2)
