"""Multi-process host-logic tests of the view sharding (gloo, world_size 2, CPU).

Checks the exchange step that makes multi-GPU global attention correct by construction:
K/V rows of every rank are gathered in rank-major (= view-major) token order, local queries attending
to the gathered K/V reproduce the corresponding rows of the unsharded attention (oracle math), and the
camera-token gather restores the full [S, 2C] matrix."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from iggt_official_amd.dist import ViewShard, view_partition

        torch.manual_seed(0)
        S, P, H, D = 4, 9, 2, 64
        C = H * D
        q = torch.randn(S * P, C)
        kv = torch.randn(S * P, 2 * C)          # [K | V] rows, token-major, as the qk-norm kernel writes them
        shard = ViewShard()
        v0, v1 = shard.local_views(S)
        assert (v0, v1) == view_partition(S, world, rank) and (v1 - v0) * world == S
        kv_all = shard.all_gather_kv(kv[v0 * P:v1 * P].contiguous())
        assert torch.equal(kv_all, kv)          # rank-major == view-major order
        # the key bounds travel with the keys: every rank's 32 norm maxima land in row `rank` of a [world, 32] buffer
        stats = torch.arange(32, dtype=torch.float32) + 100.0 * rank
        kv2, st_all = shard.all_gather_kv(kv[v0 * P:v1 * P].contiguous(), stats)
        want = torch.stack([torch.arange(32, dtype=torch.float32) + 100.0 * r for r in range(world)])
        assert torch.equal(kv2, kv) and torch.equal(st_all, want)
        kv3, st3, fin = shard.all_gather_kv_begin(kv[v0 * P:v1 * P].contiguous(), stats + 1.0)
        stats.fill_(-1.0)                        # the caller may rewrite its buffer at once: the shard keeps its own copy
        fin()
        assert torch.equal(kv3, kv) and torch.equal(st3, want + 1.0)

        def attn(qr, kvm):
            qh = qr.view(-1, H, D).transpose(0, 1)
            kh = kvm[:, :C].reshape(-1, H, D).transpose(0, 1)
            vh = kvm[:, C:].reshape(-1, H, D).transpose(0, 1)
            return torch.nn.functional.scaled_dot_product_attention(qh, kh, vh).transpose(0, 1).reshape(-1, C)

        full = attn(q, kv)
        mine = attn(q[v0 * P:v1 * P], kv_all)
        assert torch.allclose(mine, full[v0 * P:v1 * P], atol=1e-6)
        cam = torch.randn(S, 2 * C)
        assert torch.equal(shard.all_gather_rows(cam[v0:v1]), cam)
        # round 6: host decisions that change the launch sequence (a block takes the estimated-shift launches; captured graphs are
        # invalidated) are OR-ed over the ranks, so every rank switches, warms up and re-captures in step (models/aggregator.py)
        assert shard.agree_any([rank == 0, False, rank == 1, True], "cpu") == [1, 0, 1, 1]
        assert shard.agree_any([False] * 48, "cpu") == [0] * 48

        # head-group pipelined gather: kv_local[g] = [K heads of group g | V heads of group g]; per-group attention
        # over the gathered group buffers reproduces the same rows
        G = 2
        sh2 = ViewShard(kv_groups=G)
        assert sh2.kv_groups == G
        hg = H // G
        loc = kv[v0 * P:v1 * P]
        kv_local = torch.stack([torch.cat([loc[:, g * hg * D:(g + 1) * hg * D],
                                           loc[:, C + g * hg * D:C + (g + 1) * hg * D]], 1) for g in range(G)], 0)
        handles = sh2.gather_kv_groups(kv_local.contiguous())
        assert len(handles) == G
        outs = []
        for g, (work, kv_all_g) in enumerate(handles):
            if work is not None:
                work.wait()
            assert kv_all_g.shape == (S * P, 2 * hg * D)
            qg = q[v0 * P:v1 * P, g * hg * D:(g + 1) * hg * D].reshape(-1, hg, D).transpose(0, 1)
            kg = kv_all_g[:, :hg * D].reshape(-1, hg, D).transpose(0, 1)
            vg = kv_all_g[:, hg * D:].reshape(-1, hg, D).transpose(0, 1)
            outs.append(torch.nn.functional.scaled_dot_product_attention(qg, kg, vg).transpose(0, 1).reshape(-1, hg * D))
        assert torch.allclose(torch.cat(outs, 1), full[v0 * P:v1 * P], atol=1e-6)
        assert ViewShard(kv_groups=None).kv_groups == 1   # pipelining is opt-in (dist.py)
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_view_shard_gloo_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_view_partition_contract():
    from iggt_official_amd.dist import view_partition

    assert [view_partition(32, 8, r) for r in (0, 7)] == [(0, 4), (28, 32)]
    with pytest.raises(ValueError):
        view_partition(30, 8, 0)


def _worker_one(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from iggt_official_amd.dist import ViewShard

        plain, forced = ViewShard(), ViewShard(force=True, kv_groups=4)
        assert plain.world == 1 and not plain.active and plain.kv_groups == 1
        assert forced.active and forced.kv_groups == 4       # a world of one that issues its collectives anyway
        kv = torch.randn(7, 16)
        assert torch.equal(forced.all_gather_kv(kv), kv)
        out, finish = forced.all_gather_kv_begin(kv)
        finish()
        assert torch.equal(out, kv)
        fm = torch.randn(3, 5, 6, 8)                         # the track head gathers its NHWC feature maps by rows
        assert torch.equal(forced.all_gather_rows(fm), fm)
        ret[0] = "ok"
    finally:
        dist.destroy_process_group()


def test_view_shard_forced_collectives_world_of_one():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_one, args=(1, _free_port(), ret), nprocs=1, join=True)
    assert dict(ret) == {0: "ok"}


def test_query_points_argument_forms():
    """reference vggt.py:59-60 / 192-193: [N, 2] -> [1, N, 2]; anything else is rejected before any computation."""
    from iggt.models.vggt import IGGT

    q = torch.zeros(5, 2)
    assert IGGT._query(None, 1) is None
    assert IGGT._query(q, 1).shape == (1, 5, 2)
    assert IGGT._query(q[None].repeat(3, 1, 1), 3).shape == (3, 5, 2)
    for bad, b in ((torch.zeros(5, 3), 1), (torch.zeros(2, 5, 2), 1), (torch.zeros(5), 1)):
        with pytest.raises(ValueError):
            IGGT._query(bad, b)


def _worker_mixed_forms(rank, world, port, ret):
    """Round 5: whether a rank's call site runs the overlapped form (all_gather_kv_begin ... finish) or the gather-first form of
    the estimated shift (all_gather_kv with stats) is that RANK'S OWN decision (layers/blocks.py use_est).  Ranks that disagree
    must still be issuing the same collectives in the same order: here rank 0 takes one form and rank 1 the other, block after
    block, alternating -- both must end with identical gathered buffers, and the pair-operand message ([T_local, 4C]) of the x3
    rung travels through the same call."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from iggt_official_amd.dist import ViewShard

        torch.manual_seed(1)
        S, P, C = 4, 7, 128
        shard = ViewShard()
        v0, v1 = shard.local_views(S)
        for blk in range(6):
            kv = torch.randn(S * P, 2 * C, generator=torch.Generator().manual_seed(100 + blk)).half()
            stats_all = torch.stack([torch.arange(32, dtype=torch.float32) * (blk + 1) + 1000.0 * r for r in range(world)])
            mine, st = kv[v0 * P:v1 * P].contiguous(), stats_all[rank].clone()
            if (blk + rank) % 2 == 0:                 # this rank: overlapped form
                out, sout, fin = shard.all_gather_kv_begin(mine, st)
                fin()
            else:                                     # this rank: gather-first (estimated-shift) form
                out, sout = shard.all_gather_kv(mine, st)
            assert torch.equal(out, kv) and torch.equal(sout, stats_all), (rank, blk)
            pairs = torch.randn(S * P, 4 * C, generator=torch.Generator().manual_seed(200 + blk)).half()   # x3: [k_hi|v_hi|k_lo|v_lo]
            assert torch.equal(shard.all_gather_kv(pairs[v0 * P:v1 * P].contiguous()), pairs)
        ret[rank] = True
    finally:
        dist.destroy_process_group()


def test_ranks_may_disagree_on_the_attention_form():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_mixed_forms, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}
