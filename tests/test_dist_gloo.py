"""N > 1 path on CPU: two gloo ranks shard a canvas by interleaved row bands exactly as bench.py does
on GPUs (svgrasterize.py_amd/dist.py), each rank rendering ONLY its bands through the reference's own
viewport mechanism (here with the CPU oracle standing in for the device), then all_gather + assemble.
The result must equal the single-process render (to double rounding: the reference's viewport restart,
SURVEY 8e), and exactly after float32 rounding."""
import os
import socket

import numpy as np
import pytest

from tests.util import assert_f32_1ulp

TILE_ROWS = 16
SIZE, N_PATHS = 200, 40


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    from oracle import oracle as orc
    from svgrasterize_amd import dist as sdist, synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = synth.make_scene(SIZE, N_PATHS)
    pres = synth.presentation_segs(sc)
    parts = []
    for r0, r1 in sdist.owned_row_ranges(SIZE, TILE_ROWS, rank, world):
        band, _, _ = orc.render_solid(pres, sc["seg_kind"], sc["path_seg_off"], sc["path_rule"], sc["path_paint"],
                                      (r0, 0, r1 - r0, SIZE), clip01=True)
        pad = np.zeros((TILE_ROWS - (r1 - r0), SIZE, 4))
        parts.append(np.concatenate([band, pad]))
    local = torch.from_numpy(np.concatenate(parts))
    full = sdist.gather_canvas(local, SIZE, TILE_ROWS)
    assert torch.equal(full, sdist.gather_canvas_staged(local, SIZE, TILE_ROWS))  # (direct rounds + ragged tail == padded gather + copy)
    # ... and with strips of several bands, complete rounds only / a short last strip / ranks without a last strip
    for rows_, strip_ in ((192, 3), (200, 3), (200, 2), (40, 2), (16, 1)):
        k = len(sdist.owned_bands(rows_, TILE_ROWS, rank, world, strip_))
        mine = torch.arange(k * TILE_ROWS * 3 * 2, dtype=torch.float64).reshape(k * TILE_ROWS, 3, 2) + 1000.0 * rank
        a_ = sdist.gather_canvas(mine, rows_, TILE_ROWS, strip=strip_)
        b_ = sdist.gather_canvas_staged(mine, rows_, TILE_ROWS, strip=strip_)
        assert a_.shape[0] == rows_ and torch.equal(a_, b_), (rows_, strip_)
    # max-over-ranks reduction of a per-rank clock, as bench.py does
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    if rank == 0:
        np.save(os.path.join(out_dir, "full.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_band_partition_helpers():
    from svgrasterize_amd import dist as sdist

    for rows, world in [(200, 2), (4096, 8), (17, 3), (16, 4), (1, 2)]:
        seen = []
        for r in range(world):
            seen += sdist.owned_row_ranges(rows, TILE_ROWS, r, world)
        seen.sort()
        assert seen[0][0] == 0 and seen[-1][1] == rows
        assert all(a[1] == b[0] for a, b in zip(seen, seen[1:]))
        assert max(len(sdist.owned_bands(rows, TILE_ROWS, r, world)) for r in range(world)) == sdist.max_owned_bands(rows, TILE_ROWS, world)


@pytest.mark.timeout(300)
def test_two_rank_gloo_render_matches_single(tmp_path):
    import torch.multiprocessing as mp

    from oracle import oracle as orc
    from svgrasterize_amd import synth

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "full.npy")
    sc = synth.make_scene(SIZE, N_PATHS)
    ref, _, _ = orc.render_solid(synth.presentation_segs(sc), sc["seg_kind"], sc["path_seg_off"], sc["path_rule"],
                                 sc["path_paint"], sc["viewport"], clip01=True)
    assert np.abs(got - ref).max() < 1e-11
    assert_f32_1ulp(got.astype(np.float32), ref, what="2-rank gloo canvas")


GPU_SIZE, GPU_PATHS = 1024, 384


def _gpu_worker(rank, world, port, out_dir):
    """One rank of a 2-rank render in which EVERY rank drives libsvgr_hip.so (both on GPU 0, as a one-GPU box allows): its
    strips through svgr_batch_set_bands, gloo for the collectives -- the launcher path of bench.py --gpus N minus RCCL."""
    import torch
    import torch.distributed as dist

    import svgrasterize_amd as S
    from svgrasterize_amd import _abi, dist as sdist, synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = S.Context.get(0)
    sc = synth.make_scene(GPU_SIZE, GPU_PATHS)
    tr = _abi.tile_rows()
    strip = sdist.default_strip_bands(GPU_SIZE, tr, world)
    batch = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"],
                       viewport=sc["viewport"])
    batch.set_bands(rank, world, strip)
    batch.plan()
    own = batch.owned_rows()
    out = ctx.alloc(max(own, 1) * GPU_SIZE * 16)
    for _ in range(2):   # (the second render recomputes the geometry instead of reusing the plan's)
        batch.render(out, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
    local = torch.from_numpy(out.download((own, GPU_SIZE, 4), np.float32))
    dist.barrier()
    full = sdist.gather_canvas(local, GPU_SIZE, tr, strip=strip)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    if rank == 0:
        np.save(os.path.join(out_dir, "full_gpu.npy"), full.numpy())
    dist.barrier()
    batch.destroy()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_rank_gloo_render_on_the_gpu_matches_the_oracle(tmp_path):
    """Two processes, each a rank that renders ITS strips with the HIP library on GPU 0; the assembled canvas against the CPU
    oracle under the float32 contract."""
    import torch.multiprocessing as mp

    from oracle import oracle as orc
    from svgrasterize_amd import synth

    port = _free_port()
    mp.spawn(_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "full_gpu.npy")
    sc = synth.make_scene(GPU_SIZE, GPU_PATHS)
    ref, _, _ = orc.render_solid(synth.presentation_segs(sc), sc["seg_kind"], sc["path_seg_off"], sc["path_rule"],
                                 sc["path_paint"], sc["viewport"], clip01=True)
    assert_f32_1ulp(got, ref, what="2-rank canvas rendered on the GPU")


def test_rows_subscene_keeps_exactly_what_can_reach_the_block():
    """Host-side distribution of the weak-scaling bench (synth.rows_subscene): a path is given to a row block iff the hull
    of its control points (+-2 rows) reaches the block; order is kept; the blocks together cover every path."""
    import numpy as np

    from svgrasterize_amd import synth

    size, n, blocks = 300, 50, 4
    tall = synth.make_tall_scene(size, n, blocks)
    off = tall["path_seg_off"]
    assert len(off) == n * blocks + 1 and tall["viewport"] == (0, 0, size * blocks, size)
    # block b of the tall scene is the single scene of seed b, moved down
    one = synth.make_scene(size, n, seed=synth.SEED + 2)
    s2 = tall["segs"][off[2 * n]: off[3 * n]].copy()
    s2[:, 1::2] -= 2 * size
    assert np.allclose(s2, one["segs"], rtol=0, atol=1e-9)
    seen = np.zeros(n * blocks, dtype=int)
    for b in range(blocks):
        sub, kept = synth.rows_subscene(tall, b * size, (b + 1) * size)
        assert np.all(np.diff(kept) > 0) and sub["viewport"] == (b * size, 0, size, size)
        assert len(sub["path_seg_off"]) == len(kept) + 1 and sub["path_seg_off"][-1] == len(sub["segs"])
        seen[kept] += 1
        for p in range(n * blocks):
            y = tall["segs"][off[p]: off[p + 1], 1::2]
            reaches = np.floor(y.min()) - 2 < (b + 1) * size and np.ceil(y.max()) + 2 > b * size
            assert reaches == (p in set(kept.tolist()))
        # the kept paths carry their own segments, paints and rules
        k0 = int(kept[0])
        assert np.array_equal(sub["segs"][: sub["path_seg_off"][1]], tall["segs"][off[k0]: off[k0 + 1]])
        assert np.array_equal(sub["path_paint"], tall["path_paint"][kept]) and np.array_equal(sub["path_rule"], tall["path_rule"][kept])
    assert seen.min() >= 1 and seen.max() >= 2  # every path somewhere, border paths in two blocks


def _rccl_worker(rank, world, port, out_dir):
    """bench.py's N > 1 branch on the real back-end, with the one rank a one-GPU box can give RCCL: torch first, the process
    group on `nccl` with a device id, the render straight into a torch CUDA tensor (ctx.wrap), barrier / all_reduce on GPU
    tensors, and the strips gathered into the canvas rows by all_gather_into_tensor on the device."""
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    import svgrasterize_amd as S
    from svgrasterize_amd import _abi, dist as sdist, synth

    ctx = S.Context.get(0)
    sc = synth.make_scene(GPU_SIZE, GPU_PATHS)
    tr = _abi.tile_rows()
    strip = sdist.default_strip_bands(GPU_SIZE, tr, world)
    batch = _abi.Batch(ctx, sc["segs"], sc["seg_kind"], sc["path_seg_off"], sc["path_m6"], sc["path_rule"], sc["path_paint"],
                       viewport=sc["viewport"])
    batch.set_bands(rank, world, strip)
    batch.plan()
    own = batch.owned_rows()
    out_t = torch.empty((own, GPU_SIZE, 4), dtype=torch.float32, device="cuda:0")
    out = ctx.wrap(out_t.data_ptr(), out_t.numel() * 4)
    batch.render(out, _abi.OUT_CANVAS_F32, _abi.RENDER_CLIP01)
    ctx.sync()
    dist.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64, device="cuda:0")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    full = sdist.gather_canvas(out_t, GPU_SIZE, tr, strip=strip)
    staged = sdist.gather_canvas_staged(out_t, GPU_SIZE, tr, strip=strip)
    torch.cuda.synchronize()
    assert torch.equal(full, staged)
    if rank == 0:
        np.save(os.path.join(out_dir, "full_rccl.npy"), full.cpu().numpy())
    dist.barrier()
    batch.destroy()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_one_rank_rccl_group_runs_the_bench_collectives(tmp_path):
    """`init_process_group("nccl")` and every collective bench.py --gpus N issues, on RCCL itself -- with world size 1, which is
    what a box with one GPU can give it (RCCL refuses two ranks on one device; the N = 2 data path runs under gloo above)."""
    import torch.multiprocessing as mp

    from oracle import oracle as orc
    from svgrasterize_amd import synth

    port = _free_port()
    mp.spawn(_rccl_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    got = np.load(tmp_path / "full_rccl.npy")
    sc = synth.make_scene(GPU_SIZE, GPU_PATHS)
    ref, _, _ = orc.render_solid(synth.presentation_segs(sc), sc["seg_kind"], sc["path_seg_off"], sc["path_rule"],
                                 sc["path_paint"], sc["viewport"], clip01=True)
    assert_f32_1ulp(got, ref, what="1-rank RCCL canvas")
