"""The N > 1 path on CPU: two ranks over gloo run the product's partitioning and
compositing code (raytracer_amd/dist.py) with the ORACLE standing in for the GPU
renderer, and rank 0 checks the composited frame against a single-process render.
On the GPU box the same code runs over nccl (= RCCL) with the HIP renderer."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, DEPTH, SEED = 40, 26, 70, 6, 99


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mode, result_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracer_amd import CameraConfiguration, RenderChunk, RenderSettings, SceneBuilder, sample_segment
        from raytracer_amd.dist import StripCompositor, reduce_accumulators, sample_range_of_rank, strip_chunks
        from tests.oracle import ORC_RNG_COUNTER, OracleScene
        b = SceneBuilder.sample("Random Spheres", CameraConfiguration(W, H))
        o = OracleScene(b)
        st = RenderSettings(W, H, DEPTH, SPP)
        whole = RenderChunk(0, 0, W, H)
        if mode == "tiles":
            chunks = strip_chunks(W, H, world, strips_per_rank=3)
            comp = StripCompositor(W, H, chunks, rank, world, torch.device("cpu"))
            assert sum(c.height for c in chunks) == H and len(chunks) == 3 * world
            for row, c in comp.slots:
                rgb, _, _ = o.render_chunk(st, c, ORC_RNG_COUNTER, SEED)
                comp.packed[row:row + c.height] = torch.from_numpy(rgb)
            frame = comp.composite()
            if rank == 0:
                full, _, _ = o.render_chunk(st, whole, ORC_RNG_COUNTER, SEED)
                np.save(result_path, np.array([float(np.array_equal(frame.numpy(), full))]))
        else:
            first, count = sample_range_of_rank(SPP, rank, world)
            assert first % sample_segment() == 0
            acc = torch.from_numpy(o.accumulate_chunk(st, whole, SEED, first, count))
            total = reduce_accumulators(acc, rank, world)
            summed = reduce_accumulators(acc, rank, world, how="allreduce")        # the collective of BASELINE configs[4]
            if rank == 0:
                assert np.abs(summed.numpy() - total.numpy()).max() <= 1e-5 * SPP      # same sum, the collective's association
                parts = [o.accumulate_chunk(st, whole, SEED, *sample_range_of_rank(SPP, r, world)) for r in range(world)]
                expect = parts[0].copy()
                for p in parts[1:]:
                    expect += p
                full = o.accumulate_chunk(st, whole, SEED, 0, SPP)
                ok = np.array_equal(total.numpy(), expect) and np.abs(total.numpy() - full).max() < 1e-4
                covered = sum(sample_range_of_rank(SPP, r, world)[1] for r in range(world)) == SPP
                np.save(result_path, np.array([float(ok and covered)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["tiles", "spp"])
def test_two_ranks_gloo(mode, tmp_path):
    result = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(2, _free_port(), mode, result), nprocs=2, join=True)
    assert np.load(result)[0] == 1.0


def test_partition_helpers():
    sys.path.insert(0, ROOT)
    from raytracer_amd import sample_segment
    from raytracer_amd.dist import chunks_of_rank, sample_range_of_rank, strip_chunks
    for world in (1, 2, 4, 8):
        chunks = strip_chunks(1920, 1080, world)
        assert len(chunks) == 8 * world and sum(c.height for c in chunks) == 1080
        got = sorted(c.top for r in range(world) for c in chunks_of_rank(chunks, r, world))
        assert got == sorted(c.top for c in chunks)
        for spp in (1, 16, 32, 100, 256, 4096):
            ranges = [sample_range_of_rank(spp, r, world) for r in range(world)]
            assert sum(c for _, c in ranges) == spp
            pos = 0
            for f, c in ranges:
                if c:
                    assert f == pos and f % sample_segment() == 0
                    pos += c
    assert len(strip_chunks(64, 5, 8)) == 5        # never more strips than rows


def _worker_interleaved(rank, world, port, result_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracer_amd import CameraConfiguration, RenderChunk, RenderSettings, SceneBuilder
        from raytracer_amd.dist import InterleavedCompositor
        from tests.oracle import ORC_RNG_COUNTER, OracleScene
        b = SceneBuilder.sample("Random Spheres", CameraConfiguration(W, H))
        o = OracleScene(b)
        st = RenderSettings(W, H, DEPTH, 3)
        comp = InterleavedCompositor(W, H, rank, world, torch.device("cpu"), strip_height=8)
        rows = comp.image_rows(rank)
        assert len(rows) == comp.my_rows
        # the oracle stands in for rt_render_strips_device: strip by strip into packed rows
        i = 0
        while i < len(rows):
            n = 1
            while i + n < len(rows) and rows[i + n] == rows[i] + n:
                n += 1
            rgb, _, _ = o.render_chunk(st, RenderChunk(0, rows[i], W, n), ORC_RNG_COUNTER, SEED)
            comp.packed[i:i + n] = torch.from_numpy(rgb)
            i += n
        frame = comp.composite()
        if rank == 0:
            full, _, _ = o.render_chunk(st, RenderChunk(0, 0, W, H), ORC_RNG_COUNTER, SEED)
            np.save(result_path, np.array([float(np.array_equal(frame.numpy(), full))]))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo_interleaved_strips(tmp_path):
    result = str(tmp_path / "ok.npy")
    mp.spawn(_worker_interleaved, args=(2, _free_port(), result), nprocs=2, join=True)
    assert np.load(result)[0] == 1.0


def test_strip_rows_cover_the_image():
    sys.path.insert(0, ROOT)
    from raytracer_amd import strip_rows
    for h in (1, 5, 26, 1080, 2160):
        for world in (1, 2, 3, 8):
            for sh in (8, 16):
                assert sum(strip_rows(h, sh, r, world) for r in range(world)) == h


def _worker_columns(rank, world, port, result_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracer_amd import CameraConfiguration, RenderChunk, RenderSettings, SceneBuilder
        from raytracer_amd.dist import ColumnCompositor
        from tests.oracle import ORC_RNG_COUNTER, OracleScene
        b = SceneBuilder.sample("Random Spheres", CameraConfiguration(W, H))
        o = OracleScene(b)
        st = RenderSettings(W, H, DEPTH, 3)
        comp = ColumnCompositor(W, H, rank, world, torch.device("cpu"))
        cols = comp.image_columns(rank)
        assert len(cols) == comp.my_width
        # the oracle stands in for rt_render_tile_columns_device: tile column by tile column into the packed columns
        i = 0
        while i < len(cols):
            n = 1
            while i + n < len(cols) and cols[i + n] == cols[i] + n and n < 8:
                n += 1
            rgb, _, _ = o.render_chunk(st, RenderChunk(cols[i], 0, n, H), ORC_RNG_COUNTER, SEED)
            comp.packed[:, i:i + n] = torch.from_numpy(rgb)
            i += n
        frame = comp.composite()
        if rank == 0:
            full, _, _ = o.render_chunk(st, RenderChunk(0, 0, W, H), ORC_RNG_COUNTER, SEED)
            np.save(result_path, np.array([float(np.array_equal(frame.numpy(), full))]))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo_dealt_tile_columns(tmp_path):
    result = str(tmp_path / "ok.npy")
    mp.spawn(_worker_columns, args=(2, _free_port(), result), nprocs=2, join=True)
    assert np.load(result)[0] == 1.0


def test_tile_column_widths_cover_the_image():
    sys.path.insert(0, ROOT)
    from raytracer_amd import tile_columns_width
    for w in (1, 5, 8, 26, 63, 64, 65, 1920, 3840):
        for world in (1, 2, 3, 8, 16):
            widths = [tile_columns_width(w, r, world) for r in range(world)]
            assert sum(widths) == w and all(x >= 0 for x in widths)
            # only the owner of the frame's last tile column can hold a narrower one
            assert sum(1 for x in widths if x % 8) <= 1
