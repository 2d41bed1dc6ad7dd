"""The N>1 path on CPU: world_size-2 (and 3) gloo processes shard a frame into row bands,
render their band (with the oracle standing in for the GPU renderer: same y_offset /
full_height contract as rc_render) and assemble it with sharding.gather_bands; frames mode
deals frames round-robin.  The assembled frame must equal the single-process frame bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cudaraytracing_amd import configs, sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, n, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import binding as ob
        pos, rad, col = ob.generate_scene(n, 0)
        cam = ob.camera_data((0, 0, 20), (0.05, -0.1, 0), w, h)
        b, e = configs.band_rows(h, world, rank)
        band = ob.render(pos, rad, col, cam, w, e - b, y_offset=b, full_height=h, threads=2)
        full = torch.zeros((h, w, 4), dtype=torch.float32) if rank == 0 else None
        sharding.gather_bands(torch.from_numpy(band), full, h, world, rank, dst=0)
        t = sharding.max_over_ranks(float(rank + 1), "cpu")
        assert t == float(world)
        assert sharding.gather_floats(0.5 * rank, "cpu") == [0.5 * r for r in range(world)]
        dist.barrier()
        if rank == 0:
            np.save(out_path, full.numpy())
    finally:
        dist.destroy_process_group()


def _worker_overlap(rank, world, port, w, h, n, parts, layout, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import binding as ob
        pos, rad, col = ob.generate_scene(n, 0)
        cam = ob.camera_data((0, 0, 20), (0.05, -0.1, 0), w, h)
        g = sharding.BandGather(h, world, rank, parts=parts, dst=0, layout=layout)
        full = torch.zeros((h, w, 4), dtype=torch.float32) if rank == 0 else None
        band = torch.zeros((g.rows, w, 4), dtype=torch.float32)
        g.post_receives(full)
        for k, (sb, se) in enumerate(g.pieces):
            if se > sb:
                piece = ob.render(pos, rad, col, cam, w, se - sb, y_offset=sb, full_height=h, threads=2)
                band[g.offsets[k]: g.offsets[k] + (se - sb)] = torch.from_numpy(piece)
            g.piece_rendered(k, band, full)
        g.wait()
        dist.barrier()
        if rank == 0:
            np.save(out_path, full.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,h,parts,layout", [(2, 96, 2, "contiguous"), (3, 101, 3, "contiguous"), (2, 5, 4, "contiguous"),
                                                  (2, 96, 2, "interleaved"), (3, 101, 2, "interleaved"), (2, 3, 2, "interleaved")])
def test_band_gather_with_overlap_pieces_gloo(oracle, tmp_path, world, h, parts, layout):
    w, n = 96, 30
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker_overlap, args=(world, _free_port(), w, h, n, parts, layout, out), nprocs=world, join=True)
    got = np.load(out)
    pos, rad, col = oracle.generate_scene(n, 0)
    cam = oracle.camera_data((0, 0, 20), (0.05, -0.1, 0), w, h)
    want = oracle.render(pos, rad, col, cam, w, h)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_sub_bands_partition():
    for b, e, parts in [(0, 10, 3), (5, 6, 4), (7, 7, 2), (100, 640, 2)]:
        sp = sharding.sub_bands(b, e, parts)
        assert sp[0][0] == b and sp[-1][1] == e
        for x, y in zip(sp, sp[1:]):
            assert x[1] == y[0]


@pytest.mark.parametrize("world,h", [(2, 96), (3, 100)])
def test_band_gather_gloo(oracle, tmp_path, world, h):
    w, n = 128, 40
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, n, out), nprocs=world, join=True)
    got = np.load(out)
    pos, rad, col = oracle.generate_scene(n, 0)
    cam = oracle.camera_data((0, 0, 20), (0.05, -0.1, 0), w, h)
    want = oracle.render(pos, rad, col, cam, w, h)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("layout", ["contiguous", "interleaved"])
def test_rank_pieces_tile_the_frame(layout):
    """Every row belongs to exactly one piece of one rank, whatever the height."""
    for world in (1, 2, 3, 8):
        for h in (0, 1, 5, 16, 101, 2160):
            owner = np.zeros(h, dtype=np.int32)
            for r in range(world):
                g = sharding.BandGather(h, world, r, parts=3, layout=layout)
                assert g.rows == sum(e - b for b, e in g.pieces)
                for b, e in g.pieces:
                    owner[b:e] += 1
            assert (owner == 1).all()


def test_interleaved_pieces_pair_outer_with_inner():
    assert sharding.rank_pieces(160, 4, 0, layout="interleaved") == [(0, 20), (80, 100)]
    assert sharding.rank_pieces(160, 4, 3, layout="interleaved") == [(60, 80), (140, 160)]


def test_frames_round_robin():
    for world in (1, 2, 4, 8):
        seen = sorted(f for r in range(world) for f in sharding.frames_for_rank(300, world, r))
        assert seen == list(range(300))
        sizes = [len(sharding.frames_for_rank(300, world, r)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("gpus", [2, 3])
def test_bench_starts_its_own_ranks_without_a_launcher(gpus):
    """`python bench.py --gpus N` outside torch.distributed.run starts the N ranks itself (a child
    launcher, 127.0.0.1 rendezvous) and relays rank 0's one line.  RC_BENCH_DRY_RUN=1: the ranks
    rendezvous over gloo, run the barrier and the MAX reduction, and render nothing (no GPU here)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RC_BENCH_DRY_RUN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--steps", "6", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == gpus and d["steps"] == 6 and d["max_rank"] == gpus - 1


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", RC_BENCH_DRY_RUN="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
