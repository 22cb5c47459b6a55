"""The N > 1 path of bench.py on two CPU processes over gloo: ray_tracer_amd.distributed
splits the frame (interleaved rows, or sample shares summed by a reduce) and puts it back
together.  No GPU here, so the CPU oracle stands in for the device renderer; what is under
test is the split, the gather and the reduce, which are the same code on GPUs (where the
reduce runs on RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest

W, H, SPP = 48, 26, 4  # H is not a multiple of the rank counts: ragged shares


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world_size, port, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch
    import torch.distributed as dist

    import oracle_lib
    from ray_tracer_amd import configs, distributed

    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        cfg = configs.C1
        world = cfg.world()
        cam = cfg.camera(W, H)
        # tile split: this rank's rows only
        out = np.zeros((H, W, 3), dtype=np.uint8)
        tile_size, tile_count, tile_index = distributed.row_tiles(W, world_size, rank)
        assert (tile_size, tile_count, tile_index) == (W, world_size, rank)
        for row in distributed.rows_of(rank, world_size, H):
            rgb, _, _, _ = oracle_lib.render(world.desc, cam, cfg.background, W, H, SPP, pixel_range=(row * W, (row + 1) * W),
                                             nthreads=1)
            out[row] = rgb[row]
        distributed.gather_rows(out, rank, world_size)
        # sample split: all pixels, this rank's samples, summed in f64
        lo, hi = distributed.sample_range(SPP, rank, world_size)
        _, sums, _, _ = oracle_lib.render(world.desc, cam, cfg.background, W, H, SPP, sample_range=(lo, hi), nthreads=1)
        accum = torch.from_numpy(sums.reshape(-1).copy())
        distributed.reduce_accum(accum)
        if rank == 0:
            np.save(os.path.join(tmpdir, "tiles.npy"), out)
            np.save(os.path.join(tmpdir, "sums.npy"), accum.numpy().reshape(H, W, 3))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size", [2, 3])
def test_tile_and_sample_split_over_gloo(world_size, tmp_path, oracle):
    import torch.multiprocessing as mp

    from ray_tracer_amd import configs

    mp.spawn(_worker, args=(world_size, _free_port(), str(tmp_path)), nprocs=world_size, join=True)
    cfg = configs.C1
    world = cfg.world()  # (kept in a variable: its desc points into it)
    full, sums, _, _ = oracle.render(world.desc, cfg.camera(W, H), cfg.background, W, H, SPP)
    assert np.array_equal(np.load(tmp_path / "tiles.npy"), full)
    # shares summed in another order: equal to rounding, and the bytes agree to one level
    got = np.load(tmp_path / "sums.npy")
    assert np.allclose(got, sums, rtol=1e-13, atol=0)
    assert np.abs(oracle.resolve(got, SPP).astype(int) - full.astype(int)).max() <= 1


def test_sample_shares_cover_the_range_exactly():
    from ray_tracer_amd import distributed

    for spp in (1, 7, 500):
        for n in (1, 2, 3, 8):
            shares = [distributed.sample_range(spp, r, n) for r in range(n)]
            if spp < n:  # fewer samples than ranks: empty shares are None, never (0, 0) = "all samples"
                assert sum(s is None for s in shares) == n - spp
                assert sorted(s for s in shares if s is not None) == [(k, k + 1) for k in range(spp)]
                continue
            assert shares[0][0] == 0 and shares[-1][1] == spp
            assert all(a[1] == b[0] for a, b in zip(shares, shares[1:]))
            sizes = [b - a for a, b in shares]
            assert max(sizes) - min(sizes) <= 1
    for h in (1, 5, 800):
        for n in (1, 2, 3, 8):
            rows = sorted(r for k in range(n) for r in distributed.rows_of(k, n, h))
            assert rows == list(range(h))


def test_bench_starts_its_own_ranks_from_a_plain_shell(monkeypatch):
    """`python bench.py --gpus N` without torch.distributed.run around it: the script must start N fresh
    processes itself (before it touches a GPU) with the arguments it was given, on the loopback address."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, *a, **kw):
        seen["cmd"] = cmd
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as stop:
        bench.main()
    assert stop.value.code == 7  # the children's exit status is the script's
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == [os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "2"]
