"""The N>1 path on CPU: two gloo ranks each take the band the harness assigns (band_for_rank), fill it (with
the oracle standing in for the GPU render — test infrastructure), and one all-gather (gather_bands, the same
function bench.py calls over RCCL) must reproduce the single-rank image exactly."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ptlib

pkg = importlib.import_module("path-tracer-rust_amd")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, spp, seed, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
        npix = w * h
        b, e = pkg.band_for_rank(npix, rank, world)
        img, cnt, _ = ptlib.oracle_render(sc, w, h, spp, seed, idx_begin=b, idx_end=e, threads=2)
        local = torch.from_numpy(img[b:e].copy())
        full = pkg.gather_bands(local, npix, rank, world, dist)
        # the interleaved partition bench.py uses: rank r owns chunks r, r+world, ... (chunk = one row here)
        counts, index = pkg.chunk_owner_map(npix, world, w)
        mine = np.zeros((npix, 3), dtype=np.float32)
        n_chunks = (npix + w - 1) // w
        for c in range(rank, n_chunks, world):
            lo, hi = c * w, min((c + 1) * w, npix)
            part, _, _ = ptlib.oracle_render(sc, w, h, spp, seed, idx_begin=lo, idx_end=hi, threads=2)
            mine[lo:hi] = part[lo:hi]
        local2 = torch.from_numpy(mine[index[rank].numpy()].copy())
        assert local2.shape[0] == counts[rank]
        full2 = pkg.gather_chunks(local2, npix, rank, world, w, dist)
        assert torch.equal(full2, full)
        total = torch.tensor([cnt.ray_bounces], dtype=torch.int64)
        dist.all_reduce(total)
        np.save(os.path.join(out_dir, "rank%d.npy" % rank), full.numpy())
        if rank == 0:
            np.save(os.path.join(out_dir, "bounces.npy"), total.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("w,h,world", [(32, 24, 2), (31, 23, 2), (30, 7, 3)])
def test_two_rank_bands_reassemble_the_frame(tmp_path, w, h, world):
    spp, seed = 2, 6
    mp.spawn(_worker, args=(world, free_port(), w, h, spp, seed, str(tmp_path)), nprocs=world, join=True)
    sc = ptlib.load_scene_py(ptlib.scene_path("cornell"))
    want, cnt, _ = ptlib.oracle_render(sc, w, h, spp, seed, threads=2)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert got.shape == want.shape and np.array_equal(got, want)
    assert int(np.load(os.path.join(str(tmp_path), "bounces.npy"))[0]) == cnt.ray_bounces


def test_band_partition_properties():
    for npix in (1, 7, 786432, 4096 * 4096):
        for world in (1, 2, 3, 4, 8):
            bands = [pkg.band_for_rank(npix, r, world) for r in range(world)]
            assert bands[0][0] == 0 and bands[-1][1] == npix
            assert all(bands[i][1] == bands[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in bands]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        pkg.band_for_rank(10, 3, 2)


def test_chunk_counts_equal_the_owner_map():
    """chunk_counts (arithmetic, used per frame) against chunk_owner_map (explicit index lists) on ragged cases."""
    import random

    rnd = random.Random(5)
    cases = [(786432, 8, 1024), (786432, 3, 1024), (1, 1, 1), (5, 8, 2), (1000, 7, 33)]
    cases += [(rnd.randint(1, 5000), rnd.randint(1, 9), rnd.randint(1, 300)) for _ in range(300)]
    for npix, world, chunk in cases:
        counts, index = pkg.chunk_owner_map(npix, world, chunk)
        assert pkg.chunk_counts(npix, world, chunk) == counts
        assert sum(counts) == npix and all(int(i.numel()) == c for i, c in zip(index, counts))


def test_bench_launches_its_own_ranks_and_relays_their_failure():
    """`python bench.py --gpus 2` outside a launcher starts the two ranks itself (torch.distributed.run as a child, the
    parent never touching a GPU) and relays their return code.  Here the ranks cannot run - no GPU at all in the build
    container, a single one on the GPU box - and each says so itself: the failure is the children's message, not a
    usage error of the parent."""
    import subprocess
    import sys

    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ptlib.ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--spp", "1", "--width", "32", "--height", "24", "--no-variants", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env)
    n_dev = torch.cuda.device_count()
    if n_dev >= 2:
        assert r.returncode == 0, r.stderr[-3000:]
        return
    assert r.returncode != 0
    text = r.stdout + r.stderr
    if n_dev == 0:
        assert "bench.py needs a GPU" in text and "rank " in text, text[-3000:]
    else:
        assert "--gpus 2 needs 2 devices" in text, text[-3000:]
    assert "usage:" not in text
