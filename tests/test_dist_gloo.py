"""N > 1 plumbing on CPU (gloo): row-band ownership + the single gather of bench.py
(workloads.gather_bands).  The bands are rendered by the oracle here -- the GPU kernel's
own band rendering is covered by test_gpu_parity.test_bands_tile_the_frame_bitwise."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, w, h, depth, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as G
    import workloads
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        O = G.load_oracle()
        n_rows = h // 32
        band = workloads.patch_rows_for_rank(n_rows, rank, world)
        frame = np.full((h, w, 3), -1.0)                       # sentinel: untouched rows stay -1
        if band[1] > band[0]:
            O.render(O.OracleScene.create_default(), w, h, max_depth=depth, n_threads=2, frame=frame, band=band)
        t = torch.from_numpy(frame)
        workloads.gather_bands(dist, t, n_rows, rank, world)
        dist.barrier()
        if rank == 0:
            np.save(out_path, t.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h", [(2, 128, 100), (3, 96, 128), (2, 64, 40)])
def test_band_gather_reassembles_the_frame(O, tmp_path, world, w, h):
    import torch.multiprocessing as mp
    depth = 3
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, depth, out), nprocs=world, join=True)
    got = np.load(out)
    ref = np.full((h, w, 3), -1.0)
    O.render(O.OracleScene.create_default(), w, h, max_depth=depth, frame=ref)
    assert np.array_equal(got, ref)
    assert np.all(got[h - h % 32:] == -1.0) or h % 32 == 0     # rows below the last patch row untouched


def test_band_partition_properties():
    sys.path.insert(0, ROOT)
    import workloads
    for n_rows in (0, 1, 7, 33, 128, 135):
        for world in (1, 2, 3, 4, 8):
            bands = [workloads.patch_rows_for_rank(n_rows, r, world) for r in range(world)]
            assert bands[0][0] == 0 and bands[-1][1] == n_rows
            assert all(bands[i][1] == bands[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in bands]
            assert max(sizes) - min(sizes) <= 1
    # 8K has 135 patch rows -> 16 or 17 per rank on 8 GPUs
    assert sorted(set(e - b for b, e in (workloads.patch_rows_for_rank(135, r, 8) for r in range(8)))) == [16, 17]


def _worker_allgather(rank, world, port, w, h, depth, payload, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as G
    import workloads
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        O = G.load_oracle()
        n_rows = h // 32
        c, bands = workloads.equal_bands(n_rows, world)
        b, e = bands[rank]
        frame = np.zeros((max(h, world * c * 32), w, 3))
        if e > b:
            O.render(O.OracleScene.create_default(), w, h, max_depth=depth, n_threads=2, frame=frame[:h], band=(b, e))
        if payload == "u8":
            mine = torch.from_numpy(O.to_vec(frame.copy()).reshape(frame.shape))
            # only this rank's rows may be trusted before the gather
            padded = torch.full_like(mine, 77)
            padded[rank * c * 32:(rank + 1) * c * 32] = mine[rank * c * 32:(rank + 1) * c * 32]
        else:
            padded = torch.full(frame.shape, -5.0, dtype=torch.float64)
            padded[rank * c * 32:(rank + 1) * c * 32] = torch.from_numpy(frame)[rank * c * 32:(rank + 1) * c * 32]
        workloads.allgather_bands(dist, padded[:world * c * 32], rank, world)
        dist.barrier()
        if rank == world - 1:            # every rank ends up with the frame; check a non-root one
            np.save(out_path, padded.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h,payload", [(2, 128, 100, "u8"), (3, 96, 128, "f64"), (4, 64, 96, "u8")])
def test_equal_band_allgather(O, tmp_path, world, w, h, payload):
    """bench.py's N > 1 path: equal bands + one in-place all_gather_into_tensor."""
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    import workloads
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker_allgather, args=(world, _free_port(), w, h, 3, payload, out), nprocs=world, join=True)
    got = np.load(out)
    n_rows = h // 32
    ref = O.render(O.OracleScene.create_default(), w, h, max_depth=3)
    if payload == "u8":
        assert np.array_equal(got[:n_rows * 32].reshape(-1), O.to_vec(ref[:n_rows * 32].copy()))
    else:
        assert np.array_equal(got[:n_rows * 32], ref[:n_rows * 32])


def test_equal_bands_properties():
    sys.path.insert(0, ROOT)
    import workloads
    for P in (0, 1, 7, 33, 128, 135):
        for N in (1, 2, 3, 4, 8):
            c, bands = workloads.equal_bands(P, N)
            assert c * N >= P and (c - 1) * N < P or P == 0
            assert bands[0][0] == 0 and bands[-1][1] == P
            assert all(bands[i][1] == bands[i + 1][0] for i in range(N - 1))
            assert all(0 <= e - b <= c for b, e in bands)
            assert all(b == min(r * c, P) for r, (b, e) in enumerate(bands))
    assert workloads.equal_bands(33, 8)[1] == [(0, 5), (5, 10), (10, 15), (15, 20), (20, 25), (25, 30), (30, 33), (33, 33)]


def _worker_cyclic(rank, world, port, w, h, depth, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as G
    import workloads
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        O = G.load_oracle()
        n_rows = h // 32
        c, owned = workloads.cyclic_rows(n_rows, world)
        frame = np.zeros((h, w, 3))
        scene = O.OracleScene.create_default()
        for row in owned[rank]:
            O.render(scene, w, h, max_depth=depth, n_threads=1, frame=frame, band=(row, row + 1))
        # pack this rank's display rows into its chunk of the gather buffer
        gathered = torch.full((world * c * 32, w, 3), 55, dtype=torch.uint8)
        u8 = torch.from_numpy(O.to_vec(frame.copy()).reshape(h, w, 3))
        for k, row in enumerate(owned[rank]):
            gathered[(rank * c + k) * 32:(rank * c + k + 1) * 32] = u8[row * 32:(row + 1) * 32]
        work = dist.all_gather_into_tensor(gathered, gathered[rank * c * 32:(rank + 1) * c * 32], async_op=True)
        work.wait()
        display = workloads.deinterleave_rows(gathered, world, torch.zeros_like(gathered))
        dist.barrier()
        if rank == 0:
            np.save(out_path, display.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h", [(2, 96, 160), (3, 64, 128), (4, 64, 100)])
def test_cyclic_rows_allgather_and_deinterleave(O, tmp_path, world, w, h):
    """bench.py's default N > 1 path: cyclic row ownership, packed display bytes, one
    in-place (async) all-gather, de-interleave at the consumer."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "display.npy")
    mp.spawn(_worker_cyclic, args=(world, _free_port(), w, h, 3, out), nprocs=world, join=True)
    got = np.load(out)
    n_rows = h // 32
    ref = O.render(O.OracleScene.create_default(), w, h, max_depth=3)
    assert np.array_equal(got[:n_rows * 32].reshape(-1), O.to_vec(ref[:n_rows * 32].copy()))


def _worker_gather_chunks(rank, world, port, w, h, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as G
    import workloads
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        O = G.load_oracle()
        n_rows = h // 32
        c, owned = workloads.cyclic_rows(n_rows, world)
        full = np.zeros((h, w, 3))
        O.render(O.OracleScene.create_default(), w, h, max_depth=3, n_threads=2, frame=full)
        u8 = torch.from_numpy(O.to_vec(full[:n_rows * 32].copy()).reshape(n_rows * 32, w, 3))
        gathered = torch.full((world * c * 32, w, 3), 7, dtype=torch.uint8)
        for j, prow in enumerate(owned[rank]):                       # this rank's rows, packed into its chunk
            gathered[(rank * c + j) * 32:(rank * c + j + 1) * 32] = u8[prow * 32:(prow + 1) * 32]
        for req in workloads.gather_chunks(dist, gathered, rank, world):
            req.wait()
        dist.barrier()
        if rank == 0:
            image = workloads.deinterleave_rows(gathered, world, torch.zeros_like(gathered))
            np.save(out_path, np.stack([image[:n_rows * 32].numpy(), u8.numpy()]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h", [(2, 64, 100), (3, 96, 224), (4, 64, 40)])
def test_chunks_gathered_at_rank_0_reassemble_the_display_frame(tmp_path, world, w, h):
    """bench.py's default N > 1 exchange (workloads.gather_chunks: every peer sends its packed
    cyclic rows straight to rank 0, one grouped isend / irecv) + the de-interleave at the
    consumer: rank 0 ends up with the display frame in image order."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker_gather_chunks, args=(world, _free_port(), w, h, out), nprocs=world, join=True)
    got, want = np.load(out)
    assert np.array_equal(got, want)
