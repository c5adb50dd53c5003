"""CPU, world_size 2, gloo: the N > 1 path (PRN-sharded grid + ONE all-gather of {max, argmax, sum} + decision
replay on every rank) gives bit-for-bit the single-process result.  Metrics come from the oracle here (no GPU);
the exchange, regrouping and decision are the product's."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from gnss_sdr_rs_amd import distributed as Dm, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = O.ca_code_table()
    fs, N, M = 2.048e6, 2048, 2
    dop = np.array([-500.0, 0.0, 500.0, 1000.0], np.float32)
    sats = [dict(prn_row=2, cn0_dbhz=52.0, doppler_hz=480.0, code_start=100),
            dict(prn_row=6, cn0_dbhz=51.0, doppler_hz=-20.0, code_start=1999)]
    x = synth.to_c32(synth.make_scene(t, fs, 0.0, M * N, sats, config_id=31))
    all_prns = [1, 3, 5, 7, 9, 11] if world == 2 else [1, 3, 5, 7, 9, 11, 13, 15]     # equal blocks per rank (one all-gather of equal-sized blocks)
    mine = Dm.shard_prns(all_prns, world, rank)
    tables = [O.DopplerShiftTable(0.0, float(d), fs, N) for d in dop]
    tf = np.array([tb.doppler_freq_hz for tb in tables], np.float32)
    planes = []
    for p in mine:
        _, (bmax, barg, bsum, _) = O.AcquisitionWorker(p, N, fs).search_satellite(x, tables, 0, M, want_planes=True, no_early_exit=True)
        planes.append((bmax, barg, bsum))
    P, D = len(mine), dop.size
    block = Dm.pack_metrics(np.stack([a for a, _, _ in planes]), np.stack([b for _, b, _ in planes]), np.stack([c for _, _, c in planes]))
    g = Dm.all_gather_metrics(torch.from_numpy(block.copy()), world, P, D)
    res = Dm.decide_host(g.numpy(), all_prns, tf, N, fs, local_tail=5)
    single = [O.AcquisitionWorker(p, N, fs).search_satellite(x, tables, 5, M) for p in all_prns] if rank == 0 else None
    q.put((rank, res, single))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_equals_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    (_, r0, single), (_, r1, _) = got
    assert r0 == r1 == single
    assert [r["prn"] for r in r0 if r] == [3, 7]


def test_shard_prns_blocks():
    from gnss_sdr_rs_amd import distributed as Dm
    ids = list(range(1, 91))
    blocks = [Dm.shard_prns(ids, 8, r) for r in range(8)]
    assert sum(blocks, []) == ids and [len(b) for b in blocks] == [12, 12, 11, 11, 11, 11, 11, 11]
    # BASELINE configs[4]: 36 tracking channels over 1 / 2 / 4 / 8 ranks (bench.py cfg5_leg: contiguous blocks, no data-path collective)
    ch = list(range(36))
    for world, want in ((1, [36]), (2, [18, 18]), (4, [9, 9, 9, 9]), (8, [5, 5, 5, 5, 4, 4, 4, 4])):
        parts = [Dm.shard_prns(ch, world, r) for r in range(world)]
        assert sum(parts, []) == ch and [len(b) for b in parts] == want
    # more ranks than items: trailing ranks hold nothing
    assert [len(Dm.shard_prns([1, 2, 3], 8, r)) for r in range(8)] == [1, 1, 1, 0, 0, 0, 0, 0]


def _grid_families():
    """A small mixed grid: two code families with DIFFERENT transform sizes (like GPS N = 8000 next to Galileo N = 32000)."""
    import numpy as np
    from gnss_sdr_rs_amd import distributed as Dm
    rng = np.random.default_rng(5)
    dop = np.array([-500.0, 0.0, 500.0], np.float32)
    codes_b = np.where(rng.integers(0, 2, (4, 2046)) > 0, 1, -1).astype(np.int8)
    return [Dm.GridFamily("gps", 2.048e6, 0.0, 2048, 2, dop, [1, 3, 5, 7, 9]),
            Dm.GridFamily("b", 4.096e6, 0.0, 4096, 1, dop, [1, 2, 3, 4], codes=codes_b, code_rate=2.046e6)]


def _grid_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from gnss_sdr_rs_amd import distributed as Dm, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fams = _grid_families()
    t = O.ca_code_table()
    x_by = {}
    x_by[0] = synth.to_c32(synth.make_scene(t, fams[0].fs, 0.0, fams[0].M * fams[0].fft_size,
                                            [dict(prn_row=2, cn0_dbhz=52.0, doppler_hz=480.0, code_start=100)], config_id=33))
    x_by[1] = synth.to_c32(synth.make_scene(fams[1].codes, fams[1].fs, 0.0, fams[1].M * fams[1].fft_size,
                                            [dict(prn_row=3, cn0_dbhz=56.0, doppler_hz=-20.0, code_start=1999)], config_id=34,
                                            code_rate=fams[1].code_rate))

    def planes(fi, idx):
        f = fams[fi]
        tables = [O.DopplerShiftTable(f.f_if, float(d), f.fs, f.fft_size) for d in f.doppler_hz]
        w = O.AcquisitionWorker(f.prn_ids[idx], f.fft_size, f.fs, code=None if f.codes is None else f.codes[idx],
                                code_rate=f.code_rate)
        _, (bmax, barg, bsum, _) = w.search_satellite(x_by[fi], tables, 0, f.M, want_planes=True, no_early_exit=True)
        return bmax, barg, bsum
    D, pmax = 3, Dm.grid_pmax(fams, world)
    block = np.zeros((3, pmax, D), np.int32)
    row = 0
    for fi, first, cnt in Dm.shard_grid(fams, world, rank):
        for k in range(cnt):
            a, b, c = planes(fi, first + k)
            block[0, row] = a.view(np.int32); block[1, row] = b.view(np.int32); block[2, row] = c.view(np.int32)
            row += 1
    out = torch.empty(world * block.size, dtype=torch.int32)
    dist.all_gather_into_tensor(out, torch.from_numpy(block.ravel().copy()))
    res = Dm.grid_decide(Dm.grid_assemble(out.numpy(), fams, world, D), fams, local_tail=7)
    single = None
    if rank == 0:     # the same grid in one process
        one = np.zeros((1, 3, sum(f.n for f in fams), D), np.int32)
        row = 0
        for fi, f in enumerate(fams):
            for k in range(f.n):
                a, b, c = planes(fi, k)
                one[0, 0, row] = a.view(np.int32); one[0, 1, row] = b.view(np.int32); one[0, 2, row] = c.view(np.int32)
                row += 1
        single = Dm.grid_decide(Dm.grid_assemble(one, fams, 1, D), fams, local_tail=7)
    q.put((rank, res, single, Dm.shard_grid(fams, world, rank)))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_mixed_family_grid_equals_single_process():
    """configs[3]'s shape in small: 9 codes of two families with different fft sizes, sharded 5 + 4 — rank 0 holds only the
    first family, rank 1 the second — one all-gather of padded blocks, decision per family on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grid_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    (_, r0, single, s0), (_, r1, _, s1) = got
    assert r0 == r1 == single
    assert s0 == [(0, 0, 5)] and s1 == [(1, 0, 4)]
    assert [r["prn"] for r in r0["gps"] if r] == [3], r0["gps"]
    assert 4 in [r["prn"] for r in r0["b"] if r] and r0["b"][3]["code_phase_samples"] == 1999, r0["b"]


def test_world4_gloo_equals_single_process():
    """The same two paths at world size 4 (two codes per rank; the mixed grid in blocks of 3, 2, 2, 2 codes): every rank's
    decision equals the single-process one."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    for worker in (_worker, _grid_worker):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=worker, args=(r, 4, port, q)) for r in range(4)]
        for p in procs:
            p.start()
        got = [q.get(timeout=240) for _ in range(4)]
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        got.sort(key=lambda t: t[0])
        single = got[0][2]
        for g in got:
            assert g[1] == single, (worker.__name__, g[0])
        if worker is _grid_worker:
            shards = [g[3] for g in got]
            sizes = [sum(cnt for _, _, cnt in sh) for sh in shards]
            assert sum(sizes) == 9 and max(sizes) - min(sizes) <= 1, shards


def test_shard_grid_of_the_90_code_baseline_grid():
    from gnss_sdr_rs_amd import distributed as Dm
    import numpy as np
    dop = np.zeros(41, np.float32)
    fams = [Dm.GridFamily("gps", 8e6, 0, 8000, 10, dop, list(range(1, 33))),
            Dm.GridFamily("e1", 8e6, 0, 32000, 2, dop, list(range(1, 37)), codes=np.ones((36, 4092), np.int8)),
            Dm.GridFamily("b1i", 8e6, 0, 8000, 10, dop, list(range(1, 23)), codes=np.ones((22, 2046), np.int8), code_rate=2.046e6)]
    shards = [Dm.shard_grid(fams, 8, r) for r in range(8)]
    assert [sum(c for _, _, c in s) for s in shards] == [12, 12, 11, 11, 11, 11, 11, 11]
    assert shards[2] == [(0, 24, 8), (1, 0, 3)]          # rank 2 straddles GPS and Galileo: two transform sizes on one rank
    assert Dm.grid_pmax(fams, 8) == 12
    covered = {(fi, first + k) for s in shards for fi, first, c in s for k in range(c)}
    assert len(covered) == 90
