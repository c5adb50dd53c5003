"""BASELINE configs[3] AS WRITTEN, on the GPU: the 90-code GPS + Galileo-E1-geometry + BeiDou B1I acquisition grid at its
full sizes (32 x N=8000 x 10 ms, 36 x N=32000 x 2 x 4 ms, 22 x N=8000 x 10 ms, 41 Doppler bins, one 10 ms snapshot at 8 Msps
int8) through MixedGrid.search_dev -> the exchange block -> assemble -> decision, for world = 1 and for every one of the
eight rank shards in turn on the one GPU (ranks are independent given the snapshot: the reference fans the PRNs out the same
way, do_acquisition.rs:302-313).  Checked:
  (a) the 8-shard grid equals the 1-shard grid word for word, through the host assemble + decision and through the
      device one (gm_grid_assemble_dev + gm_acq_decide_planes_dev), on a stream of the grid's own and on a stream of the caller's;
  (b) per-(code, bin) {max, first argmax, sum} equal the generalised oracle's for codes of every family, including both
      halves of the rank whose block straddles two transform sizes (rank 2 of 8: GPS 25-32 + E1 1-3);
  (c) the simulated satellites are found at their code phase.
Multi-GPU hardware is not available to this suite: what is NOT covered here is RCCL itself (tests/test_distributed_gloo.py
covers the N > 1 exchange logic on gloo)."""
import numpy as np
import pytest
import torch   # before the HIP library: torch brings its own HIP runtime, which must be the first one the process loads

pytestmark = pytest.mark.gpu
REL = 1e-5


@pytest.fixture(scope="module")
def setup(gpu):
    import torch
    from gnss_sdr_rs_amd import acquisition as A, distributed as Dm, synth
    ca = A.ca_code_table()
    b1i = A.b1i_codes(range(1, 23))
    sc = synth.cfg4_grid_scene(ca, b1i)
    fams = Dm.baseline_grid_families(sc, b1i)
    assert [f.n for f in fams] == [32, 36, 22] and [f.fft_size for f in fams] == [8000, 32000, 8000]
    xi8 = synth.to_i8_iq(sc["x"])
    d_x = torch.from_numpy(xi8).cuda()
    return dict(sc=sc, fams=fams, xi8=xi8, d_x=d_x, ca=ca, b1i=b1i)


def _gather_world(fams, world, d_x, stream=None):
    """run every rank's shard in turn on this GPU -> (gathered int32 tensor [world][3][pmax][D], rank 0's MixedGrid)"""
    import torch
    from gnss_sdr_rs_amd import acquisition as A, distributed as Dm
    ptrs = {0: d_x.data_ptr(), 1: d_x.data_ptr(), 2: d_x.data_ptr()}
    blocks, keep = [], None
    for r in range(world):
        g = Dm.MixedGrid(fams, world, r, stream=stream)
        blk = g.search_dev(ptrs, A.FMT_I8_IQ)
        torch.cuda.synchronize()
        blocks.append(blk.clone())
        if r == 0:
            keep = g
        else:
            g.close()
    return torch.cat(blocks), keep


def test_grid_8_shards_equal_1_shard_and_device_decision_equals_host(setup):
    import torch
    from gnss_sdr_rs_amd import distributed as Dm
    fams, d_x, D = setup["fams"], setup["d_x"], setup["sc"]["D"]
    g1, grid1 = _gather_world(fams, 1, d_x)
    g8, grid8 = _gather_world(fams, 8, d_x)
    assert g1.numel() == 3 * 90 * D and g8.numel() == 8 * 3 * 12 * D
    a1 = Dm.grid_assemble(g1.cpu().numpy(), fams, 1, D)
    a8 = Dm.grid_assemble(g8.cpu().numpy(), fams, 8, D)
    for fi in range(3):
        assert (a1[fi] == a8[fi]).all(), fams[fi].name                       # (a) word for word
    host1, host8 = Dm.grid_decide(a1, fams, local_tail=1234), Dm.grid_decide(a8, fams, local_tail=1234)
    assert host1 == host8
    # the device decision on the gathered blocks, world = 8 (rank 0's grid) and world = 1
    grid8.decide_dev(g8, local_tail=1234)
    dev8, planes8 = grid8.fetch()
    grid1.decide_dev(g1, local_tail=1234)
    dev1, planes1 = grid1.fetch()
    assert dev8 == host8 and dev1 == host8
    for fi in range(3):
        assert (planes8[fi] == a8[fi]).all() and (planes1[fi] == a1[fi]).all()
    # (c) every simulated satellite sits at its code phase on the strongest bin of its plane, within a bin of its Doppler
    sc = setup["sc"]
    for fi, f in enumerate(fams):
        for prn, cp in sc["truth"][f.name].items():
            mx, am = a8[fi][0][prn - 1].view(np.float32), a8[fi][1][prn - 1].view(np.uint32)
            b = int(np.argmax(mx))
            assert int(am[b]) == cp, (f.name, prn)
            assert abs(float(f.doppler_hz[b]) - sc["truth_doppler"][f.name][prn]) <= 250.0
    # GPS and B1I (10 integrations): the reference's early-exit decision itself reports them
    for fam in ("gps", "b1i"):
        found = {r["prn"]: r["code_phase_samples"] for r in host8[fam] if r}
        assert found == sc["truth"][fam], (fam, found)
        for r in host8[fam]:
            if r:
                assert r["sample_global_index"] == 1234 + r["code_phase_samples"]
    grid1.close(); grid8.close()
    # the same grid on a stream of the caller's (search, block copy and decision all ordered on it); above: the grid's own
    side = torch.cuda.Stream()
    gs, grids = _gather_world(fams, 8, d_x, stream=side.cuda_stream)
    assert (gs.cpu() == g8.cpu()).all()
    grids.decide_dev(gs, local_tail=1234)
    devs, _ = grids.fetch()
    assert devs == host8
    grids.close()


def test_grid_metrics_equal_generalised_oracle(setup, oracle):
    """(b): rank 2 of 8 holds GPS codes 25-32 and E1 codes 1-3 (two transform sizes on one rank): rows of both halves, the
    simulated satellites' rows and one more row per family are compared with the oracle over all 41 bins."""
    from gnss_sdr_rs_amd import distributed as Dm
    fams, d_x, sc, xi8 = setup["fams"], setup["d_x"], setup["sc"], setup["xi8"]
    D = sc["D"]
    assert Dm.shard_grid(fams, 8, 2) == [(0, 24, 8), (1, 0, 3)]
    g8, grid8 = _gather_world(fams, 8, d_x)
    grid8.close()
    a8 = Dm.grid_assemble(g8.cpu().numpy(), fams, 8, D)
    xc = (xi8[:, 0] + 1j * xi8[:, 1]).astype(np.complex64)
    rows = {0: [0, 4, 24, 31], 1: [0, 2, 6, 35], 2: [2, 21]}
    for fi, f in enumerate(fams):
        tables = [oracle.DopplerShiftTable(0.0, float(d), f.fs, f.fft_size) for d in f.doppler_hz]
        for w in rows[fi]:
            ow = oracle.AcquisitionWorker(f.prn_ids[w], f.fft_size, f.fs, code=(None if f.codes is None else f.codes[w]),
                                          code_rate=f.code_rate)
            _, (bmax, barg, bsum, done) = ow.search_satellite(xc, tables, 0, f.M, want_planes=True, no_early_exit=True)
            assert done == D
            mx, am, sm = a8[fi][0][w].view(np.float32), a8[fi][1][w].view(np.uint32), a8[fi][2][w].view(np.float32)
            assert np.allclose(mx, bmax, rtol=REL, atol=0.0), (f.name, w, float(np.max(np.abs(mx - bmax) / bmax)))
            assert np.allclose(sm, bsum, rtol=REL, atol=0.0), (f.name, w)
            assert (am == barg).all(), (f.name, w, np.nonzero(am != barg)[0])
