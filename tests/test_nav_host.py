"""SURVEY §8 f4 — bit sync, 20 ms nav-bit accumulation, preamble correlation and word parity of the legacy
src/decoding.rs: product (host code in the C-ABI library) vs the oracle's restatement, on a synthetic prompt-I stream.
Integer / flag work: every field compared exactly.  No GPU needed (host-side step)."""
import numpy as np
import pytest


def gps_word(d24, d29s, d30s):
    """IS-GPS-200 20.3.5.2 encoder in 0/1 form: source bits d1..d24 -> transmitted D1..D30."""
    d = [0] + list(d24)
    D = [0] * 31
    for i in range(1, 25):
        D[i] = d[i] ^ d30s
    D[25] = d29s ^ d[1] ^ d[2] ^ d[3] ^ d[5] ^ d[6] ^ d[10] ^ d[11] ^ d[12] ^ d[13] ^ d[14] ^ d[17] ^ d[18] ^ d[20] ^ d[23]
    D[26] = d30s ^ d[2] ^ d[3] ^ d[4] ^ d[6] ^ d[7] ^ d[11] ^ d[12] ^ d[13] ^ d[14] ^ d[15] ^ d[18] ^ d[19] ^ d[21] ^ d[24]
    D[27] = d29s ^ d[1] ^ d[3] ^ d[4] ^ d[5] ^ d[7] ^ d[8] ^ d[12] ^ d[13] ^ d[14] ^ d[15] ^ d[16] ^ d[19] ^ d[20] ^ d[22]
    D[28] = d30s ^ d[2] ^ d[4] ^ d[5] ^ d[6] ^ d[8] ^ d[9] ^ d[13] ^ d[14] ^ d[15] ^ d[16] ^ d[17] ^ d[20] ^ d[21] ^ d[23]
    D[29] = d30s ^ d[1] ^ d[3] ^ d[5] ^ d[6] ^ d[7] ^ d[9] ^ d[10] ^ d[14] ^ d[15] ^ d[16] ^ d[17] ^ d[18] ^ d[21] ^ d[22] ^ d[24]
    D[30] = d29s ^ d[3] ^ d[5] ^ d[6] ^ d[8] ^ d[9] ^ d[10] ^ d[11] ^ d[13] ^ d[15] ^ d[19] ^ d[22] ^ d[23] ^ d[24]
    return D[1:]


def _stream(edge_ms, polarity, n_bits, seed):
    """prompt I per 1 ms epoch: random data bits with the preamble embedded, bit edges at cnt % 20 == edge_ms."""
    rng = np.random.default_rng(seed)
    bits = list(rng.integers(0, 2, n_bits) * 2 - 1)
    for at in (70, 130, 180, 230):
        bits[at:at + 8] = [polarity * b for b in (1, -1, -1, -1, 1, -1, 1, 1)]
    cnt = np.arange(1, 20 * (n_bits - 1))
    which = (cnt - edge_ms) // 20 + 1
    ip = np.array([bits[w] for w in which], np.float32) * 900.0 + rng.standard_normal(cnt.size).astype(np.float32) * 150.0
    return cnt, ip, bits


@pytest.mark.parametrize("mode,edge", [(0, 0), (0, 7), (1, 0), (1, 7), (1, 19)])
def test_nav_sync_product_equals_oracle(gm, oracle, mode, edge):
    from gnss_sdr_rs_amd import decoding as Dm
    cnt, ip, bits = _stream(edge, -1 if edge == 7 else 1, 260, seed=edge + 3 * mode)
    prod, orc = Dm.NavSyncStatus(mode), oracle.NavSyncStatus(fixed=bool(mode))
    old = np.float32(0.0)
    emitted = 0
    for c, v in zip(cnt, ip):
        st = prod.update(float(old), float(v), int(c), buff_loc=int(c) * 8000)
        sw = orc.update(float(old), float(v), int(c), int(c) * 8000)
        o = orc.s
        assert (st["sync_sw"], st["flag_bit_sync"], st["flag_frame_sync"]) == (sw, o.flag_bit_sync, o.flag_frame_sync), c
        assert (st["frame_sync_ind"], st["n_frame_bits"], st["polarity"]) == (o.frame_sync_ind, o.n_frame_bits, o.polarity)
        assert np.float32(st["i_p"]) == np.float32(o.i_p)
        assert (st["sf_cnt"], st["sf_start_biti"], st["tow_expected_ind"]) == (o.sf_cnt, o.sf_start_biti, o.tow_expected_ind)
        if sw:
            assert st["bit"] == o.last_bit
            emitted += 1
        old = v
    assert (prod.histogram() == np.array(orc.s.bit_sync_buff[:], np.uint64)).all()
    assert (prod.frame_bits() == orc.frame_bits()).all()
    assert orc.s.flag_bit_sync == 1 and orc.s.frame_sync_ind == edge          # the histogram finds the bit edge
    if mode == 0 and edge != 0:
        assert emitted == 0                  # the reference's missing modulo (:203-205): no bit is ever completed
    else:
        assert emitted > 120
        fb = prod.frame_bits()
        # the emitted bits are the transmitted ones from the first complete bit after sync
        first = next(k for k in range(len(bits) - len(fb) + 1) if (np.array(bits[k:k + len(fb)]) == fb).all())
        assert first > 0
    if mode == 1:
        assert orc.s.flag_frame_sync == 1 and orc.s.polarity == (-1 if edge == 7 else 1)
    prod.close()


def test_parity_check_against_is_gps_200_encoder(gm, oracle):
    from gnss_sdr_rs_amd import decoding as Dm
    rng = np.random.default_rng(0)
    n_ref_disagrees = 0
    for trial in range(400):
        d29s, d30s = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        word = gps_word(rng.integers(0, 2, 24), d29s, d30s)
        # the reference multiplies the RECEIVED D1..D24 (not the source bits): that is the IS-GPS-200 equation only when
        # D30* = 0; symbols map 0 -> +1, 1 -> -1 so that XOR becomes a product
        sym = lambda b: 1 - 2 * b
        bits = np.array([sym(d29s), sym(d30s)] + [sym(b) for b in word], np.int8)
        ok, ref = Dm.parity_check(bits)
        assert (ok, ref) == oracle.nav_parity_check(bits)
        if d30s == 0:
            assert ok and ref
        k = int(rng.integers(2, 32))
        bad = bits.copy()
        bad[k] = -bad[k]
        okb, refb = Dm.parity_check(bad)
        assert (okb, refb) == oracle.nav_parity_check(bad)
        if d30s == 0:
            assert not okb and not refb
        k2 = (k - 2 + 1 + int(rng.integers(0, 29))) % 30 + 2
        bad[k2] = -bad[k2]                    # a second, different symbol
        okc, refc = Dm.parity_check(bad)
        assert (okc, refc) == oracle.nav_parity_check(bad)
        if d30s == 0:
            assert not okc
            n_ref_disagrees += int(refc)      # the reference's SUMMED criterion (:348-350) lets some double errors through
    assert n_ref_disagrees > 0
    with pytest.raises(Exception):
        Dm.parity_check(np.zeros(32, np.int8))


def test_update_many_equals_the_per_epoch_steps(gm):
    """gm_nav_sync_update_many = nav_decoding (decoding.rs:102-145) called once per epoch: same final status, same bits, and the
    epochs at which bit sync / frame sync first appear are the ones the per-epoch loop sees — whatever the block boundaries."""
    from gnss_sdr_rs_amd import decoding as Dm
    rng = np.random.default_rng(3)
    data = rng.integers(0, 2, 150) * 2 - 1
    for at in (12, 45, 80, 110):
        data[at:at + 8] = Dm.GPS_CA_PREAMBLE
    edge, n = 7, 2900
    ip = np.array([600.0 * data[((e - edge) // 20) % data.size] + 40.0 * rng.standard_normal() for e in range(n)], np.float32)
    one = Dm.NavSyncStatus(Dm.NAV_FIXED)
    old, first_bit, first_frame, st1 = 0.0, -1, -1, None
    for e in range(n):
        st1 = one.update(float(old), float(ip[e]), e)
        old = ip[e]
        if st1["flag_bit_sync"] and first_bit < 0:
            first_bit = e
        if st1["flag_frame_sync"] and first_frame < 0:
            first_frame = e
    assert first_bit > 0 and first_frame > first_bit
    many = Dm.NavSyncStatus(Dm.NAV_FIXED)
    old, fb, ff, st2, e = 0.0, -1, -1, None, 0
    for blk in (1, 17, 400, 3, 1000, 16, 16, 2000):
        blk = min(blk, n - e)
        if blk <= 0:
            break
        st2, b, f = many.update_many(old, ip[e:e + blk], e)
        if b >= 0 and fb < 0:
            fb = e + b
        if f >= 0 and ff < 0:
            ff = e + f
        old = float(ip[e + blk - 1])
        e += blk
    assert e == n and st2 == st1 and (fb, ff) == (first_bit, first_frame)
    assert (many.frame_bits() == one.frame_bits()).all() and (many.histogram() == one.histogram()).all()
