"""Multi-GPU sharding of the PRN x Doppler grid: contiguous PRN blocks per rank, ONE
all-gather of the per-(PRN, bin) {max, argmax, sum} metrics (RCCL over xGMI on GPUs, `nccl` backend;
gloo on CPU for tests), then every rank replays the reference's decision on the gathered grid.
torch.distributed is plumbing here; the metrics block is 3*P*D 32-bit words per rank (15.7 KB at P=32, D=41)."""
import ctypes as C

import numpy as np

from ._lib import AcqResult, check, lib


def shard_prns(prn_ids, world, rank):
    """Contiguous PRN-major blocks, sizes differing by at most one (90 PRNs on 8 ranks -> 12,12,11,...)."""
    n = len(prn_ids)
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return list(prn_ids[start:start + base + (1 if rank < extra else 0)])


def pack_metrics(mx, am, sm):
    """[P][D] planes -> one int32 block [3][P][D] (bit patterns preserved)."""
    return np.concatenate([np.ascontiguousarray(mx, np.float32).view(np.int32).ravel(),
                           np.ascontiguousarray(am, np.uint32).view(np.int32).ravel(),
                           np.ascontiguousarray(sm, np.float32).view(np.int32).ravel()])


def regroup_gathered(gathered, world, P, D):
    """torch tensor [world][3][P][D] (as all_gather_into_tensor lays it out) -> [3][world*P][D]."""
    return gathered.view(world, 3, P * D).permute(1, 0, 2).contiguous()


def all_gather_metrics(local_block, world, P, D):
    """local_block: torch int32 tensor of 3*P*D words on this rank's device.  Equal P on every rank
    (pad the last block).  Returns the regrouped [3][world*P][D] int32 tensor."""
    import torch
    import torch.distributed as dist
    out = torch.empty(world * local_block.numel(), dtype=local_block.dtype, device=local_block.device)
    dist.all_gather_into_tensor(out, local_block)
    return regroup_gathered(out, world, P, D)


COMM_ID_BYTES = 128


class NativeComm:
    """gm_comm_* — the C-ABI exchange a host without PyTorch (the reference is Rust) uses: RCCL all-gather of the
    metrics block on the acquisition handle's stream + regroup, no torch types.  The 128-byte id is created on
    rank 0 and carried to the other ranks by whatever channel the host has; `from_torch_dist` uses a
    torch.distributed broadcast for that (plumbing only)."""

    def __init__(self, nranks, rank, unique_id):
        assert len(unique_id) == COMM_ID_BYTES
        self._h = C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        check(lib().gm_comm_init(nranks, rank, C.cast(buf, C.c_void_p), C.byref(self._h)), "gm_comm_init")
        self.nranks, self.rank = nranks, rank

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        check(lib().gm_comm_get_unique_id(C.cast(buf, C.c_void_p)), "gm_comm_get_unique_id")
        return bytes(buf)

    @classmethod
    def from_torch_dist(cls):
        import torch.distributed as dist
        box = [cls.unique_id() if dist.get_rank() == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(dist.get_world_size(), dist.get_rank(), box[0])

    def allgather_metrics(self, engine, d_all_ptr, d_local_ptr=None):
        """engine: AcquisitionEngine; d_all_ptr: device pointer to 3*nranks*P*D words; asynchronous on the handle's stream."""
        check(lib().gm_acq_allgather_metrics(engine._h, self._h, d_local_ptr, d_all_ptr), "gm_acq_allgather_metrics")

    def allgather_metrics_async(self, engine, d_all_ptr, d_local_ptr=None):
        """the same exchange on the communicator's own stream (overlaps the next dwell); wait(stream) before d_all is read"""
        check(lib().gm_acq_allgather_metrics_async(engine._h, self._h, d_local_ptr, d_all_ptr), "gm_acq_allgather_metrics_async")

    def wait(self, stream_ptr):
        check(lib().gm_comm_wait(self._h, C.c_void_p(stream_ptr)), "gm_comm_wait")

    def allgather_words(self, d_local_ptr, d_all_ptr, words, stream_ptr):
        """raw all-gather of `words` 32-bit words per rank (the mixed grid's padded blocks), asynchronous on stream_ptr"""
        check(lib().gm_comm_allgather_words(self._h, d_local_ptr, d_all_ptr, int(words), C.c_void_p(stream_ptr)),
              "gm_comm_allgather_words")

    def close(self):
        if self._h:
            lib().gm_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decide_host(block, prn_ids, table_freq, fft_size, fs, local_tail=0, code_rate=1.023e6, threshold=7.0):
    """Decision replay (do_acquisition.rs:195-238) on a host int32 block [3][P][D]."""
    P = len(prn_ids)
    b = np.ascontiguousarray(block, np.int32).reshape(3, P, -1)
    D = b.shape[2]
    mx, am, sm = b[0].view(np.float32).copy(), b[1].view(np.uint32).copy(), b[2].view(np.float32).copy()
    tf = np.ascontiguousarray(table_freq, np.float32)
    ids = np.ascontiguousarray(prn_ids, np.uint8)
    res = (AcqResult * P)()
    found = np.zeros(P, np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib().gm_acq_decide_host(p(mx), p(am), p(sm), p(tf), P, D, p(ids), fft_size, fs, code_rate, threshold,
                                   int(local_tail), C.cast(res, C.c_void_p), p(found)), "gm_acq_decide_host")
    return [res[i].as_dict() if found[i] else None for i in range(P)]


# ---------------------------------------------------------------------------------------------- mixed-constellation grid
# BASELINE configs[3]: "GPS + Galileo E1 + BeiDou B1I ~90-PRN acquisition grid sharded 8 x MI355X, RCCL all-gather of
# peaks".  The codes of all families form ONE list (family-major); ranks take contiguous blocks of it whose sizes differ by at
# most one (shard_prns: 90 codes on 8 ranks -> 12, 12, 11, ...), so a rank may hold pieces of two families with different
# transform sizes.  Every rank searches its codes, ONE all-gather of the padded [3][P_max][D] metrics blocks, and every rank
# replays the reference's decision per family on the gathered grid.
class GridFamily:
    def __init__(self, name, fs, f_if, fft_size, n_integrations, doppler_hz, prn_ids, codes=None, code_rate=1.023e6):
        self.name, self.fs, self.f_if, self.fft_size, self.M = name, float(fs), float(f_if), int(fft_size), int(n_integrations)
        self.doppler_hz = np.ascontiguousarray(doppler_hz, np.float32)
        self.prn_ids = list(prn_ids)
        self.codes = None if codes is None else np.ascontiguousarray(codes, np.int8)
        self.code_rate = float(code_rate)
        assert self.codes is None or self.codes.shape[0] == len(self.prn_ids)

    @property
    def n(self):
        return len(self.prn_ids)

    def table_freq(self):
        """DopplerShiftTable.doppler_freq_hz = f_if + doppler (doppler_shift.rs:11,20), f32 like the reference"""
        return np.array([np.float32(self.f_if) + np.float32(d) for d in self.doppler_hz], np.float32)


def shard_grid(families, world, rank):
    """-> [(family index, first local code, count)] of this rank's contiguous block of the family-major code list."""
    total = sum(f.n for f in families)
    mine = shard_prns(list(range(total)), world, rank)
    out, base = [], 0
    for fi, f in enumerate(families):
        loc = [g - base for g in mine if base <= g < base + f.n]
        if loc:
            out.append((fi, loc[0], len(loc)))
        base += f.n
    return out


def grid_pmax(families, world):
    total = sum(f.n for f in families)
    return (total + world - 1) // world


def grid_assemble(gathered, families, world, D):
    """gathered: int32 array [world][3][P_max][D] (one padded block per rank) -> {family index: int32 [3][n_family][D]}."""
    pmax = grid_pmax(families, world)
    g = np.ascontiguousarray(gathered, np.int32).reshape(world, 3, pmax, D)
    out = {fi: np.zeros((3, f.n, D), np.int32) for fi, f in enumerate(families)}
    for r in range(world):
        row = 0
        for fi, first, cnt in shard_grid(families, world, r):
            out[fi][:, first:first + cnt, :] = g[r][:, row:row + cnt, :]
            row += cnt
    return out


def grid_decide(assembled, families, local_tail=0, threshold=7.0):
    """The reference's decision (do_acquisition.rs:195-238) per family on the assembled grid -> {family: [result | None]}."""
    return {f.name: decide_host(assembled[fi], f.prn_ids, f.table_freq(), f.fft_size, f.fs, local_tail, f.code_rate, threshold)
            for fi, f in enumerate(families)}


class MixedGrid:
    """This rank's engines for its block of the grid (GPU).  search_dev() fills the padded local block; the exchange is the
    caller's (torch.distributed, or NativeComm.allgather_words: gm_comm); decide_dev() assembles the gathered blocks into the
    family-major grid and replays the reference's decision per family ON THE DEVICE (gm_grid_assemble_dev +
    gm_acq_decide_planes_dev); fetch() is the one device-to-host copy.  grid_assemble + grid_decide remain as the host form.

    Every engine, the block copy and the decision run on ONE explicit stream: `stream` (a non-zero HIP stream handle, e.g. a
    torch.cuda.Stream's .cuda_stream), or a stream of the grid's own when none is given (never the NULL stream: its implicit
    ordering with torch's copies proved unreliable for the first dwell of a multi-engine grid, tools/grid_probe2.py).
    search_dev() makes the caller's current torch stream wait for the block, decide_dev() makes the grid's stream wait for
    the caller's (the all-gather that produced `gathered`), so the hand-offs are ordered whatever stream the caller uses."""

    def __init__(self, families, world, rank, stream=None, threshold=7.0, decision_modes=None):
        import torch
        from . import acquisition as A
        self.families, self.world, self.rank = families, world, rank
        self.D = int(families[0].doppler_hz.size)
        assert all(f.doppler_hz.size == self.D for f in families), "one Doppler grid for the whole exchange block"
        self.pmax = grid_pmax(families, world)
        self.threshold = float(threshold)
        self.decision_modes = list(decision_modes) if decision_modes is not None else [0] * len(families)
        self._ext = torch.cuda.ExternalStream(int(stream)) if stream else torch.cuda.Stream()
        self.stream = int(self._ext.cuda_stream)
        self.parts = []
        row = 0
        for fi, first, cnt in shard_grid(families, world, rank):
            f = families[fi]
            eng = A.AcquisitionEngine(f.fs, f.f_if, f.fft_size, doppler_hz=f.doppler_hz, prn_ids=f.prn_ids[first:first + cnt],
                                      n_integrations=f.M, codes=None if f.codes is None else f.codes[first:first + cnt],
                                      code_rate=f.code_rate)
            eng.set_stream(self.stream)
            met = torch.zeros(3 * cnt * self.D, dtype=torch.int32, device="cuda")
            self.parts.append(dict(fi=fi, eng=eng, met=met, row=row, cnt=cnt))
            row += cnt
        self.block = torch.zeros(3 * self.pmax * self.D, dtype=torch.int32, device="cuda")
        # device-side decision state: where each code of the family-major list sits in the gathered blocks
        self.total = sum(f.n for f in families)
        rmap = np.zeros(self.total, np.uint32)
        base = [0]
        for f in families:
            base.append(base[-1] + f.n)
        self.base = base
        for r in range(world):
            row = 0
            for fi, first, cnt in shard_grid(families, world, r):
                rmap[base[fi] + first:base[fi] + first + cnt] = r * self.pmax + row + np.arange(cnt, dtype=np.uint32)
                row += cnt
        self.d_row_map = torch.from_numpy(rmap.view(np.int32)).cuda()
        self.d_grid = torch.zeros(3 * self.total * self.D, dtype=torch.int32, device="cuda")
        self.d_ids = [torch.from_numpy(np.ascontiguousarray(f.prn_ids, np.uint8)).cuda() for f in families]
        self.d_tf = [torch.from_numpy(f.table_freq()).cuda() for f in families]
        self.d_res = torch.zeros(self.total * C.sizeof(AcqResult), dtype=torch.uint8, device="cuda")
        self.d_found = torch.zeros(self.total, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()

    def set_stream(self, stream_ptr):
        """move the whole grid to another (non-zero) HIP stream handle"""
        import torch
        assert stream_ptr, "an explicit stream: the NULL stream is not accepted"
        self._ext.synchronize()
        self._ext = torch.cuda.ExternalStream(int(stream_ptr))
        self.stream = int(stream_ptr)
        for p in self.parts:
            p["eng"].set_stream(self.stream)

    def search_dev(self, samples_ptr_by_family, fmt):
        """samples_ptr_by_family: {family index: device pointer to that family's n_integrations * fft_size samples}"""
        import torch
        b = self.block.view(3, self.pmax, self.D)
        caller = torch.cuda.current_stream()
        self._ext.wait_stream(caller)                # the samples (and whoever last read the block) on the caller's stream
        with torch.cuda.stream(self._ext):          # the copies follow the searches on the engines' stream
            for p in self.parts:
                p["eng"].search_dev(samples_ptr_by_family[p["fi"]], fmt, p["met"].data_ptr())
                b[:, p["row"]:p["row"] + p["cnt"], :] = p["met"].view(3, p["cnt"], self.D)
        caller.wait_stream(self._ext)                # the block is complete for whatever the caller enqueues next
        return self.block

    def decide_dev(self, gathered, local_tail=0):
        """gathered: int32 device tensor [world][3][P_max][D] (this rank's own block when world == 1).  Asynchronous."""
        import torch
        L = lib()
        self._ext.wait_stream(torch.cuda.current_stream())      # `gathered` was produced on the caller's stream
        check(L.gm_grid_assemble_dev(gathered.data_ptr(), self.world, self.pmax, self.D, self.d_row_map.data_ptr(), self.total,
                                     self.d_grid.data_ptr(), self.stream), "gm_grid_assemble_dev")
        g, TD = self.d_grid.data_ptr(), self.total * self.D
        for fi, f in enumerate(self.families):
            o = self.base[fi] * self.D * 4
            check(L.gm_acq_decide_planes_dev(g + o, g + TD * 4 + o, g + 2 * TD * 4 + o, f.n, self.D, self.d_ids[fi].data_ptr(),
                                             self.d_tf[fi].data_ptr(), f.fft_size, f.fs, f.code_rate, self.threshold,
                                             self.decision_modes[fi], int(local_tail),
                                             self.d_res.data_ptr() + self.base[fi] * C.sizeof(AcqResult),
                                             self.d_found.data_ptr() + self.base[fi], self.stream), "gm_acq_decide_planes_dev")

    def fetch(self):
        """-> ({family name: [result | None]}, {family index: int32 [3][n_family][D]}) of the last decide_dev (synchronises)."""
        import torch
        with torch.cuda.stream(self._ext):
            raw, found, grid = self.d_res.cpu().numpy(), self.d_found.cpu().numpy(), self.d_grid.cpu().numpy()
        res = (AcqResult * self.total).from_buffer_copy(raw.tobytes())
        g = grid.reshape(3, self.total, self.D)
        out, planes = {}, {}
        for fi, f in enumerate(self.families):
            b = self.base[fi]
            out[f.name] = [res[b + i].as_dict() if found[b + i] else None for i in range(f.n)]
            planes[fi] = g[:, b:b + f.n, :].copy()
        return out, planes

    def cells(self):
        return sum(f.n * self.D * f.fft_size for f in self.families)

    def close(self):
        for p in self.parts:
            p["eng"].close()
        self.parts = []


def baseline_grid_families(scene, b1i_codes):
    """The GridFamily list of BASELINE configs[3] at its full sizes, for a synth.cfg4_grid_scene: 32 + 36 + 22 = 90 codes."""
    fs, dop = scene["fs"], scene["doppler_hz"]
    return [GridFamily("gps", fs, 0.0, 8000, 10, dop, list(range(1, 33))),
            GridFamily("e1", fs, 0.0, 32000, 2, dop, list(range(1, 37)), codes=scene["e1"], code_rate=1.023e6),
            GridFamily("b1i", fs, 0.0, 8000, 10, dop, list(range(1, 23)), codes=b1i_codes, code_rate=2.046e6)]
