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
