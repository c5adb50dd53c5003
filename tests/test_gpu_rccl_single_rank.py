"""torch.distributed's `nccl` backend (= RCCL on ROCm) on the GPU box, ONE rank: the carrier bench.py's N > 1 path uses by default has no
multi-GPU box on this pool, so the world-2 / world-4 tests run over gloo on the CPU — this one at least proves that the backend
loads, creates a communicator on cuda:0 and runs, on a non-default current stream, the very calls the bench makes (all_gather_into_tensor
synchronous and async_op=True, all_reduce MIN / MAX, barrier) with the metric block's shape.  In a child process: a process group
is global state."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import os, sys
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", world_size=1, rank=0, device_id=dev)
torch.cuda.set_device(dev)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
P, D = 32, 41
met = torch.arange(3 * P * D, dtype=torch.int32, device=dev)           # {max, argmax, sum}[P][D] as 32-bit words
out = torch.zeros(3 * P * D, dtype=torch.int32, device=dev)
dist.all_gather_into_tensor(out, met)
assert torch.equal(out, met)
out.zero_()
w = dist.all_gather_into_tensor(out, met, async_op=True)                # the timed loop's one exchange step
w.wait()
assert torch.equal(out, met)
flag = torch.tensor([1], dtype=torch.int32, device=dev)
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
tt = torch.tensor([0.25], dtype=torch.float64, device=dev)
dist.all_reduce(tt, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert int(flag.item()) == 1 and float(tt.item()) == 0.25
dist.destroy_process_group()
print("rccl single rank ok")
"""


def test_torch_nccl_backend_runs_the_benchs_collectives_on_one_rank(gpu):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", _CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0 and "rccl single rank ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
