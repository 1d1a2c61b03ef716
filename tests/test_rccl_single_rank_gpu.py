"""The RCCL backend, executed once: a one-rank `torch.distributed` process group with backend "nccl" (= RCCL on ROCm) on the GPU box's single MI355X
runs the collectives the multi-GPU path uses -- `ResultGatherer.gather` (gather to rank 0 and all_gather into the preallocated receive buffer),
`barrier`, the MAX all-reduce of `max_over_ranks` -- on device tensors of the result-map shape.  No 8-GPU node was available in any round, so until
the driver's scaling run this is the only evidence that the RCCL side of `genpercept_amd/distributed.py` initialises and launches on this software
stack (HSA_ENABLE_IPC_MODE_LEGACY=0, device_id binding); the N > 1 logic itself (shards, padding, order) is covered by the world-size-2 gloo tests
on CPU (tests/test_host.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys, torch
sys.path.insert(0, {root!r})
import torch.distributed as dist
from genpercept_amd import distributed as gd
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
d = torch.device("cuda", 0)
local = torch.arange(4 * 1 * 96 * 128, dtype=torch.float32, device=d).reshape(4, 1, 96, 128)
for dst in (0, None):
    g = gd.ResultGatherer(local, 4, dst)          # (gather_results() short-cuts a world of one; the class does not)
    out = g.gather(local)
    torch.cuda.synchronize()
    assert out.shape == local.shape and torch.equal(out, local) and out.data_ptr() == g.recv.data_ptr()
    out2 = g.gather(local + 1.0)                  # buffers reused
    torch.cuda.synchronize()
    assert out2.data_ptr() == out.data_ptr() and torch.equal(out2, local + 1.0)
dist.barrier()
t = torch.tensor([3.25], dtype=torch.float64, device=d)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 3.25
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK", flush=True)
"""


def test_rccl_process_group_with_one_rank(tmp_path):
    script = tmp_path / "rccl_worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 1000), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]
