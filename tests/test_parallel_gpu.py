"""The RCCL code paths on the GPU box.  A box of the pool has ONE GPU, so the process group here has one rank: what this covers
is that `init_process_group("nccl", device_id=...)`, the barrier, the MAX / SUM reductions of bench.py and the AVG all-reduce of
the training step's gradient buckets are accepted and executed by RCCL on device memory (the 8-GPU run itself is the driver's)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, torch
sys.path.insert(0, os.environ["MUDG_ROOT"])
from mudg_amd import parallel
from mudg_amd.train.step import GradientAllReducer
rank, world, local, dist = parallel.init_from_env("nccl", force=True)
assert dist is not None and dist.get_backend() == "nccl" and world == 1
dev = torch.device("cuda", local)
parallel.barrier(dist)
assert parallel.max_over_ranks(2.5, dist, dev) == 2.5 and parallel.sum_over_ranks(3.0, dist, dev) == 3.0
params = [torch.nn.Parameter(torch.zeros(n, device=dev)) for n in (1000, 300000, 17)]
for i, p in enumerate(params):
    p.grad = torch.full_like(p, float(i + 1))
red = GradientAllReducer(params, bucket_mb=1, always=True)
assert red() == len(red.buckets) >= 2                      # one-rank group: the average of one value is the value
assert all(torch.equal(p.grad, torch.full_like(p, float(i + 1))) for i, p in enumerate(params))
# the same collectives launched from gradient hooks, under a backward pass on the GPU (RCCL on its own stream beside the kernels)
net = torch.nn.Sequential(torch.nn.Linear(512, 512), torch.nn.Tanh(), torch.nn.Linear(512, 512)).to(dev)
red = GradientAllReducer(list(net.parameters()), bucket_mb=0.5, always=True, overlap=True)
x = torch.randn(64, 512, device=dev)
net(x).square().mean().backward()
launched = red._next
want = [p.grad.clone() for p in net.parameters()]
assert red() == len(red.buckets) >= 2 and launched == len(red.buckets)     # every bucket went out before backward() returned
assert all(torch.allclose(p.grad, w) for p, w in zip(net.parameters(), want))
dist.destroy_process_group()
print("RCCL-OK")
"""


def test_rccl_process_group_barrier_reductions_and_gradient_buckets(cuda):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MUDG_ROOT=ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
