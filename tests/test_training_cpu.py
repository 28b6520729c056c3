"""Host logic of the training step that needs no GPU: the data-parallel gradient all-reduce (bucket layout, averaging, parameters
without gradients) over two gloo ranks — the N > 1 path of SURVEY §8 f4 (one process per GPU, gradients averaged after
backward: main/utils_train.py:126-137), and the boundary methods that route into mudg_amd.train."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mudg_amd.train.step import GradientAllReducer
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (5, 300000, 7, 120000, 3)]
    frozen = torch.nn.Parameter(torch.zeros(4), requires_grad=False)
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    params[2].grad = None if rank == 0 else params[2].grad          # a parameter one rank did not touch
    red = GradientAllReducer(params + [frozen], bucket_mb=1)         # 1 MiB buckets: 262144 floats -> several buckets
    n = red()
    ok = n == len(red.buckets) and len(red.buckets) >= 2 and frozen.grad is None
    for i, p in enumerate(params):
        want = (1 + 2) / 2 * (i + 1) if i != 2 else (0 + 2 * 3) / 2
        ok = ok and torch.allclose(p.grad, torch.full_like(p, want))
    q.put((rank, bool(ok), [len(b) for b in red.buckets]))
    dist.destroy_process_group()


def _worker_overlap(rank, world, port, q):
    """The collectives launched from post-accumulate-grad hooks, under a real backward pass, must give what the after-backward
    reducer gives; a micro-batch with .sync = False must leave the local gradients alone."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mudg_amd.train.step import GradientAllReducer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(300, 400), torch.nn.Tanh(), torch.nn.Linear(400, 300), torch.nn.Tanh(), torch.nn.Linear(300, 5))
    unused = torch.nn.Parameter(torch.ones(7))                       # never reaches the loss: its bucket (the last one) completes only at the end
    params = [unused] + list(net.parameters())
    red = GradientAllReducer(params, bucket_mb=0.25, overlap=True)    # 65536 floats per bucket: several buckets
    launched, snap = [], {}
    orig = red._launch

    def launch(i):                                                   # the bucket as this rank filled it, before the collective averages it in place
        launched.append((i, red._next))
        snap[i] = red.flats[i].clone()
        return orig(i)
    red._launch = launch
    x = torch.randn(16, 300, generator=torch.Generator().manual_seed(10 + rank))
    # micro-batch 1 of 2: local accumulation only
    red.sync = False
    net(x).square().mean().backward()
    local = [p.grad.clone() for p in net.parameters()]
    ok = red() == 0 and not launched
    ok = ok and all(p.grad.data_ptr() == red._views[id(p)].data_ptr() for p in params)       # gradients live in the buckets
    # micro-batch 2 of 2: the hooks launch the collectives while backward is still running
    red.sync = True
    net(2 * x).square().mean().backward()
    under_backward = len(launched)
    n = red()
    ok = ok and n == len(red.buckets) >= 3 and under_backward >= 1 and [i for i, _ in launched] == list(range(n))
    # reference: gather every rank's local bucket (both micro-batches accumulated) and average
    for i in range(n):
        both = [torch.zeros_like(snap[i]) for _ in range(world)]
        dist.all_gather(both, snap[i])
        ok = ok and torch.allclose(red.flats[i], sum(both) / world, rtol=1e-6, atol=1e-7)
    ok = ok and unused.grad is not None and float(unused.grad.abs().sum()) == 0.0
    ok = ok and all(not torch.equal(a, p.grad) for a, p in zip(local, net.parameters()))
    # zero_grad keeps the views and clears them with one fill per bucket
    red.zero_grad()
    ok = ok and all(float(f.abs().sum()) == 0.0 for f in red.flats) and all(p.grad.data_ptr() == red._views[id(p)].data_ptr() for p in params)
    red.remove_hooks()
    q.put((rank, bool(ok), under_backward))
    dist.destroy_process_group()


def test_overlapped_gradient_all_reduce_matches_and_respects_accumulation():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_overlap, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res), res


def test_gradient_all_reduce_over_two_gloo_ranks():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2]                 # both ranks built the same bucket layout


def test_training_entry_points_exist_and_reject_what_is_not_built():
    import pytest
    from helpers import cfgs
    from lvdm.models.ddpm3d import LatentVisualDiffusion
    ident = {"target": "torch.nn.Identity"}
    model = LatentVisualDiffusion(img_cond_stage_config=ident, image_proj_stage_config=ident, cond_stage_config=ident,
                                  first_stage_config=ident, unet_config={"target": "lvdm.modules.networks.openaimodel3d.UNetModel",
                                                                         "params": cfgs.UNET_B}, **cfgs.DIFFUSION)
    assert torch.equal(model.lvlb_weights, torch.ones(1000)) and model.logvar.shape == (1000,)       # v-prediction: ones (ddpm3d.py:178-180)
    assert "lvlb_weights" not in model.state_dict() and "logvar" not in model.state_dict()          # as in the reference
    with pytest.raises(NotImplementedError, match="data"):
        model.shared_step({})
    with pytest.raises(RuntimeError, match="GPU"):
        model.configure_optimizers().step() if False else model.p_losses(
            torch.zeros(1, 4, 4, 8, 8), {"c_crossattn": [torch.zeros(1, 141, 64)], "c_concat": [torch.zeros(1, 8, 4, 8, 8)]},
            torch.zeros(1, dtype=torch.long), class_label=torch.zeros(1, 1, dtype=torch.long))


def test_one_rank_reducer_leaves_untouched_parameters_without_a_gradient():
    """the reducer binds every .grad to a zero-filled bucket view; in a one-rank job a parameter that never received a
    gradient must look to the optimiser as it does without a reducer (grad None: no state, no step, no weight decay)."""
    from mudg_amd.train.step import GradientAllReducer
    used, unused = torch.nn.Parameter(torch.ones(6)), torch.nn.Parameter(torch.ones(4))
    red = GradientAllReducer([used, unused], bucket_mb=1)
    opt = torch.optim.AdamW([used, unused], lr=0.1, weight_decay=0.5)
    for _ in range(2):
        red.zero_grad()
        assert used.grad is not None and unused.grad is not None            # views into the bucket while backward runs
        red.sync = False                                                    # a local micro-batch: nothing is dropped yet
        (used * 2).sum().backward()
        assert red() == 0 and unused.grad is not None
        red.sync = True
        (used * 3).sum().backward()
        assert red() == 0
        assert unused.grad is None and torch.equal(used.grad, torch.full((6,), 5.0))
        assert used.grad.data_ptr() == red._views[id(used)].data_ptr()
        opt.step()
    assert torch.equal(unused.detach(), torch.ones(4)) and unused not in opt.state
    assert not torch.equal(used.detach(), torch.ones(6))
