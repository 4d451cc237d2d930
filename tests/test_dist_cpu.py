"""CPU, world_size 2 over gloo: the data-parallel gradient buckets (rlmg_amd.dist.GradSync) give every
rank the mean gradient, equal to the single-process gradient of the concatenated batch."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 8),
                               torch.nn.Linear(8, 1, bias=False))


class _DirectLinear(torch.autograd.Function):
    """y = x W^T + b whose backward writes dW, db straight into .grad (ops.deliver_grads) and returns None for them --
    the pattern of the product's encoder layers.  Autograd still runs W's and b's AccumulateGrad with an undefined
    gradient afterwards and fires their post-accumulate hooks: GradSync must take that as an echo, not a delivery."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.params = (w, b)
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, g):
        from rlmg_amd import ops
        x, w = ctx.saved_tensors
        ops.deliver_grads(((ctx.params[0], g.t() @ x), (ctx.params[1], g.sum(0))))
        return g @ w, None, None


def _direct_forward(net, x):
    lin1, act, lin2, lin3 = net
    h = act(_DirectLinear.apply(x, lin1.weight, lin1.bias))
    return lin3(_DirectLinear.apply(h, lin2.weight, lin2.bias))          # lin3 goes through autograd


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import rlmg_amd  # noqa: F401
    from rlmg_amd import dist as rdist
    r, _, w = rdist.init_from_env()
    assert (r, w) == (rank, world)
    net = _model()
    unused = torch.nn.Linear(4, 4)                      # never receives a gradient (like project_concat_type)
    params = list(net.parameters()) + list(unused.parameters())
    sync = rdist.GradSync(params, bucket_bytes=256)     # several small buckets
    assert len(sync.buckets) >= 2
    g = torch.Generator().manual_seed(100)
    x = torch.randn(8, 16, generator=g)
    for step in range(2):                                # second step checks zero_grad / re-arming
        sync.zero_grad()
        xs = x[rank * 4:(rank + 1) * 4]
        net(xs).pow(2).mean().backward()
        sync.finish()
    total = sync.clip_grad_norm_(1e9)
    res = {"grads": [p.grad.clone() for p in net.parameters()], "norm": total.item(),
           "unused": [p.grad.clone() for p in unused.parameters()]}
    # gradient accumulation: two backward passes per step, one deferred reduction in finish()
    xs = x[rank * 4:(rank + 1) * 4]
    sync.zero_grad()
    sync.defer = True
    (net(xs[:2]).pow(2).sum() / 4).backward()
    (net(xs[2:]).pow(2).sum() / 4).backward()
    sync.finish()
    sync.defer = False
    res["accum"] = [p.grad.clone() for p in net.parameters()]
    # direct delivery: gradients written into the buckets by ops.deliver_grads, buckets notified per parameter
    from rlmg_amd import ops
    ps = list(net.parameters())
    gs = torch.autograd.grad(net(xs).pow(2).mean(), ps)
    sync.zero_grad()
    assert all(ops.direct_grads(p) for p in ps)
    ops.deliver_grads(tuple(zip(ps[:3], gs[:3])))       # two calls, as two layers' backward would do
    ops.deliver_grads(tuple(zip(ps[3:], gs[3:])))
    sync.finish()
    res["direct"] = [p.grad.clone() for p in net.parameters()]
    # the same through autograd Functions that deliver directly (hook echoes; small buckets so that a bucket which
    # launched after only part of its gradients -- counting an echo as a delivery -- gives a wrong result)
    for step in range(2):
        sync.zero_grad()
        _direct_forward(net, xs).pow(2).mean().backward()
        sync.finish()
    res["direct_fn"] = [p.grad.clone() for p in net.parameters()]
    # A step that runs the network TWICE before one backward (DQN.update: TD pass + train_step; PPO's inner step:
    # select_udpate + train_step) delivers every directly-written gradient twice.  With the default overlap=True
    # sync that must fail loudly (a bucket would go on the wire after the first pass's gradients only) ...
    g1 = torch.autograd.grad(net(xs[:2]).pow(2).sum() / 4, ps)
    g2 = torch.autograd.grad(net(xs[2:]).pow(2).sum() / 4, ps)
    sync.zero_grad()
    ops.deliver_grads(tuple(zip(ps, g1)))
    try:
        ops.deliver_grads(tuple(zip(ps, g2)))
        res["double_raises"] = False
    except RuntimeError as e:
        res["double_raises"] = "twice" in str(e)
    sync.finish()                                        # drain whatever was launched before the error
    # ... and with overlap=False (what DQN / PPO / RewardDiscri build) the two deliveries accumulate and finish()
    # reduces once: equal to the single-process gradient
    net2 = _model()
    sync2 = rdist.GradSync(list(net2.parameters()), bucket_bytes=256, overlap=False)
    ps2 = list(net2.parameters())
    for step in range(2):
        h1 = torch.autograd.grad(net2(xs[:2]).pow(2).sum() / 4, ps2)
        h2 = torch.autograd.grad(net2(xs[2:]).pow(2).sum() / 4, ps2)
        sync2.zero_grad()
        ops.deliver_grads(tuple(zip(ps2[:3], h1[:3])))   # pass 1, layer by layer
        ops.deliver_grads(tuple(zip(ps2[3:], h1[3:])))
        ops.deliver_grads(tuple(zip(ps2[:3], h2[:3])))   # pass 2
        ops.deliver_grads(tuple(zip(ps2[3:], h2[3:])))
        assert all(b.work is None and not b.launched for b in sync2.buckets)   # nothing on the wire before finish()
        sync2.finish()
    res["double"] = [p.grad.clone() for p in net2.parameters()]
    torch.save(res, out % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_gradsync_world2_matches_single_process(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    out = str(tmp_path / "r%d.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    net = _model()
    g = torch.Generator().manual_seed(100)
    x = torch.randn(8, 16, generator=g)
    # mean over ranks of per-rank means == mean over the whole batch (equal shard sizes)
    net(x).pow(2).mean().backward()
    ref = [p.grad for p in net.parameters()]
    for a, b, c in zip(r0["grads"], r1["grads"], ref):
        assert torch.allclose(a, b, atol=0, rtol=0)
        assert torch.allclose(a, c, atol=1e-6)
    assert r0["double_raises"] is True and r1["double_raises"] is True
    for key in ("accum", "direct", "direct_fn", "double"):
        for a, b, c in zip(r0[key], r1[key], ref):
            assert torch.allclose(a, b, atol=0, rtol=0), key
            assert torch.allclose(a, c, atol=1e-6), key
    assert all(t.abs().sum().item() == 0 for t in r0["unused"])
    ref_norm = torch.linalg.vector_norm(torch.cat([t.flatten() for t in ref])).item()
    assert abs(r0["norm"] - ref_norm) < 1e-5 and abs(r1["norm"] - ref_norm) < 1e-5


def test_gradsync_single_process_clip():
    sys.path.insert(0, ROOT)
    import rlmg_amd  # noqa: F401
    from rlmg_amd import dist as rdist
    net = _model()
    sync = rdist.GradSync(net.parameters())
    sync.zero_grad()
    net(torch.ones(3, 16)).sum().backward()
    sync.finish()
    n0 = sync.grad_norm().item()
    sync.clip_grad_norm_(n0 / 2)
    assert abs(sync.grad_norm().item() - n0 / 2) < 1e-4 * n0
    ref = torch.nn.utils.clip_grad_norm_(net.parameters(), 1e9).item()
    assert abs(ref - n0 / 2) < 1e-4 * n0
