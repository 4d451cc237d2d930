"""GPU, BASELINE configs[3] (IRL_dqn_train data-parallel): two ranks (gloo, both on the one GPU of the test box --
RCCL refuses duplicate devices) run the product `DQN._update_device` on their own batches.  `DQN.update` runs
eval_net twice before its one backward (TD pass + train_step, IRL_dqn_train.py:285-336), so every encoder
gradient is delivered twice; the all-reduced gradient must still be the mean of the two ranks' single-process
gradients, and the replicas must stay identical after the Adam step."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
pytestmark = pytest.mark.gpu
N_CLASS = [56, 135, 18, 87, 18, 25]


def _batch(rank):
    g = torch.Generator().manual_seed(500 + rank)
    B = 30
    tok = lambda T: torch.stack([torch.randint(0, n, (B, T), generator=g) for n in N_CLASS], -1)
    return (tok(50), tok(50), tok(25), torch.rand(B, 1, generator=g), torch.zeros(B, 1, dtype=torch.int64),
            tok(50), torch.ones(B, 50))


def _make_agent():
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from fill import fill_params
    from rlmg_amd.dqn_policy import IRL_dqn_train as T, config
    import contextlib
    import io
    old = dict(config.AgentConfig)
    config.AgentConfig.update({"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2})
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            agent = T.DQN(N_CLASS, Pretrain=False)
    finally:
        config.AgentConfig.update(old)
    fill_params(agent.eval_net, seed=61)
    fill_params(agent.target_net, seed=62)
    agent.eval_net.eval()                       # dropout off: the comparison must be deterministic
    agent.target_net.eval()
    return agent


def _grads(agent):
    return {n: p.grad.detach().cpu().clone() for n, p in agent.eval_net.named_parameters() if p.grad is not None}


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), CWLT_DIST_BACKEND="gloo", CWLT_SINGLE_DEVICE="1")
    import rlmg_amd  # noqa: F401
    from rlmg_amd import dist as rdist
    rdist.init_from_env()
    torch.cuda.set_device(0)
    agent = _make_agent()
    assert agent.sync.world == 2 and agent.sync.defer
    args = [t.cuda() for t in _batch(rank)]
    agent.optim.step = lambda *a, **k: None     # first call: keep the all-reduced gradient for inspection
    agent._update_device(*args)
    res = {"grads": _grads(agent)}
    del agent.optim.step
    agent._update_device(*args)                  # second call: real Adam step
    res["param_sum"] = torch.stack([p.detach().double().sum() for p in agent.eval_net.parameters()]).cpu()
    torch.save(res, out % rank)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_dqn_update_two_ranks_matches_mean_of_single_process_gradients(cuda, tmp_path):
    port = 29500 + (os.getpid() % 2000)
    out = str(tmp_path / "r%d.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    sys.path.insert(0, ROOT)
    import rlmg_amd  # noqa: F401
    singles = []
    for rank in range(2):
        agent = _make_agent()
        agent.optim.step = lambda *a, **k: None
        agent._update_device(*[t.cuda() for t in _batch(rank)])
        singles.append(_grads(agent))
    scale = max(v.abs().max().item() for v in singles[0].values())
    assert scale > 1e-5
    for name in singles[0]:
        want = (singles[0][name] + singles[1][name]) / 2
        assert torch.equal(r0["grads"][name], r1["grads"][name]), name             # replicas hold the SAME gradient
        assert (r0["grads"][name] - want).abs().max().item() <= 1e-5 * max(1.0, scale) + 1e-6 * scale, name
    assert torch.equal(r0["param_sum"], r1["param_sum"])                           # and stay identical after Adam
