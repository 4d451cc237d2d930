"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed backend "nccl")
over xGMI.  New functionality relative to the reference, which is single-GPU only.

Gradients live in a few flat f32 buckets (parameter .grad tensors are views into them), filled in
reverse registration order so the buckets complete in the order backward produces them.  A bucket's
all-reduce is launched asynchronously the moment its last gradient has been accumulated, so the
exchange of layer i overlaps the backward of layers < i; xGMI is point-to-point, so a few large
buckets (default 32 MiB) beat many small ones.  With world_size == 1 the same flat buffers serve the
fused gradient-norm clip, and nothing is communicated.

Two ways a gradient arrives: through autograd (post-accumulate hook) or written straight into the bucket by a
layer backward (`ops.deliver_grads`, which then calls the parameter's `_cwlt_ready` callback).

Two modes:
  overlap=True  (the pretrain step: ONE forward, ONE backward): every parameter reports exactly once per backward and
                a bucket's all-reduce starts when its last parameter has reported.  A parameter that reports a SECOND
                time before `zero_grad()` raises: a bucket may already be on the wire with only part of its gradient
                (a network run twice in one step delivers each encoder-layer gradient once per forward pass).
  overlap=False (`defer`; every RL step: DQN.update and PPO's inner step run the trainable net twice -- the TD /
                `select_udpate` pass and `train_step` -- and PPO.update_rollouts accumulates over rollout groups):
                nothing is launched from the callbacks; `finish()` reduces every bucket once, after the last
                backward.  The RL updates are small next to their all-reduce-free compute, so nothing is lost.
"""
import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("flat", "params", "seen", "n_seen", "work", "launched")

    def __init__(self, flat, params):
        self.flat, self.params = flat, params
        self.seen, self.n_seen, self.work, self.launched = set(), 0, None, False


class GradSync:
    def __init__(self, params, bucket_bytes=32 << 20, group=None, overlap=True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("no trainable parameters")
        self.buckets = []
        cur, cur_n = [], 0
        for p in reversed(params):
            cur.append(p)
            cur_n += p.numel()
            if cur_n * 4 >= bucket_bytes:
                self._close(cur)
                cur, cur_n = [], 0
        if cur:
            self._close(cur)
        self.defer = not overlap
        self._by_param = {}
        for b in self.buckets:
            for p in b.params:
                self._by_param[p] = b
                p.register_post_accumulate_grad_hook(self._on_hook)
                p._cwlt_ready = self._on_ready         # called by ops.deliver_grads after a direct write

    def _close(self, params):
        n = sum(p.numel() for p in params)
        flat = torch.zeros(n, dtype=torch.float32, device=params[0].device)
        o = 0
        for p in params:
            p.grad = flat[o:o + p.numel()].view_as(p)
            o += p.numel()
        self.buckets.append(_Bucket(flat, list(params)))

    def _launch(self, b):
        b.launched = True
        if self.world > 1:
            b.flat.div_(self.world)
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _report(self, p, direct):
        # A directly delivered gradient reports through `_cwlt_ready`; autograd then still runs the parameter's
        # AccumulateGrad node with an undefined gradient (the layer backward returned None for it) and, on this
        # PyTorch, fires the post-accumulate hook as well: that ONE echo per direct delivery is expected and ignored.
        b = self._by_param[p]
        key = (id(p), direct)
        if key in b.seen:
            raise RuntimeError(
                "GradSync(overlap=True): a parameter reported its gradient twice in one step (the network ran more "
                "than one forward pass before backward, or backward ran twice).  Its bucket may already be in flight "
                "with a partial gradient -- build the GradSync with overlap=False (or set defer) for such steps")
        b.seen.add(key)
        if not direct and (id(p), True) in b.seen:
            return                                     # echo of a direct delivery
        b.n_seen += 1
        if b.n_seen == len(b.params):
            self._launch(b)

    def _on_hook(self, p):
        if not self.defer:
            self._report(p, False)

    def _on_ready(self, p):
        if not self.defer:
            self._report(p, True)

    def zero_grad(self):
        """Replaces net.zero_grad(): keeps the .grad views, zeroes the flat storage."""
        for b in self.buckets:
            b.flat.zero_()
            b.seen, b.n_seen, b.work, b.launched = set(), 0, None, False
            o = 0
            for p in b.params:       # re-attach in case an optimizer / user dropped the view
                if p.grad is None or p.grad.data_ptr() != b.flat.data_ptr() + 4 * o:
                    p.grad = b.flat[o:o + p.numel()].view_as(p)
                o += p.numel()

    def finish(self):
        """Wait for every bucket (launching those whose parameters never received a gradient)."""
        for b in self.buckets:
            if not b.launched:
                self._launch(b)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.work = None

    def grad_norm(self):
        return torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(b.flat) for b in self.buckets]))

    def clip_grad_norm_(self, max_norm):
        """clip_grad_norm_(parameters, max_norm) on the flat buckets (dqn_policy/agent_pretrain.py:563-564)."""
        total = self.grad_norm()
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        for b in self.buckets:
            b.flat.mul_(coef)
        return total


def init_from_env():
    """(rank, local_rank, world) from torchrun's env; initialises RCCL when world > 1."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test-only overrides: CWLT_DIST_BACKEND=gloo and CWLT_SINGLE_DEVICE=1 let several ranks share one GPU
    # (RCCL refuses duplicate devices), to rehearse the multi-rank code path on a one-GPU box
    if os.environ.get("CWLT_SINGLE_DEVICE") == "1":
        local = 0
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CWLT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        import datetime
        # explicit collective timeout (CWLT_DIST_TIMEOUT_S, default 10 min): a rank that died leaves the others in a
        # collective; with the timeout they raise instead of hanging, exit non-zero, and the launcher tears the job down
        kw = {"timeout": datetime.timedelta(seconds=float(os.environ.get("CWLT_DIST_TIMEOUT_S", "600")))}
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
            if backend == "nccl":
                # bind the RCCL communicator to THIS rank's GPU at creation (eager init; without it the first
                # collective picks the device lazily and barrier() warns / may use device 0 for every rank)
                kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local, world
