"""Python bindings of the libcwlt kernels: raw launch wrappers + torch.autograd Functions.

Everything here runs on the GPU through the C-ABI in include/cwlt.h; there is no CPU path.
"""
import ctypes
import math
import os

import torch

from . import _lib

CLA_EPS = 1e-6  # fast_transformers CausalLinearAttention default eps
LN_EPS = 1e-5   # nn.LayerNorm default, used by fast_transformers' encoder layers


# --------------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------------
def _row_stride(t):
    """(N, L, H, D) view whose last two dims are dense and whose batch stride is L*row_stride."""
    N, L, H, D = t.shape
    if t.numel() == 0:
        return H * D                 # empty batch / sequence: strides are irrelevant
    if t.stride(3) != 1 or t.stride(2) != D:
        return None
    ld = t.stride(1)
    if N > 1 and t.stride(0) != L * ld:
        return None
    if ld % 4 != 0 or t.data_ptr() % 16 != 0:
        return None
    return ld


def _as_rows(t):
    ld = _row_stride(t)
    if ld is None:
        t = t.contiguous()
        ld = _row_stride(t)
    return t, ld


def _f32(t):
    """Parameter as a dense f32 device tensor (the kernels read gamma/beta/bias/tables in f32)."""
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.float().contiguous()
    return t


class KernelTimer:
    """Optional per-launch timing with HIP events on the launch stream (torch's current stream).

    bench.py switches it on for the timed region; each C-ABI call is then bracketed by two events.
    `summary()` (after a device sync) -> {entry point: (calls, mean ms)}."""
    enabled = False
    records = {}
    pool = []

    @classmethod
    def reset(cls, enabled):
        cls.enabled = bool(enabled)
        cls.records = {}

    @classmethod
    def reserve(cls, n):
        """Create (and once record) n timing events ahead of the timed region: hipEventCreate is slow, and the
        first few hundred creations are the slowest."""
        while len(cls.pool) < n:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            cls.pool.append(e)

    @classmethod
    def event(cls):
        return cls.pool.pop() if cls.pool else torch.cuda.Event(enable_timing=True)

    @classmethod
    def summary(cls):
        """{entry point: (calls, mean ms, mean work per call or None)} -- `work` is what the caller declared for
        the launch (FLOPs of a weight-gradient GEMM of that very shape), so rates use the real per-call shapes."""
        out = {}
        for name, evs in cls.records.items():
            ms = [a.elapsed_time(b) for a, b, _ in evs]
            works = [w for _, _, w in evs if w is not None]
            out[name] = (len(ms), sum(ms) / max(1, len(ms)), (sum(works) / len(works)) if works else None)
        return out


def _call(name, *args, work=None):
    """Invoke one libcwlt entry point on the current stream and raise on a non-zero status."""
    fn = getattr(_lib.load(), name)
    if KernelTimer.enabled:
        a, b = KernelTimer.event(), KernelTimer.event()
        a.record()
        st = fn(*args)
        b.record()
        KernelTimer.records.setdefault(name, []).append((a, b, work))
    else:
        st = fn(*args)
    _lib.check(st, name)


def direct_grads(param=None):
    """True when parameter gradients may be written straight into .grad by the layer backward: always in a
    single process; under data parallelism only for parameters owned by a `dist.GradSync` (it is told through
    the parameter's `_cwlt_ready` callback) -- with any other gradient hook (e.g. DDP) gradients go through
    autograd so that the hooks fire."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return True
    return param is not None and getattr(param, "_cwlt_ready", None) is not None


def deliver_grad(p, g):
    """p.grad += g with one fused convert + add (g may be bf16, p.grad is f32)."""
    if not p.requires_grad:
        return
    if p.grad is None:
        p.grad = g.float().clone() if g.dtype == torch.float32 else g.float()
    else:
        p.grad.add_(g)


def deliver_grads(pairs):
    """deliver_grad for a list of (parameter, gradient) pairs with ONE multi-tensor launch instead of one small
    device op per parameter (16 per encoder layer, ~9 us each).  Always an accumulation, as autograd's own
    AccumulateGrad: the buffers are zeroed by zero_grad().
    Never allocates while a hipGraph is being captured: a captured step must find every `.grad` in place (the flat
    buckets of dist.GradSync) -- a gradient tensor created inside the capture would live in the graph's private
    pool while `.grad` kept pointing at it from outside; that is refused here instead of being left to chance."""
    dst, src = [], []
    for p, g in pairs:
        if not p.requires_grad:
            continue
        if p.grad is None:
            if p.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("deliver_grads under graph capture: a parameter has no .grad buffer yet -- attach "
                                   "the gradients (dist.GradSync) and run the step eagerly once before capturing it")
            p.grad = g.float().clone() if g.dtype == torch.float32 else g.float()
        else:
            dst.append(p.grad)
            src.append(g.view(p.grad.shape) if g.dtype == p.grad.dtype else g.view(p.grad.shape).to(p.grad.dtype))
    if dst:
        torch._foreach_add_(dst, src)
    for p, _ in pairs:                       # data parallel: tell the gradient buckets (dist.GradSync)
        ready = getattr(p, "_cwlt_ready", None)
        if ready is not None and p.requires_grad:
            ready(p)


def deliver_grads_flat(params, grads, all_need_grad):
    """deliver_grads for a cached flat parameter list and its gradient views (encoder._EncoderStackFn): no per-pair
    Python work on the common path (every parameter trainable and already holding a .grad buffer)."""
    dst = [q.grad for q in params]
    if not all_need_grad or any(g is None for g in dst):
        return deliver_grads(list(zip(params, grads)))
    torch._foreach_add_(dst, grads)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for q in params:
            ready = getattr(q, "_cwlt_ready", None)
            if ready is not None:
                ready(q)


_seed_counter = [0]


def next_seed():
    """Fresh 62-bit dropout key from torch's CPU generator (reproducible under torch.manual_seed)."""
    _seed_counter[0] += 1
    return int(torch.empty((), dtype=torch.int64).random_().item()) & ((1 << 62) - 1)


def next_seeds(k):
    """k dropout keys in one draw: the same values as k calls of next_seed() (one generator call per element, in order)."""
    _seed_counter[0] += k
    return [v & ((1 << 62) - 1) for v in torch.empty(k, dtype=torch.int64).random_().tolist()]


# A captured hipGraph bakes every kernel argument, dropout seeds included.  The dropout kernels therefore accept
# an optional device pointer to one uint64 that is ADDED to the seed when the kernel runs (include/cwlt.h,
# "seed_base").  It is passed only while `GraphedCall` captures (and warms up) a function; the captured graph
# starts by bumping the value, so every replay draws fresh masks.  Eager launches pass NULL.
_SEED_BASE = {}
_USE_SEED_BASE = False
_SEED_STEP = 0x9E3779B97F4A7C15 - (1 << 64)     # golden-ratio increment as a signed 64-bit integer


def seed_base_tensor(device=None):
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    t = _SEED_BASE.get(idx)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("seed base must exist before graph capture starts")
        t = torch.zeros(1, dtype=torch.int64, device=torch.device("cuda", idx))
        _SEED_BASE[idx] = t
    return t


def _seed_base():
    return _lib.dev(seed_base_tensor()) if _USE_SEED_BASE else None


GRAPHS_ENABLED = os.environ.get("CWLT_GRAPHS", "1") != "0"     # CWLT_GRAPHS=0: RL rollout / update steps launch eagerly


TRAIN_GRAPHS = os.environ.get("CWLT_TRAIN_GRAPHS", "0") == "1"


def train_graphs_enabled():
    """Whole-training-step graphs (forward + backward + Adam of DQN.update / PPO.update_policy in one launch): OPT-IN
    with CWLT_TRAIN_GRAPHS=1 (and not CWLT_GRAPHS=0), single process only (under data parallelism the gradient
    all-reduce stays eager).  The no-grad rollout graphs (GRAPHS_ENABLED) are unaffected and on by default.

    Why opt-in: in round 1 a captured DQN.update produced NaN losses from its third replay on --
    once hipErrorIllegalAddress -- whenever large eager GEMMs ran between replays.  The change that made it go away
    (accumulate-only gradient delivery instead of copy-on-first-delivery) is arithmetically equivalent to what it
    replaced and explains neither symptom; in round 2 neither the old delivery re-created on the current tree
    (tools/diag_fresh_copy.py) nor the tree that preceded the change, run with its own failing recipe, reproduces
    the fault on this round's boxes (gpurun_out/r02_diag_*.log).  The cause is therefore NOT known.  What is in
    place instead of a diagnosis: captured steps never allocate gradient storage (deliver_grads), a captured step
    checks on every replay that the gradient / parameter storage it was captured against is still where it was
    (GraphedCall(params=...)), and the feature stays off unless asked for."""
    import torch.distributed as dist
    single = not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
    return GRAPHS_ENABLED and TRAIN_GRAPHS and single


FUSED_ADAM = os.environ.get("CWLT_FUSED_ADAM", "1") != "0"


def graph_adam(params, lr, **kw):
    """torch.optim.Adam as the reference constructs it; when training steps may be captured, `capturable=True`
    with the learning rate held in a device tensor, so an LR scheduler's updates reach the captured step.
    `fused=True` there: the capturable multi-tensor ("foreach") path divides every state tensor by 0-dim tensors one
    parameter at a time (2 x 438 launches per PPO step, 14 % of the window-50 update's GPU time); the fused kernel
    does the whole step in a few launches."""
    params = list(params)
    if train_graphs_enabled() and params and params[0].is_cuda:
        return torch.optim.Adam(params, lr=torch.tensor(float(lr), device=params[0].device), capturable=True,
                                fused=True, **kw)
    # eager steps: the single-kernel ("fused") implementation of the same update -- the default multi-tensor one is
    # eight passes over every parameter and state tensor per step (9 % of the GPU time of a window-50 PPO iteration,
    # profiles/r04_ppo_w50_kernels.txt).  CWLT_FUSED_ADAM=0: torch's default implementation.
    if FUSED_ADAM and params and all(q.is_cuda and q.dtype == torch.float32 for q in params):
        return torch.optim.Adam(params, lr=lr, fused=True, **kw)
    return torch.optim.Adam(params, lr=lr, **kw)


def capture_hip_graph(body, mode=torch.no_grad, name="step"):
    """Capture `body()` on the current stream into a hipGraph that is safe to replay on this runtime.
    -> (torch.cuda.CUDAGraph or None, body's result, node census before the rewrite, memset nodes rewritten).

    A captured hipMemsetAsync replays a WRONG fill pattern on ROCm 7.2 from the second replay on when other work runs
    in between (tools/probes/graph_memset_probe.py), and library code inside a captured step issues such memsets
    (PyTorch's reduction semaphores, hipBLASLt's split-K workspaces: tools/diag_memset_sites.py).  Before the graph is
    instantiated every memset node is therefore replaced by a fill-kernel node (cwlt_graph_replace_memset_nodes).
    keep_graph: the hipGraph_t stays accessible after the capture so that its nodes can be counted and edited; the
    executable graph is instantiated by the first replay.  Should any memset node remain (inside a child graph), None
    is returned in place of the graph: the caller must run the step eagerly -- the capture itself executed nothing."""
    import warnings
    if not _graph_editing_available():
        # a PyTorch without CUDAGraph(keep_graph=True) / raw_cuda_graph(): the graph's nodes can be neither counted nor
        # rewritten, so nothing is captured and the caller runs the step eagerly (one warning per process)
        if not _GRAPH_API[1]:
            _GRAPH_API[1] = True
            warnings.warn("capture_hip_graph: this PyTorch has no CUDAGraph(keep_graph=True).raw_cuda_graph(); "
                          "hipGraph capture is off, steps run eagerly")
        return None, None, {}, 0
    graph = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(graph), mode():
        out = body()
    raw = graph.raw_cuda_graph()
    census = _lib.graph_node_census(raw)
    replaced = 0
    if census.get("memset", 0):
        n = ctypes.c_int(0)
        st = _lib.load().cwlt_graph_replace_memset_nodes(ctypes.c_void_p(int(raw)), ctypes.byref(n))
        replaced = n.value
        left = -1 if st else _lib.graph_node_census(raw).get("memset", 0)
        if left:
            # a rewrite that failed (status != 0: e.g. an element size the fill kernel does not take) or left a node
            # inside a child graph: the step runs eagerly rather than raising out of a training loop
            warnings.warn("capture_hip_graph: the captured %s %s (%s); a replayed hipMemsetAsync is not reliable on this "
                          "runtime -- running it eagerly instead"
                          % (name, "could not be rewritten (status %d)" % st if st
                             else "still holds %d memset node(s) after the rewrite" % left, census))
            graph.reset()
            graph = None
    return graph, out, census, replaced


_GRAPH_API = [None, False]          # [keep_graph / raw_cuda_graph available, warned]


def _graph_editing_available():
    if _GRAPH_API[0] is None:
        try:
            import inspect
            ok = hasattr(torch.cuda.CUDAGraph, "raw_cuda_graph")
            if ok:
                try:
                    ok = "keep_graph" in inspect.signature(torch.cuda.CUDAGraph.__new__).parameters
                except (TypeError, ValueError):
                    ok = True                      # signature not introspectable: trust the method's presence
            _GRAPH_API[0] = bool(ok)
        except Exception:
            _GRAPH_API[0] = False
    return _GRAPH_API[0]


class GraphedCall:
    """Run `fn(*tensors)` -- a sync-free function of device tensors -- as ONE hipGraph launch.

    grad=False: a no-grad forward (an RL rollout step: trunk forward + heads + action gather, a few hundred launches
    for a 50-token window).  Warmed up on a side stream, then captured on the first call.
    grad=True: a whole training step -- forward, backward and optimizer.step() (optimizers built with
    capturable=True) -- for the launch-bound small-batch updates of the RL loops.  The first `eager_calls` calls
    run eagerly (they are real steps and create the optimizer state); the next call is captured, then replayed.

    Captured once per input signature; replays copy the arguments into the graph's static inputs, bump the dropout
    seed base (a captured op) and launch.  Parameters, gradients and optimizer state are used in place.
    Outputs are the graph's static buffers: valid until the next call with the same signature."""

    def __init__(self, fn, warmup=2, grad=False, eager_calls=2, params=None):
        self.fn, self.warmup, self.grad, self.eager_calls = fn, warmup, grad, eager_calls
        self.graphs, self.calls = {}, {}
        self.census = None                                 # node kinds of the last capture (_lib.graph_node_census)
        self.memsets_replaced = 0
        # grad=True: the parameters the step trains.  Their storage and their .grad storage are baked into the graph
        # as raw addresses; `_storage()` is compared before every replay so that a moved buffer (a dropped / re-made
        # .grad, a re-allocated parameter) raises instead of letting the graph write through a stale address.
        self.params = [p for p in params if p.requires_grad] if params is not None else None
        self._addr = {}

    def _storage(self):
        if not self.params:
            return None
        if any(p.grad is None for p in self.params):
            raise RuntimeError("GraphedCall(grad=True): a trained parameter has no .grad buffer -- gradients must live "
                               "in persistent storage (dist.GradSync buckets) before the step is captured")
        return tuple((p.data_ptr(), p.grad.data_ptr()) for p in self.params)

    def _capture(self, args):
        global _USE_SEED_BASE
        dev = args[0].device
        base = seed_base_tensor(dev)
        static = [a.clone() for a in args]
        was_timing = KernelTimer.enabled
        KernelTimer.enabled = False
        _USE_SEED_BASE = True
        mode = torch.enable_grad if self.grad else torch.no_grad
        try:
            if not self.grad:
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side), torch.no_grad():
                    for _ in range(self.warmup):
                        self.fn(*static)
                torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            before = self._storage()
            def body():
                base.add_(_SEED_STEP)
                return self.fn(*static)
            graph, out, self.census, self.memsets_replaced = capture_hip_graph(
                body, mode, getattr(self.fn, "__name__", "step"))
            if self._storage() != before:
                raise RuntimeError("GraphedCall(grad=True): parameter / gradient storage changed DURING capture")
        finally:
            _USE_SEED_BASE = False
            KernelTimer.enabled = was_timing
        return graph, static, out

    def __call__(self, *args):
        key = tuple((tuple(a.shape), a.dtype, a.device.index) for a in args)
        ent = self.graphs.get(key)
        if ent is None:
            n = self.calls.get(key, 0)
            if self.grad and n < self.eager_calls:
                self.calls[key] = n + 1
                return self.fn(*args)
            ent = self.graphs[key] = self._capture(args)
            self._addr[key] = self._storage()
        graph, static, out = ent
        if graph is None:                                  # a capture that was refused (memset nodes): eager
            return self.fn(*args)
        if self.params and self._storage() != self._addr[key]:
            raise RuntimeError("GraphedCall(grad=True): parameter / gradient storage moved since the step was "
                               "captured (a .grad was dropped or re-created, or a parameter re-allocated)")
        for s, a in zip(static, args):
            s.copy_(a)
        graph.replay()
        return out



# --------------------------------------------------------------------------------------------------
# compute-dtype copies of the master parameters
# --------------------------------------------------------------------------------------------------
# The f32 master weights are used in bf16 (and Q/K/V row-stacked) by every layer.  Casting them per call is seven
# 1-4 MB copy kernels + two cat kernels per layer per forward: nothing at B*T = 524 288 rows, but a third of the GPU
# time of the launch-bound RL steps (window 50: 108 such kernels per 12-layer trunk forward).  A ShadowSet keeps
# persistent copies and refreshes ALL of them with one multi-tensor copy per forward.
#
# The refresh is unconditional: nothing cheap says whether a parameter changed (fused optimizers and replays of a
# captured optimizer step move parameters WITHOUT bumping autograd's version counters).  Only inside
# `with ops.frozen_weights():` -- a scope in which the caller guarantees that no parameter changes, e.g. the loop
# over buffer batches of one reward-scoring call -- is a set refreshed once and then reused.
_FROZEN = [0]            # id of the innermost frozen_weights scope, 0 outside any
_FROZEN_IDS = [0]


class frozen_weights:
    """Scope in which no parameter of any model changes: ShadowSets refresh at most once inside it."""

    def __enter__(self):
        self.prev = _FROZEN[0]
        _FROZEN_IDS[0] += 1
        _FROZEN[0] = _FROZEN_IDS[0]
        return self

    def __exit__(self, *exc):
        _FROZEN[0] = self.prev
        return False


def _no_shadow():
    return None


# CWLT_DGRAD_WT_CACHE=0: the per-op layer transposes a weight at every use in the backward instead of once per forward
DGRAD_WT_CACHE = os.environ.get("CWLT_DGRAD_WT_CACHE", "1") != "0"
# CWLT_CAST_MANY=0: refresh the bf16 weight copies with torch's multi-tensor copy instead of cwlt_cast_bf16_many
CAST_MANY = os.environ.get("CWLT_CAST_MANY", "1") != "0"


class ShadowSet:
    """`groups`: tuples of parameters; group i becomes ONE buffer of `dtype` with its members stacked along dim 0
    (a single-member group is a plain cast).  `refresh()` -> list of buffers, up to date with the parameters."""

    def __init__(self, groups, dtype):
        self.dtype = dtype
        self.device = groups[0][0].device
        self.bufs, self.dst, self.src = [], [], []
        for g in groups:
            rows = sum(p.shape[0] for p in g)
            buf = torch.empty((rows,) + tuple(g[0].shape[1:]), dtype=dtype, device=self.device)
            o = 0
            for p in g:
                self.dst.append(buf[o:o + p.shape[0]])
                self.src.append(p)
                o += p.shape[0]
            self.bufs.append(buf)
        self.fresh_in = -1           # id of the frozen_weights scope the buffers were last refreshed in
        self._fast = None            # (parameter addresses, device table or None): the one-launch cast, see _cast_table

    def _cast_table(self):
        """bf16 copies of dense f32 GPU parameters: (table, source base, destination base, n) for cwlt_cast_bf16_many --
        one launch where torch's multi-tensor copy takes four for the 84 tensors of an encoder -- or None (other dtypes,
        or parameters that moved while a capture is running: the multi-tensor copy then).  Rebuilt when a parameter's
        storage moves."""
        ptrs = tuple(p.data_ptr() for p in self.src)
        if self._fast is None or self._fast[0] != ptrs:
            ok = (self.dtype == torch.bfloat16 and self.device.type == "cuda"
                  and all(p.dtype == torch.float32 and p.is_contiguous() for p in self.src)
                  and all(d.is_contiguous() for d in self.dst))
            if ok and torch.cuda.is_current_stream_capturing():
                return None                                  # the table is a host-to-device copy: not inside a capture
            entry = None
            if ok:
                sbase = min(ptrs)
                dbase = min(d.data_ptr() for d in self.dst)
                rows = [((p.data_ptr() - sbase) // 4, (d.data_ptr() - dbase) // 2, p.numel())
                        for p, d in zip(self.src, self.dst)]
                entry = (torch.tensor(rows, dtype=torch.int64).to(self.device), sbase, dbase, len(rows))
            self._fast = (ptrs, entry)
        return self._fast[1]

    def matches(self, dtype, device):
        return self.dtype == dtype and self.device == device and all(p.device == device for p in self.src)

    def __reduce__(self):
        # a cache, not state: copy.deepcopy(model) / torch.save(model) get None and rebuild it on first use (a copied
        # set would have its views detached from its buffers)
        return (_no_shadow, ())

    def refresh(self):
        scope = _FROZEN[0]
        if scope and self.fresh_in == scope and not (self.device.type == "cuda"
                                                     and torch.cuda.is_current_stream_capturing()):
            return self.bufs
        fast = self._cast_table() if CAST_MANY else None
        if fast is not None:
            table, sbase, dbase, n = fast
            _call("cwlt_cast_bf16_many", ctypes.c_void_p(sbase), ctypes.c_void_p(dbase), _lib.dev(table), n,
                  _lib.stream_ptr())
        else:
            with torch.no_grad():
                torch._foreach_copy_(self.dst, [p.detach() for p in self.src])
        self.fresh_in = scope if scope else -1
        return self.bufs


# --------------------------------------------------------------------------------------------------
# causal linear attention
# --------------------------------------------------------------------------------------------------
def scan_segments(N, H, L, dtype, ld_ok=True):
    """Segments per stream the scan kernels should use (1 unless N * H is far below the CU count; bf16 only).
    CWLT_SCAN_SEGMENTS=<n> forces a count (tests, A/B runs): clipped to the number of 64-token chunks."""
    if dtype != torch.bfloat16 or not ld_ok or N * L == 0:
        return 1
    forced = os.environ.get("CWLT_SCAN_SEGMENTS")
    if forced:
        nch = (L + 63) // 64
        want = max(1, min(int(forced), nch))
        cps = -(-nch // want)
        return -(-nch // cps)
    return int(_lib.load().cwlt_scan_segments(N, H, L, _lib.dtype_code(dtype)))


def _seg_ws(N, H, P, backward, device):
    if P <= 1:
        return None
    n = int(_lib.load().cwlt_scan_seg_floats(N, H, P, 1 if backward else 0))
    return torch.empty(n, dtype=torch.float32, device=device)


# One-sweep backward of the bf16 scan (cwlt_causal_linear_bwd_sweep); CWLT_SCAN_SWEEP=0 keeps the dkdv + dq pair.
SCAN_SWEEP = os.environ.get("CWLT_SCAN_SWEEP", "1") != "0"


def cla_fwd(q, k, v, eps=CLA_EPS, final_state=None):
    """q, k, v: (N, L, H, 64) views (row-strided ok) -> out (N, L, H, 64) dense, zinv (N, L, H) f32
    [, fin: the scan's final state for `cla_bwd(final_state=fin)`, or None where the one-sweep backward does not apply
    (f32, odd row strides, segmented few-stream launches) or final_state is False -- a sixth element whenever
    final_state is given (True / False) at all]."""
    lib = _lib.load()
    N, L, H, D = q.shape
    if k.shape != q.shape or v.shape != q.shape:
        raise ValueError("causal linear attention needs q, k, v of one shape (N, L, H, D)")
    if q.dtype != k.dtype or q.dtype != v.dtype:
        raise TypeError("q, k, v dtypes differ")
    q, ldq = _as_rows(q)
    k, ldk = _as_rows(k)
    v, ldv = _as_rows(v)
    out = torch.empty((N, L, H, D), dtype=q.dtype, device=q.device)
    zinv = torch.empty((N, L, H), dtype=torch.float32, device=q.device)
    fast = q.dtype == torch.bfloat16 and all(x % 8 == 0 for x in (ldq, ldk, ldv))
    P = scan_segments(N, H, L, q.dtype, fast)
    ws = _seg_ws(N, H, P, False, q.device)
    fin = None
    if final_state and SCAN_SWEEP and fast and P == 1 and N * L > 0:
        fin = torch.empty(int(lib.cwlt_scan_final_state_floats(N, H)), dtype=torch.float32, device=q.device)
    _call("cwlt_causal_linear_fwd", _lib.dev(q, "q"), _lib.dev(k, "k"), _lib.dev(v, "v"), _lib.dev(out), _lib.dev(zinv),
        N, H, L, D, ldq, ldk, ldv, H * D, float(eps), P, _lib.opt(ws), _lib.opt(fin), _lib.dtype_code(q.dtype),
        _lib.stream_ptr())
    if final_state is not None:
        return q, k, v, out, zinv, fin
    return q, k, v, out, zinv


def cla_bwd(q, k, v, out, zinv, dout, want_colsum=False, final_state=None):
    """-> dqkv (N, L, 3, H, 64): dq | dk | dv side by side (= gradient of a fused QKV projection)
    [, dbias (3*H*64) f32 = its column sums, fused into the kernels on the bf16 path].
    final_state: what `cla_fwd(final_state=True)` returned for these q, k, v -- the backward then runs as one sweep."""
    lib = _lib.load()
    N, L, H, D = q.shape
    dout, lddo = _as_rows(dout)
    q, ldq = _as_rows(q)
    k, ldk = _as_rows(k)
    v, ldv = _as_rows(v)
    dqkv = torch.empty((N, L, 3, H, D), dtype=q.dtype, device=q.device)
    ld = 3 * H * D
    common = (_lib.dev(q), _lib.dev(k), _lib.dev(v), _lib.dev(out), _lib.dev(zinv), _lib.dev(dout, "dout"))
    code, st = _lib.dtype_code(q.dtype), _lib.stream_ptr()
    fast = q.dtype == torch.bfloat16 and all(x % 8 == 0 for x in (ldq, ldk, ldv, lddo))
    fused = want_colsum and fast
    if final_state is not None and not fast:
        # dout arrived as a view whose row stride the bf16 kernels cannot take (a legal autograd input of the public
        # causal_linear_attention): the final state is simply not used -- the dkdv + dq pair below handles any stride
        final_state = None
    if final_state is not None:
        cs = torch.empty((3, N, H * D), dtype=torch.float32, device=q.device) if want_colsum else None
        _call("cwlt_causal_linear_bwd_sweep", *common, _lib.dev(final_state), _lib.dev(dqkv[:, :, 0]),
              _lib.dev(dqkv[:, :, 1]), _lib.dev(dqkv[:, :, 2]), *([_lib.dev(cs[i]) for i in range(3)] if want_colsum
                                                                  else [None] * 3),
              N, H, L, D, ldq, ldk, ldv, H * D, lddo, ld, ld, ld, code, st)
        if not want_colsum:
            return dqkv
        return dqkv, (cs.sum(1).reshape(3 * H * D) if N > 1 else cs.reshape(3 * H * D))
    P = scan_segments(N, H, L, q.dtype, fast)
    ws = _seg_ws(N, H, P, True, q.device)          # shared by the two calls: dkdv fills it, dq reads it
    cs = torch.empty((3, N * P, H * D), dtype=torch.float32, device=q.device) if fused else None
    # dden = -(dout . out) * zinv: written by the reverse scan, read by the dq scan instead of the whole `out` stream
    dden = torch.empty((N, L, H), dtype=torch.float32, device=q.device) if fast else None
    _call("cwlt_causal_linear_bwd_dkdv", *common, _lib.dev(dqkv[:, :, 1]), _lib.dev(dqkv[:, :, 2]),
          _lib.dev(cs[1]) if fused else None, _lib.dev(cs[2]) if fused else None, _lib.opt(dden), N, H, L, D,
          ldq, ldk, ldv, H * D, lddo, ld, ld, P, _lib.opt(ws), code, st)
    _call("cwlt_causal_linear_bwd_dq", *common, _lib.dev(dqkv[:, :, 0]), _lib.dev(cs[0]) if fused else None,
          _lib.opt(dden), N, H, L, D, ldq, ldk, ldv, H * D, lddo, ld, P, _lib.opt(ws), code, st)
    if not want_colsum:
        return dqkv
    if fused:
        dbias = cs.sum(1).reshape(3 * H * D) if N * P > 1 else cs.reshape(3 * H * D)
    else:
        dbias = colsum(dqkv.view(N * L, 3 * H * D))
    return dqkv, dbias


class CausalLinearAttentionFn(torch.autograd.Function):
    """out = CLA(q, k, v): q, k, v (N, L, H, 64) raw projections, elu+1 applied inside the kernel.

    Replaces fast_transformers CausalLinearAttention.forward + causal_dot_product
    (reference call sites: dqn_policy/model.py:128-137,231-232).
    """

    @staticmethod
    def forward(ctx, q, k, v, eps=CLA_EPS, grad_mode=True):
        # grad_mode = torch.is_grad_enabled() at the call site (inside forward() it is off; see encoder._EncoderLayerFn)
        q, k, v, out, zinv, fin = cla_fwd(q, k, v, eps, final_state=bool(grad_mode) and any(ctx.needs_input_grad))
        ctx.save_for_backward(q, k, v, out, zinv)
        ctx.fin = fin
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, zinv = ctx.saved_tensors
        dqkv = cla_bwd(q, k, v, out, zinv, dout, final_state=ctx.fin)
        return dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2], None, None


def causal_linear_attention(q, k, v, eps=CLA_EPS):
    return CausalLinearAttentionFn.apply(q, k, v, eps, torch.is_grad_enabled())


# --------------------------------------------------------------------------------------------------
# residual + dropout + LayerNorm
# --------------------------------------------------------------------------------------------------
def ln_fwd(x, a, gamma, beta, eps=LN_EPS, p=0.0, seed=0, save_s=True):
    """s = x + dropout(a); y = LN(s).  x may be None.  a: (rows, D) dense.  -> s (or None), y, mean, rstd."""
    lib = _lib.load()
    rows, D = a.shape
    a = a.contiguous()
    if x is not None:
        x = x.contiguous()
    y = torch.empty_like(a)
    need_s = save_s and (x is not None or p > 0)
    s = torch.empty_like(a) if need_s else None
    mean = torch.empty(rows, dtype=torch.float32, device=a.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=a.device)
    _call("cwlt_add_dropout_layernorm_fwd", _lib.opt(x), _lib.dev(a, "a"), _lib.dev(gamma), _lib.dev(beta), _lib.opt(s), _lib.dev(y),
        _lib.dev(mean), _lib.dev(rstd), rows, D, float(eps), float(p), int(seed), _seed_base(),
        _lib.dtype_code(a.dtype), _lib.stream_ptr())
    if save_s and s is None:
        s = a
    return s, y, mean, rstd


def ln_bwd(dy, dy2, s, gamma, mean, rstd, p=0.0, seed=0, want_dbias=True):
    """-> ds, da, dgamma, dbeta, dbias.  da is ds itself when p == 0."""
    lib = _lib.load()
    rows, D = s.shape
    dy = dy.contiguous()
    if dy2 is not None:
        dy2 = dy2.contiguous()
    ds = torch.empty_like(s)
    da = torch.empty_like(s) if p > 0 else ds
    nb = lib.cwlt_ln_blocks(rows)
    part = torch.empty(nb * 3 * D, dtype=torch.float32, device=s.device)
    stats = torch.empty((3, D), dtype=torch.float32, device=s.device)
    _call("cwlt_add_dropout_layernorm_bwd", _lib.dev(dy, "dy"), _lib.opt(dy2), _lib.dev(s), _lib.dev(gamma), _lib.dev(mean), _lib.dev(rstd),
        _lib.dev(ds), _lib.dev(da) if p > 0 else None, _lib.dev(part), _lib.dev(stats), rows, D, float(p), int(seed), _seed_base(),
        _lib.dtype_code(s.dtype), _lib.stream_ptr())
    return ds, da, stats[0], stats[1], (stats[2] if want_dbias else None)


class LayerNormFn(torch.autograd.Function):
    """Plain LayerNorm over the last dim through the fused kernel (x == NULL, p == 0)."""

    @staticmethod
    def forward(ctx, a, gamma, beta, eps):
        shape = a.shape
        a2 = a.reshape(-1, shape[-1])
        g, b = _f32(gamma), _f32(beta)
        s, y, mean, rstd = ln_fwd(None, a2, g, b, eps, 0.0, 0, save_s=True)
        ctx.save_for_backward(s, g, mean, rstd)
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        s, g, mean, rstd = ctx.saved_tensors
        ds, _, dg, db, _ = ln_bwd(dy.reshape(s.shape), None, s, g, mean, rstd, 0.0, 0, want_dbias=False)
        return ds.view(dy.shape), dg, db, None


def layer_norm(a, gamma, beta, eps=LN_EPS):
    return LayerNormFn.apply(a, gamma, beta, eps)


# --------------------------------------------------------------------------------------------------
# FFN activation, column sums, positional encoding
# --------------------------------------------------------------------------------------------------
def gelu_fwd(h, bias, p=0.0, seed=0, gd_inplace=False):
    """g = dropout(gelu(h + bias)).  gd_inplace: h is OVERWRITTEN with gd = mask / (1 - p) * gelu'(h + bias), the factor
    the backward multiplies the upstream gradient with (gemm_nt_mul); h itself is gone afterwards."""
    lib = _lib.load()
    rows, F = h.shape
    g = torch.empty_like(h)
    _call("cwlt_bias_gelu_dropout_fwd", _lib.dev(h, "h"), _lib.opt(bias), _lib.dev(g), _lib.dev(h) if gd_inplace else None,
          rows, F, float(p), int(seed), _seed_base(), _lib.dtype_code(h.dtype), _lib.stream_ptr())
    return g


# rows from which the hand-written projection GEMM is used (256-row tiles, one workgroup per CU: a step of a few
# thousand token rows leaves most of the chip idle; those keep the library GEMM)
GEMM_BF16_MIN_ROWS = int(os.environ.get("CWLT_GEMM_BF16_MIN_ROWS", 32768))
GEMM_BF16 = os.environ.get("CWLT_GEMM_BF16", "1") != "0"


def gemm_bf16_supported(a, w, c=None, transposed_w=False):
    """Whether `gemm_bf16(a, w, out=c)` can run.  transposed_w: `w` is a transposed VIEW of a stored weight that the
    caller will make contiguous first (the input-gradient forms): only its shape is looked at."""
    return (GEMM_BF16 and a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and a.dim() == 2 and w.dim() == 2
            and a.is_cuda and a.shape[1] == w.shape[1] and a.shape[1] % 64 == 0 and a.shape[1] >= 128
            and w.shape[0] % 8 == 0 and w.shape[0] <= 8192 and a.shape[0] >= GEMM_BF16_MIN_ROWS
            and a.stride(1) == 1 and a.stride(0) % 8 == 0 and a.data_ptr() % 16 == 0
            and (transposed_w or (w.stride(1) == 1 and w.stride(0) % 8 == 0 and w.data_ptr() % 16 == 0))
            and (c is None or (c.dtype == torch.bfloat16 and c.shape == (a.shape[0], w.shape[0]) and c.stride(1) == 1
                               and c.stride(0) % 8 == 0 and c.data_ptr() % 16 == 0)))


def gemm_bf16(a, w, bias=None, out=None, accumulate=False):
    """out (M, N) [+]= a (M, K) @ w (N, K).T [+ bias]: the projection GEMM (csrc/gemm_bf16.hip).  a, w bf16; bias (N)
    f32 or None; out bf16 (allocated when None; accumulate adds onto its contents)."""
    _lib.load()
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        if accumulate:
            raise ValueError("gemm_bf16: accumulate needs the tensor to add onto")
        out = torch.empty((M, N), dtype=a.dtype, device=a.device)
    _call("cwlt_gemm_bf16", _lib.dev(a, "a"), _lib.dev(w, "w"), _lib.opt(bias), _lib.dev(out, "out"), M, N, K, a.stride(0),
          w.stride(0), out.stride(0), 1 if accumulate else 0, _lib.stream_ptr(), work=2.0 * M * N * K)
    return out


# one encoder layer per host call (csrc/layer.hip, below): CWLT_LAYER_C=0 keeps the per-op path at every size
LAYER_C = os.environ.get("CWLT_LAYER_C", "1") != "0"
LAYER_C_MAX_ROWS = int(os.environ.get("CWLT_LAYER_C_MAX_ROWS", 8192))
# ... and the whole stack of an encoder per host call (one autograd node); CWLT_LAYER_C_STACK=0: one call per layer
LAYER_C_STACK = os.environ.get("CWLT_LAYER_C_STACK", "1") != "0"
# CWLT_GEMM_SMALL_PER_OP=1 (tests): the per-op encoder layer takes its plain projections from cwlt_gemm_bf16_small instead
# of hipBLASLt whenever its rows are few enough for the one-call layer -- then the two paths run the same kernels
GEMM_SMALL_PER_OP = os.environ.get("CWLT_GEMM_SMALL_PER_OP", "0") == "1"


def gemm_small_per_op(x):
    return GEMM_SMALL_PER_OP and x.dtype == torch.bfloat16 and x.dim() == 2 and 0 < x.shape[0] <= LAYER_C_MAX_ROWS


def gemm_bf16_small(a, w, bias=None, out=None, accumulate=False):
    """`gemm_bf16` for few rows (csrc/gemm_small.hip: 64 x 64 output tiles, operands straight from L2): K % 32 == 0."""
    _lib.load()
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        if accumulate:
            raise ValueError("gemm_bf16_small: accumulate needs the tensor to add onto")
        out = torch.empty((M, N), dtype=a.dtype, device=a.device)
    _call("cwlt_gemm_bf16_small", _lib.dev(a, "a"), _lib.dev(w, "w"), _lib.opt(bias), _lib.dev(out, "out"), M, N, K,
          a.stride(0), w.stride(0), out.stride(0), 1 if accumulate else 0, _lib.stream_ptr(), work=2.0 * M * N * K)
    return out


def gemm_bf16_small_gelu(x, w, bias, p=0.0, seed=0, want_gd=True):
    """`ffn1_gelu_dropout` on the split-K small tiles (csrc/gemm_small.hip), for passes of a few hundred rows at most:
    g = dropout(gelu(x @ w.T + bias)) [, gd = mask / (1 - p) * gelu'(.)] -- bit for bit what `gemm_bf16_small` followed by
    `gelu_fwd(..., gd_inplace=True)` produce.  K % 128 == 0."""
    _lib.load()
    M, K = x.shape
    N = w.shape[0]
    g = torch.empty((M, N), dtype=x.dtype, device=x.device)
    gd = torch.empty((M, N), dtype=x.dtype, device=x.device) if want_gd else None
    _call("cwlt_gemm_bf16_small_gelu", _lib.dev(x, "x"), _lib.dev(w, "w"), _lib.dev(bias), _lib.dev(g), _lib.opt(gd), M, N, K,
          x.stride(0), w.stride(0), float(p), int(seed), _seed_base(), _lib.stream_ptr(), work=2.0 * M * N * K)
    return g, gd


# --------------------------------------------------------------------------------------------------
# one encoder layer per host call (csrc/layer.hip): steps of a few thousand token rows are bound by the HOST when every
# kernel is its own Python-level call (the reference's RL updates: 30 windows x 50 tokens, ~970 launches)
# --------------------------------------------------------------------------------------------------
_LAYER_PLANS = {}


def layer_plan(N, L, D, F, H, p, want_backward):
    """cwlt_encoder_layer_plan, cached: buffer sizes and gradient offsets of one layer call."""
    key = (N, L, D, F, H, p > 0, bool(want_backward))
    plan = _LAYER_PLANS.get(key)
    if plan is None:
        plan = _lib.EncoderLayerPlan()
        _lib.check(_lib.load().cwlt_encoder_layer_plan(N, L, D, F, H, float(p), 1 if want_backward else 0,
                                                       ctypes.byref(plan)), "cwlt_encoder_layer_plan")
        _LAYER_PLANS[key] = plan
    return plan


class LayerCache:
    """What the one-call layers of ONE encoder share: the transposed bf16 weight copies the input-gradient products
    read (all matrices refreshed by one launch, cwlt_transpose_bf16_many) and a scratch buffer (every layer call of
    an encoder runs on the launch stream, one after the other).  `mats`: the bf16 weight buffers, four per layer."""

    def __init__(self, mats, owner):
        self.owner = owner                       # the ShadowSet the buffers belong to (identity = validity)
        dev = mats[0].device
        self.flat = torch.empty(sum(m.numel() for m in mats), dtype=torch.bfloat16, device=dev)
        self.base = mats[0]
        rows, self.views, o = [], [], 0
        for m in mats:
            if m.dtype != torch.bfloat16 or not m.is_contiguous() or m.dim() != 2:
                raise ValueError("LayerCache: dense 2-d bf16 weight buffers expected")
            rows.append(((m.data_ptr() - self.base.data_ptr()) // 2, o, m.shape[0], m.shape[1]))
            self.views.append(self.flat[o:o + m.numel()].view(m.shape[1], m.shape[0]))
            o += m.numel()
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.n = len(mats)
        self.scratch = None
        self.fresh_in = -1

    def __reduce__(self):
        return (_no_shadow, ())                  # a cache, as ShadowSet: copies of a model rebuild it

    def refresh_transposed(self):
        scope = _FROZEN[0]
        if scope and self.fresh_in == scope and not torch.cuda.is_current_stream_capturing():
            return
        _call("cwlt_transpose_bf16_many", _lib.dev(self.base), _lib.dev(self.flat), _lib.dev(self.table), self.n,
              _lib.stream_ptr())
        self.fresh_in = scope if scope else -1

    def setup_stack(self, layer_params, weights, qkv_bias, dims):
        """Whole-stack calls (encoder._EncoderStackFn): a persistent HOST array of cwlt_encoder_layer structs with
        everything that does not change from call to call filled in once -- dims = (d_model, d_ff, n_heads), the bf16
        weight buffers (four per layer, persistent ShadowSet storage), their transposed copies, the stacked f32 Q/K/V
        biases -- and the flat list of the layers' parameters (16 per layer, encoder._layer_params order)."""
        n = len(qkv_bias)
        arr = (_lib.EncoderLayer * n)()
        D, F, H = dims
        for i in range(n):
            st = arr[i]
            st.d_model, st.d_ff, st.n_heads, st.ln_eps, st.attn_eps = D, F, H, LN_EPS, CLA_EPS
            st.wqkv, st.wo, st.w1, st.w2 = (t.data_ptr() for t in weights[4 * i:4 * i + 4])
            st.wqkv_t, st.wo_t, st.w1_t, st.w2_t = (t.data_ptr() for t in self.views[4 * i:4 * i + 4])
            st.bqkv = qkv_bias[i].data_ptr()
        self.arr = arr
        self.params = list(layer_params)
        self.all_need_grad = all(q.requires_grad for q in self.params)
        self.dims = dims
        self.qkv_bias_owner = qkv_bias
        self._grads = None

    def grad_buffer(self, plan):
        """(grads, views): a persistent f32 buffer for the parameter gradients of one backward of the whole stack and the
        view of each parameter's gradient in it, in the order of self.params."""
        n = len(self.arr)
        gf = plan.grad_floats
        got = self._grads
        if got is None or got[0].numel() != n * gf or torch.cuda.is_current_stream_capturing():
            D, F, _ = self.dims
            grads = torch.empty(n * gf, dtype=torch.float32, device=self.flat.device)
            go = list(plan.grad_off)
            views = []
            for i in range(n):
                def gv(j, *shape):
                    m = 1
                    for d_ in shape:
                        m *= d_
                    return grads[i * gf + go[j]:i * gf + go[j] + m].view(shape)
                wqkv, bqkv = gv(0, 3 * D, D), gv(1, 3 * D)
                # q / k / v weight (bias) gradients are the row blocks of the stacked ones
                views += [wqkv[:D], bqkv[:D], wqkv[D:2 * D], bqkv[D:2 * D], wqkv[2 * D:], bqkv[2 * D:], gv(2, D, D), gv(3, D),
                          gv(4, F, D), gv(5, F), gv(6, D, F), gv(7, D), gv(8, D), gv(9, D), gv(10, D), gv(11, D)]
            got = (grads, views)
            if not torch.cuda.is_current_stream_capturing():
                self._grads = got
        return got

    def get_scratch(self, nbytes):
        dev = self.flat.device
        if torch.cuda.is_current_stream_capturing():
            return torch.empty(max(1, nbytes), dtype=torch.uint8, device=dev)     # lives in the graph's pool
        if self.scratch is None or self.scratch.numel() < nbytes:
            self.scratch = torch.empty(max(1, nbytes), dtype=torch.uint8, device=dev)
        return self.scratch


def encoder_fwd(arr, n):
    _call("cwlt_encoder_fwd", arr, n, _lib.stream_ptr())


def encoder_bwd(arr, n):
    _call("cwlt_encoder_bwd", arr, n, _lib.stream_ptr())


def encoder_layer_fwd(st):
    _call("cwlt_encoder_layer_fwd", ctypes.byref(st), _lib.stream_ptr())


def encoder_layer_bwd(st):
    _call("cwlt_encoder_layer_bwd", ctypes.byref(st), _lib.stream_ptr())


def gemm_nt_mul_supported(a, w, g):
    return (a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and g.dtype == torch.bfloat16 and a.dim() == 2
            and w.dim() == 2 and g.dim() == 2 and w.shape[0] % 256 == 0 and a.shape[1] % 64 == 0
            and a.shape[1] == w.shape[1] and g.shape == (a.shape[0], w.shape[0])
            and all(t.stride(1) == 1 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0 for t in (a, w, g)))


def gemm_nt_mul(a, w, g, want_colsum=True):
    """c = (a @ w.T) * g in one kernel, + column sums of c.  a (M, K), w (N, K), g (M, N) bf16 -> c (M, N) bf16
    [, colsum (N) f32].  The FFN backward: a = dy, w = linear2.weight.T (contiguous), g = gd."""
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    c = torch.empty((M, N), dtype=a.dtype, device=a.device)
    part = cs = None
    if want_colsum:
        part = torch.empty(lib.cwlt_gemm_nt_tiles(M) * N, dtype=torch.float32, device=a.device)
        cs = torch.empty(N, dtype=torch.float32, device=a.device)
    _call("cwlt_gemm_nt_mul", _lib.dev(a, "a"), _lib.dev(w, "w"), _lib.dev(g, "g"), _lib.dev(c), _lib.opt(part),
          _lib.opt(cs), M, N, K, a.stride(0), w.stride(0), g.stride(0), N, _lib.stream_ptr(), work=2.0 * M * N * K)
    return (c, cs) if want_colsum else c


def ffn1_fused_supported(x, w, bias):
    return (x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and bias.dtype == torch.float32 and x.dim() == 2
            and w.dim() == 2 and w.shape[0] % 256 == 0 and x.shape[1] % 64 == 0 and x.shape[1] == w.shape[1]
            and bias.numel() == w.shape[0] and bias.is_contiguous() and bias.data_ptr() % 16 == 0
            and all(t.stride(1) == 1 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0 for t in (x, w)))


def ffn1_gelu_dropout(x, w, bias, p=0.0, seed=0):
    """g = dropout(gelu(x @ w.T + bias)) and gd = mask / (1 - p) * gelu'(x @ w.T + bias) in ONE kernel (the GEMM's
    epilogue): what `torch.mm` + `gelu_fwd(..., gd_inplace=True)` produce, without the pre-activation's round trip
    through HBM.  x (M, K), w (N, K) bf16, bias (N) f32 -> g, gd (M, N) bf16."""
    _lib.load()
    M, K = x.shape
    N = w.shape[0]
    g = torch.empty((M, N), dtype=x.dtype, device=x.device)
    gd = torch.empty((M, N), dtype=x.dtype, device=x.device)
    _call("cwlt_gemm_nt_bias_gelu_dropout", _lib.dev(x, "x"), _lib.dev(w, "w"), _lib.dev(bias), _lib.dev(g), _lib.dev(gd),
          M, N, K, x.stride(0), w.stride(0), float(p), int(seed), _seed_base(), _lib.stream_ptr(), work=2.0 * M * N * K)
    return g, gd


# rows from which the one-kernel residual block is used (128-row tiles, one workgroup per CU: below ~2 tiles per CU the
# hipBLASLt GEMM + LayerNorm kernel pair fills the chip better)
LINEAR_LN_MIN_ROWS = int(os.environ.get("CWLT_LINEAR_LN_MIN_ROWS", 65536))
# ... and the reduction lengths: at K = 512 (the out-projection) the one-kernel form takes 0.61 ms against 0.36 + 0.44
# for the pair at R = 524 288; at K = 2048 (linear2) its serial epilogue (one workgroup per CU: nothing runs beside
# it) costs more than the pair's extra stream, 1.43 against 0.96 + 0.42 ms (profiles/r03_linear_ln_microbench.txt)
LINEAR_LN_MAX_K = int(os.environ.get("CWLT_LINEAR_LN_MAX_K", 1024))
FUSED_LINEAR_LN = os.environ.get("CWLT_FUSED_LINEAR_LN", "1") != "0"


def linear_ln_supported(a, w, x):
    return (FUSED_LINEAR_LN and a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x is not None
            and x.dtype == torch.bfloat16 and a.dim() == 2 and w.dim() == 2 and w.shape[0] == 512
            and a.shape[1] == w.shape[1] and a.shape[1] % 64 == 0 and x.shape == (a.shape[0], 512) and x.is_contiguous()
            and a.shape[0] >= LINEAR_LN_MIN_ROWS and a.shape[1] <= LINEAR_LN_MAX_K
            and all(t.stride(1) == 1 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0 for t in (a, w)))


def linear_ln(a, w, bias, x, gamma, beta, eps=LN_EPS, p=0.0, seed=0):
    """s = x + dropout(a @ w.T + bias); y = LayerNorm(s) in ONE kernel (the GEMM's epilogue): what `torch.addmm` +
    `ln_fwd` produce without the projection's output ever reaching HBM.  a (M, K), w (512, K), x (M, 512) bf16;
    bias, gamma, beta (512) f32 -> s, y (M, 512) bf16, mean, rstd (M) f32."""
    _lib.load()
    M, K = a.shape
    N = w.shape[0]
    if x.shape != (M, N) or not x.is_contiguous() or x.dtype != a.dtype:
        raise ValueError("linear_ln: the residual must be a dense (%d, %d) tensor of the operands' dtype" % (M, N))
    s = torch.empty((M, N), dtype=a.dtype, device=a.device)
    y = torch.empty((M, N), dtype=a.dtype, device=a.device)
    mean = torch.empty(M, dtype=torch.float32, device=a.device)
    rstd = torch.empty(M, dtype=torch.float32, device=a.device)
    _call("cwlt_gemm_nt_bias_dropout_add_layernorm", _lib.dev(a, "a"), _lib.dev(w, "w"), _lib.dev(bias), _lib.dev(x, "x"),
          _lib.dev(gamma), _lib.dev(beta), _lib.dev(s), _lib.dev(y), _lib.dev(mean), _lib.dev(rstd), M, N, K, a.stride(0),
          w.stride(0), float(eps), float(p), int(seed), _seed_base(), _lib.stream_ptr(), work=2.0 * M * N * K)
    return s, y, mean, rstd


def gelu_bwd(dg, h, bias, p=0.0, seed=0, want_dbias=True):
    lib = _lib.load()
    rows, F = h.shape
    dg = dg.contiguous()
    dh = torch.empty_like(h)
    part = dbias = None
    if want_dbias:
        part = torch.empty(lib.cwlt_rowslab_blocks(rows) * F, dtype=torch.float32, device=h.device)
        dbias = torch.empty(F, dtype=torch.float32, device=h.device)
    _call("cwlt_bias_gelu_dropout_bwd", _lib.dev(dg, "dg"), _lib.dev(h), _lib.opt(bias), _lib.dev(dh),
                                              _lib.opt(part), _lib.opt(dbias), rows, F, float(p), int(seed), _seed_base(),
                                              _lib.dtype_code(h.dtype), _lib.stream_ptr())
    return dh, dbias


def wgrad_supported(a, b):
    return (a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.dim() == 2 and b.dim() == 2
            and a.shape[1] % 8 == 0 and b.shape[1] % 8 == 0 and a.stride(1) == 1 and b.stride(1) == 1
            and a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0)


def wgrad(a, b, out=None, accumulate=False):
    """out (N1, N2) f32 (+)= a^T b;  a (M, N1), b (M, N2) bf16 row-major.  The weight gradient of a Linear with
    input b and output gradient a.  `out` may be a dense f32 .grad view (written in place)."""
    lib = _lib.load()
    M, N1 = a.shape
    N2 = b.shape[1]
    S = lib.cwlt_wgrad_splits(M, N1, N2)
    part = torch.empty(S * N1 * N2, dtype=torch.float32, device=a.device)
    if out is None:
        out = torch.empty((N1, N2), dtype=torch.float32, device=a.device)
        accumulate = False
    _call("cwlt_wgrad_bf16", _lib.dev(a, "a"), _lib.dev(b, "b"), _lib.dev(part), _lib.dev(out), M, N1, N2,
          a.stride(0), b.stride(0), 1 if accumulate else 0, _lib.stream_ptr(), work=2.0 * M * N1 * N2)
    return out


def wgrad_group(pairs, accumulate=False, outs=None):
    """[(a_i, b_i)] (at most four, all with the same number of rows, widths multiples of 256) -> [a_i^T b_i] f32, as ONE
    launch + one reduce launch (cwlt_wgrad_bf16_group): bit-identical to `wgrad` on every pair."""
    lib = _lib.load()
    n = len(pairs)
    M = pairs[0][0].shape[0]
    parts, res = [], []
    for i, (a, b) in enumerate(pairs):
        N1, N2 = a.shape[1], b.shape[1]
        parts.append(torch.empty(lib.cwlt_wgrad_splits(M, N1, N2) * N1 * N2, dtype=torch.float32, device=a.device))
        res.append(outs[i] if outs is not None else torch.empty((N1, N2), dtype=torch.float32, device=a.device))
    vp = ctypes.c_void_p * n
    _call("cwlt_wgrad_bf16_group", vp(*[a.data_ptr() for a, _ in pairs]), vp(*[b.data_ptr() for _, b in pairs]),
          vp(*[t.data_ptr() for t in parts]), vp(*[t.data_ptr() for t in res]),
          (ctypes.c_int * n)(*[a.shape[1] for a, _ in pairs]), (ctypes.c_int * n)(*[b.shape[1] for _, b in pairs]),
          (ctypes.c_int64 * n)(*[a.stride(0) for a, _ in pairs]), (ctypes.c_int64 * n)(*[b.stride(0) for _, b in pairs]),
          n, M, 1 if accumulate else 0, _lib.stream_ptr())
    return res


class LinearWgradFn(torch.autograd.Function):
    """y = x @ w.T + b for a bf16 activation x and f32 master parameters (in_linear, the fused head projection).
    Same forward as F.linear on the bf16 casts; the backward takes the weight gradient with the split-K MFMA kernel
    straight in f32 (no bf16 rounding of the gradient, no .float() pass) and the bias gradient with the
    deterministic column sum."""

    @staticmethod
    def forward(ctx, x, w, b):
        w16, b16 = _cast_pair(w, b, x.dtype)
        ctx.save_for_backward(x)
        ctx.w16 = w16              # a persistent shadow buffer when w is a Parameter: not version-tracked on purpose
        if gemm_bf16_supported(x, w16):
            return gemm_bf16(x, w16, _f32(b))
        return torch.addmm(b16, x, w16.t())

    @staticmethod
    def backward(ctx, dy):
        (x,), w16 = ctx.saved_tensors, ctx.w16
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = (gemm_bf16(dy, w16.t().contiguous()) if gemm_bf16_supported(dy, w16.t(), transposed_w=True)
                  else torch.mm(dy, w16))
        dw = wgrad(dy, x) if wgrad_supported(dy, x) else torch.mm(dy.t(), x).float()
        return dx, dw, colsum(dy)


def _cast_pair(w, b, dtype):
    """(w, b) in `dtype`; for a Parameter pair through a ShadowSet kept on the weight (no per-call cast kernels)."""
    if w.dtype == dtype and b.dtype == dtype:
        return w, b
    if isinstance(w, torch.nn.Parameter) and isinstance(b, torch.nn.Parameter):
        sh = getattr(w, "_cwlt_shadow", None)
        if sh is None or not sh.matches(dtype, w.device) or sh.src[1] is not b:
            sh = w._cwlt_shadow = ShadowSet([(w,), (b,)], dtype)
        w16, b16 = sh.refresh()
        return w16, b16
    return w.to(dtype), b.to(dtype)


def linear(x, w, b):
    """F.linear(x, w.to(x.dtype), b.to(x.dtype)) with the f32-gradient backward above when x is bf16."""
    if x.dtype == torch.bfloat16 and x.dim() == 2 and x.is_contiguous() and torch.is_grad_enabled():
        return LinearWgradFn.apply(x, w, b)
    if not torch.is_grad_enabled():
        w16, b16 = _cast_pair(w, b, x.dtype)
        if x.dim() == 2 and gemm_bf16_supported(x, w16):
            return gemm_bf16(x, w16, _f32(b))
        return torch.nn.functional.linear(x, w16, b16)
    return torch.nn.functional.linear(x, w.to(x.dtype), b.to(x.dtype))


def colsum(x):
    """Deterministic column sums of a (rows, ncols) matrix (row-strided ok) -> (ncols) f32."""
    lib = _lib.load()
    rows, ncols = x.shape
    if x.stride(1) != 1 or x.stride(0) % 8 != 0 or x.data_ptr() % 16 != 0 or ncols % 8 != 0:
        return x.float().sum(0)
    part = torch.empty(lib.cwlt_colsum_blocks(rows) * ncols, dtype=torch.float32, device=x.device)
    out = torch.empty(ncols, dtype=torch.float32, device=x.device)
    _call("cwlt_colsum", _lib.dev(x, "x"), _lib.dev(part), _lib.dev(out), rows, ncols, x.stride(0),
                               _lib.dtype_code(x.dtype), _lib.stream_ptr())
    return out


def posenc_dropout(x, pe, T, p=0.0, seed=0):
    """x (rows, D) dense; pe (max_len, D) f32 or None -> dropout(x + pe[r % T])."""
    lib = _lib.load()
    rows, D = x.shape
    x = x.contiguous()
    y = torch.empty_like(x)
    _call("cwlt_posenc_dropout", _lib.dev(x, "x"), _lib.opt(pe), _lib.dev(y), rows, int(T), D, float(p),
                                       int(seed), _seed_base(), _lib.dtype_code(x.dtype), _lib.stream_ptr())
    return y


class PosEncDropoutFn(torch.autograd.Function):
    """PositionalEncoding.forward (dqn_policy/model.py:90-92): dropout(x + pe[:, :T])."""

    @staticmethod
    def forward(ctx, x, pe, p, seed):
        N, T, D = x.shape
        if pe is not None and T > pe.shape[-2]:
            raise RuntimeError("sequence length %d exceeds the positional table (%d)" % (T, pe.shape[-2]))
        ctx.p, ctx.seed, ctx.T = p, seed, T
        return posenc_dropout(x.reshape(N * T, D), None if pe is None else pe.reshape(-1, D), T, p,
                              seed).view(N, T, D)

    @staticmethod
    def backward(ctx, dy):
        if ctx.p == 0:
            return dy, None, None, None
        N, T, D = dy.shape
        return posenc_dropout(dy.reshape(N * T, D), None, T, ctx.p, ctx.seed).view(N, T, D), None, None, None


# --------------------------------------------------------------------------------------------------
# compound-word embedding
# --------------------------------------------------------------------------------------------------
class CWEmbedFn(torch.autograd.Function):
    """tokens (..., A) int64 + A tables -> (..., sum widths): lut_f(x_f) * sqrt(d_f), concatenated."""

    @staticmethod
    def forward(ctx, tokens, out_dtype, *tables):
        lib = _lib.load()
        A = len(tables)
        if tokens.shape[-1] != A:
            raise ValueError("tokens carry %d attributes, model has %d tables" % (tokens.shape[-1], A))
        if tokens.dtype != torch.int64:
            raise TypeError("tokens must be int64 (the reference indexes nn.Embedding with .long())")
        tabs = [_f32(t) for t in tables]
        widths = [t.shape[1] for t in tabs]
        nrows = [t.shape[0] for t in tabs]
        tok = tokens.reshape(-1, A).contiguous()
        rows = tok.shape[0]
        dcat = sum(widths)
        out = torch.empty((rows, dcat), dtype=out_dtype, device=tok.device)
        _call("cwlt_cw_embed_fwd", _lib.dev(tok, "tokens"), _lib.ptr_array(tabs), _lib.int_array(widths),
                                         _lib.int_array(nrows), A, _lib.dev(out), rows, dcat,
                                         _lib.dtype_code(out_dtype), _lib.stream_ptr())
        ctx.save_for_backward(tok)
        ctx.widths, ctx.nrows = widths, nrows
        return out.view(*tokens.shape[:-1], dcat)

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        (tok,) = ctx.saved_tensors
        widths, nrows = ctx.widths, ctx.nrows
        A = len(widths)
        dcat = sum(widths)
        d2 = dout.reshape(-1, dcat).contiguous()
        rows = d2.shape[0]
        total = sum(w * n for w, n in zip(widths, nrows))
        part = torch.empty(lib.cwlt_embed_splits(rows) * total, dtype=torch.float32, device=d2.device)
        flat = torch.empty(total, dtype=torch.float32, device=d2.device)
        _call("cwlt_cw_embed_bwd", _lib.dev(tok), _lib.int_array(widths), _lib.int_array(nrows), A,
                                         _lib.dev(d2, "dout"), _lib.dev(part), _lib.dev(flat), rows, dcat,
                                         _lib.dtype_code(d2.dtype), _lib.stream_ptr())
        grads, o = [], 0
        for w, n in zip(widths, nrows):
            grads.append(flat[o:o + w * n].view(n, w))
            o += w * n
        return (None, None) + tuple(grads)


def cw_embed(tokens, tables, out_dtype=torch.float32):
    return CWEmbedFn.apply(tokens, out_dtype, *tables)


EMBED_PROJ = os.environ.get("CWLT_EMBED_PROJ", "1") != "0"
# below this many token rows the front's three kernels cost nothing on the GPU and the ~14 small launches that build the
# projected tables are a net loss for the launch-bound RL steps
EMBED_PROJ_MIN_ROWS = int(os.environ.get("CWLT_EMBED_PROJ_MIN_ROWS", "8192"))


class EmbedProjFn(torch.autograd.Function):
    """The model's input front in one pass (dqn_policy/model.py:206-223 + 90-92):
    dropout(in_linear(cat_f(lut_f(x_f) * sqrt(d_f))) + pe[:, :T]) = dropout(sum_f P_f[x_f] + b + pe) with the projected
    tables P = cat_f(sqrt(d_f) lut_f . W_in[:, cols_f]^T) ((sum n_f, D) f32, built by the caller with differentiable
    torch ops, so autograd carries dP on to the tables and to in_linear.weight).  tokens (N, T, A) int64 ->
    (N, T, D) of `out_dtype`.  Backward: dP = the scatter-add of the gradient in front of the dropout over the ids (the
    one-hot MFMA GEMM of cwlt_cw_embed_bwd), db = the column sum of one attribute's rows of dP."""

    @staticmethod
    def forward(ctx, tokens, tproj, bias, pe, nrows, p, seed, out_dtype):
        if tokens.dtype != torch.int64:
            raise TypeError("tokens must be int64 (the reference indexes nn.Embedding with .long())")
        N, T, A = tokens.shape
        if A != len(nrows):
            raise ValueError("tokens carry %d attributes, model has %d tables" % (A, len(nrows)))
        D = tproj.shape[1]
        if tproj.shape[0] != sum(nrows) or bias.shape != (D,):
            raise ValueError("projected tables / bias do not match the vocabularies")
        if pe is not None:
            pe = pe.reshape(-1, D)
            if T > pe.shape[0]:
                raise RuntimeError("sequence length %d exceeds the positional table (%d)" % (T, pe.shape[0]))
        tok = tokens.reshape(-1, A).contiguous()
        rows = tok.shape[0]
        tp = tproj.detach().to(out_dtype).contiguous()
        out = torch.empty((rows, D), dtype=out_dtype, device=tok.device)
        _call("cwlt_cw_embed_proj_fwd", _lib.dev(tok, "tokens"), _lib.dev(tp), _lib.int_array(nrows), A,
              _lib.dev(_f32(bias)), _lib.opt(None if pe is None else _f32(pe)), _lib.dev(out), rows, int(T), D, float(p),
              int(seed), _seed_base(), _lib.dtype_code(out_dtype), _lib.stream_ptr())
        ctx.save_for_backward(tok)
        ctx.cfg = (list(nrows), T, D, p, seed)
        return out.view(N, T, D)

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        (tok,) = ctx.saved_tensors
        nrows, T, D, p, seed = ctx.cfg
        A = len(nrows)
        d2 = dout.reshape(-1, D)
        dpre = posenc_dropout(d2, None, T, p, seed) if p > 0 else d2.contiguous()
        rows = dpre.shape[0]
        total = sum(nrows) * D
        part = torch.empty(lib.cwlt_embed_splits(rows) * total, dtype=torch.float32, device=dpre.device)
        dtp = torch.empty((sum(nrows), D), dtype=torch.float32, device=dpre.device)
        _call("cwlt_cw_embed_proj_bwd", _lib.dev(tok), _lib.int_array(nrows), A, D, _lib.dev(dpre, "dpre"), _lib.dev(part),
              _lib.dev(dtp), rows, D, _lib.dtype_code(dpre.dtype), _lib.stream_ptr())
        dbias = dtp[:nrows[0]].sum(0)             # every token row has exactly one id of attribute 0
        return None, dtp, dbias, None, None, None, None, None


def embed_proj(tokens, tables, w_in, b_in, pe, p, seed, out_dtype):
    """tokens (N, T, A); tables: A embedding weights (n_f, d_f) f32; w_in (D, sum d_f), b_in (D); pe (.., max_len, D)."""
    cols, parts = 0, []
    for t in tables:
        d = t.shape[1]
        parts.append(torch.nn.functional.linear(t * math.sqrt(d), w_in[:, cols:cols + d]))
        cols += d
    tproj = torch.cat(parts, 0)
    return EmbedProjFn.apply(tokens, tproj, b_in, pe, tuple(t.shape[0] for t in tables), p, seed, out_dtype)


# --------------------------------------------------------------------------------------------------
# per-attribute heads
# --------------------------------------------------------------------------------------------------
def heads_forward(logits, n_class, target=None, mask=None, want_argmax=False, want_pmax=False, want_probs=False):
    """logits (rows, ld).  -> dict(loss_sum (A) f32 | argmax (rows, A) | pmax | probs (rows, sum n))."""
    lib = _lib.load()
    rows, ld = logits.shape[0], logits.stride(0)
    A = len(n_class)
    dev = logits.device
    res = {}
    loss_part = loss_sum = None
    if target is not None:
        target = target.reshape(rows, A).contiguous()
        loss_part = torch.empty(lib.cwlt_heads_blocks(rows) * A, dtype=torch.float32, device=dev)
        loss_sum = res["loss_sum"] = torch.empty(A, dtype=torch.float32, device=dev)
    if mask is not None:
        mask = mask.reshape(rows).float().contiguous()
    am = res["argmax"] = torch.empty((rows, A), dtype=torch.int64, device=dev) if want_argmax else None
    pm = res["pmax"] = torch.empty((rows, A), dtype=torch.float32, device=dev) if want_pmax else None
    ncol = sum(n_class)
    pr = res["probs"] = torch.empty((rows, ncol), dtype=torch.float32, device=dev) if want_probs else None
    _call("cwlt_heads_fwd", _lib.dev(logits, "logits"), _lib.int_array(n_class), A, _lib.opt(target),
                                  _lib.opt(mask), _lib.opt(loss_part), _lib.opt(loss_sum), _lib.opt(am),
                                  _lib.opt(pm), _lib.opt(pr), rows, ld, ncol, _lib.dtype_code(logits.dtype),
                                  _lib.stream_ptr())
    return res


class HeadsCEFn(torch.autograd.Function):
    """6 x compute_loss (dqn_policy/model.py:163-197): returns the (A) vector of masked-mean CE losses."""

    @staticmethod
    def forward(ctx, logits, target, mask, n_class):
        logits = logits.contiguous()   # dense (rows, W >= sum n_class): the bwd kernel zero-fills cols up to W
        rows = logits.shape[0]
        mask_f = mask.reshape(rows).float().contiguous()
        target = target.reshape(rows, len(n_class)).contiguous()
        res = heads_forward(logits, n_class, target, mask_f)
        msum = mask_f.sum()
        ctx.save_for_backward(logits, target, mask_f, msum)
        ctx.n_class = n_class
        return res["loss_sum"] / msum

    @staticmethod
    def backward(ctx, gloss):
        lib = _lib.load()
        logits, target, mask_f, msum = ctx.saved_tensors
        n_class = ctx.n_class
        coef = (gloss.float() / msum).contiguous()
        dlogits = torch.empty_like(logits)
        _call("cwlt_heads_ce_bwd", _lib.dev(logits), _lib.int_array(n_class), len(n_class), _lib.dev(target),
                                         _lib.dev(mask_f), _lib.dev(coef), _lib.dev(dlogits), logits.shape[0],
                                         logits.stride(0), _lib.dtype_code(logits.dtype), _lib.stream_ptr())
        return dlogits, None, None, None


def heads_ce(logits, target, mask, n_class):
    """logits (rows, >= sum n_class) dense rows; -> (A) losses = sum(mask * nll) / sum(mask) per attribute."""
    return HeadsCEFn.apply(logits, target, mask, tuple(int(n) for n in n_class))


# --------------------------------------------------------------------------------------------------
# banded attention (AIRL discriminator)
# --------------------------------------------------------------------------------------------------
def band_attention(q, k, v, mask, window, p=0.0, seed=0, want_lse=False):
    """q, k, v: (B, L, H, 64) views; mask (B, L) nonzero = attend or None; window = one-sided width.
    -> (B, L, H*64)  [, lse (B, H, L) f32 when want_lse]."""
    B, L, H, D = q.shape
    q, ldq = _as_rows(q)
    k, ldk = _as_rows(k)
    v, ldv = _as_rows(v)
    out = torch.empty((B, L, H * D), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, H, L), dtype=torch.float32, device=q.device) if want_lse else None
    if mask is not None:
        mask = mask.reshape(B, L).float().contiguous()
    _call("cwlt_band_attn_fwd", _lib.dev(q, "q"), _lib.dev(k, "k"), _lib.dev(v, "v"), _lib.opt(mask), _lib.dev(out),
          _lib.opt(lse), B, H, L, D, int(window), ldq, ldk, ldv, H * D, 1.0 / math.sqrt(D), float(p), int(seed), _seed_base(),
          _lib.dtype_code(q.dtype), _lib.stream_ptr())
    return (out, lse) if want_lse else out


class BandAttentionFn(torch.autograd.Function):
    """Band attention over one fused projection buffer qkv (B, L, 3, H, 64) -> (B, L, H*64); the backward
    writes the three gradients straight into the column blocks of one (B, L, 3, H, 64) buffer."""

    @staticmethod
    def forward(ctx, qkv, mask, window, p, seed):
        qkv = qkv.contiguous()
        B, L, _, H, D = qkv.shape
        if mask is not None:
            mask = mask.reshape(B, L).float().contiguous()
        out, lse = band_attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], mask, window, p, seed, want_lse=True)
        ctx.save_for_backward(qkv, out, lse, mask)
        ctx.cfg = (int(window), float(p), int(seed))
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, mask = ctx.saved_tensors
        window, p, seed = ctx.cfg
        B, L, _, H, D = qkv.shape
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        ld = 3 * H * D
        views = [_as_rows(t[:, :, i])[0] for t in (qkv, dqkv) for i in range(3)]
        q, k, v, dq, dk, dv = views
        if any(t.data_ptr() != src[:, :, i].data_ptr() for src, ts in ((qkv, views[:3]), (dqkv, views[3:]))
               for i, t in enumerate(ts)):
            raise RuntimeError("band attention backward needs the fused (B, L, 3, H, 64) layout")
        _call("cwlt_band_attn_bwd", _lib.dev(q, "q"), _lib.dev(k, "k"), _lib.dev(v, "v"), _lib.opt(mask),
              _lib.dev(out), _lib.dev(lse), _lib.dev(dout, "dout"), _lib.dev(dq, "dq"), _lib.dev(dk, "dk"),
              _lib.dev(dv, "dv"), B, H, L, D, window, ld, ld, ld, H * D, H * D, ld, ld, ld, 1.0 / math.sqrt(D), p, seed, _seed_base(),
              _lib.dtype_code(qkv.dtype), _lib.stream_ptr())
        return dqkv, None, None, None, None


class AddDropoutLayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x + dropout(a)) (x may be None) -- the post-LN residual block of BertSelfOutput /
    LongformerOutput, differentiable.  gamma / beta f32."""

    @staticmethod
    def forward(ctx, x, a, gamma, beta, eps, p, seed):
        g, b = _f32(gamma), _f32(beta)
        s, y, mean, rstd = ln_fwd(x, a, g, b, eps, p, seed, save_s=True)
        ctx.save_for_backward(s, g, mean, rstd)
        ctx.cfg = (x is not None, float(p), int(seed))
        return y

    @staticmethod
    def backward(ctx, dy):
        s, g, mean, rstd = ctx.saved_tensors
        has_x, p, seed = ctx.cfg
        ds, da, dg, db, _ = ln_bwd(dy, None, s, g, mean, rstd, p, seed, want_dbias=False)
        return (ds if has_x else None), da, dg, db, None, None, None


class BiasGeluDropoutFn(torch.autograd.Function):
    """g = dropout(gelu(h + bias)), exact-erf gelu; bias f32."""

    @staticmethod
    def forward(ctx, h, bias, p, seed):
        b = _f32(bias)
        h = h.contiguous()
        ctx.save_for_backward(h, b)
        ctx.cfg = (float(p), int(seed))
        return gelu_fwd(h, b, p, seed)

    @staticmethod
    def backward(ctx, dg):
        h, b = ctx.saved_tensors
        p, seed = ctx.cfg
        dh, dbias = gelu_bwd(dg, h, b, p, seed, want_dbias=True)
        return dh, dbias, None, None


def recurrent_cla_step(qkv, S, Z, H, eps=CLA_EPS):
    """qkv (N, 3*H*64) fused projections of ONE token per sequence; S (N, H, 64, 64), Z (N, H, 64) f32 states,
    updated in place.  -> (N, H*64)."""
    N, W = qkv.shape
    D = W // 3
    qkv = qkv.contiguous()
    out = torch.empty((N, D), dtype=qkv.dtype, device=qkv.device)
    esz = qkv.element_size()
    base = qkv.data_ptr()
    import ctypes
    _call("cwlt_recurrent_cla_step", ctypes.c_void_p(base), ctypes.c_void_p(base + D * esz),
          ctypes.c_void_p(base + 2 * D * esz), _lib.dev(S, "S"), _lib.dev(Z, "Z"), _lib.dev(out), N, H, D // H,
          W, W, W, D, float(eps), _lib.dtype_code(qkv.dtype), _lib.stream_ptr())
    return out


def sqrt_width(w):
    return math.sqrt(w)


# --------------------------------------------------------------------------------------------------
# generation: the decode step's GEMV building block (csrc/decode.hip); the whole step is generation.DecodeSession
# --------------------------------------------------------------------------------------------------
def decode_gemv(w, bias, x, ln=None, ln2=None, eps=1e-5, res=None, act=None, want_normed=False):
    """out[n] = epi(W . pro(x[n]) + bias): pro = LayerNorm(*ln) then LayerNorm(*ln2) when given; epi = exact GELU when
    act == "gelu", then + res[n].  x (n, K) f32, w (n_out, K) f32 -> (n, n_out) f32 [, the normalised x]."""
    if x.dtype != torch.float32 or w.dtype != torch.float32:
        raise TypeError("decode_gemv computes in f32")
    n, K = x.shape
    n_out = w.shape[0]
    x, w = x.contiguous(), w.contiguous()
    out = torch.empty((n, n_out), dtype=torch.float32, device=x.device)
    xn = torch.empty_like(x) if (want_normed and ln is not None) else None
    if res is not None:
        res = res.contiguous()
    p = lambda pair, i: _lib.opt(None if pair is None else pair[i].contiguous())
    _call("cwlt_decode_gemv", _lib.dev(w, "w"), _lib.opt(bias), _lib.dev(x, "x"), p(ln, 0), p(ln, 1), p(ln2, 0),
          p(ln2, 1), float(eps), _lib.opt(res), _lib.dev(out), _lib.opt(xn), n_out, K, 1 if act == "gelu" else 0, n,
          K, n_out, n_out, K, _lib.stream_ptr())
    return (out, xn) if want_normed else out


def sample_categorical(logits, n_class, tokens, seed, counter=None, song=None, temperature=None, top_p=None):
    """tokens[row, a] ~ Categorical(softmax(logits[row, segment a] / temperature[a])) on the device
    (csrc/sample.hip; ppo_policy/inference.py:115-141).  logits (rows, >= sum n_class) f32; tokens (rows, A) int64
    written in place; counter: device int64 scalar tensor that keys the draw (and indexes `song` (T, rows, A));
    top_p[a] < 1 (or None = off) samples attribute a from its nucleus (dqn_policy/model.py:33-47)."""
    if logits.dtype != torch.float32 or tokens.dtype != torch.int64:
        raise TypeError("sample_categorical takes f32 logits and int64 tokens")
    rows, A = logits.shape[0], len(n_class)
    if tokens.numel() != rows * A or not tokens.is_contiguous():
        raise ValueError("tokens must be a contiguous (rows, n_attr) buffer")
    if logits.stride(-1) != 1:
        logits = logits.contiguous()
    temp = None if temperature is None else (ctypes.c_float * A)(*[float(t) for t in temperature])
    topp = None if top_p is None else (ctypes.c_float * A)(*[1.0 if p is None else float(p) for p in top_p])
    _call("cwlt_sample_categorical", _lib.dev(logits, "logits"), _lib.int_array(n_class), temp, topp, A, rows,
          logits.stride(0), int(seed) & 0xFFFFFFFFFFFFFFFF, _lib.opt(counter), _lib.dev(tokens, "tokens"), _lib.opt(song),
          0 if song is None else song.shape[0], _lib.stream_ptr())
    return tokens

