"""GPU: the split-K MFMA weight-gradient GEMM vs an fp64 reference."""
import pytest
import torch

import rlmg_amd  # noqa: F401
from rlmg_amd import ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N1,N2", [(32, 256, 256), (1000, 512, 256), (4096, 2048, 512), (777, 256, 1536),
                                     (1000, 384, 512), (4096, 512, 1216), (333, 8, 264), (64, 1216, 384)])   # edge tiles
def test_wgrad_matches_fp64(cuda, M, N1, N2):
    g = torch.Generator().manual_seed(M)
    a = torch.randn(M, N1, generator=g).bfloat16()
    b = torch.randn(M, N2, generator=g).bfloat16()
    ref = a.double().t() @ b.double()
    got = ops.wgrad(a.to(cuda), b.to(cuda))
    err = (got.cpu().double() - ref).abs().max().item()
    assert err <= 1e-4 * max(1.0, ref.abs().max().item()), err       # f32 accumulation of exact bf16 products
    acc = torch.ones(N1, N2, device=cuda)
    ops.wgrad(a.to(cuda), b.to(cuda), out=acc, accumulate=True)
    assert (acc.cpu().double() - (ref + 1)).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


def test_wgrad_strided_operands_and_determinism(cuda):
    big = torch.randn(3000, 3, 512, device=cuda).bfloat16()
    x = torch.randn(3000, 512, device=cuda).bfloat16()
    a = big.view(3000, 1536)
    r1 = ops.wgrad(a, x)
    r2 = ops.wgrad(a, x)
    assert torch.equal(r1, r2)
    ref = a.double().t() @ x.double()
    assert (r1.double() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


def test_wgrad_edge_tile_on_column_slices(cuda):
    """Partial 256-tiles read on into the next row: with operands that are column slices of wider tensors the
    neighbouring columns must not leak into the result."""
    wide_a = torch.randn(900, 640, device=cuda).bfloat16()
    wide_b = torch.randn(900, 1400, device=cuda).bfloat16()
    a, b = wide_a[:, 128:128 + 384], wide_b[:, 64:64 + 1216]
    got = ops.wgrad(a, b)
    ref = a.double().t() @ b.double()
    assert got.shape == (384, 1216)
    assert (got.double() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


def test_wgrad2_switch_matches_fp64_in_a_process_of_its_own(cuda):
    """CWLT_WGRAD_V2=1 (read once per process): the 8-wave form on gemm_bf16.hip's main loop (csrc/wgrad2.hip) against the
    f64 reference -- ragged token counts (slices ending inside a K-tile), strided operands (column blocks of a wider
    tensor), accumulate, and bit-for-bit determinism."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys
sys.path.insert(0, %r)
import torch
import rlmg_amd
from rlmg_amd import ops
dev = torch.device("cuda:0")
for M, N1, N2 in ((70000, 512, 512), (33333, 2048, 512), (40000, 512, 1536), (2048, 256, 256)):
    g = torch.Generator().manual_seed(M)
    wide = torch.randn(M, N1 + 256, generator=g).bfloat16().to(dev)
    a = wide[:, 128:128 + N1]                       # row stride N1 + 256, 16-byte aligned start
    b = torch.randn(M, N2, generator=g).bfloat16().to(dev)
    ref = a.double().t() @ b.double()
    got = ops.wgrad(a, b)
    assert torch.equal(got, ops.wgrad(a, b))
    scale = max(1.0, ref.abs().max().item())
    assert (got.double() - ref).abs().max().item() <= 1e-4 * scale, (M, N1, N2)
    acc = torch.ones(N1, N2, device=dev)
    ops.wgrad(a, b, out=acc, accumulate=True)
    assert (acc.double() - (ref + 1)).abs().max().item() <= 1e-4 * scale, (M, N1, N2)
print("wgrad2 ok")
''' % root
    env = dict(os.environ, CWLT_WGRAD_V2="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "wgrad2 ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("M", [50, 1500, 8192])
def test_grouped_launch_equals_the_four_separate_ones(cuda, M):
    """cwlt_wgrad_bf16_group: the four weight gradients of an encoder layer (linear2, linear1, out-projection, Q|K|V) over
    the same token rows in ONE launch + one reduce: the same slices, the same summation order -> bit for bit the results
    of four cwlt_wgrad_bf16 calls (M = 8 192: the products cut the rows into different numbers of slices); accumulate
    form; fewer than four products."""
    g = torch.Generator().manual_seed(M)
    mk = lambda n: torch.randn(M, n, generator=g).bfloat16().to(cuda)
    pairs = [(mk(512), mk(2048)), (mk(2048), mk(512)), (mk(512), mk(512)), (mk(1536), mk(512))]
    one = [ops.wgrad(a, b) for a, b in pairs]
    grp = ops.wgrad_group(pairs)
    for x, y in zip(one, grp):
        assert torch.equal(x, y)
    outs = [torch.full_like(x, 0.5) for x in one]
    ops.wgrad_group(pairs, accumulate=True, outs=outs)
    for x, y in zip(one, outs):
        assert torch.equal(x + 0.5, y)
    two = ops.wgrad_group(pairs[2:])
    assert torch.equal(two[0], one[2]) and torch.equal(two[1], one[3])

