"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/cwlt.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    import rlmg_amd  # noqa: F401
    from rlmg_amd import _lib
    return _lib


def _declared():
    text = open(os.path.join(ROOT, "include", "cwlt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t)\s+(cwlt_\w+)\s*\(", text)))


def test_header_symbols_are_exported(built):
    lib = built.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libcwlt.so does not export %s" % n


def test_binding_covers_header(built):
    assert sorted(built.exported_names()) == _declared()


def test_abi_version(built):
    assert built.load().cwlt_abi_version() == built.ABI_VERSION


def test_argument_validation_without_gpu(built):
    """Entry points refuse bad arguments before touching the device."""
    lib = built.load()
    null = ctypes.c_void_p(0)
    assert lib.cwlt_causal_linear_fwd(null, null, null, null, null, 1, 8, 16, 64, 512, 512, 512, 512, 1e-6, 1, null, null, 0, null) == 1001
    buf = ctypes.c_void_p(16)   # non-null dummy; rejected on head_dim before any launch
    assert lib.cwlt_causal_linear_fwd(buf, buf, buf, buf, buf, 1, 8, 16, 32, 512, 512, 512, 512, 1e-6, 1, null, null, 0, null) == 1001
    assert lib.cwlt_add_dropout_layernorm_fwd(null, null, null, null, null, null, null, null, 4, 512, 1e-5, 0.0, 0, null, 0, null) == 1001
    assert lib.cwlt_ln_blocks(65536) == 1024 and lib.cwlt_ln_blocks(1) == 1
    # weight-gradient split counts: ~one workgroup per CU in multiples of 8 at training sizes, slices of >= 256 token
    # rows at RL sizes (the caller sizes the partial-tile workspace from this)
    assert lib.cwlt_wgrad_splits(524288, 2048, 512) == 16 and lib.cwlt_wgrad_splits(524288, 512, 512) == 64
    assert lib.cwlt_wgrad_splits(1500, 2048, 512) == 5 and lib.cwlt_wgrad_splits(32, 256, 256) == 1
    assert lib.cwlt_wgrad_splits(4096, 2048, 512) == 16
    # scan segments: one workgroup per stream once N * H fills the chip; the reference's own batch (4 x 8 streams of
    # 3584 tokens = 56 chunks) is cut into 14 runs of 4 chunks; f32 (dtype 0) never
    assert lib.cwlt_scan_segments(512, 8, 1024, 1) == 1 and lib.cwlt_scan_segments(4, 8, 3584, 1) == 14
    assert lib.cwlt_scan_segments(4, 8, 3584, 0) == 1 and lib.cwlt_scan_segments(1, 8, 50, 1) == 1
    assert lib.cwlt_scan_seg_floats(4, 8, 14, 0) == 4 * 8 * 14 * 6 * 3 * 1024 and lib.cwlt_scan_seg_floats(4, 8, 1, 1) == 0
    # a segment count that leaves trailing segments empty (9 chunks cut 4 ways: 3 + 3 + 3 + 0) is refused by all three
    # scan entry points before any launch: the prefix / suffix pass would add the empty segment's unwritten workspace
    L9 = 9 * 64
    assert lib.cwlt_causal_linear_fwd(buf, buf, buf, buf, buf, 1, 8, L9, 64, 512, 512, 512, 512, 1e-6, 4, buf, null, 1, null) == 1001
    assert lib.cwlt_causal_linear_bwd_dkdv(buf, buf, buf, buf, buf, buf, buf, buf, null, null, null, 1, 8, L9, 64,
                                           512, 512, 512, 512, 512, 512, 512, 4, buf, 1, null) == 1001
    assert lib.cwlt_causal_linear_bwd_dq(buf, buf, buf, buf, buf, buf, buf, null, buf, 1, 8, L9, 64,
                                         512, 512, 512, 512, 512, 512, 4, buf, 1, null) == 1001
    assert lib.cwlt_causal_linear_fwd(buf, buf, buf, buf, buf, 1, 8, L9, 64, 512, 512, 512, 512, 1e-6, 10, buf, null, 1, null) == 1001
    # (3 segments of 3 chunks is what cwlt_scan_segments would hand out for such a length; not launched here: no GPU)
    assert lib.cwlt_sample_categorical(buf, (ctypes.c_int * 2)(5, 300), None, None, 2, 1, 305, 0, null, buf, null, 0, null) == 1001
    # one-pass input front: D must be a multiple of 64, T positive, p < 1, vocabularies positive -- all before any launch
    nr = (ctypes.c_int * 6)(56, 135, 18, 87, 18, 25)
    buf16 = ctypes.c_void_p(4096)
    assert lib.cwlt_cw_embed_proj_fwd(buf16, buf16, nr, 6, buf16, null, buf16, 8, 4, 500, 0.0, 0, null, 1, null) == 1001
    assert lib.cwlt_cw_embed_proj_fwd(buf16, buf16, nr, 6, buf16, null, buf16, 8, 0, 512, 0.0, 0, null, 1, null) == 1001
    assert lib.cwlt_cw_embed_proj_fwd(buf16, buf16, nr, 6, buf16, null, buf16, 8, 4, 512, 1.0, 0, null, 1, null) == 1001
    assert lib.cwlt_cw_embed_proj_fwd(buf16, buf16, nr, 9, buf16, null, buf16, 8, 4, 512, 0.0, 0, null, 1, null) == 1001
    assert lib.cwlt_cw_embed_proj_fwd(buf16, null, nr, 6, buf16, null, buf16, 8, 4, 512, 0.0, 0, null, 1, null) == 1001
    assert lib.cwlt_cw_embed_proj_fwd(buf16, buf16, nr, 6, buf16, null, buf16, 0, 4, 512, 0.0, 0, null, 1, null) == 0   # no rows
    assert lib.cwlt_cw_embed_proj_bwd(buf16, nr, 6, 512, buf16, buf16, buf16, 8, 256, 1, null) == 1001      # ldd < D
    assert lib.cwlt_cw_embed_proj_bwd(buf16, nr, 6, 512, buf16, buf16, null, 8, 512, 1, null) == 1001
    # generation step: an incomplete model description is refused before any launch
    m = built.DecodeModel()
    assert lib.cwlt_decode_workspace_floats(ctypes.byref(m)) == -1
    assert lib.cwlt_decode_step(ctypes.byref(m), buf, buf, null, buf, 1, null) == 1001
    assert lib.cwlt_decode_gemv(buf, null, buf, null, null, null, null, 1e-5, null, buf, null, 8, 6, 0, 1, 6, 8, 8, 6, null) == 1001


def test_product_has_no_cpu_fallback(built):
    """CPU tensors are refused loudly instead of being routed to a fallback."""
    import torch
    from rlmg_amd import ops
    q = torch.zeros(1, 4, 1, 64)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.causal_linear_attention(q, q, q)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "reinforcement-learning-in-music-generation_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)


def test_shadow_set_host_logic():
    """ops.ShadowSet: stacked compute-dtype copies follow the parameters on every refresh, are reused only inside an
    ops.frozen_weights() scope, and are dropped (not copied) by copy.deepcopy / pickling of the owning module."""
    import copy
    import pickle
    import torch
    from rlmg_amd import ops
    q, k, b = (torch.nn.Parameter(torch.randn(4, 3)) for _ in range(3))
    sh = ops.ShadowSet([(q, k), (b,)], torch.bfloat16)
    w, bb = sh.refresh()
    assert w.shape == (8, 3) and torch.equal(w[4:], k.detach().bfloat16()) and torch.equal(bb, b.detach().bfloat16())
    with torch.no_grad():
        k.mul_(2.0)
    assert torch.equal(sh.refresh()[0][4:], k.detach().bfloat16())
    with ops.frozen_weights():
        sh.refresh()
        with torch.no_grad():
            k.mul_(2.0)
        assert not torch.equal(sh.refresh()[0][4:], k.detach().bfloat16())      # reused, by contract
    assert torch.equal(sh.refresh()[0][4:], k.detach().bfloat16())
    holder = torch.nn.Module()
    holder.q, holder.k, holder.b, holder._shadow = q, k, b, sh
    assert copy.deepcopy(holder)._shadow is None and pickle.loads(pickle.dumps(sh)) is None


def test_public_header_is_plain_c(tmp_path):
    """include/cwlt.h is the contract for non-Python hosts: it must compile as C99 (and as C++) on its own, and a C
    host linking the library must resolve the symbols it declares."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "host.c"
    src.write_text('#include "cwlt.h"\n'
                   "int main(void) { cwlt_decode_model m; cwlt_decode_layer l; (void)m; (void)l;\n"
                   "  return cwlt_abi_version() > 0 && cwlt_ln_blocks(1) == 1 ? 0 : 1; }\n")
    inc = os.path.join(ROOT, "include")
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)],
                ["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, "-x", "c++", str(src)]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_encoder_layer_plan_and_refusals_without_gpu(built):
    """cwlt_encoder_layer_plan is host arithmetic; the layer / stack calls refuse null pointers and shapes they do not
    run before touching the device (include/cwlt.h, "a whole encoder layer per host call")."""
    lib = built.load()
    plan = built.EncoderLayerPlan()
    assert lib.cwlt_encoder_layer_plan(30, 50, 512, 2048, 8, 0.1, 1, ctypes.byref(plan)) == 0
    R = 1500
    # saved: qkv (3D) + attention out, s1, x1, s2 (4D) + g, gd (2F) bf16 per row at least
    assert plan.saved_bytes >= R * (3 * 512 + 4 * 512 + 2 * 2048) * 2
    assert plan.saved_bytes % 256 == 0 and plan.fwd_scratch_bytes >= R * 512 * 2
    assert plan.bwd_scratch_bytes >= R * (2048 + 3 * 512 + 2 * 512) * 2
    offs, sizes = list(plan.grad_off), [3 * 512 * 512, 3 * 512, 512 * 512, 512, 2048 * 512, 2048, 512 * 2048, 512, 512, 512,
                                        512, 512]
    spans = sorted(zip(offs, sizes))
    assert all(a + n <= b for (a, n), (b, _) in zip(spans, spans[1:]))          # the 12 gradients do not overlap
    assert spans[-1][0] + spans[-1][1] <= plan.grad_floats
    assert lib.cwlt_encoder_layer_plan(30, 50, 256, 2048, 8, 0.1, 1, ctypes.byref(plan)) == 1001       # d_model
    assert lib.cwlt_encoder_layer_plan(30, 50, 512, 2000, 8, 0.1, 1, ctypes.byref(plan)) == 1001       # d_ff % 256
    assert lib.cwlt_encoder_layer_plan(0, 50, 512, 2048, 8, 0.1, 1, ctypes.byref(plan)) == 1001        # no rows
    st = built.EncoderLayer(n_seq=30, len=50, d_model=512, d_ff=2048, n_heads=8, want_backward=1, p_drop=0.1)
    assert lib.cwlt_encoder_layer_fwd(ctypes.byref(st), None) == 1001
    assert lib.cwlt_encoder_layer_bwd(ctypes.byref(st), None) == 1001
    arr = (built.EncoderLayer * 2)()
    assert lib.cwlt_encoder_fwd(arr, 2, None) == 1001 and lib.cwlt_encoder_bwd(arr, 2, None) == 1001
    assert lib.cwlt_encoder_fwd(arr, 0, None) == 0                                # nothing to do
    null = ctypes.c_void_p(0)
    assert lib.cwlt_gemm_bf16_small(null, null, null, null, 8, 512, 512, 512, 512, 512, 0, null) == 1001
    assert lib.cwlt_gemm_bf16_small(null, null, null, null, 0, 512, 512, 512, 512, 512, 0, null) == 0  # no rows
    assert lib.cwlt_transpose_bf16_many(null, null, null, 3, null) == 1001

