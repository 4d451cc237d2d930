"""ctypes binding of libcwlt.so -- the only way product code reaches the HIP kernels.

Fails loudly: a missing library raises ImportError at first use, a non-zero status from any entry
point raises RuntimeError.  No fallback path exists.
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libcwlt.so")

CWLT_F32 = 0
CWLT_BF16 = 1
ABI_VERSION = 20

_c_int = ctypes.c_int
_c_i64 = ctypes.c_int64
_c_f32 = ctypes.c_float
_c_u64 = ctypes.c_uint64
_ptr = ctypes.c_void_p



class DecodeLayer(ctypes.Structure):
    """cwlt_decode_layer (include/cwlt.h)."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("wqkv", "bqkv", "wo", "bo", "ln1_w", "ln1_b", "w1", "b1", "w2", "b2",
                                               "ln2_w", "ln2_b", "S", "Z")]


class DecodeModel(ctypes.Structure):
    """cwlt_decode_model (include/cwlt.h)."""
    _fields_ = ([(n, ctypes.c_int) for n in ("n_layer", "n_head", "d_model", "d_ff", "n_attr", "emb_width",
                                             "n_logits")] +
                [("eps_ln", ctypes.c_float), ("eps_attn", ctypes.c_float),
                 ("tables", ctypes.POINTER(ctypes.c_void_p)), ("widths", ctypes.POINTER(ctypes.c_int)),
                 ("nrows", ctypes.POINTER(ctypes.c_int)), ("w_in", ctypes.c_void_p), ("b_in", ctypes.c_void_p),
                 ("pe0", ctypes.c_void_p), ("layers", ctypes.POINTER(DecodeLayer)), ("lnf_w", ctypes.c_void_p),
                 ("lnf_b", ctypes.c_void_p), ("w_heads", ctypes.c_void_p), ("b_heads", ctypes.c_void_p)])


LAYER_NGRADS = 12       # CWLT_LAYER_NGRADS
LAYER_NSAVED = 13       # CWLT_LAYER_NSAVED


class EncoderLayer(ctypes.Structure):
    """cwlt_encoder_layer (include/cwlt.h)."""
    _fields_ = ([("n_seq", ctypes.c_int64), ("len", ctypes.c_int64), ("d_model", ctypes.c_int32), ("d_ff", ctypes.c_int32),
                 ("n_heads", ctypes.c_int32), ("want_backward", ctypes.c_int32), ("p_drop", ctypes.c_float),
                 ("ln_eps", ctypes.c_float), ("attn_eps", ctypes.c_float), ("reserved", ctypes.c_int32),
                 ("seed", ctypes.c_uint64 * 3), ("seed_base", ctypes.c_void_p)] +
                [(n, ctypes.c_void_p) for n in ("wqkv", "wo", "w1", "w2", "bqkv", "bo", "b1", "b2", "gamma1", "beta1",
                                                "gamma2", "beta2", "wqkv_t", "wo_t", "w1_t", "w2_t", "x", "y", "saved",
                                                "scratch", "dy", "dx", "grads")])


class EncoderLayerPlan(ctypes.Structure):
    """cwlt_encoder_layer_plan_t (include/cwlt.h)."""
    _fields_ = [("saved_bytes", ctypes.c_int64), ("fwd_scratch_bytes", ctypes.c_int64),
                ("bwd_scratch_bytes", ctypes.c_int64), ("grad_floats", ctypes.c_int64),
                ("grad_off", ctypes.c_int64 * LAYER_NGRADS), ("saved_off", ctypes.c_int64 * LAYER_NSAVED)]


# name -> argtypes; restype is always int (status)
_SIGNATURES = {
    "cwlt_abi_version": [],
    "cwlt_scan_segments": [_c_int, _c_int, _c_int, _c_int],
    "cwlt_scan_seg_floats": [_c_int, _c_int, _c_int, _c_int],
    "cwlt_scan_final_state_floats": [_c_int, _c_int],
    "cwlt_causal_linear_fwd": [_ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int,
                               _c_i64, _c_i64, _c_i64, _c_i64, _c_f32, _c_int, _ptr, _ptr, _c_int, _ptr],
    "cwlt_causal_linear_bwd_sweep": [_ptr] * 13 + [_c_int] * 4 + [_c_i64] * 8 + [_c_int, _ptr],
    "cwlt_causal_linear_bwd": [_ptr] * 9 + [_c_int] * 4 + [_c_i64] * 8 + [_c_int, _ptr],
    "cwlt_causal_linear_bwd_dkdv": [_ptr] * 11 + [_c_int] * 4 + [_c_i64] * 7 + [_c_int, _ptr, _c_int, _ptr],
    "cwlt_causal_linear_bwd_dq": [_ptr] * 9 + [_c_int] * 4 + [_c_i64] * 6 + [_c_int, _ptr, _c_int, _ptr],
    "cwlt_ln_blocks": [_c_i64],
    "cwlt_add_dropout_layernorm_fwd": [_ptr] * 8 + [_c_i64, _c_int, _c_f32, _c_f32, _c_u64, _ptr, _c_int, _ptr],
    "cwlt_add_dropout_layernorm_bwd": [_ptr] * 10 + [_c_i64, _c_int, _c_f32, _c_u64, _ptr, _c_int, _ptr],
    "cwlt_colsum_blocks": [_c_i64],
    "cwlt_colsum": [_ptr, _ptr, _ptr, _c_i64, _c_int, _c_i64, _c_int, _ptr],
    "cwlt_rowslab_blocks": [_c_i64],
    "cwlt_bias_gelu_dropout_fwd": [_ptr, _ptr, _ptr, _ptr, _c_i64, _c_int, _c_f32, _c_u64, _ptr, _c_int, _ptr],
    "cwlt_gemm_nt_tiles": [_c_i64],
    "cwlt_gemm_nt_mul": [_ptr] * 6 + [_c_i64, _c_int, _c_int] + [_c_i64] * 4 + [_ptr],
    "cwlt_gemm_nt_bias_gelu_dropout": [_ptr] * 5 + [_c_i64, _c_int, _c_int, _c_i64, _c_i64, _c_f32, _c_u64, _ptr, _ptr],
    "cwlt_gemm_nt_bias_dropout_add_layernorm": [_ptr] * 10 + [_c_i64, _c_int, _c_int, _c_i64, _c_i64, _c_f32, _c_f32, _c_u64,
                                                _ptr, _ptr],
    "cwlt_gemm_bf16": [_ptr, _ptr, _ptr, _ptr, _c_i64, _c_int, _c_int, _c_i64, _c_i64, _c_i64, _c_int, _ptr],
    "cwlt_gemm_bf16_tune": [_c_int, _ptr],
    "cwlt_gemm_bf16_small": [_ptr, _ptr, _ptr, _ptr, _c_i64, _c_int, _c_int, _c_i64, _c_i64, _c_i64, _c_int, _ptr],
    "cwlt_transpose_bf16_many": [_ptr, _ptr, _ptr, _c_int, _ptr],
    "cwlt_cast_bf16_many": [_ptr, _ptr, _ptr, _c_int, _ptr],
    "cwlt_gemm_bf16_small_gelu": [_ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_int, _c_int, _c_i64, _c_i64, _c_f32, _c_u64, _ptr,
                                  _ptr],
    "cwlt_encoder_layer_plan": [_c_i64, _c_i64, _c_int, _c_int, _c_int, _c_f32, _c_int, ctypes.POINTER(EncoderLayerPlan)],
    "cwlt_encoder_layer_fwd": [ctypes.POINTER(EncoderLayer), _ptr],
    "cwlt_encoder_layer_bwd": [ctypes.POINTER(EncoderLayer), _ptr],
    "cwlt_encoder_fwd": [ctypes.POINTER(EncoderLayer), _c_int, _ptr],
    "cwlt_encoder_bwd": [ctypes.POINTER(EncoderLayer), _c_int, _ptr],
    "cwlt_graph_replace_memset_nodes": [_ptr, _ptr],
    "cwlt_bias_gelu_dropout_bwd": [_ptr] * 6 + [_c_i64, _c_int, _c_f32, _c_u64, _ptr, _c_int, _ptr],
    "cwlt_posenc_dropout": [_ptr, _ptr, _ptr, _c_i64, _c_int, _c_int, _c_f32, _c_u64, _ptr, _c_int, _ptr],
    "cwlt_embed_splits": [_c_i64],
    "cwlt_cw_embed_fwd": [_ptr, _ptr, _ptr, _ptr, _c_int, _ptr, _c_i64, _c_i64, _c_int, _ptr],
    "cwlt_cw_embed_bwd": [_ptr, _ptr, _ptr, _c_int, _ptr, _ptr, _ptr, _c_i64, _c_i64, _c_int, _ptr],
    "cwlt_cw_embed_proj_fwd": [_ptr, _ptr, _ptr, _c_int, _ptr, _ptr, _ptr, _c_i64, _c_int, _c_int, _c_f32, _c_u64, _ptr,
                               _c_int, _ptr],
    "cwlt_cw_embed_proj_bwd": [_ptr, _ptr, _c_int, _c_int, _ptr, _ptr, _ptr, _c_i64, _c_i64, _c_int, _ptr],
    "cwlt_band_attn_fwd": [_ptr] * 6 + [_c_int] * 5 + [_c_i64] * 4 + [_c_f32, _c_f32, _c_u64, _ptr, _c_int, _ptr],
    "cwlt_band_attn_bwd": [_ptr] * 10 + [_c_int] * 5 + [_c_i64] * 8 + [_c_f32, _c_f32, _c_u64, _ptr, _c_int, _ptr],
    "cwlt_wgrad_splits": [_c_i64, _c_int, _c_int],
    "cwlt_wgrad_bf16": [_ptr, _ptr, _ptr, _ptr, _c_i64, _c_int, _c_int, _c_i64, _c_i64, _c_int, _ptr],
    "cwlt_wgrad_bf16_group": [_ptr] * 8 + [_c_int, _c_i64, _c_int, _ptr],
    "cwlt_recurrent_cla_step": [_ptr] * 6 + [_c_int] * 3 + [_c_i64] * 4 + [_c_f32, _c_int, _ptr],
    "cwlt_decode_workspace_floats": [ctypes.POINTER(DecodeModel)],
    "cwlt_decode_step": [ctypes.POINTER(DecodeModel), _ptr, _ptr, _ptr, _ptr, _c_int, _ptr],
    "cwlt_decode_gemv": [_ptr] * 7 + [_c_f32] + [_ptr] * 3 + [_c_int] * 4 + [_c_i64] * 4 + [_ptr],
    "cwlt_sample_categorical": [_ptr, _ptr, _ptr, _ptr, _c_int, _c_i64, _c_i64, _c_u64, _ptr, _ptr, _ptr, _c_i64, _ptr],
    "cwlt_heads_blocks": [_c_i64],
    "cwlt_heads_fwd": [_ptr, _ptr, _c_int] + [_ptr] * 7 + [_c_i64, _c_i64, _c_i64, _c_int, _ptr],
    "cwlt_heads_ce_bwd": [_ptr, _ptr, _c_int, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_i64, _c_int, _ptr],
    "cwlt_heads_logp_bwd": [_ptr, _ptr, _c_int, _ptr, _ptr, _ptr, _c_i64, _c_i64, _c_int, _ptr],
    "cwlt_rollout_gather": [_ptr, _ptr, _ptr, _c_int, _ptr, _ptr, _c_int, _c_int, _c_int, _c_i64, _c_int, _ptr],
    "cwlt_ppo_returns_adv": [_ptr, _ptr, _ptr, _ptr, _c_int, _c_f32, _c_int, _ptr],
    "cwlt_ppo_policy_loss": [_ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_f32, _ptr],
    "cwlt_dqn_td_fwd": [_ptr, _ptr, _ptr, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_i64,
                        _c_f32, _ptr],
    "cwlt_dqn_td_bwd": [_ptr, _ptr, _ptr, _c_int, _ptr, _ptr, _c_int, _c_int, _c_i64, _ptr],
}

_lib = None


def load():
    """Load libcwlt.so (once) and bind every entry point of include/cwlt.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libcwlt.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    # Bind libcwlt's hip* symbols to the HIP runtime PyTorch uses (its bundled libamdhip64.so), so
    # torch streams / device pointers are valid inside the kernels' launches: load that runtime into
    # the global symbol scope BEFORE libcwlt.so (which carries no DT_NEEDED on a runtime of its own).
    import torch
    rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if not os.path.exists(rt):
        raise ImportError("PyTorch-ROCm's HIP runtime not found at %s" % rt)
    global _hip
    _hip = ctypes.CDLL(rt, mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.argtypes = argtypes
        fn.restype = _c_i64 if name in ("cwlt_decode_workspace_floats", "cwlt_gemm_nt_tiles", "cwlt_scan_seg_floats",
                                         "cwlt_scan_final_state_floats") else _c_int
    got = lib.cwlt_abi_version()
    if got != ABI_VERSION:
        raise ImportError("libcwlt.so ABI version %d, binding expects %d -- rebuild" % (got, ABI_VERSION))
    _lib = lib
    return lib


_hip = None
# hipGraphNodeType (hip_runtime_api.h): the kinds of node a stream capture can record
GRAPH_NODE_TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event",
                    7: "event_record"}


def graph_node_census(raw_graph):
    """{node kind: count} of a hipGraph_t (an integer handle, e.g. torch.cuda.CUDAGraph(keep_graph=True)
    .raw_cuda_graph()), child graphs included.  Used to keep memset nodes out of replayed graphs: a captured
    hipMemsetAsync replays a wrong fill pattern on ROCm 7.2 (tools/probes/graph_memset_probe.py)."""
    load()
    counts = {}

    def walk(g):
        n = ctypes.c_size_t(0)
        if _hip.hipGraphGetNodes(ctypes.c_void_p(g), None, ctypes.byref(n)) != 0:
            raise RuntimeError("hipGraphGetNodes failed")
        if n.value == 0:
            return
        nodes = (ctypes.c_void_p * n.value)()
        if _hip.hipGraphGetNodes(ctypes.c_void_p(g), nodes, ctypes.byref(n)) != 0:
            raise RuntimeError("hipGraphGetNodes failed")
        for node in nodes:
            t = ctypes.c_int(-1)
            if _hip.hipGraphNodeGetType(ctypes.c_void_p(node), ctypes.byref(t)) != 0:
                raise RuntimeError("hipGraphNodeGetType failed")
            kind = GRAPH_NODE_TYPES.get(t.value, "type%d" % t.value)
            counts[kind] = counts.get(kind, 0) + 1
            if t.value == 4:
                child = ctypes.c_void_p(0)
                if _hip.hipGraphChildGraphNodeGetGraph(ctypes.c_void_p(node), ctypes.byref(child)) == 0 and child.value:
                    walk(child.value)

    walk(int(raw_graph))
    return counts


def exported_names():
    return sorted(_SIGNATURES)


def check(status, what):
    if status != 0:
        raise RuntimeError("%s failed with status %d (1001 = bad argument, 1002 = bad dtype, "
                           "otherwise a hipError_t)" % (what, status))


def dtype_code(t):
    import torch
    if t == torch.float32:
        return CWLT_F32
    if t == torch.bfloat16:
        return CWLT_BF16
    raise TypeError("libcwlt kernels take float32 or bfloat16 activations, got %s" % t)


_raw_stream = None


def stream_ptr():
    """torch's current HIP stream (of the current device) as a void*.  Uses the raw accessor: building a
    torch.cuda.Stream object per launch cost ~10 us of host time, 1.5 ms per launch-bound RL update."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def opt(t, name="tensor"):
    """Device pointer or NULL."""
    return None if t is None else dev(t, name)


def int_array(values):
    return (ctypes.c_int * len(values))(*[int(v) for v in values])


def ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[dev(t).value for t in tensors])


def dev(t, name="tensor"):
    """Device pointer of a tensor that must already live on the GPU."""
    if not t.is_cuda:
        raise RuntimeError("%s must be a GPU tensor: the cwlt hot path has no CPU implementation" % name)
    return ctypes.c_void_p(t.data_ptr())
