"""hipBLASLt / rocBLAS solution selection for the dense projections (the MFMA work of the path).

The GEMMs are plain library GEMMs through torch.mm/addmm; for this model's skinny shapes
(M = B*T rows, N, K in {384, 512, 1216, 1536, 2048}, or K = B*T for the weight gradients) the
libraries' default heuristics are 10-25 % off their own best kernels.  PyTorch's TunableOp picks the
best solution per shape; the table tuned on MI355X ships in tuning/ and is loaded read-only
(no tuning at run time).  `tune()` regenerates it on a GPU box.
"""
import os

import torch

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning")
TABLE = os.path.join(_DIR, "gemm_gfx950.csv")


def enable(table=TABLE):
    """Use the shipped table (if present); shapes not in it fall back to the library default."""
    if not os.path.exists(table) or not torch.cuda.is_available():
        return False
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.tuning_enable(False)
    tunable.record_untuned_enable(False)
    try:
        ok = tunable.read_file(table)
    except Exception:   # validator mismatch (different library build): keep defaults
        ok = False
    if not ok:
        tunable.enable(False)
    return bool(ok)


def tune(out=TABLE, max_ms=100, iters=50, rotating_mb=1024, preload=TABLE):
    """Switch TunableOp to tuning mode; run the workload once afterwards, then call `save()`.  Entries of
    `preload` (the shipped table) are kept, so only shapes it lacks are tuned and `out` holds the union."""
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    if os.path.exists(out):          # several tuning runs in a row accumulate into `out`
        preload = out
    if preload and os.path.exists(preload):
        try:
            tunable.read_file(preload)
        except Exception:
            pass
    tunable.tuning_enable(True)
    tunable.set_max_tuning_duration(max_ms)
    tunable.set_max_tuning_iterations(iters)
    tunable.set_rotating_buffer_size(rotating_mb)
    tunable.set_filename(out)
    if hasattr(tunable, "write_file_on_exit"):
        tunable.write_file_on_exit(False)


def save(out=TABLE):
    import torch.cuda.tunable as tunable
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if hasattr(tunable, "write_file"):
        tunable.write_file(out)     # older builds write the file at interpreter exit instead
