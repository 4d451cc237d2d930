"""Deterministic, name-keyed parameter fill shared by make_golden.py (applied to the REFERENCE's
classes) and the tests (applied to the oracle / product classes), so fixtures need not store weights.
"""
import zlib

import torch


def fill_params(module, seed=0):
    """Overwrite every parameter of `module` with values that depend only on (seed, name, shape)."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
            r = torch.randn(p.shape, generator=g, dtype=torch.float32)
            if p.dim() >= 2:
                val = r * (0.5 / (p.shape[-1] ** 0.5)) if "lut" not in name else r * 0.05
            elif "norm" in name.lower() and name.endswith("weight"):
                val = 1.0 + 0.1 * r
            else:
                val = 0.05 * r
            p.copy_(val.to(p.dtype))
    return module
