"""Build-owned stand-in exposing the three fast_transformers names the reference imports
(dqn_policy/model.py:9-11), backed by the oracle restatement in oracle/ft_encoder.py.

Used ONLY by tests/golden/make_golden.py, in the build container, to run the reference's own
model.py wrappers (embeddings, in_linear, positional encoding, output heads, CE loss) on CPU and
record golden vectors.  It never ships as product code and is not a copy of the real package.
"""
import sys
import types

from .. import ft_encoder


def install():
    """Register `fast_transformers`, `.builders`, `.masking` in sys.modules."""
    root = types.ModuleType("fast_transformers")
    builders = types.ModuleType("fast_transformers.builders")
    masking = types.ModuleType("fast_transformers.masking")
    builders.TransformerEncoderBuilder = ft_encoder.TransformerEncoderBuilder
    builders.RecurrentEncoderBuilder = ft_encoder.RecurrentEncoderBuilder
    masking.TriangularCausalMask = ft_encoder.TriangularCausalMask
    root.builders = builders
    root.masking = masking
    sys.modules["fast_transformers"] = root
    sys.modules["fast_transformers.builders"] = builders
    sys.modules["fast_transformers.masking"] = masking
