"""Run only the bf16 scan kernels (forward, dQ, dK/dV, one-sweep backward) a few times at the bench shape -- target for rocprofv3
--pmc passes.  usage: python tools/scan_only.py [B] [T] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import rlmg_amd  # noqa: F401
from rlmg_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
qkv = torch.randn(B, T, 3, 8, 64, device=dev).bfloat16()
dout = torch.randn(B, T, 8, 64, device=dev).bfloat16()
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
for _ in range(reps):
    _, _, _, out, zinv, fin = ops.cla_fwd(q, k, v, final_state=True)
    ops.cla_bwd(q, k, v, out, zinv, dout, want_colsum=True)                       # the dkdv + dq pair
    if fin is not None:
        ops.cla_bwd(q, k, v, out, zinv, dout, want_colsum=True, final_state=fin)  # one sweep
torch.cuda.synchronize()
print("done")
