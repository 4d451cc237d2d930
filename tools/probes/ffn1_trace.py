"""GPU box: per-workgroup timeline of the one-kernel FFN forward (needs a libcwlt built with the trace hook of
tools/probes/ffn1_trace.patch; CWLT_GEMM_TRACE=1).  Prints how many workgroups are in their epilogue at once, how long
epilogues and main loops take, and how both depend on the start spread (CWLT_GEMM_NT_SPREAD)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import rlmg_amd  # noqa: F401
from rlmg_amd import _lib, ops

M = 524288
x = (torch.randn(M, 512, device="cuda") * 0.5).bfloat16()
w = (torch.randn(2048, 512, device="cuda") * 0.05).bfloat16()
b = torch.randn(2048, device="cuda") * 0.1
for _ in range(3):
    g, gd = ops.ffn1_gelu_dropout(x, w, b, 0.1, 1234)
torch.cuda.synchronize()
lib = _lib.load()
lib.cwlt_debug_gemm_trace.restype = ctypes.c_void_p
ptr = lib.cwlt_debug_gemm_trace()
nwg = (M // 128) * 8
host = np.zeros(nwg * 6, dtype=np.uint64)
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
assert hip.hipMemcpy(host.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), ctypes.c_size_t(host.nbytes), 2) == 0
t = host.reshape(nwg, 6).astype(np.int64)
t0 = t[:, 0].min()
start, main, issued, done = [(t[:, i] - t0) / 100.0 for i in range(4)]      # us (100 MHz clock)
print("workgroups %d, kernel span %.1f us" % (nwg, done.max()))
print("main loop   : median %.2f us  p10 %.2f  p90 %.2f" % tuple(np.percentile(main - start, [50, 10, 90])))
print("epilogue    : median %.2f us  p10 %.2f  p90 %.2f   (until the last store is issued)" % tuple(np.percentile(issued - main, [50, 10, 90])))
tl = (t[:, 4] - t0) / 100.0
print("  acc -> LDS tile + barrier: median %.2f us  p10 %.2f  p90 %.2f" % tuple(np.percentile(tl - main, [50, 10, 90])))
print("  row loop (LDS read, GELU, hash, two stores) x 8: median %.2f us  p10 %.2f  p90 %.2f" % tuple(np.percentile(issued - tl, [50, 10, 90])))
print("store drain : median %.2f us  p10 %.2f  p90 %.2f   (issued -> vmcnt(0))" % tuple(np.percentile(done - issued, [50, 10, 90])))
# how many workgroups are between main-loop end and drain end at a time, sampled over the middle half of the kernel
lo, hi = done.max() * 0.25, done.max() * 0.75
ts = np.linspace(lo, hi, 2000)
ne = np.array([np.count_nonzero((main <= x_) & (done > x_)) for x_ in ts])
nm = np.array([np.count_nonzero((start <= x_) & (main > x_)) for x_ in ts])
print("in epilogue/drain at once: mean %.1f  min %d  max %d  std %.1f   (resident %.1f)" % (ne.mean(), ne.min(), ne.max(), ne.std(), (ne + nm).mean()))
# phase coherence: histogram of main-loop end times modulo the median tile period
period = np.median(done - start)
ph = np.mod(main[(main > lo) & (main < hi)], period) / period
hist = np.histogram(ph, bins=10, range=(0, 1))[0]
print("tile period %.2f us; main-loop ends by phase decile: %s" % (period, hist.tolist()))
