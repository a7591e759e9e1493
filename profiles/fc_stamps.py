#!/usr/bin/env python3
"""In-kernel cycle stamps of k_fc_wide_f16 (diagnostic build: make -C chinesechesszero_amd/csrc ab NAME=fcdiag ABFLAGS=-DCCZ_FC_DIAG,
library copied to build/ab/ for a GPU-box run): wave 0 of every tile with n0 = 0 stamps entry / first fragments in registers / loop
end / tile end into the pad columns of C. Prints the median phase lengths in cycles and in us at the clock the kernel ran at."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chinesechesszero_amd import _lib  # noqa: E402

for d in ("ab", "diag"):
    p = os.path.join(ROOT, "build", d, "libcczero_ab_fcdiag.so")
    if os.path.exists(p):
        _lib.LIB_PATH = p
        break
L = _lib.lib()
dev = torch.device("cuda", 0)
s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
out = []
N, K = 2086, 1536
for M in (128, 4096):
    a = torch.relu(torch.randn(M, K, device=dev)).half()
    w = (torch.randn(2176, K, device=dev) * 0.03).half()
    b = torch.randn(2176, device=dev)
    ldc = N + 16
    c = torch.zeros(M, ldc, dtype=torch.float16, device=dev)
    for dbg, what in ((32, "full"), (32 | 1 | 2 | 4, "nothing in the loop"), (32 | 1 | 2 | 4 | 16, "nothing in the loop, no barrier"), (32 | 4, "no DMA in the loop"),
                      (32 | 1, "no MFMA"), (32 | 16, "no barrier (wrong results)"), (32 | 2 | 4, "MFMA + barrier only"), (32 | 2 | 4 | 16, "MFMA only")):
        for _ in range(3):
            _lib.check(L.ccz_fc_f16(s, P(a), K, P(w), P(b), P(c), ldc, M, N, K, (dbg << 8) | 4, None))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.ccz_fc_f16(s, P(a), K, P(w), P(b), P(c), ldc, M, N, K, (dbg << 8) | 4, None))
        e1.record()
        torch.cuda.synchronize()
        st = c[0::256, N:N + 16].contiguous().view(torch.int32).cpu().to(torch.int64) & 0xffffffff   # [tiles_m, 8]: 4 stamps x (lo, hi)
        t = st[:, 0::2] + (st[:, 1::2] << 32)
        d = (t[:, 1:] - t[:, :-1]).double()
        row = {"M": M, "what": what, "event_us": e0.elapsed_time(e1) * 1e3, "tiles_stamped": int(t.shape[0]),
               "cycles_prologue_loop_epilogue_median": [float(x) for x in d.median(0).values], "cycles_total_median": float((t[:, 3] - t[:, 0]).double().median()),
               "span_first_entry_to_last_end_cycles": float(t[:, 3].max() - t[:, 0].min())}
        out.append(row)
        print(row, file=sys.stderr, flush=True)
print(json.dumps({"lib": os.path.basename(_lib.LIB_PATH), "counter": "s_memtime (__builtin_readcyclecounter)", "rows": out}))
