#!/usr/bin/env python3
"""Dev probe: does an exhaustive MIOpen search find a faster kernel for the tower conv [B,256,10,9] fp16 NHWC?
Run with a hard timeout; prints the conv time before/after tuning and lists the user perf-db files written."""
import glob
import os
import sys
import time

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mode = sys.argv[2] if len(sys.argv) > 2 else "search"
layout = sys.argv[3] if len(sys.argv) > 3 else "nhwc"
dt_name = sys.argv[4] if len(sys.argv) > 4 else "fp16"
db = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "miopen_db")
os.makedirs(db, exist_ok=True)
os.environ["MIOPEN_USER_DB_PATH"] = db
os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = os.path.join(db, "cache")
if mode == "search":
    os.environ["MIOPEN_FIND_ENFORCE"] = "3"      # SEARCH: tune every applicable solver
    os.environ["MIOPEN_FIND_MODE"] = "1"         # NORMAL find
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda")
fmt = torch.channels_last if layout == "nhwc" else torch.contiguous_format
dt = torch.float16 if dt_name == "fp16" else torch.bfloat16
x = torch.randn(B, 256, 10, 9, device=dev, dtype=dt).contiguous(memory_format=fmt)
w = (torch.randn(256, 256, 3, 3, device=dev, dtype=dt) * 0.02).contiguous(memory_format=fmt)
t0 = time.time()
with torch.no_grad():
    y = F.conv2d(x, w, None, padding=1)
torch.cuda.synchronize()
print(f"first call (find/tune) took {time.time() - t0:.1f} s", flush=True)
with torch.no_grad():
    for _ in range(5):
        F.conv2d(x, w, None, padding=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        F.conv2d(x, w, None, padding=1)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 50
fl = 2 * B * 90 * 256 * 256 * 9
print(f"layout={layout} dtype={dt_name} mode={mode} conv {dt * 1e6:.1f} us  {fl / dt / 1e12:.1f} TFLOP/s", flush=True)
print("db files:", [(os.path.relpath(p, db), os.path.getsize(p)) for p in glob.glob(db + "/**/*", recursive=True) if os.path.isfile(p)][:20])
