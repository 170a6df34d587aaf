#!/usr/bin/env python3
"""Host-to-device copy rates on the GPU box: pageable source through the runtime, pinned source, 1.2 GB."""
import json
import time

import torch

n = 1264 * 1000 * 1000
dev = torch.device("cuda", 0)
src = torch.empty(n, dtype=torch.uint8)
src.random_(0, 255)
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
pin.copy_(src)
dst = torch.empty(n, dtype=torch.uint8, device=dev)
out = {}
for name, s in (("pageable", src), ("pinned", pin)):
    ts = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dst.copy_(s, non_blocking=False)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    out[name + "_GB_per_s"] = round(n / min(ts) / 1e9, 1)
t0 = time.perf_counter(); pin.copy_(src); out["host_memcpy_1thread_GB_per_s"] = round(n / (time.perf_counter() - t0) / 1e9, 1)
import os
out["cpus"] = os.cpu_count(); out["affinity"] = len(os.sched_getaffinity(0))
print(json.dumps(out))
