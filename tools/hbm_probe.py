#!/usr/bin/env python
"""What the HBM of this box sustains for plain streaming (context for the roofline fractions):
read-only reduction, copy (1 read + 1 write stream), and a 4-read / 1-write elementwise kernel
(the access mix of the backward gate kernel), each over buffers far larger than the 256 MB MALL."""
import json
import time

import torch

dev = torch.device("cuda:0")
n = 512 * 1024 * 1024 // 4  # 512 MB per buffer
a, b, c, d = (torch.randn(n, device=dev) for _ in range(4))
out = torch.empty_like(a)


def bench(fn, nbytes, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return nbytes * reps / (time.perf_counter() - t0) / 1e12


res = {
    "read_only_sum_TBps": bench(lambda: a.sum(), n * 4),
    "copy_TBps": bench(lambda: out.copy_(a), 2 * n * 4),
    "read4_write1_TBps": bench(lambda: torch.add(torch.addcmul(a, b, c), d, out=out), 0),
}
# the fused 4-read/1-write case needs one kernel: addcmul(a,b,c) writes a temp, so time it as two kernels
# and report the bytes both move (3 reads + 1 write, then 2 reads + 1 write)
t = bench(lambda: torch.add(torch.addcmul(a, b, c), d, out=out), 7 * n * 4)
res["read4_write1_TBps"] = t
res["note"] = "torch elementwise kernels; buffers 512 MB each; bytes = tensors read + written per call"
print(json.dumps(res))
