"""configs[2]-shaped search: one GPU's shard (1.25M x 512 rows) against query batches of 48 ... 1000; per-kernel event times.

    python tools/bench_search_q.py [nq ...]
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "intelligent-video-analysis-retrieval-system_amd"))
import torch
from ivr_amd import _ffi
from ivr_amd.index import FlatIPIndex
N = 1_250_000     # configs[2]: 10M rows over 8 GPUs
idx = FlatIPIndex(512, capacity=N)
for i in range(0, N, 250_000):
    idx.add(torch.randn((250_000, 512), device="cuda"), normalize=True)
for nq in ([int(v) for v in sys.argv[1:]] or [48, 100, 256, 1000]):
    q = torch.randn((nq, 512), device="cuda")
    idx.reserve_search(nq, 10)
    for _ in range(2):
        idx.search_device(q, 10, normalize=True)
    torch.cuda.synchronize()
    _ffi.profile_reset(); _ffi.profile_enable(2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        idx.search_device(q, 10, normalize=True)
    e1.record()
    torch.cuda.synchronize(); _ffi.profile_enable(False)
    ms = e0.elapsed_time(e1) / 5
    pr = _ffi.profile_read()
    print(f"nq={nq:5d} N={N}: {ms:8.3f} ms  {N*nq/ms/1e6:8.1f} G pairs/s  {2*512*N*nq/ms/1e9:7.1f} TFLOP/s f32",
          {k: round(v['ms']/5, 3) for k, v in pr.items()})
