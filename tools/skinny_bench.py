"""The adaLN table launch of a generated frame (csrc/skinny.hip, fp32 MFMA; N = 198 656 modulation features, K = 1024) at 101 rows (batch 1: all noise steps of a frame)
and 808 rows (batch 8).  Experiments build: GTAV_SKINNY_VARIANT=41 / 42 / 82 forces (feature tiles per wave, 16-row tiles per slab), GTAV_SKINNY_FT_MIN_BLOCKS=100000000 the
round-2 kernel (1, 1).  Usage (GPU box): python tools/skinny_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtav_amd import lib as L
lib = L.load_experiments()
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
N, K = 198656, 1024
w = torch.randn(N, K, device=dev) * 0.03; b = torch.randn(N, device=dev)
for M in (80, 101, 808):
    x = torch.randn(M, K, device=dev); y = torch.empty(M, N, device=dev)
    run = lambda: L.check(lib.gtav_op_skinny_f32(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), N, M, N, K, 0, st))
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    print(f"adaLN table M={M}: {e0.elapsed_time(e1)/5:.3f} ms", flush=True)
