"""LayerNorm + modulate launch WITHOUT a pending split-K update (what follows the in-place residual GEMM at large M), through gtav_op_ln_modulate; run once per
GTAV_LN_FLAGS value (experiments build: bits 8.. switch pieces off, elementwise.hip LN_DBG).  Usage (GPU box): GTAV_LN_FLAGS=7 python tools/ln_plain_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtav_amd import lib as L
lib = L.load_experiments()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
for M in (720, 5760):
    D, P = 1024, 144
    x = torch.randn(M, D, device=dev)
    out = torch.zeros((M + 127) // 128 * 128, D, device=dev, dtype=torch.float16)
    mod = torch.randn(M // P, 2 * D, device=dev) * 0.1
    def run():
        L.check(lib.gtav_op_ln_modulate(x.data_ptr(), out.data_ptr(), M, D, mod.data_ptr(), mod[:, D:].data_ptr(), 2 * D, P, st))
    for _ in range(10): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): run()
    e1.record(); torch.cuda.synchronize()
    print(f"GTAV_LN_FLAGS={os.environ.get('GTAV_LN_FLAGS','default')} M={M}: LayerNorm without pending update {e0.elapsed_time(e1)*1e3/300:.2f} us", flush=True)
