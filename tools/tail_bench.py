"""How much of a large-M GEMM launch is round quantisation?  Times the same kernel (GELU epilogue, K = 1024, M = 5760) at output widths whose grids
fill the 512 block slots once, 1.4 times and 1.9 times.  Usage (GPU box): python tools/tail_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gtav_amd import lib as L  # noqa: E402


def main():
    lib = L.load()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    M, K = 5760, 1024
    Mp = (M + 127) // 128 * 128
    x = (torch.randn(Mp, K, device=dev) * 0.5).half()
    for N, wm in ((3072, 12), (2048, 12), (1024, 13), (1024, 12), (4096, 12), (2560, 12), (1536, 12), (3072, 13)):
        ws = [(torch.randn(N, K, device=dev) * 0.03).half() for _ in range(8)]
        bias = torch.randn(N, device=dev)
        out = torch.empty(Mp, N, device=dev, dtype=torch.float16)
        lib.gtav_op_gemm_set_wm(wm)

        def run(i):
            L.check(lib.gtav_op_gemm_f16(x.data_ptr(), K, ws[i % 8].data_ptr(), bias.data_ptr(), out.data_ptr(), N, M, N, K, 2, 0, 0, 1, st))
        for i in range(8):
            run(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(64):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 64
        tm = 192 if wm == 12 else 96
        tiles = -(-M // tm) * (N // 128)
        print(f"N={N:5d} shape {wm}: {tiles:5d} tiles = {tiles / 512:.2f} rounds of 512 slots  {us:7.2f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
    lib.gtav_op_gemm_set_wm(0)


if __name__ == "__main__":
    main()
