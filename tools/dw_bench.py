"""Weight-gradient GEMM shapes of the training step (dW[n][k] += sum_m dY[m][n] X[m][k], contraction over the 11 520 tokens of a batch-16 step):
M = layer outputs, N = layer inputs, K = tokens, accumulating fp32 epilogue.  Which block shape runs the long K loop fastest?
Usage (GPU box): python tools/dw_bench.py [--wm 0 3 7 12]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gtav_amd import lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wm", type=int, nargs="+", default=[0, 3, 7, 12])
    ap.add_argument("--tokens", type=int, default=11520)
    a = ap.parse_args()
    lib = L.load()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    K = a.tokens
    shapes = {"fc1 dW": (4096, 1024), "fc2 dW": (1024, 4096), "qkv dW": (3072, 1024), "out dW": (1024, 1024)}
    for name, (M, N) in shapes.items():
        nb = 4
        xs = [torch.randn(M * K, device=dev).half() * 0.05 for _ in range(nb)]     # tile-major images are opaque here: any fp16 bytes time the same
        ws = [torch.randn(N * K, device=dev).half() * 0.05 for _ in range(nb)]
        out = torch.zeros(M, N, device=dev)
        for wm in a.wm:
            lib.gtav_op_gemm_set_wm(wm)
            try:
                for i in range(nb):
                    L.check(lib.gtav_op_gemm_f16(xs[i].data_ptr(), K, ws[i].data_ptr(), 0, out.data_ptr(), N, M, N, K, 4, 0, 0, 0, st))
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 5 * nb
                e0.record()
                for i in range(n):
                    lib.gtav_op_gemm_f16(xs[i % nb].data_ptr(), K, ws[i % nb].data_ptr(), 0, out.data_ptr(), N, M, N, K, 4, 0, 0, 0, st)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / n
                print(f"{name} M={M} N={N} K={K} wm={wm:2d}: {us:8.2f} us {2.0 * M * N * K / us * 1e-6:7.1f} TFLOP/s", flush=True)
            except Exception as e:  # a shape the library refuses for this epilogue
                print(f"{name} wm={wm}: {str(e)[:100]}")
            finally:
                lib.gtav_op_gemm_set_wm(0)


if __name__ == "__main__":
    main()
