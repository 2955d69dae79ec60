"""Yardstick only (never on the product path): what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on the four DiT GEMM
shapes, fp16 in / fp32 accumulate, no epilogue.  Tells how far the hand-written kernels of csrc/gemm.hip are from a tuned library on the
same box.  Usage (GPU box): python tools/blas_yardstick.py [--ms 720 5760]"""
import argparse
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", type=int, nargs="+", default=[720, 5760])
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    shapes = {"qkv": (3072, 1024), "out": (1024, 1024), "fc1": (4096, 1024), "fc2": (1024, 4096)}
    for M in a.ms:
        for name, (N, K) in shapes.items():
            nbuf = 8   # rotate operands: 8 x (X + W) > 32 MiB of L2, like consecutive layers of the forward
            xs = [torch.randn(M, K, device=dev, dtype=torch.float16) for _ in range(nbuf)]
            ws = [torch.randn(N, K, device=dev, dtype=torch.float16) for _ in range(nbuf)]
            for i in range(3 * nbuf):
                torch.matmul(xs[i % nbuf], ws[i % nbuf].t())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 40 * nbuf
            e0.record()
            for i in range(n):
                torch.matmul(xs[i % nbuf], ws[i % nbuf].t())
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            print(f"{name:4s} M={M:6d} N={N:5d} K={K:5d}: {us:8.2f} us  {2.0 * M * N * K / us * 1e-6:8.1f} TFLOP/s (back-to-back launches, no epilogue)", flush=True)


if __name__ == "__main__":
    main()
