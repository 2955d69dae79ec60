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
            outs = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(nbuf)]
            for i in range(3 * nbuf):
                torch.matmul(xs[i % nbuf], ws[i % nbuf].t(), out=outs[i % nbuf])
            torch.cuda.synchronize()
            # one hipGraph of 8 x nbuf launches: no per-call host cost (torch.matmul costs ~18 us of CPU per call, more than the small-M kernels)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for i in range(nbuf):
                    torch.matmul(xs[i], ws[i].t(), out=outs[i])
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            per_graph = 8 * nbuf
            with torch.cuda.graph(g):
                for i in range(per_graph):
                    torch.matmul(xs[i % nbuf], ws[i % nbuf].t(), out=outs[i % nbuf])
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (reps * per_graph)
            print(f"{name:4s} M={M:6d} N={N:5d} K={K:5d}: {us:8.2f} us  {2.0 * M * N * K / us * 1e-6:8.1f} TFLOP/s (hipGraph of {per_graph} launches, fp16 out, no epilogue)", flush=True)


if __name__ == "__main__":
    main()
