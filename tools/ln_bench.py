"""LayerNorm launch of the residual path in isolation (elementwise.hip ln_row_block_kernel with a pending split-K update) through the C-ABI:
gtav_op_gemm_splitk_ln runs the slab GEMM and the LayerNorm; with K = 64 the GEMM is a few microseconds, measured alone (epilogue 6) and subtracted.
Run once per GTAV_LN_FLAGS value (experiments build reads it): bit 0 residual store sc1, bit 1 fp16 operand store sc1, bit 2 non-temporal loads.
Usage (GPU box): GTAV_LN_FLAGS=7 python tools/ln_bench.py [--ms 720 5760] [--splitk 1]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gtav_amd import lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", type=int, nargs="+", default=[720, 5760])
    ap.add_argument("--splitk", type=int, nargs="+", default=[1, 4])
    a = ap.parse_args()
    lib = L.load_experiments()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    N, K, P = 1024, 64, 144
    for M in a.ms:
        for sk in a.splitk:
            if K // 64 % sk:
                Kk = 64 * sk
            else:
                Kk = K
            nb = 4   # rotate residual / output buffers like consecutive layers do not (the model reuses ONE residual): keep one, as in the model
            x = torch.randn(((M + 127) // 128 * 128) * Kk, device=dev).half()
            w = torch.randn(N * Kk, device=dev).half() * 0.03
            b = torch.randn(N, device=dev)
            resid = torch.randn(M, N, device=dev)
            mod = torch.randn(M // P, 3 * N, device=dev) * 0.1
            parts = torch.zeros(sk, M, N, device=dev)
            out = torch.zeros((M + 127) // 128 * 128, N, device=dev, dtype=torch.float16)

            def both():
                L.check(lib.gtav_op_gemm_splitk_ln(x.data_ptr(), Kk, w.data_ptr(), b.data_ptr(), M, N, Kk, sk, parts.data_ptr(), resid.data_ptr(), mod.data_ptr(), 3 * N, P,
                                                   out.data_ptr(), mod[:, N:].data_ptr(), mod[:, 2 * N:].data_ptr(), 3 * N, st))

            def gemm_only():
                L.check(lib.gtav_op_gemm_f16(x.data_ptr(), Kk, w.data_ptr(), 0, parts.data_ptr(), N, M, N, Kk, 6, 0, sk, 1, st))

            def timeit(fn, n=200):
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    fn()
                    resid.mul_(0.5) if False else None
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e3 / n
            tb, tg = timeit(both), timeit(gemm_only)
            print(f"GTAV_LN_FLAGS={os.environ.get('GTAV_LN_FLAGS', 'default')} M={M} slabs={sk}: GEMM+LN {tb:.2f} us, GEMM {tg:.2f} us, LayerNorm {tb - tg:.2f} us", flush=True)


if __name__ == "__main__":
    main()
