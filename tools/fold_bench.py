"""Micro-benchmark of the LayerNorm-fold producer GEMM (EPI_RESID_FOLD) against the split-K slab GEMM it replaces, back to back, with pieces
of its epilogue switched off (experiments build: gemm debug bits 0x10000 no residual store, 0x20000 no operand copy-out, 0x40000 no statistics
store, 0x80000 plain instead of write-through residual stores, 0x100000 no gate / scale loads).  Timing only: results are wrong with bits set.
Usage (GPU box): python tools/fold_bench.py [--m 5760] [--k 1024 4096]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gtav_amd import lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, nargs="+", default=[5760])
    ap.add_argument("--k", type=int, nargs="+", default=[1024, 4096])
    ap.add_argument("--iters", type=int, default=64)
    ap.add_argument("--wm", type=int, nargs="+", default=[0])
    ap.add_argument("--bits", type=lambda v: int(v, 0), nargs="+", default=[0, 0x10000, 0x20000, 0x40000, 0x80000, 0x100000, 0x70000, 0x170000])
    a = ap.parse_args()
    lib = L.load_experiments()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    N, P, copies = 1024, 144, 8
    for M in a.m:
        for K in a.k:
            Mp = (M + 127) // 128 * 128
            x = (torch.randn(Mp, K, device=dev) * 0.5).half()
            ws = [(torch.randn(N, K, device=dev) * 0.03).half() for _ in range(copies)]
            bias = torch.randn(N, device=dev)
            resid = torch.randn(M, N, device=dev)
            mod = torch.randn(M // P, 2 * N, device=dev) * 0.1
            aout = torch.empty(Mp, N, device=dev, dtype=torch.float16)
            stats = torch.empty(M, N // 64, 2, device=dev)
            sk = lib.gtav_op_gemm_choose_splitk(M, N, K)
            parts = torch.empty(sk * Mp, N, device=dev)

            def timed(fn):
                for i in range(4):
                    fn(i)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for i in range(a.iters):
                    fn(i)
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e3 / a.iters
            for wm in a.wm:
                lib.gtav_op_gemm_set_wm(wm)
                lib.gtav_op_gemm_set_debug(0)
                us = timed(lambda i: L.check(lib.gtav_op_gemm_f16(x.data_ptr(), K, ws[i % copies].data_ptr(), 0, parts.data_ptr(), N, M, N, K, 6, 0, sk, 1, st)))
                print(f"M={M} K={K} wm={wm}: split-K slabs (splitk {sk})                 {us:8.2f} us", flush=True)
                us = timed(lambda i: L.check(lib.gtav_op_gemm_f16(x.data_ptr(), K, ws[i % copies].data_ptr(), 0, parts.data_ptr(), N, M, N, K, 6, 0, 1, 1, st)))
                print(f"M={M} K={K} wm={wm}: one slab, full K                          {us:8.2f} us", flush=True)
                us = timed(lambda i: L.check(lib.gtav_op_gemm_f16(x.data_ptr(), K, ws[i % copies].data_ptr(), bias.data_ptr(), resid.data_ptr(), N, M, N, K, 4,
                                                                   mod.data_ptr(), 2 * N, P, st)))
                print(f"M={M} K={K} wm={wm}: in-place gated residual (EPI_RESID)        {us:8.2f} us", flush=True)
                for bits in a.bits:
                    lib.gtav_op_gemm_set_debug(bits)
                    try:
                        us = timed(lambda i: L.check(lib.gtav_op_gemm_fold_producer(x.data_ptr(), ws[i % copies].data_ptr(), bias.data_ptr(), resid.data_ptr(), M, N, K,
                                                                                     mod.data_ptr(), mod[:, N:].data_ptr(), 2 * N, P, aout.data_ptr(), stats.data_ptr(), st)))
                        print(f"M={M} K={K} wm={wm}: fold producer, debug bits {bits:#9x}          {us:8.2f} us", flush=True)
                    except L.GtavError as e:
                        print(f"M={M} K={K} wm={wm}: fold producer skipped: {e}")
    lib.gtav_op_gemm_set_debug(0)
    lib.gtav_op_gemm_set_wm(0)


if __name__ == "__main__":
    main()
