"""Micro-benchmark of the spatial attention kernel through the C-ABI (GPU): python tools/attn_bench.py [--nb 5 40 80]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gtav_amd import lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nb", type=int, nargs="+", default=[5, 40, 80])
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--s", type=int, default=144)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--exp", action="store_true", help="experiments build (libgtav_amd_exp.so): honours GTAV_ATTN_FLASH_NQT / GTAV_ATTN_FLASH_OCC3")
    a = ap.parse_args()
    lib = L.load_experiments() if a.exp else L.load()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    for NB in a.nb:
        S, H = a.s, a.heads
        # rotate over several buffer sets so inputs are not L2-resident from the previous launch
        sets = []
        for _ in range(4):
            q = (torch.randn(NB, H, S, 64, device=dev) * 0.5).half()
            k = (torch.randn(NB, H, S, 64, device=dev) * 0.5).half()
            vt = torch.randn(NB, H, 64, S, device=dev).half()
            out = torch.zeros((NB * S + 127) // 128 * 128, H * 64, device=dev, dtype=torch.float16)
            sets.append((q, k, vt, out))

        def run(i):
            q, k, vt, out = sets[i % 4]
            L.check(lib.gtav_op_attn_spatial(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), NB, H, S, st))
        for i in range(8):
            run(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(a.iters):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        mb = NB * H * S * 64 * 2 * 4 / 1e6
        print(f"attn_spatial NB={NB:4d} heads={H} S={S}: {us:8.2f} us   {mb:7.1f} MB moved -> {mb / us:5.2f} TB/s   "
              f"{4.0 * NB * H * S * S * 64 / us / 1e6:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
