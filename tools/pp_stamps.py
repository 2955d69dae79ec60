"""Timeline of the persistent ping-pong GEMM (experiments build): per block, s_memrealtime at entry, at the end of each tile's MAIN
loop and at the end of the last epilogue (csrc/gemm.hip gemm_pp_kernel).  Prints the phase durations.
Usage (GPU box): python tools/pp_stamps.py [--m 5760] [--only fc1,qkv]"""
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
    ap.add_argument("--only", type=str, default="fc1,qkv,out,fc2")
    a = ap.parse_args()
    lib = L.load_experiments()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    shapes = [("qkv", 3072, 1024, 5), ("out", 1024, 1024, 4), ("fc1", 4096, 1024, 2), ("fc2", 1024, 4096, 4)]
    stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    for M in a.m:
        for name, N, K, epi in shapes:
            if name not in a.only.split(","):
                continue
            x = (torch.randn((M + 127) // 128 * 128, K, device=dev) * 0.5).half()
            w = (torch.randn((N + 127) // 128 * 128, K, device=dev) * 0.03).half()
            bias = torch.randn(N, device=dev)
            out = torch.zeros(((M + 127) // 128 * 128), N, device=dev, dtype=torch.float32 if epi == 4 else torch.float16)
            q = torch.empty(3, M, 1024, device=dev, dtype=torch.float16)
            cs = torch.ones(144, 64, device=dev)
            lib.gtav_op_gemm_set_wm(16)

            def run():
                if epi == 5:
                    L.check(lib.gtav_op_gemm_qkv(x.data_ptr(), K, w.data_ptr(), 0, (M // 144) * 144, 1024, 0, q[0].data_ptr(), q[1].data_ptr(),
                                                 q[2].data_ptr(), 144, 0, 0, 0, cs.data_ptr(), st))
                else:
                    L.check(lib.gtav_op_gemm_f16(x.data_ptr(), K, w.data_ptr(), bias.data_ptr(), out.data_ptr(), N, M, N, K, epi, 0, 0, 1, st))
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            lib.gtav_op_gemm_set_stamps(stamps.data_ptr(), 4096)
            stamps.zero_()
            torch.cuda.synchronize()
            run()
            torch.cuda.synchronize()
            lib.gtav_op_gemm_set_stamps(None, 0)
            lib.gtav_op_gemm_set_wm(0)
            s = stamps.cpu().reshape(-1, 8)
            s = s[s[:, 0] != 0].double()
            t0 = s[:, 0].min()
            f = lambda v: (v - t0) / 100.0
            nt = s[:, 1]
            print(f"{name} M={M} N={N} K={K}: {us:.2f} us/launch; {len(s)} blocks, tiles per block {sorted(set(nt.int().tolist()))}, span {float(f(s[:, 7]).max()):.2f} us")
            print(f"    entry after first: med {float(f(s[:, 0]).median()):.2f} max {float(f(s[:, 0]).max()):.2f}")
            prev = s[:, 0]
            for i in range(5):
                col = s[:, 2 + i]
                ok = col != 0
                if not ok.any():
                    break
                dur = (col[ok] - prev[ok]) / 100.0
                print(f"    MAIN({i}) end: med {float(f(col[ok]).median()):6.2f} us   (since previous mark: med {float(dur.median()):5.2f} p90 {float(dur.quantile(0.9)):5.2f})  [{int(ok.sum())} blocks]")
                prev = torch.where(ok, col, prev)
            tail = (s[:, 7] - prev) / 100.0
            print(f"    last epilogue after the last MAIN: med {float(tail.median()):.2f} p90 {float(tail.quantile(0.9)):.2f} us; block end med {float(f(s[:, 7]).median()):.2f} max {float(f(s[:, 7]).max()):.2f}")


if __name__ == "__main__":
    main()
