"""Per-K-step timeline of the loader-wave GEMM kernels (experiments build, debug bit 5): for every block, loader wave 0 stamps
s_memrealtime when it passes barrier t and compute wave 0 when it ARRIVES at barrier t.  Prints the median time of each K-step and
how long the compute wave waited at each barrier.   Usage (GPU box): python tools/kstep_stamps.py [--m 720] [--wm 20] [--only fc1]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gtav_amd import lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=720)
    ap.add_argument("--wm", type=int, default=20)
    ap.add_argument("--only", type=str, default="fc1")
    ap.add_argument("--debug", type=int, default=0, help="extra debug bits (1 = no refills, 2 = no MFMA)")
    a = ap.parse_args()
    lib = L.load_experiments()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    shapes = {"qkv": (3072, 1024, 5), "out": (1024, 1024, 6), "fc1": (4096, 1024, 2), "fc2": (1024, 4096, 6)}
    N, K, epi = shapes[a.only]
    M = a.m
    maxb = 4096
    stamps = torch.zeros(maxb * 64, dtype=torch.int64, device=dev)
    x = (torch.randn((M + 127) // 128 * 128, K, device=dev) * 0.5).half()
    ws = [(torch.randn((N + 127) // 128 * 128, K, device=dev) * 0.03).half() for _ in range(8)]
    bias = torch.randn(N, device=dev)
    sk = lib.gtav_op_gemm_choose_splitk(M, N, K) if epi == 6 else 1
    Mp = (M + 127) // 128 * 128
    out = torch.empty((max(sk, 1) * Mp, N), device=dev, dtype=torch.float32 if epi in (0, 6) else torch.float16)
    q = torch.empty(3, M, 1024, device=dev, dtype=torch.float16)
    cs = torch.ones(144, 64, device=dev)
    lib.gtav_op_gemm_set_wm(a.wm)
    lib.gtav_op_gemm_set_debug(32 | a.debug)

    def run(i):
        w = ws[i % 8]
        if epi == 5:
            L.check(lib.gtav_op_gemm_qkv(x.data_ptr(), K, w.data_ptr(), 0, (M // 144) * 144, 1024, 0, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(),
                                         144, 0, 0, 0, cs.data_ptr(), st))
        elif epi == 6:
            L.check(lib.gtav_op_gemm_f16(x.data_ptr(), K, w.data_ptr(), 0, out.data_ptr(), N, M, N, K, 6, 0, sk, 1, st))
        else:
            L.check(lib.gtav_op_gemm_f16(x.data_ptr(), K, w.data_ptr(), bias.data_ptr(), out.data_ptr(), N, M, N, K, epi, 0, 0, 1, st))

    for i in range(8):
        run(i)
    torch.cuda.synchronize()
    lib.gtav_op_gemm_set_stamps(stamps.data_ptr(), maxb)
    for r in range(3):
        stamps.zero_()
        torch.cuda.synchronize()
        run(r)
        torch.cuda.synchronize()
    s = stamps.cpu().reshape(-1, 64)
    s = s[s[:, 0] != 0]
    lib.gtav_op_gemm_set_stamps(None, 0)
    lib.gtav_op_gemm_set_debug(0)
    lib.gtav_op_gemm_set_wm(0)
    nk = K // 64 // max(sk, 1)
    t0 = s[:, 0:1]
    ld = (s[:, 8:8 + nk] - t0).double() / 100.0          # loader 0 past barrier t (us after block entry)
    cw = (s[:, 32:32 + nk] - t0).double() / 100.0        # compute wave 0 arrives at barrier t
    med = lambda v: float(v.median())
    print(f"{a.only} M={M} wm={a.wm} splitk={sk} debug={a.debug}: {s.shape[0]} blocks, {nk} K-steps; first fill at {med((s[:, 7] - s[:, 0]).double() / 100):.2f} us, "
          f"end of main loop {med((s[:, 2] - s[:, 0]).double() / 100):.2f}, end of block {med((s[:, 3] - s[:, 0]).double() / 100):.2f}")
    print("   t : barrier passed (loader 0) | step time | compute wave 0 arrived | waited at the barrier")
    for t in range(nk):
        dt = med(ld[:, t] - ld[:, t - 1]) if t else float("nan")
        print(f"  {t:2d} : {med(ld[:, t]):6.2f} | {dt:5.2f} | {med(cw[:, t]):6.2f} | {med(ld[:, t] - cw[:, t]):5.2f}")


if __name__ == "__main__":
    main()
