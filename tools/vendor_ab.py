"""Same-box, same-process A/B of the product's large-M GEMMs (csrc/gemm.hip, WITH their fused epilogues, through the C-ABI) against the vendor library
(torch.matmul -> hipBLASLt, fp16 in / fp16 out, NO epilogue) at the token counts of BASELINE configs[4]: M = 11 520 (DiT, batch 16 x 5 x 144) and
M = 23 040 / 46 080 (VAE encode of 40 / 80 frames x 576).  Round 6, VERDICT r5 item 3: the round-3 yardstick was taken on another box and never at the
VAE's sizes.  Per (M, class): `--rounds` alternations of {ours x iters, vendor x iters}, best of each; weights rotate over `--copies` buffers.
Under `rocprofv3 --kernel-trace --stats` the vendor's kernel names (macro-tile, depth-U, wave grid) land in the statistics.
  usage (GPU box): python tools/vendor_ab.py [--ms 11520 23040 46080]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gtav_amd import lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", type=int, nargs="+", default=[11520, 23040, 46080])
    ap.add_argument("--copies", type=int, default=12)
    ap.add_argument("--iters", type=int, default=48)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    lib = L.load()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    print(f"{'class':>9} {'M':>6} {'N':>5} {'K':>5} {'ours us':>9} {'ours TF':>8} {'vendor us':>10} {'vendor TF':>9} {'vendor/ours':>11}")
    for M in a.ms:
        vae = M % 576 == 0 and M >= 23040
        # (name, N, K, kind): kinds — qkv (RoPE + head layout; S = 144 or 576 + bias), gelu (tanh for the DiT, erf for the VAE), resid (in place, gated for the DiT)
        classes = [("qkv", 3072, 1024, "qkv"), ("out/proj", 1024, 1024, "resid"), ("fc1", 4096, 1024, "gelu"), ("fc2", 1024, 4096, "resid")]
        for name, N, K, kind in classes:
            Mp = (M + 127) // 128 * 128
            x = (torch.randn(Mp, K, device=dev) * 0.5).half()
            ws = [(torch.randn(N, K, device=dev) * 0.03).half() for _ in range(a.copies)]
            bias = torch.randn(N, device=dev)
            S = 576 if vae else 144
            q = torch.empty(3, Mp, 1024, device=dev, dtype=torch.float16)
            cs = torch.ones(S, 64, device=dev)
            out16 = torch.zeros(Mp, N, device=dev, dtype=torch.float16)
            resid = torch.zeros(M, N, device=dev)
            gate = torch.full((M // S + 1, N), 1e-3, device=dev)
            vout = torch.empty(M, N, device=dev, dtype=torch.float16)
            xr = x[:M]

            def ours(i):
                w = ws[i % a.copies]
                if kind == "qkv":
                    L.check(lib.gtav_op_gemm_qkv(x.data_ptr(), K, w.data_ptr(), bias.data_ptr() if vae else 0, M, 1024, 0, q[0].data_ptr(), q[1].data_ptr(),
                                                 q[2].data_ptr(), S, 0, 0, 0, cs.data_ptr(), st))
                elif kind == "gelu":
                    L.check(lib.gtav_op_gemm_f16(x.data_ptr(), K, w.data_ptr(), bias.data_ptr(), out16.data_ptr(), N, M, N, K, 3 if vae else 2, 0, 0, 1, st))
                else:
                    L.check(lib.gtav_op_gemm_f16(x.data_ptr(), K, w.data_ptr(), bias.data_ptr(), resid.data_ptr(), N, M, N, K, 4,
                                                 0 if vae else gate.data_ptr(), 0 if vae else N, 0 if vae else S, st))

            def vendor(i):
                torch.matmul(xr, ws[i % a.copies].t(), out=vout)

            best = {}
            for fn_name, fn in (("ours", ours), ("vendor", vendor)):
                for i in range(6):
                    fn(i)
            torch.cuda.synchronize()
            for _ in range(a.rounds):
                for fn_name, fn in (("ours", ours), ("vendor", vendor)):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for i in range(a.iters):
                        fn(i)
                    e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) * 1e3 / a.iters
                    best[fn_name] = min(best.get(fn_name, 1e30), us)
            fl = 2.0 * M * N * K
            print(f"{name:>9} {M:6d} {N:5d} {K:5d} {best['ours']:9.2f} {fl / best['ours'] / 1e6:8.1f} {best['vendor']:10.2f} {fl / best['vendor'] / 1e6:9.1f} "
                  f"{best['vendor'] / best['ours']:11.3f}", flush=True)
            del x, ws, q, out16, resid, vout
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
