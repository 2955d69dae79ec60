"""LayerNorm without a pending update at the row counts of the batched paths (B = 8 window 5 760, trainer 11 520, VAE 23 040 / 46 080), from HBM (three
rotating residual buffers).  The experiments build reads GTAV_LN_FLAGS (bit 0 residual store sc1 — unused without a pending update —, bit 1 fp16 operand
store as paired 16-byte sc1 stores, bit 2 non-temporal loads of a pending update), GTAV_LN_WAVE_ROW_MIN (first M of the wave-per-row kernel): run once per setting (the checksum changes by rounding between the two kernels, not with the flags).
  GTAV_LN_FLAGS=7 python tools/ln_large_m_bench.py ; GTAV_LN_FLAGS=5 python tools/ln_large_m_bench.py"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gtav_amd import lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", type=int, nargs="+", default=[5760, 11520, 23040, 46080])
    a = ap.parse_args()
    lib = L.load_experiments()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    D = 1024
    tag = f"LN_FLAGS={os.environ.get('GTAV_LN_FLAGS', 'default')} WAVE_ROW_MIN={os.environ.get('GTAV_LN_WAVE_ROW_MIN', 'default')}"
    for M in a.ms:
        g = torch.Generator(device="cpu").manual_seed(M)
        xs = [torch.randn(M, D, generator=g).to(dev) * 3 + 0.5 for _ in range(3)]
        gamma, beta = torch.randn(D, generator=g).to(dev), torch.randn(D, generator=g).to(dev)
        P = 144
        shift, scale = (torch.randn(M // P, D, generator=g) * 0.1).to(dev), (torch.randn(M // P, D, generator=g) * 0.1).to(dev)
        out = torch.zeros((M + 127) // 128 * 128, D, device=dev, dtype=torch.float16)
        for name in ("affine", "modulate"):
            def run(i):
                if name == "affine":
                    L.check(lib.gtav_op_ln_affine(xs[i % 3].data_ptr(), out.data_ptr(), M, D, gamma.data_ptr(), beta.data_ptr(), st))
                else:
                    L.check(lib.gtav_op_ln_modulate(xs[i % 3].data_ptr(), out.data_ptr(), M, D, shift.data_ptr(), scale.data_ptr(), D, P, st))
            for i in range(6):
                run(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = 60
            for i in range(n):
                run(i)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            run(0)
            torch.cuda.synchronize()
            ck = int(out.view(torch.int16).to(torch.int64).sum().item())
            print(f"{tag} {name:8s} M={M:6d}: {us:7.2f} us  {6.0 * M * D / us / 1e6:5.2f} TB/s  checksum {ck}", flush=True)
        # the same bytes by torch's copy kernels, for scale: fp32 -> fp16 conversion of the same matrix
        o2 = torch.empty(M, D, device=dev, dtype=torch.float16)
        for i in range(6):
            o2.copy_(xs[i % 3])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(60):
            o2.copy_(xs[i % 3])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 60
        print(f"{tag} torch fp32->fp16 copy M={M:6d}: {us:7.2f} us  {6.0 * M * D / us / 1e6:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
