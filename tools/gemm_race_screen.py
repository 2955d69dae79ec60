"""Race screen for the GEMM block shapes (GPU): every shape runs the same problem `--reps` times, every output must be bitwise
identical to the first run's and match fp32 math on the same fp16 operands.  An LDS-DMA read placed before the wait that retires
it passes a single parity check whenever the copy happens to land first; repeated runs at several sizes are what catch it.
  python tools/gemm_race_screen.py [--reps 200]"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gtav_amd import lib as L  # noqa: E402
from helpers import dev, gemm, pad_weight_f16, rel_l2, to_tiled_f16, untile  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--shapes", type=int, nargs="*", default=None, help="block shapes to screen (default: every shape the library has)")
    a = ap.parse_args()
    lib = L.load()
    cases = [(720, 4096, 1024), (5760, 1024, 1024), (144, 3072, 1024), (333, 512, 4096), (2880, 3072, 1024), (1152, 4096, 1024)]
    bad = 0
    ap_shapes = a.shapes or [2, 3, 7, 11, 12, 13, 14, 20, 24, 26, 29, 31]
    for shape in ap_shapes:
        for (M, N, K) in cases:
            g = torch.Generator().manual_seed(M + N + K)
            x = (torch.randn(M, K, generator=g)).half()
            w = torch.randn(N, K, generator=g) / math.sqrt(K)
            b = torch.randn(N, generator=g)
            xd, wd, bd = to_tiled_f16(x), pad_weight_f16(w), b.to(dev())
            ref = x.float() @ w.half().float().t() + b
            resid = torch.randn(M, N, generator=g).to(dev())
            for epi in ((0, 2, 4, 6) if shape != 8 else (0,)):        # fp32 row-major, GELU tile-major, in-place residual (round 3), one K slice of slabs
                lib.gtav_op_gemm_set_wm(shape)
                outs = []
                first = None
                ok = True
                for r in range(a.reps):
                    if epi in (0, 6):
                        out = torch.full((M, N), float("nan"), device=dev())
                    elif epi == 4:
                        out = resid.clone()
                    else:
                        out = torch.zeros(((M + 127) // 128 * 128, N), device=dev(), dtype=torch.float16)
                    try:
                        if epi == 6:
                            L.check(lib.gtav_op_gemm_f16(xd.data_ptr(), K, wd.data_ptr(), 0, out.data_ptr(), N, M, N, K, 6, 0, 1, 1, L.current_stream()))
                        elif epi == 4:
                            L.check(lib.gtav_op_gemm_f16(xd.data_ptr(), K, wd.data_ptr(), bd.data_ptr(), out.data_ptr(), N, M, N, K, 4, 0, 0, 1, L.current_stream()))
                        else:
                            gemm(xd, wd, bd, M, N, K, epi, out, N)
                    except L.GtavError as e:        # a shape that does not take this problem (e.g. the ping-pong kernel's K >= 768)
                        err = float("nan")
                        print(f"shape {shape:2d} epi {epi} M={M:5d} N={N:5d} K={K:5d}: skipped ({e})")
                        ok = None
                        break
                    if first is None:
                        first = out.clone()
                        got = untile(first, M, N).float() if epi == 2 else first.float().cpu()
                        want = {0: ref, 2: torch.nn.functional.gelu(ref, approximate="tanh"), 4: resid.cpu() + ref, 6: ref - b}[epi]
                        err = rel_l2(got, want)
                        ok = err < (6e-4 if epi == 2 else 2e-5)
                    elif not torch.equal(out, first):
                        ok = False
                        break
                lib.gtav_op_gemm_set_wm(0)
                if ok is None:
                    continue
                if not ok:
                    bad += 1
                print(f"shape {shape:2d} epi {epi} M={M:5d} N={N:5d} K={K:5d}: {'ok' if ok else 'MISMATCH'} (rel-L2 {err:.2e}, {a.reps} reps)")
    print("race screen:", "clean" if bad == 0 else f"{bad} failing cases")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
