"""A/B of GEMM shape selections on the REAL DiT forward in ONE process on ONE GPU (boxes differ by up to ~10 % in MFMA-bound kernels,
so numbers from different gpurun calls do not compare).  Experiments build; alternates the variants `--rounds` times.
Usage (GPU box): python tools/forward_ab.py [--batch 1] [--variants 0 64]   (variant = gemm debug bits; 64 = round-1 shapes)"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gtav_amd import lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--variants", type=int, nargs="+", default=[0, 64])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--actions", action="store_true")
    ap.add_argument("--fused-ab", action="store_true", help="A/B the fused temporal QKV + attention kernel instead: variant 0 = two-kernel path, 1 = fused")
    ap.add_argument("--depth", type=int, default=16, help="fewer blocks: the weights then stay in the 256 MiB Infinity Cache between forwards")
    ap.add_argument("--fold-ab", action="store_true", help="A/B the LayerNorm fold instead: variant = fold mode (0 off, 1 default policy, 2 every seam, "
                    "3 = seam A only, 4 = seam B only); the PRODUCT library unless --exp")
    ap.add_argument("--exp", action="store_true")
    a = ap.parse_args()
    lib = L.load_experiments()   # (the LayerNorm fold and the A/B knobs exist in the experiments build only)
    import gtav_amd.weights as W
    from gtav_amd.model.dit import DiT_models
    dev = torch.device("cuda", 0)
    B = a.batch
    from gtav_amd.model.dit import DiT
    dit = DiT(depth=a.depth, init_weights=False, max_batch=B)
    dit.load_state_dict(W.synth_state_dict(W.dit_param_shapes(depth=a.depth), seed=0))
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 5, 16, 18, 32, generator=g).to(dev)
    t = torch.tensor([[15, 15, 15, 15, 500]] * B)
    act = None
    if a.actions:
        act = torch.zeros(B, 5, 25, device=dev)
        act[:, :, 3] = 1
    ref = None
    for r in range(a.rounds):
        for v in a.variants:
            if a.fold_ab:
                if v == 3: dit.set_fold(1, 0, 1 << 30)
                elif v == 4: dit.set_fold(1, 1 << 30, 0)
                else: dit.set_fold(v, 1024, 1024)   # (variant 1 = mode 1 with 1024-token thresholds)
            elif a.fused_ab:
                dit.set_fused_temporal(bool(v & 1))
                lib.gtav_op_gemm_set_debug(v & ~1)    # e.g. 513 = fused + debug bit 9 (no K/V cache rows: timing only)
            else:
                lib.gtav_op_gemm_set_debug(v)
            for _ in range(3):
                out = dit(x, t, act)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            err = ((out - ref).norm() / ref.norm()).item()
            t0 = time.perf_counter()
            n = 20
            for _ in range(n):
                dit(x, t, act)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            dit.profile(True)
            for _ in range(4):
                dit(x, t, act)
            torch.cuda.synchronize()
            prof = dit.profile_read()
            dit.profile(False)
            cls = " ".join(f"{k.replace('gemm_', '')}={v[0] / max(v[1], 1) * 1e3:.2f}x{v[1] // 4}" for k, v in prof.items() if v[1] and k != "empty_event_pair")
            tot = sum(v[0] for k, v in prof.items() if k != "empty_event_pair") / 4
            print(f"round {r} variant {v:3d}: forward {ms:.3f} ms (rel diff vs first {err:.1e}) kernels {tot:.3f} ms | us per launch x launches: {cls}", flush=True)
    if hasattr(lib, "gtav_op_gemm_set_debug"):
        lib.gtav_op_gemm_set_debug(0)


if __name__ == "__main__":
    main()
