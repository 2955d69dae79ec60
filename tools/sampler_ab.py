"""A/B of GEMM debug bits on the CAPTURED sampler step (hipGraph replay, the path the headline measures) in ONE process on ONE GPU: a fresh
DiT handle per variant (a captured step bakes the kernel parameters in), variants alternating.  Experiments build.
Usage (GPU box): python tools/sampler_ab.py [--variants 0 8388608] [--batch 1] [--frames 3] [--steps 50] [--cached]"""
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
    ap.add_argument("--variants", type=int, nargs="+", default=[0, 8388608])
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--frames", type=int, default=3, help="generated frames per measurement")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--cached", action="store_true")
    ap.add_argument("--actions", action="store_true")
    ap.add_argument("--g256", action="store_true", help="the 256x256-frame preset: 16x16 latents, 64 tokens per frame (M = 320 per window step)")
    ap.add_argument("--product", action="store_true", help="the product library (no debug bits: one variant) instead of the experiments build")
    ap.add_argument("--fused-ab", action="store_true", help="every variant also with the fused temporal QKV + attention kernel")
    ap.add_argument("--fused-spatial-ab", action="store_true", help="every variant also with the fused spatial QKV + attention kernel, and with both fused kernels")
    ap.add_argument("--graph-ab", action="store_true", help="every variant also with eager (stream-ordered) launches instead of the captured graph")
    a = ap.parse_args()
    lib = L.load() if a.product else L.load_experiments()
    if a.product:
        a.variants = [0]
        lib.gtav_op_gemm_set_debug = lambda v: None
    import gtav_amd.weights as W
    from gtav_amd.generate import generate_latents
    from gtav_amd.model.dit import DiT, DiT_models
    dev = torch.device("cuda", 0)
    B = a.batch
    gkw = dict(input_h=16, input_w=16, patch_size=2, in_channels=16, hidden_size=1024, depth=16, num_heads=16, external_cond_dim=25)
    sd = W.synth_state_dict(W.dit_param_shapes(**gkw) if a.g256 else W.dit_param_shapes(depth=16), seed=0)
    lat = (16, 16) if a.g256 else (18, 32)
    g = torch.Generator().manual_seed(3)
    n_prompt, total = 4, 4 + a.frames
    x0 = torch.randn(B, n_prompt, 16, *lat, generator=g) * 0.5
    nz = torch.randn(B, total - n_prompt, 16, *lat, generator=g)
    act = None
    if a.actions:
        act = torch.zeros(B, total, 25)
        act[:, :, 3] = 1
    models = {}
    for v in a.variants:
        for graph in ((True, False) if a.graph_ab else ((True, "fused") if a.fused_ab else (True, "fused", "fused-s", "fused-st") if a.fused_spatial_ab else (True,))):
            lib.gtav_op_gemm_set_debug(v)
            m = DiT(**gkw, init_weights=False, max_batch=B) if a.g256 else DiT_models["DiT-S/2"](init_weights=False, max_batch=B)
            m.load_state_dict(sd)
            if graph in ("fused", "fused-st"):
                m.set_fused_temporal(True)
            if a.fused_spatial_ab:                      # explicit either way: the fused spatial launch is the library's default at 144 tokens per frame
                m.set_fused_spatial(graph in ("fused-s", "fused-st"))
            generate_latents(m, x0, total, 4, nz, act, ctx_cache=a.cached)   # builds the handle; warm-up + capture under this variant's bits
            if not graph:
                m.set_graph(False)
                generate_latents(m, x0, total, 4, nz, act, ctx_cache=a.cached)
            torch.cuda.synchronize()
            models[(v, graph)] = m
    ref = None
    for r in range(a.rounds):
        for (v, graph), m in models.items():
            lib.gtav_op_gemm_set_debug(v)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = generate_latents(m, x0, total, a.steps, nz, act, ctx_cache=a.cached)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if ref is None:
                ref = out.clone()
            nf = a.frames * (a.steps + 1)
            print(f"round {r} variant {v:9d} { {'fused': 'graph+fused-temporal', 'fused-s': 'graph+fused-spatial', 'fused-st': 'graph+fused-both'}.get(graph, ('graph, both split' if a.fused_spatial_ab else 'graph') if graph else 'eager') }: {dt / nf * 1e3:.4f} ms per sampler step ({nf} steps, batch {B}, {'cached' if a.cached else 'window'}), "
                  f"rel diff vs first {((out - ref).norm() / ref.norm()).item():.1e}", flush=True)
    lib.gtav_op_gemm_set_debug(0)


if __name__ == "__main__":
    main()
