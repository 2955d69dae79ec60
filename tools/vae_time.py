"""VAE encode / decode time as the generation harness runs them (ViT-L/20, 360x640 frames): product library.  Usage (GPU box): python tools/vae_time.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gtav_amd import lib as L  # noqa: E402


def main():
    L.load()
    import gtav_amd.weights as W
    from gtav_amd.model.vae import VAE_models
    dev = torch.device("cuda", 0)
    vae = VAE_models["vit-l-20-shallow-encoder"](init_weights=False, max_frames_per_call=32)
    vae.load_state_dict(W.synth_state_dict(W.vae_param_shapes(), seed=1))
    g = torch.Generator().manual_seed(0)
    img = (torch.rand(4, 3, 360, 640, generator=g) * 2 - 1).to(dev)
    z = torch.randn(32, 576, 16, generator=g).to(dev)
    for name, fn in (("encode 4 frames", lambda: vae.encode(img)), ("decode 32 frames", lambda: vae.decode(z))):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms", flush=True)


if __name__ == "__main__":
    main()
