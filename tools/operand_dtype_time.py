"""Forward time of the full-size DiT with fp16 operands (default) and with every layer group on bf16 operands (DiT.set_operand_dtype; the same kernels compiled for
v_mfma_f32_16x16x32_bf16), batch 1 and batch 8, and the captured batch-1 sampler step; plus the VAE encode of 8 frames.  Product library.
  usage (GPU box): python tools/operand_dtype_time.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import gtav_amd.weights as W  # noqa: E402
from gtav_amd.model.dit import DiT_models  # noqa: E402
from gtav_amd.model.vae import VAE_models  # noqa: E402
from gtav_amd.utils import alphas_cumprod  # noqa: E402


def timeit(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    m = DiT_models["DiT-S/2"](init_weights=False, max_batch=8)
    m.load_state_dict(W.synth_state_dict(W.dit_param_shapes(depth=16), seed=0))
    m.set_schedule(alphas_cumprod(1e-4))
    g = torch.Generator().manual_seed(1)
    v = VAE_models["vit-l-20-shallow-encoder"](init_weights=False, max_frames_per_call=8)
    v.load_state_dict(W.synth_state_dict(W.vae_param_shapes(), seed=1))
    img = (torch.rand(8, 3, 360, 640, generator=g) * 2 - 1).to(dev)
    for dt in (torch.float16, torch.bfloat16, torch.float16):
        m.set_operand_dtype(dt)
        v.set_operand_dtype(dt)
        row = [str(dt)]
        for B in (1, 8):
            x = torch.randn(B, 5, 16, 18, 32, generator=g).to(dev)
            t = torch.tensor([[15, 15, 15, 15, 500]] * B)
            row.append("forward B=%d %.3f ms" % (B, timeit(lambda: m(x, t, None), 20)))
        xs = (torch.randn(1, 6, 16, 18, 32, generator=g) * 0.5).to(dev)
        ts = [999 - 9 * k for k in range(40)]
        m.prepare_frame_(1, 6, 1, 5, 15, ts, None)
        for k in range(3):
            m.denoise_step_(xs, 1, 5, 15, ts[k], ts[k + 1], False, None, cond_step=k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(3, 39):
            m.denoise_step_(xs, 1, 5, 15, ts[k], ts[k + 1], False, None, cond_step=k)
        torch.cuda.synchronize()
        row.append("captured step B=1 %.3f ms" % ((time.perf_counter() - t0) / 36 * 1e3))
        row.append("VAE encode 8 frames %.3f ms" % timeit(lambda: v.encode(img), 10))
        print("  ".join(row), flush=True)
        m.check()
        v.check()


if __name__ == "__main__":
    main()
