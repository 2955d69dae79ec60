"""Host-side schedule tables (reference utils.py:30-48). float64 table construction on the CPU —
constants, not model compute."""
import torch


def sigmoid_beta_schedule(timesteps, start=-3, end=3, tau=1.0, clamp_min=1e-4):
    """utils.py:30-48: sigmoid alpha-bar rescaled to [clamp_min, 1]; returns float64 betas."""
    steps = timesteps + 1
    t = torch.linspace(0, timesteps, steps, dtype=torch.float64) / timesteps
    v_start = torch.tensor(start / tau).sigmoid()
    v_end = torch.tensor(end / tau).sigmoid()
    ac = (-((t * (end - start) + start) / tau).sigmoid() + v_end) / (v_end - v_start)
    ac = ac / ac[0]
    ac = ac * (1 - clamp_min) + clamp_min
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


def alphas_cumprod(clamp_min=1e-4, max_noise_level=1000):
    """generate.py:195-198 / train_dit.py:292-297: fp32 cumprod of (1 - betas.float()), shape (1000,)."""
    return torch.cumprod(1.0 - sigmoid_beta_schedule(max_noise_level, clamp_min=clamp_min).float(), dim=0)


def visualize_step(x_curr, x_noisy, noise, v, step, vae, alphas_cumprod, pred=None, scaling_factor=0.07843137255, name=None, dtype=None,
                   out_dir="debug_visualizations"):
    """utils.py:104-211 `visualize_step`: the trainer's debug grid — rows: original / noisy / noise / predicted v / denoised, one column per
    frame of the window — written to <out_dir>/<name> (default sequence_step_<step>.png).  Same signature as the reference (`dtype` is
    accepted for call compatibility).  The three image rows are VAE decodes (HIP kernels: vae.decode((lat / s)) -> (x + 1) / 2 -> clamp),
    the denoised latents x_start = (x_noisy - sqrt(1 - a_t) v) / sqrt(a_t) when `pred` is not given (two axpy launches); figure assembly is
    host work (matplotlib, Agg).  Returns (path, {"orig", "noisy", "denoised"}: float (B, t, 3, H, W) CPU tensors in [0, 1])."""
    import os

    from . import lib as _lib
    from .generate import vae_decode_frames
    dev = vae.device
    L = _lib.load()

    def decode(lat):
        img = vae_decode_frames(lat.to(dev, torch.float32).contiguous(), vae, scaling_factor, to_uint8=False)    # (B, t, 3, H, W), (decode + 1) / 2
        with torch.cuda.device(dev):
            _lib.check(L.gtav_clamp_frames(img.data_ptr(), 1, 1, 0, img.numel(), 0.0, 1.0, _lib.current_stream()))
        return img.cpu()
    orig, noisy = decode(x_curr), decode(x_noisy)
    if pred is None:
        a_t = float(alphas_cumprod.reshape(-1)[int(step)])
        xs = x_noisy.to(dev, torch.float32).clone().contiguous()
        vd = v.to(dev, torch.float32).contiguous()
        with torch.cuda.device(dev):
            _lib.check(L.gtav_axpy_f32(xs.data_ptr(), vd.data_ptr(), -(1.0 - a_t) ** 0.5, xs.numel(), _lib.current_stream()))
            _lib.check(L.gtav_axpy_f32(xs.data_ptr(), xs.data_ptr(), 1.0 / a_t ** 0.5 - 1.0, xs.numel(), _lib.current_stream()))
        den = decode(xs)
    else:
        den = decode(pred)
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"sequence_step_{step}.png" if name is None else name)
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    T = x_curr.shape[1]
    fig, axes = plt.subplots(5, T, figsize=(5 * T, 25), squeeze=False)
    col = lambda z, t: torch.cat(list(z[:, t]), dim=1)                     # the batch stacked vertically (make_grid(nrow=1) without padding)
    for t in range(T):
        for r, (imgs, lat, title) in enumerate(((orig, x_curr, "Original"), (noisy, x_noisy, "Noisy"))):
            axes[r, t].imshow(col(imgs, t).permute(1, 2, 0).numpy())
            axes[r, t].set_title(f"{title} Frame {t}\nRange: [{float(lat[0, t].min()):.3f}, {float(lat[0, t].max()):.3f}]")
        for r, (z, title) in ((2, (noise, "Noise")), (3, (v, "Predicted Noise"))):
            g = col(z.float().cpu(), t).mean(0)
            im = axes[r, t].imshow(g.numpy(), cmap="RdBu", interpolation="nearest")
            plt.colorbar(im, ax=axes[r, t])
            axes[r, t].set_title(f"{title} Frame {t}\nRange: [{float(g.min()):.3f}, {float(g.max()):.3f}]")
        axes[4, t].imshow(col(den, t).permute(1, 2, 0).numpy())
        axes[4, t].set_title(f"Denoised Frame {t}\nRange: [{float(den[0, t].min()):.3f}, {float(den[0, t].max()):.3f}]")
        for r in range(5):
            axes[r, t].axis("off")
    plt.suptitle(f"Step {step}", y=1.02, fontsize=16)
    plt.tight_layout()
    plt.savefig(path, bbox_inches="tight")
    plt.close(fig)
    return path, {"orig": orig, "noisy": noisy, "denoised": den}
