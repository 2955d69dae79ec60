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
