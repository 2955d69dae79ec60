"""Times the DiT optimisation step (forward + loss, backward, AdamW) at batch 16 on latents through the EXPERIMENTS library, so that the GTAV_* overrides apply
(e.g. GTAV_DW_TN=0: transposed operand copies in front of the grouped weight-gradient launch; GTAV_DW_GROUPED=0).  One configuration per process: run it
twice in one gpurun call for an A/B on one box.   Usage (GPU box): GTAV_DW_TN=0 python tools/train_step_time.py [--steps 8] [--batch 16]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gtav_amd import lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    L.load_experiments()
    import gtav_amd.weights as W
    from gtav_amd.model.dit import DiT_models
    from gtav_amd.train import training_step
    dev = torch.device("cuda", 0)
    B = a.batch
    dit = DiT_models["DiT-S/2"](init_weights=False, max_batch=B, trainable=True)
    dit.load_state_dict(W.synth_state_dict(W.dit_param_shapes(depth=16), seed=0))
    g = torch.Generator().manual_seed(7)
    lat = (torch.randn(B, 5, 16, 18, 32, generator=g) * 0.5).to(dev)
    actions = torch.zeros(B, 5, 25, device=dev)
    actions[:, :, 3] = 1
    tgt = torch.randint(1, 51, (B,), generator=g)
    ctx = torch.randint(1, 41, (B,), generator=g)
    ctx_noise = torch.randn(B, 4, 16, 18, 32, generator=g).to(dev)
    noise = torch.randn(B, 1, 16, 18, 32, generator=g).to(dev)

    def step():
        return training_step(dit, lat, actions, tgt, ctx, ctx_noise, noise, lr=1e-5, weight_decay=0.01, max_grad_norm=1.0, world_size=1)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    applied, skipped, gnorm = dit.train_stats()
    print(json.dumps({"batch": B, "ms_per_step": round(ms, 3), "loss": float(loss), "grad_norm": gnorm, "skipped": skipped,
                      "env": {k: v for k, v in os.environ.items() if k.startswith("GTAV_")}}))


if __name__ == "__main__":
    main()
