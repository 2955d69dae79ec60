#!/usr/bin/env python3
"""CPU emulation of the LayerNorm fold (docs/LABNOTES.md 4.7) on the full-size DiT: which fp16 operand policy keeps the forward
within 1e-3 relative L2 of the fp32 oracle?

  base   : today's kernels — xn = fp16(LN(x) (1 + s) + sh) is the GEMM operand
  fold0  : A = fp16(x (1 + s)), y = (A W^T - mu c1) rstd + c2            (un-centred operand)
  foldc  : A = fp16((x - mu~) (1 + s)), mu~ = row mean of the residual BEFORE the branch update (what the producer
           epilogue knows), y = (A W^T - (mu - mu~) c1) rstd + c2
c1 = sum_k (1 + s_k) W16[n, k], c2 = sum_k sh_k W16[n, k] + b[n] with the fp16-rounded weights the GEMM itself uses.
Everything else (fp16 weights, fp16 q/k/v, attention output and MLP hidden activations) is rounded the same way in all
three policies.  Runs on the CPU only (tools/, not part of the product)."""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gtav_amd  # noqa: E402,F401
import gtav_amd.weights as W  # noqa: E402
from oracle import ref_cpu as O  # noqa: E402


def h(x):
    return x.half().float()


class Emu:
    def __init__(self, sd, cfg, policy):
        self.sd, self.cfg, self.policy = sd, cfg, policy
        self.w16 = {k: h(v) for k, v in sd.items() if k.endswith("weight") and v.dim() == 2 and "adaLN" not in k and "t_embedder" not in k
                    and "external" not in k}
        self.mu_prev = None

    def ln_gemm(self, x, shift, scale, wname, bias):
        """Linear(modulate(LN(x), shift, scale)) under the policy.  x (B,T,H,W,D); shift/scale (B,T,D)."""
        w = self.w16[wname]
        sc = (scale + 1e-6)[:, :, None, None, :]
        sh = shift[:, :, None, None, :]
        mu = x.mean(-1, keepdim=True)
        var = x.var(-1, unbiased=False, keepdim=True)
        rstd = torch.rsqrt(var + 1e-6)
        if self.policy == "base":
            xn = h((x - mu) * rstd * (1 + sc) + sh)
            return F.linear(xn, w, bias)
        mut = torch.zeros_like(mu) if self.policy == "fold0" or self.mu_prev is None else self.mu_prev
        A = h((x - mut) * (1 + sc))
        acc = F.linear(A, w)
        c1 = F.linear((1 + sc), w)
        c2 = F.linear(sh, w, bias)
        return (acc - (mu - mut) * c1) * rstd + c2

    def block(self, i, x, c):
        sd, cfg = self.sd, self.cfg
        p = f"blocks.{i}."
        scv = F.silu(c)
        B, T, H, Wd, D = x.shape
        heads, d = cfg.num_heads, D // cfg.num_heads
        for half in ("s", "t"):
            m = F.linear(scv, sd[p + f"{half}_adaLN_modulation.1.weight"], sd[p + f"{half}_adaLN_modulation.1.bias"])
            shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = m.chunk(6, dim=-1)
            qkv = self.ln_gemm(x, shift_msa, scale_msa, p + f"{half}_attn.to_qkv.weight", None)
            self.mu_prev = x.mean(-1, keepdim=True)      # the producer (out-proj) knows the stats of the residual it updates
            q, k, v = qkv.chunk(3, dim=-1)
            if half == "s":
                sp = lambda z: z.reshape(B * T, H, Wd, heads, d).permute(0, 3, 1, 2, 4)
                q, k, v = sp(q), sp(k), sp(v)
                q, k = O.apply_rope(self.s_angles, q), O.apply_rope(self.s_angles, k)
                q, k, v = (h(z.reshape(B * T, heads, H * Wd, d)) for z in (q, k, v))
                o = F.scaled_dot_product_attention(q, k, v)
                o = o.reshape(B, T, heads, H, Wd, d).permute(0, 1, 3, 4, 2, 5).reshape(B, T, H, Wd, D)
            else:
                sp = lambda z: z.reshape(B, T, H, Wd, heads, d).permute(0, 2, 3, 4, 1, 5).reshape(B * H * Wd, heads, T, d)
                q, k, v = sp(q), sp(k), sp(v)
                ang = O.rope_angles_temporal(T, self.t_freqs)
                q, k = O.apply_rope(ang, q), O.apply_rope(ang, k)
                q, k, v = h(q), h(k), h(v)
                o = F.scaled_dot_product_attention(q, k, v, is_causal=True)
                o = o.reshape(B, H, Wd, heads, T, d).permute(0, 4, 1, 2, 3, 5).reshape(B, T, H, Wd, D)
            a = F.linear(h(o), self.w16[p + f"{half}_attn.to_out.weight"], sd[p + f"{half}_attn.to_out.bias"])
            x = x + O.gate(a, gate_msa)
            u = self.ln_gemm(x, shift_mlp, scale_mlp, p + f"{half}_mlp.fc1.weight", sd[p + f"{half}_mlp.fc1.bias"])
            self.mu_prev = x.mean(-1, keepdim=True)
            hh = h(F.gelu(u, approximate="tanh"))
            y = F.linear(hh, self.w16[p + f"{half}_mlp.fc2.weight"], sd[p + f"{half}_mlp.fc2.bias"])
            x = x + O.gate(y, gate_mlp)
        return x

    def forward(self, x, t, a):
        sd, cfg = self.sd, self.cfg
        B, T, C, H, Wd = x.shape
        hd = cfg.head_dim
        self.s_angles = O.rope_angles_axial(*cfg.grid, O.rope_freqs_pixel(hd // 2, 256))
        self.t_freqs = O.rope_freqs_lang(hd)
        gh, gw = cfg.grid
        hcur = O.patch_embed(h(x.reshape(B * T, C, H, Wd)), h(sd["x_embedder.proj.weight"]), sd["x_embedder.proj.bias"], cfg.patch_size)
        hcur = hcur.reshape(B, T, gh, gw, -1)
        c = O.dit_cond(sd, cfg, t, a)
        self.mu_prev = None
        stats = []
        for i in range(cfg.depth):
            hcur = self.block(i, hcur, c)
            mu, sdv = hcur.mean(-1), hcur.std(-1)
            stats.append((float((mu.abs() / sdv).mean()), float((mu.abs() / sdv).max())))
        m = F.linear(F.silu(c), sd["final_layer.adaLN_modulation.1.weight"], sd["final_layer.adaLN_modulation.1.bias"])
        shift, scale = m.chunk(2, dim=-1)
        out = self.ln_gemm(hcur, shift, scale, "final_layer.linear.weight", sd["final_layer.linear.bias"])
        out = O.dit_unpatchify(out.reshape(B * T, gh, gw, -1), cfg)
        return out.reshape(B, T, C, H, Wd), stats


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    depth = int(os.environ.get("DEPTH", "16"))
    cfg = O.dit_s_2()
    cfg.depth = depth
    sd = W.synth_state_dict(W.dit_param_shapes(depth=depth), seed=0)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 5, 16, 18, 32, generator=g)
    t = torch.tensor([[15, 15, 15, 15, 500]])
    a = torch.zeros(1, 5, 25)
    a[:, :, 3] = 1
    with torch.no_grad():
        t0 = time.time()
        ref = O.dit_forward(sd, cfg, x, t, a)
        print(f"oracle forward {time.time() - t0:.1f} s")
        for pol in ("base", "fold0", "foldc"):
            out, stats = Emu(sd, cfg, pol).forward(x, t, a)
            err = ((out - ref).norm() / ref.norm()).item()
            print(f"{pol:6s} rel-L2 vs fp32 oracle {err:.3e}   |mean|/std of the residual rows: block0 {stats[0][0]:.3f} (max {stats[0][1]:.3f}) "
                  f"last {stats[-1][0]:.3f} (max {stats[-1][1]:.3f})")


if __name__ == "__main__":
    main()
