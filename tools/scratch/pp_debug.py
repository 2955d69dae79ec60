import math, sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import dev, pad_weight_f16, stream, to_tiled_f16
from gtav_amd import lib as L
lib = L.load()
def _rand(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed); return torch.randn(*shape, generator=g) * scale
def rope_ref(x, cos, sin):
    x2 = x.reshape(*x.shape[:-1], 32, 2); rot = torch.stack((-x2[..., 1], x2[..., 0]), dim=-1).reshape(x.shape); return x * cos + rot * sin
NB, S, D = 40, 144, 1024
heads, M = D // 64, NB * S
x = _rand(M, D, seed=1).half(); w = _rand(3 * D, D, scale=1 / math.sqrt(D), seed=2); bias = _rand(3 * D, seed=7)
ang = (_rand(S, 32, seed=3) * 3).repeat_interleave(2, dim=-1); cos, sin = ang.cos(), ang.sin()
w16 = pad_weight_f16(w)
xd, bd, cd, sd_ = to_tiled_f16(x), bias.to(dev()), cos.to(dev()).contiguous(), sin.to(dev()).contiguous()
cs = torch.empty_like(cd)
L.check(lib.gtav_op_rope_interleave(cd.data_ptr(), sd_.data_ptr(), cs.data_ptr(), S, stream()))
y = (x.float() @ w.half().float().t() + bias).reshape(NB, S, 3, heads, 64)
qr = rope_ref(y[:, :, 0].permute(0, 2, 1, 3), cos[None, None], sin[None, None])
kr = rope_ref(y[:, :, 1].permute(0, 2, 1, 3), cos[None, None], sin[None, None])
vr = y[:, :, 2].permute(0, 2, 3, 1)
tot = {}
for wm in [12] + [16] * 10:
    q = torch.zeros(NB, heads, S, 64, device=dev(), dtype=torch.float16); k = torch.zeros_like(q)
    vt = torch.zeros(NB, heads, 64, S, device=dev(), dtype=torch.float16)
    lib.gtav_op_gemm_set_wm(wm)
    L.check(lib.gtav_op_gemm_qkv(xd.data_ptr(), D, w16.data_ptr(), bd.data_ptr(), M, D, 0, q.data_ptr(), k.data_ptr(), vt.data_ptr(), S, 0, 0, 0, cs.data_ptr(), stream()))
    lib.gtav_op_gemm_set_wm(0)
    torch.cuda.synchronize()
    for name, got, ref in (("q", q, qr), ("k", k, kr), ("vt", vt, vr)):
        d = (got.float().cpu() - ref)
        rel = (d.norm() / ref.norm()).item()
        bad = (d.abs() > 0.02 + 0.01 * ref.abs())
        key = (wm, name)
        t = tot.setdefault(key, [0, 0.0, 0.0])
        t[0] += int(bad.sum()); t[1] = max(t[1], rel); t[2] = max(t[2], d.abs().max().item())
        if bad.any() and t[0] <= 64:
            idx = bad.nonzero()
            print(f"   wm={wm} {name}: bad idx sample {idx[:3].tolist()} ... d values {sorted(set(idx[:, 3].tolist()))[:10]}")
for k_, v in tot.items():
    print(k_, "bad total", v[0], "max rel %.3e" % v[1], "max abs %.4f" % v[2])
