import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gtav_amd.lib as L
if len(sys.argv) > 1: L.load_experiments()
from helpers import dev, pad_weight_f16, stream, to_tiled_f16, untile
def _rand(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed); return torch.randn(*shape, generator=g) * scale
B, P, D = 1, 144, 1024
Tq, t0, Tmax = 5, 0, 5
M = B * Tq * P
lib = L.load()
x = _rand(M, D, seed=1).half()
w = _rand(3 * D, D, scale=1 / math.sqrt(D), seed=2)
ang = (_rand(Tmax, 32, seed=3) * 3).repeat_interleave(2, dim=-1)
cd, sd_ = ang.cos().to(dev()).contiguous(), ang.sin().to(dev()).contiguous()
cs = torch.empty_like(cd)
L.check(lib.gtav_op_rope_interleave(cd.data_ptr(), sd_.data_ptr(), cs.data_ptr(), Tmax, stream()))
w16 = pad_weight_f16(w)
Mp = (M + 127) // 128 * 128
q = torch.zeros(M, D, device=dev(), dtype=torch.float16)
kv = torch.zeros(B, Tmax, P, 2, D, device=dev(), dtype=torch.float16)
o = torch.zeros(Mp, D, device=dev(), dtype=torch.float16)
L.check(lib.gtav_op_gemm_qkv(to_tiled_f16(x).data_ptr(), D, w16.data_ptr(), 0, M, D, 1, q.data_ptr(), kv.data_ptr(), kv.data_ptr(), P, Tq, t0, Tmax, cs.data_ptr(), stream()))
L.check(lib.gtav_op_attn_temporal(q.data_ptr(), kv.data_ptr(), o.data_ptr(), B, P, D, Tq, t0, Tmax, stream()))
xp = x.reshape(B, Tq, P // 16, 16, D).permute(0, 2, 1, 3, 4).reshape(M, D)
w_hm = torch.empty_like(w16)
L.check(lib.gtav_op_qkv_head_major(w16.data_ptr(), w_hm.data_ptr(), D, stream()))
kv2 = torch.zeros_like(kv); o2 = torch.zeros_like(o)
L.check(lib.gtav_op_gemm_qkvt_attn(to_tiled_f16(xp).data_ptr(), w_hm.data_ptr(), M, D, P, Tq, t0, Tmax, cs.data_ptr(), kv2.data_ptr(), o2.data_ptr(), stream()))
torch.cuda.synchronize()
for name, i in (("k", 0), ("v", 1)):
    a, b = kv[:, :, :, i].float().cpu(), kv2[:, :, :, i].float().cpu()   # B T P D
    ne = (a != b)
    print(name, "mismatches", ne.sum().item(), "of", ne.numel(), "max abs diff", (a - b).abs().max().item(), "rel", ((a - b).norm() / a.norm()).item())
    if ne.any():
        idx = ne.nonzero()
        print("  frames", idx[:, 1].unique().tolist(), "positions", idx[:, 2].unique().tolist()[:20], "features (mod 64)", (idx[:, 3] % 64).unique().tolist()[:70])
        print("  heads", (idx[:, 3] // 64).unique().tolist())
        print("  sample", [(a[tuple(j)].item(), b[tuple(j)].item()) for j in idx[:6]])
oa, ob = untile(o, M, D).float(), untile(o2, M, D).float()
print("o mismatches", (oa != ob).sum().item(), "rel", ((oa - ob).norm() / oa.norm()).item())
