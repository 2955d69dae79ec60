import math, sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import dev, pad_weight_f16, stream, to_tiled_f16, gemm
from gtav_amd import lib as L
lib = L.load()
def _rand(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed); return torch.randn(*shape, generator=g) * scale
NB, S, D = 40, 144, 1024
heads, M = D // 64, NB * S
x = _rand(M, D, seed=1).half(); w = _rand(3 * D, D, scale=1 / math.sqrt(D), seed=2); bias = _rand(3 * D, seed=7)
w16 = pad_weight_f16(w); xd = to_tiled_f16(x); bd = bias.to(dev())
def run_qkv(wm, use_bias, ident):
    ang = (_rand(S, 32, seed=3) * (0 if ident else 3)).repeat_interleave(2, dim=-1)
    cd, sd_ = ang.cos().to(dev()).contiguous(), ang.sin().to(dev()).contiguous()
    cs = torch.empty_like(cd)
    L.check(lib.gtav_op_rope_interleave(cd.data_ptr(), sd_.data_ptr(), cs.data_ptr(), S, stream()))
    q = torch.zeros(NB, heads, S, 64, device=dev(), dtype=torch.float16); k = torch.zeros_like(q)
    vt = torch.zeros(NB, heads, 64, S, device=dev(), dtype=torch.float16)
    lib.gtav_op_gemm_set_wm(wm)
    L.check(lib.gtav_op_gemm_qkv(xd.data_ptr(), D, w16.data_ptr(), bd.data_ptr() if use_bias else 0, M, D, 0, q.data_ptr(), k.data_ptr(), vt.data_ptr(), S, 0, 0, 0, cs.data_ptr(), stream()))
    lib.gtav_op_gemm_set_wm(0)
    torch.cuda.synchronize()
    return q, k, vt
for use_bias, ident in ((True, False), (False, False), (True, True), (False, True)):
    ref = run_qkv(12, use_bias, ident)
    nbad = [0, 0, 0]; pats = []
    for it in range(12):
        got = run_qkv(16, use_bias, ident)
        for j in range(3):
            neq = (got[j] != ref[j])
            n = int(neq.sum())
            nbad[j] += n
            if n and len(pats) < 6:
                idx = neq.nonzero()
                pats.append((["q", "k", "vt"][j], n, idx[0].tolist(), idx[-1].tolist()))
    print(f"bias={use_bias} ident_rope={ident}: elements differing from shape 12 over 12 runs: q {nbad[0]} k {nbad[1]} vt {nbad[2]}", pats)
# GELU / RESID repeatability + vs shape 12
for (Mm, N, K) in ((5760, 4096, 1024), (5760, 1024, 4096)):
    xx = _rand(Mm, K, seed=1).half(); ww = _rand(N, K, scale=1 / math.sqrt(K), seed=2)
    w16b, xdb, bdb = pad_weight_f16(ww), to_tiled_f16(xx), _rand(N, seed=3).to(dev())
    def run(wm, epi):
        lib.gtav_op_gemm_set_wm(wm)
        if epi == 2:
            out = torch.zeros(((Mm + 127) // 128 * 128, N), device=dev(), dtype=torch.float16)
            gemm(xdb, w16b, bdb, Mm, N, K, 2, out, N)
        else:
            out = torch.ones(Mm, N, device=dev())
            L.check(lib.gtav_op_gemm_f16(xdb.data_ptr(), K, w16b.data_ptr(), bdb.data_ptr(), out.data_ptr(), N, Mm, N, K, 4, 0, 0, 1, stream()))
        lib.gtav_op_gemm_set_wm(0); torch.cuda.synchronize(); return out
    for epi in (2, 4):
        ref = run(12 if epi == 2 else 2, epi)
        bad = 0; mx = 0.0
        for it in range(15):
            o = run(16, epi)
            d = (o.float() - ref.float()).abs()
            bad += int((d > 1e-2).sum()); mx = max(mx, d.max().item())
        print(f"M={Mm} N={N} K={K} epi={epi}: elements off by > 1e-2 vs one-shot kernel over 15 runs: {bad}, max diff {mx:.4g}")
