"""Race screen of the long-sequence attention kernel: N launches on the same inputs must agree bit for bit, and agree with fp32 math.
   python tools/attn_repeat_check.py [--exp] [--s 576] [--jump 4.0] [--reps 50]   (GTAV_ATTN_FLASH_* knobs with --exp)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gtav_amd import lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--exp", action="store_true")
    ap.add_argument("--s", type=int, default=576)
    ap.add_argument("--nb", type=int, default=2)
    ap.add_argument("--heads", type=int, default=3)
    ap.add_argument("--jump", type=float, default=4.0)
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    lib = L.load_experiments() if a.exp else L.load()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    NB, H, S = a.nb, a.heads, a.s
    g = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(NB, H, S, 64, generator=g).half() for _ in range(3))
    if a.jump != 0:
        k[NB - 1, H - 1, max(S - 9, 0)] = q[NB - 1, H - 1, min(7, S - 1)] * a.jump
        k[0, min(1, H - 1), min(70, S - 1)] = q[0, min(1, H - 1), min(150, S - 1)] * a.jump
    vt = v.transpose(-1, -2).contiguous()
    rows = (NB * S + 127) // 128 * 128
    qd, kd, vd = q.to(dev), k.to(dev), vt.to(dev)
    outs = []
    for _ in range(a.reps):
        o = torch.zeros(rows, H * 64, device=dev, dtype=torch.float16)
        L.check(lib.gtav_op_attn_spatial(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), o.data_ptr(), NB, H, S, st))
        outs.append(o)
    torch.cuda.synchronize()
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float()).permute(0, 2, 1, 3).reshape(NB * S, H * 64)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import untile
    bad = 0
    for i, o in enumerate(outs):
        got = untile(o, NB * S, H * 64).float().cpu()
        err = ((got - ref).norm() / ref.norm()).item()
        same = torch.equal(o, outs[0])
        if not same or not err < 1.5e-3:
            bad += 1
            d = (got - untile(outs[0], NB * S, H * 64).float().cpu()).abs()
            rows_bad = torch.nonzero(d.amax(1) > 0).flatten().tolist()
            cols_bad = torch.nonzero(d.amax(0) > 0).flatten().tolist()
            print(f"rep {i}: equal to rep 0: {same}, rel-L2 vs fp32 {err:.3e}, differing rows {rows_bad[:12]}{'...' if len(rows_bad) > 12 else ''} ({len(rows_bad)}), "
                  f"cols {cols_bad[:4]}..{cols_bad[-1:] } ({len(cols_bad)}), max |diff| {d.max().item():.3e}")
    got0 = untile(outs[0], NB * S, H * 64).float().cpu()
    print(f"S={S} NB={NB} heads={H} jump={a.jump}: {bad} of {a.reps} launches differ / fail; rep 0 rel-L2 vs fp32 {((got0 - ref).norm() / ref.norm()).item():.3e}")


if __name__ == "__main__":
    main()
