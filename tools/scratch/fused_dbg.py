import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import gtav_amd.weights as W
from gtav_amd.model.dit import DiT
from gtav_amd import lib as L
if len(sys.argv) > 1 and sys.argv[1] == "exp":
    L.load_experiments()
depth = 16
dit = DiT(depth=depth, init_weights=False, max_batch=1)
dit.load_state_dict(W.synth_state_dict(W.dit_param_shapes(depth=depth), seed=0))
g = torch.Generator().manual_seed(3)
x = torch.randn(1, 5, 16, 18, 32, generator=g).cuda()
t = torch.tensor([[15, 15, 15, 15, 500]])
dit(x, t, None)
for act in (None, "a"):
    a = None
    if act:
        a = torch.zeros(1, 5, 25, device="cuda"); a[:, :, 3] = 1
    outs = []
    for f in (False, True, False, True):
        dit.set_fused_temporal(f)
        dit.profile(True)
        o = dit(x, t, a).clone()
        pr = dit.profile_read()
        dit.profile(False)
        outs.append(o)
        print("fused", f, "attn_temporal launches", pr["attn_temporal"][1], "qkv", pr["gemm_qkv"][1])
    tag = "exp" if len(sys.argv) > 1 else "prod"
    torch.save({"split": outs[0].cpu(), "fused": outs[1].cpu()}, os.path.join(ROOT, "gpurun_out", f"fused_dbg_{tag}_{act}.pt"))
    print("act", act, "split==split", torch.equal(outs[0], outs[2]), "fused==fused", torch.equal(outs[1], outs[3]),
          "split==fused", torch.equal(outs[0], outs[1]), "rel", ((outs[0] - outs[1]).norm() / outs[0].norm()).item())
