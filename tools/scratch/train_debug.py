"""Per-parameter gradient comparison GPU vs oracle autograd (debugging aid for tests/test_gpu_train.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_gpu_train import _setup
from helpers import rel_l2
from oracle import ref_cpu as O
act = len(sys.argv) < 2 or sys.argv[1] != "noact"
m, sd, cfg, x, t, a, vt = _setup(actions=act)
loss_ref, v_ref, grads = O.dit_loss_and_grads(sd, cfg, x, t, a, vt)
v = m.forward_train(x, t, a)
print("forward rel", rel_l2(v, v_ref), "vs inference", rel_l2(m(x, t, a), v.cpu()))
m.zero_grad(); m.backward_(v, vt)
try:
    m.check()
except Exception as e:
    print("check:", e)
for k, gref in grads.items():
    g = m.grad(k).cpu()
    print(f"{k:55s} ref norm {float(gref.norm()):.3e} gpu norm {float(g.norm()):.3e} rel {rel_l2(g, gref):.3e}")
