import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import gtav_amd.lib as L
lib = L.load_experiments() if os.environ.get("EXP") else L.load()
NB, heads, S = 80, 16, 144
dev = "cuda"
q = (torch.randn(NB * heads, S, 64, device=dev)).half(); k = torch.randn_like(q); vt = torch.randn(NB * heads, 64, S, device=dev).half()
do = torch.randn(NB * S, heads * 64, device=dev).half()
cs = torch.ones(S, 64, device=dev)
out = torch.zeros((NB * S + 127) // 128 * 128, 3 * heads * 64, device=dev, dtype=torch.float16)
st = torch.cuda.current_stream().cuda_stream
def run():
    L.check(lib.gtav_op_attn_spatial_bwd(q.data_ptr(), k.data_ptr(), vt.data_ptr(), do.data_ptr(), NB, heads, S, cs.data_ptr(), out.data_ptr(), st))
for _ in range(3): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): run()
torch.cuda.synchronize()
print(os.environ.get("GTAV_ATTN_BWD_VALU"), os.environ.get("GTAV_ATTN_BWD_DBG"), "us per call", (time.perf_counter() - t0) / 10 * 1e6)
