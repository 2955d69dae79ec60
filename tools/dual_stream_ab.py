"""Does splitting a batch-B forward into S independent sub-batches on S HIP streams beat one launch chain?  (docs/LABNOTES.md 4.8)

At large M the 512 resident blocks of a GEMM run in lockstep: everybody streams operands, then everybody computes, then everybody
stores — HBM idles during the main loops and the matrix pipes idle during the prologues / epilogues and the LayerNorm launches.
Sequences of a batch never interact, so the batch can be cut into S sub-batches whose launch chains run on S
streams: the hardware then co-schedules blocks of DIFFERENT kernels on a CU, one chain's memory-bound phases beside another's
MFMA-bound ones.  One process, one GPU, alternating variants.
Usage (GPU box): python tools/dual_stream_ab.py [--batch 8] [--splits 1 2 4] [--fold 1]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gtav_amd import lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--splits", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--fold", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--share-weights", action="store_true", help="(not implemented in the ABI yet: every sub-batch handle owns a copy)")
    a = ap.parse_args()
    L.load_experiments()   # set_fold exists in the experiments build only
    import gtav_amd.weights as W
    from gtav_amd.model.dit import DiT_models
    dev = torch.device("cuda", 0)
    B = a.batch
    sd = W.synth_state_dict(W.dit_param_shapes(depth=16), seed=0)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 5, 16, 18, 32, generator=g).to(dev)
    t = torch.tensor([[15, 15, 15, 15, 500]] * B)
    act = torch.zeros(B, 5, 25, device=dev)
    act[:, :, 3] = 1
    variants = {}
    for S in a.splits:
        assert B % S == 0
        hs = []
        for _ in range(S):
            m = DiT_models["DiT-S/2"](init_weights=False, max_batch=B // S)
            m.load_state_dict(sd)
            m.set_fold(a.fold, 1024, 1024)
            hs.append(m)
        variants[S] = (hs, [torch.cuda.Stream(device=dev) for _ in range(S)])
    ref = None
    for r in range(a.rounds):
        for S, (hs, streams) in variants.items():
            b = B // S

            def run():
                outs = []
                cur = torch.cuda.current_stream(dev)
                for k in range(S):
                    streams[k].wait_stream(cur)
                    with torch.cuda.stream(streams[k]):
                        outs.append(hs[k](x[k * b:(k + 1) * b], t[k * b:(k + 1) * b], act[k * b:(k + 1) * b]))
                for k in range(S):
                    cur.wait_stream(streams[k])
                return outs
            for _ in range(3):
                out = torch.cat(run())
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            err = ((out - ref).norm() / ref.norm()).item()
            t0 = time.perf_counter()
            n = 20
            for _ in range(n):
                run()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            print(f"round {r} batch {B} as {S} x {b} on {S} stream(s), fold {a.fold}: {ms:.3f} ms per batch-{B} forward (rel diff vs first {err:.1e})", flush=True)


if __name__ == "__main__":
    main()
