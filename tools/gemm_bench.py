"""GPU micro-benchmark of the fp16 MFMA GEMM (through the C-ABI) on the shapes of the DiT at several M.
Weights rotate over `--copies` distinct buffers so that a launch does not find its W panel in the caches.
Usage (GPU box): python tools/gemm_bench.py [--stages 0|2|4]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gtav_amd import lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stages", type=int, nargs="+", default=[2, 4])
    ap.add_argument("--copies", type=int, default=24)
    ap.add_argument("--iters", type=int, default=96)
    ap.add_argument("--ms", type=int, nargs="+", default=[144, 720, 1152, 5760, 11520])
    ap.add_argument("--debug", type=int, nargs="+", default=[0], help="gemm debug bits to sweep (1 = no fills, 2 = no MFMA)")
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--splitk", type=int, default=0, help="split-K factor for the partial-slab GEMMs (0 = heuristic)")
    ap.add_argument("--wm", type=int, nargs="+", default=[0], help="block shapes to sweep: 0 heuristic, 2 = 128x128/4 waves, 4 = 128x256/8 waves")
    a = ap.parse_args()
    lib = L.load_experiments()   # libgtav_amd_exp.so: the debug bits below exist only in that build
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    shapes = [("qkv", 3072, 1024, 5), ("qkvt", 3072, 1024, 7), ("out", 1024, 1024, 6), ("fc1", 4096, 1024, 2), ("fc2", 1024, 4096, 6)]   # epi 7 here = temporal QKV
    if "qkvs" in a.only:    # the spatial half's two forms back to back: to_qkv + attention launches (58) and the fused launch (8), frames of 144 tokens
        shapes += [("qkv+attn_s", 3072, 1024, 58), ("qkvs", 3072, 1024, 8)]
    if "train" in a.only:   # GEMMs of the training step without an activation: fc1 forward / fc2 dX (fp16 tile-major out, code 70 = EPI_F16_TILED), fc1 dX / to_qkv dX (fp32 out)
        shapes += [("train_f16t", 4096, 1024, 70), ("train_f32_k4096", 1024, 4096, 0), ("train_f32_k3072", 1024, 3072, 0)]
    if "vae" in a.only:     # the ViT-VAE's GEMMs (model/vae.py:115-157): erf-GELU fc1, un-gated in-place residual proj / fc2 (code 40 = EPI_RESID without a gate), QKV at S = 576
        shapes += [("vae_qkv", 3072, 1024, 55), ("vae_fc1", 4096, 1024, 3), ("vae_proj", 1024, 1024, 40), ("vae_fc2", 1024, 4096, 40)]
    if "floor" in a.only:   # one K-step only: launch + first tile + epilogue (the per-launch floor of each epilogue)
        shapes += [("floor_gelu", 4096, 64, 2), ("floor_f32", 4096, 64, 0), ("floor_part", 1024, 64, 6), ("floor_n128", 128, 64, 0)]
    print(f"{'shape':>5} {'M':>6} {'N':>5} {'K':>5} {'ns':>3} {'split':>5} {'us':>9} {'TFLOP/s':>9}")
    for M in a.ms:
        for name, N, K, epi in shapes:
            if a.only and name not in a.only.split(","):
                continue
            # operands are tile-major; for timing any data of the padded size will do
            x = (torch.randn((M + 127) // 128 * 128, K, device=dev) * 0.5).half()
            ws = [(torch.randn((N + 127) // 128 * 128, K, device=dev) * 0.03).half() for _ in range(a.copies)]
            bias = torch.randn(N, device=dev)
            sk = (a.splitk or lib.gtav_op_gemm_choose_splitk(M, N, K)) if epi == 6 else 1
            # fp32 row-major for epilogues 0 / 6, fp16 (tile-major, rows padded to 128) otherwise; sized for the padded M
            Mp = (M + 127) // 128 * 128
            out = torch.zeros((max(sk, 1) * Mp, N), device=dev, dtype=torch.float32 if epi in (0, 6, 40) else torch.float16)
            vq = torch.empty(3, Mp, 1024, device=dev, dtype=torch.float16) if epi == 55 else None
            vcs = torch.ones(576, 64, device=dev) if epi == 55 else None
            q = torch.empty(3, M, 1024, device=dev, dtype=torch.float16)
            cs = torch.ones(144, 64, device=dev)
            sn = torch.zeros(144, 64, device=dev)
            kvc = torch.empty(max(1, M // 720) * 5 * 144 * 2 * 1024, device=dev, dtype=torch.float16)
            for ns, dbg, wm in [(n_, d_, w_) for w_ in a.wm for n_ in a.stages for d_ in a.debug]:
                lib.gtav_op_gemm_set_stages(ns)
                lib.gtav_op_gemm_set_debug(dbg)
                lib.gtav_op_gemm_set_wm(wm)

                def run(i):
                    w = ws[i % a.copies]
                    if epi == 5:
                        Mq = (M // 144) * 144
                        L.check(lib.gtav_op_gemm_qkv(x.data_ptr(), K, w.data_ptr(), 0, Mq, 1024, 0, q[0].data_ptr(), q[1].data_ptr(),
                                                     q[2].data_ptr(), 144, 0, 0, 0, cs.data_ptr(), st))
                    elif epi in (58, 8):
                        Mq = (M // 144) * 144
                        if epi == 58:
                            L.check(lib.gtav_op_gemm_qkv(x.data_ptr(), K, w.data_ptr(), 0, Mq, 1024, 0, q[0].data_ptr(), q[1].data_ptr(),
                                                         q[2].data_ptr(), 144, 0, 0, 0, cs.data_ptr(), st))
                            L.check(lib.gtav_op_attn_spatial(q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), out.data_ptr(), Mq // 144, 16, 144, st))
                        else:
                            L.check(lib.gtav_op_gemm_qkvs_attn(x.data_ptr(), w.data_ptr(), Mq, 1024, 144, cs.data_ptr(), out.data_ptr(), st))
                    elif epi == 55:     # VAE: S = 576 tokens per attention item
                        Mq = (M // 576) * 576
                        L.check(lib.gtav_op_gemm_qkv(x.data_ptr(), K, w.data_ptr(), bias.data_ptr(), Mq, 1024, 0, vq[0].data_ptr(), vq[1].data_ptr(),
                                                     vq[2].data_ptr(), 576, 0, 0, 0, vcs.data_ptr(), st))
                    elif epi == 40:     # in-place residual without a gate
                        L.check(lib.gtav_op_gemm_f16(x.data_ptr(), K, w.data_ptr(), bias.data_ptr(), out.data_ptr(), N, M, N, K, 4, 0, 0, 0, st))
                    elif epi == 7:      # temporal layout: q [M][D], k/v into a [B][Tmax][P][2][D] cache (B = 1 here, Tq = M / 144 frames)
                        Tq = M // 144
                        Bq = max(1, Tq // 5)
                        Tq = min(Tq, 5)
                        L.check(lib.gtav_op_gemm_qkv(x.data_ptr(), K, w.data_ptr(), 0, Bq * Tq * 144, 1024, 1, q[0].data_ptr(), kvc.data_ptr(),
                                                     kvc.data_ptr(), 144, Tq, 0, 5, cs.data_ptr(), st))
                    elif epi == 6:
                        g = lib.gtav_op_gemm_splitk_ln  # noqa: F841  (partial GEMM only: use the raw op below)
                        L.check(lib.gtav_op_gemm_f16(x.data_ptr(), K, w.data_ptr(), 0, out.data_ptr(), N, M, N, K, 6, 0, sk, 1, st))
                    else:
                        L.check(lib.gtav_op_gemm_f16(x.data_ptr(), K, w.data_ptr(), bias.data_ptr(), out.data_ptr(), N, M, N, K, 7 if epi == 70 else epi, 0, 0,
                                                     1, st))
                try:
                    for i in range(8):
                        run(i)
                except L.GtavError as e:          # e.g. the 96-feature tile cannot run the spatial QKV epilogue
                    print(f"{name:>5} {M:6d} {N:5d} {K:5d}   skipped (wm={wm}): {e}")
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for i in range(a.iters):
                    run(i)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / a.iters
                print(f"{name:>5} {M:6d} {N:5d} {K:5d} {ns:3d} {sk:5d} {us:9.2f} {2.0 * M * N * K / us / 1e6:9.1f}  dbg={dbg} wm={wm}")
    lib.gtav_op_gemm_set_stages(0)
    lib.gtav_op_gemm_set_debug(0)
    lib.gtav_op_gemm_set_wm(0)


if __name__ == "__main__":
    main()
