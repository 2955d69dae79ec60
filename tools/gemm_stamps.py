"""Per-block timeline of the GEMM kernels (experiments build only): where does a launch spend its time?

Every block records s_memrealtime (100 MHz, chip-wide) at entry, when its first K-tile has landed (prologue), at the end of the
main loop and at the end of the epilogue, plus s_memtime (shader cycles, per-XCD counter) at entry / end and its XCC id (csrc/gemm.hip BlockStamps).  This tool
launches one GEMM per DiT shape, reads the stamps back and prints, per launch: the kernel's event time, the spread of block
start times, the per-phase durations (median / p90) and how the blocks' epilogues overlap other blocks' main loops.

Usage (GPU box): python tools/gemm_stamps.py [--ms 720 5760] [--only fc1,qkv] [--wm 0 7 12]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gtav_amd import lib as L  # noqa: E402


def pct(t, q):
    return float(torch.quantile(t.double(), q))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", type=int, nargs="+", default=[720, 5760])
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--wm", type=int, nargs="+", default=[0])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--debug", type=int, default=0, help="gemm debug bits (1 = no refills, 2 = no LDS reads / MFMAs: wrong results, timing only)")
    a = ap.parse_args()
    lib = L.load_experiments()
    lib.gtav_op_gemm_set_debug(a.debug)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    shapes = [("qkv", 3072, 1024, 5), ("qkvs", 3072, 1024, 8), ("out", 1024, 1024, 6), ("fc1", 4096, 1024, 2), ("fc2", 1024, 4096, 6)]   # qkvs: the fused spatial to_qkv + attention launch (frames of 144 tokens)
    maxb = 16384
    stamps = torch.zeros(maxb * 8, dtype=torch.int64, device=dev)
    for M in a.ms:
        for name, N, K, epi in shapes:
            if a.only and name not in a.only.split(","):
                continue
            x = (torch.randn((M + 127) // 128 * 128, K, device=dev) * 0.5).half()
            ws = [(torch.randn((N + 127) // 128 * 128, K, device=dev) * 0.03).half() for _ in range(4)]
            bias = torch.randn(N, device=dev)
            sk = lib.gtav_op_gemm_choose_splitk(M, N, K) if epi == 6 else 1
            Mp = (M + 127) // 128 * 128
            out = torch.empty((max(sk, 1) * Mp, N), device=dev, dtype=torch.float32 if epi in (0, 6) else torch.float16)
            q = torch.empty(3, M, 1024, device=dev, dtype=torch.float16)
            cs = torch.ones(144, 64, device=dev)
            for wm in a.wm:
                lib.gtav_op_gemm_set_wm(wm)

                def run(i):
                    w = ws[i % 4]
                    if epi == 5:
                        L.check(lib.gtav_op_gemm_qkv(x.data_ptr(), K, w.data_ptr(), 0, (M // 144) * 144, 1024, 0, q[0].data_ptr(),
                                                     q[1].data_ptr(), q[2].data_ptr(), 144, 0, 0, 0, cs.data_ptr(), st))
                    elif epi == 8:
                        L.check(lib.gtav_op_gemm_qkvs_attn(x.data_ptr(), w.data_ptr(), (M // 144) * 144, 1024, 144, cs.data_ptr(), out.data_ptr(), st))
                    elif epi == 6:
                        L.check(lib.gtav_op_gemm_f16(x.data_ptr(), K, w.data_ptr(), 0, out.data_ptr(), N, M, N, K, 6, 0, sk, 1, st))
                    else:
                        L.check(lib.gtav_op_gemm_f16(x.data_ptr(), K, w.data_ptr(), bias.data_ptr(), out.data_ptr(), N, M, N, K, epi, 0,
                                                     0, 1, st))
                try:
                    for i in range(4):
                        run(i)
                except L.GtavError as e:
                    print(f"{name} M={M} wm={wm}: skipped ({e})")
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for i in range(32):
                    run(i)
                e1.record()
                torch.cuda.synchronize()
                us_plain = e0.elapsed_time(e1) * 1e3 / 32
                lib.gtav_op_gemm_set_stamps(stamps.data_ptr(), maxb)
                rows = []
                for r in range(a.reps):
                    stamps.zero_()
                    torch.cuda.synchronize()
                    run(r)
                    torch.cuda.synchronize()
                    s = stamps.cpu().reshape(-1, 8)
                    s = s[s[:, 0] != 0]
                    rows.append(s)
                lib.gtav_op_gemm_set_stamps(None, 0)
                s = rows[-1]
                nb = s.shape[0]
                if nb == 0:            # the kernel this shape dispatches to does not write stamps (split-K slabs, persistent tiles without the hook)
                    print(f"{name:>4} M={M:5d} N={N} K={K} wm={wm} splitk={sk}: {us_plain:7.2f} us/launch back-to-back; no stamps from this kernel")
                    continue
                t0, t1, t2, t3, c0, c1, xcc = (s[:, i] for i in range(7))     # t*: s_memrealtime (100 MHz); c*: s_memtime (cycles)
                lstart, lland = s[:, 7], xcc >> 8            # loader-wave kernels: last loader's first issue / its tile-0 pieces landed
                xcc = xcc & 0xF
                span_us = float(t3.max() - t0.min()) / 100.0
                cyc_per_us = float(torch.median((c1 - c0).double() / (t3 - t0).clamp_min(1).double())) * 100.0
                f = lambda c: c.double() / 100.0            # 10 ns ticks -> us
                start = f(t0 - t0.min())
                pro, main_, epi_, tot = f(t1 - t0), f(t2 - t1), f(t3 - t2), f(t3 - t0)
                # overlap: fraction of a block's epilogue interval during which at least one OTHER block is inside its main loop is not
                # computable per CU without the CU id; report the chip-level concurrency instead: at 200 sample instants, how many
                # blocks are in prologue / main / epilogue
                ts = torch.linspace(float(t0.min()), float(t3.max()), 202)[1:-1].to(torch.int64)
                inp = ((t0[None] <= ts[:, None]) & (ts[:, None] < t1[None])).sum(1).double()
                inm = ((t1[None] <= ts[:, None]) & (ts[:, None] < t2[None])).sum(1).double()
                ine = ((t2[None] <= ts[:, None]) & (ts[:, None] < t3[None])).sum(1).double()
                print(f"{name:>4} M={M:5d} N={N} K={K} wm={wm} splitk={sk}: {us_plain:7.2f} us/launch back-to-back; stamped launch: {nb} blocks, "
                      f"span {span_us:6.2f} us, clock {cyc_per_us / 1e3:.2f} GHz, XCDs used {len(set(xcc.tolist()))}")
                print(f"      block start after first: med {pct(start, .5):6.2f} p90 {pct(start, .9):6.2f} max {float(start.max()):6.2f} us")
                print(f"      prologue (entry -> first K-tile landed): med {pct(pro, .5):5.2f} p90 {pct(pro, .9):5.2f} us")
                if int(lstart.max()) != 0:
                    ls = (lstart - t0).double() / 100.0
                    ll = ((lland - (t0 & 0xFFFFFFFFFFFF)).double()) / 100.0
                    print(f"      last loader wave: first fill issued {pct(ls, .5):5.2f} us after block entry (p90 {pct(ls, .9):5.2f}); its tile-0 pieces landed at "
                          f"{pct(ll, .5):5.2f} (p90 {pct(ll, .9):5.2f})")
                print(f"      main loop:                               med {pct(main_, .5):5.2f} p90 {pct(main_, .9):5.2f} us")
                print(f"      epilogue:                                med {pct(epi_, .5):5.2f} p90 {pct(epi_, .9):5.2f} us")
                print(f"      block total:                             med {pct(tot, .5):5.2f} p90 {pct(tot, .9):5.2f} us   (sum of block time / 256 CUs / span = "
                      f"{float(tot.sum()) / 256 / span_us:.2f} blocks resident per CU on average)")
                if int(lstart.max()) != 0 and wm in (40, 41, 42):   # persistent 256-token tiles: slot 7 = the block's end (all its units); the phases above are its FIRST unit
                    print(f"      whole block (all units): med {pct(f(lstart - t0), .5):6.2f} p90 {pct(f(lstart - t0), .9):6.2f} us; first unit's share: main {pct(main_, .5):5.2f} + epilogue {pct(epi_, .5):5.2f}")
                print(f"      resident blocks by phase, averaged over the span: prologue {inp.mean():6.1f}  main {inm.mean():6.1f}  epilogue {ine.mean():6.1f}"
                      f"   | time with NO block in its main loop: {float((inm == 0).double().mean()) * 100:4.1f} %")
                # timeline in 10 bins
                bins = 10
                seg = lambda v: " ".join(f"{float(v[i * 20:(i + 1) * 20].mean()):5.0f}" for i in range(bins))
                print(f"      timeline (10 bins) main    : {seg(inm)}")
                print(f"                         epilogue: {seg(ine)}")
                print(f"                         prologue: {seg(inp)}")
    lib.gtav_op_gemm_set_wm(0)


if __name__ == "__main__":
    main()
