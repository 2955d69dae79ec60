"""One line per GPU box: what differs between MI355X boxes for the batch-1 sampler step.  Measures, in ONE process:
  * the captured window step (batch 1, M = 720) with the next-weight L2 prefetch on and off (gtav_amd.generate.tune_weight_prefetch);
  * the in-kernel shader clock the chip holds under the step's GEMMs: s_memtime / s_memrealtime stamps of the loader-wave kernel (experiments build,
    tools/gemm_stamps.py's buffers) for fc2 at M = 720 and for fc1 at M = 5760, and the same kernels' back-to-back time;
  * a plain device-to-device copy (HBM bandwidth) and the empty-kernel launch floor as seen through torch.
Usage (GPU box): python tools/box_probe.py >> gpurun_out/box_probe.jsonl"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gtav_amd import lib as L  # noqa: E402


LAST_XCC = []


def gemm_clock(lib, M, N, K, epi, wm=0):
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    x = (torch.randn((M + 127) // 128 * 128, K, device=dev) * 0.5).half()
    ws = [(torch.randn(N, K, device=dev) * 0.03).half() for _ in range(8)]
    bias = torch.randn(N, device=dev)
    sk = lib.gtav_op_gemm_choose_splitk(M, N, K) if epi == 6 else 1
    out = torch.empty((max(sk, 1) * ((M + 127) // 128 * 128), N), device=dev, dtype=torch.float32 if epi in (0, 6) else torch.float16)
    lib.gtav_op_gemm_set_wm(wm)

    def run(i):
        if epi == 6:
            L.check(lib.gtav_op_gemm_f16(x.data_ptr(), K, ws[i % 8].data_ptr(), 0, out.data_ptr(), N, M, N, K, 6, 0, sk, 1, st))
        else:
            L.check(lib.gtav_op_gemm_f16(x.data_ptr(), K, ws[i % 8].data_ptr(), bias.data_ptr(), out.data_ptr(), N, M, N, K, epi, 0, 0, 1, st))
    for i in range(8):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(64):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 64
    maxb = 16384
    stamps = torch.zeros(maxb * 8, dtype=torch.int64, device=dev)
    lib.gtav_op_gemm_set_stamps(stamps.data_ptr(), maxb)
    # warm the clock with a burst, then one stamped launch
    for i in range(32):
        run(i)
    stamps.zero_()
    torch.cuda.synchronize()
    for i in range(4):
        run(i)
    torch.cuda.synchronize()
    s = stamps.cpu().reshape(-1, 8)
    s = s[s[:, 0] != 0]
    lib.gtav_op_gemm_set_stamps(None, 0)
    lib.gtav_op_gemm_set_wm(0)
    global LAST_XCC
    LAST_XCC = [int(v) & 0xF for v in s[:16, 6].tolist()]          # XCC id of blocks 0 .. 15 of the last stamped launch
    t0, t3, c0, c1 = s[:, 0], s[:, 3], s[:, 4], s[:, 5]
    ok = (t3 > t0) & (c1 > c0)
    ghz = float(torch.median(((c1 - c0)[ok].double() / (t3 - t0)[ok].double()))) * 0.1
    return round(us, 2), round(ghz, 3)


def main():
    lib = L.load_experiments()
    dev = torch.device("cuda", 0)
    import gtav_amd.weights as W
    from gtav_amd.generate import tune_weight_prefetch
    from gtav_amd.model.dit import DiT_models
    out = {}
    a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        b.copy_(a)
    torch.cuda.synchronize()
    out["copy_tb_s"] = round(2 * a.numel() * 10 / (time.perf_counter() - t0) / 1e12, 3)
    del a, b
    xccs = []
    for _ in range(4):       # block -> XCD placement of consecutive launches of one kernel (the prefetch assumes block b on XCD (b + c) mod 8 with ONE c for every launch)
        out["fc2_M720_us"], out["fc2_M720_clock_ghz"] = gemm_clock(lib, 720, 1024, 4096, 6)
        xccs.append(LAST_XCC)
    out["xcc_of_blocks_0_15_in_4_launches"] = xccs
    out["fc1_M720_us"], out["fc1_M720_clock_ghz"] = gemm_clock(lib, 720, 4096, 1024, 2)
    out["fc1_M5760_us"], out["fc1_M5760_clock_ghz"] = gemm_clock(lib, 5760, 4096, 1024, 2)
    dit = DiT_models["DiT-S/2"](init_weights=False, max_batch=1)
    dit.load_state_dict(W.synth_state_dict(W.dit_param_shapes(depth=16), seed=0))
    dit.reserve(1, 5, 100)
    out["step"] = tune_weight_prefetch(dit, 1, steps=60, rounds=3)
    try:
        out["gpu"] = torch.cuda.get_device_properties(0).name
        import subprocess
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20)
        if r.returncode == 0:
            out["rocm_smi"] = json.loads(r.stdout)
    except Exception as e:  # noqa: BLE001
        out["rocm_smi_error"] = str(e)[:100]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
