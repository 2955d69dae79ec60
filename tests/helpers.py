"""Thin torch wrappers over the kernel-level C-ABI entry points (test infrastructure)."""
import torch

from gtav_amd import lib as L


def dev():
    return torch.device("cuda", 0)


def stream():
    return L.current_stream()


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def tiled_index(R, K):
    """Offsets (in halves) of element (r, k) in the GEMM's tile-major operand layout (csrc/common.h tiled_off)."""
    r = torch.arange(R)[:, None]
    k = torch.arange(K)[None, :]
    return ((r >> 7) * (K >> 6) + (k >> 6)) * 8192 + (r & 127) * 64 + ((((k >> 3) & 7) ^ (r & 7)) << 3) + (k & 7)


def untile(buf, R, K):
    """tile-major fp16 device buffer -> row-major (R, K) CPU tensor."""
    flat = buf.reshape(-1).cpu()
    return flat[tiled_index(R, K).reshape(-1)].reshape(R, K)


def to_tiled_f16(x):
    """fp32/fp16 (R, K) -> tile-major fp16 (round_up(R,128) * round_up(K,64) halves, zero padded) on the GPU."""
    R, K = x.shape
    Rp, Kp = (R + 127) // 128 * 128, (K + 63) // 64 * 64
    out = torch.empty((Rp, Kp), device=dev(), dtype=torch.float16)
    src = x.to(dev(), torch.float32).contiguous()
    L.check(L.load().gtav_op_convert_f16(src.data_ptr(), K, R, K, out.data_ptr(), Rp, Kp, 1, stream()))
    return out


pad_weight_f16 = to_tiled_f16


def gemm(x16, w16, bias, M, N, K, epi, out, ldo, gate=None, gate_stride=0, rows_per_gate=1):
    L.check(L.load().gtav_op_gemm_f16(x16.data_ptr(), K, w16.data_ptr(), L.ptr(bias), out.data_ptr(), ldo, M, N, K,
                                      epi, L.ptr(gate), gate_stride, rows_per_gate, stream()))
    return out


def resize_probe_image(H, W):
    """The deterministic uint8 probe image (H, W, 3) of fixture G11 (tools/make_golden.py resize_probe_image: integer arithmetic only)."""
    import torch
    y = torch.arange(H).view(H, 1, 1)
    x = torch.arange(W).view(1, W, 1)
    c = torch.arange(3).view(1, 1, 3)
    return ((x * 7 + y * 13 + c * 29 + (x * y) % 11 + ((x // 3) * (y // 5)) % 17 * 9) % 256).to(torch.uint8)


G11_ROWS = (0, 1, 100, 179, 180, 358, 359)
G11_SIZES = {"500x333": (333, 500), "480x270": (270, 480), "1280x720": (720, 1280)}
