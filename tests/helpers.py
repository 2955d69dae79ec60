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


def pad_weight_f16(w):
    """fp32 (N, K) -> fp16 (round_up(N,128), round_up(K,64)) zero padded, on the GPU (via the convert kernel)."""
    N, K = w.shape
    Np, Kp = (N + 127) // 128 * 128, (K + 63) // 64 * 64
    out = torch.empty((Np, Kp), device=dev(), dtype=torch.float16)
    src = w.to(dev(), torch.float32).contiguous()
    L.check(L.load().gtav_op_convert_f16(src.data_ptr(), K, N, K, out.data_ptr(), Np, Kp, stream()))
    return out


def gemm(x16, w16, bias, M, N, K, epi, out, ldo, gate=None, gate_stride=0, rows_per_gate=1):
    L.check(L.load().gtav_op_gemm_f16(x16.data_ptr(), x16.shape[1], w16.data_ptr(), L.ptr(bias), out.data_ptr(), ldo, M, N, K,
                                      epi, L.ptr(gate), gate_stride, rows_per_gate, stream()))
    return out
