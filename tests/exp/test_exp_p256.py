"""Persistent 256-token-tile kernel (block shapes 40 / 41 / 42, docs/LABNOTES.md 4.11): EXPERIMENTS BUILD ONLY — it measured slower than the product's
shapes at every size and left the product library in round 5, but tools/forward_ab.py and friends still drive it, and its inter-block flag / workspace
hand-off is exactly the kind of code a race hides in.  Run with  GTAV_TEST_EXP=1 python -m pytest tests/exp -q  (the session then loads libgtav_amd_exp.so)."""
import math
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = [pytest.mark.gpu, pytest.mark.exp]

from helpers import dev, gemm, pad_weight_f16, rel_l2, stream, to_tiled_f16, untile  # noqa: E402
from gtav_amd import lib as L  # noqa: E402


def _rand(*shape, scale=1.0, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("shape", [41, 17])   # 17 (round 6): the 256 x 256 tile with its two wave groups in antiphase (mainloop256_pp) — the same ten-run bit-equality screen
@pytest.mark.parametrize("M,N,K", [(5760, 4096, 1024), (5760, 3072, 1024), (2312, 384, 896), (192, 128, 128), (11520, 1024, 1024), (100, 256, 128),
                                   (700, 192, 384), (1152, 4096, 256)])
def test_persistent_256_token_tile_kernel(shape, M, N, K):
    """One block per CU walks whole rounds of 192 x 256 tiles with the two-parity LDS ring running on across tile boundaries, epilogue straight from the
    accumulators; a remainder of at most half a round of tiles is split in two K halves whose partial sums change hands through a workspace and a flag
    (agent-scope release / acquire).  Cases: two whole rounds (506 tiles), one round + split remainder (368 tiles), fewer tiles than CUs with every tile
    split, a handful of tiles with K too short to split, ragged token / feature edges.  Epilogues: GELU-tanh fp16 tile-major, fp32 row-major (+ bias), one
    fp32 slab.  Ten runs must agree bit for bit: a partial tile read before its flag, a flag left set, or a fill that lands late shows up as run-to-run
    differences."""
    lib = L.load()
    x = _rand(M, K, seed=1).half()
    w = _rand(N, K, scale=1 / math.sqrt(K), seed=2)
    b = _rand(N, seed=3)
    w16, xd, bd = pad_weight_f16(w), to_tiled_f16(x), b.to(dev())
    pre = x.float() @ w.half().float().t()
    try:
        lib.gtav_op_gemm_set_wm(shape)
        first = None
        for r in range(10):
            out = torch.zeros(((M + 127) // 128 * 128, N), device=dev(), dtype=torch.float16)
            gemm(xd, w16, bd, M, N, K, 2, out, N)
            f32 = torch.full((M, N), float("nan"), device=dev())
            gemm(xd, w16, bd, M, N, K, 0, f32, N)
            parts = torch.full((M, N), float("nan"), device=dev())
            L.check(lib.gtav_op_gemm_f16(xd.data_ptr(), K, w16.data_ptr(), 0, parts.data_ptr(), N, M, N, K, 6, 0, 1, 1, stream()))
            if first is None:
                first = (out.clone(), f32.clone(), parts.clone())
                assert rel_l2(untile(out, M, N).float(), torch.nn.functional.gelu(pre + b, approximate="tanh")) < 6e-4
                assert rel_l2(f32, pre + b) < 2e-5
                assert rel_l2(parts, pre) < 2e-5
            else:
                assert torch.equal(out, first[0]) and torch.equal(f32, first[1]) and torch.equal(parts, first[2]), r
    finally:
        lib.gtav_op_gemm_set_wm(0)


@pytest.mark.parametrize("shape", [22, 18])
def test_weight_operand_straight_into_registers(shape):
    """Block shapes 22 / 18 (round 6; csrc/gemm.hip mainloop_lw): the 128 x 96 loader-wave tile with the weight operand loaded by the compute waves straight into a
    register ring 4 / 8 K-steps ahead (hand-counted vmcnt), X through the LDS ring as before — measured slower than shape 20, kept for the record.  Every epilogue
    through the product suite's shape sweep (K-steps per slice that are not a multiple of the ring depth fall back to shape 20)."""
    import test_gpu_ops as T
    T.test_gemm_other_tiles_all_epilogues(shape)
