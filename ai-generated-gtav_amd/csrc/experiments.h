/* Entry points that exist ONLY in libgtav_amd_exp.so (csrc/build.sh exp: -DGTAV_EXPERIMENTS), the build the timing tools
 * under tools/ load.  They are not part of the product ABI (include/gtav_amd.h) because they change results.
 * That build also reads the GTAV_* environment variables listed in DESIGN.md "Experiment knobs". */
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* bit 0: skip the LDS fills after the prologue; bit 1: skip LDS reads + MFMA (results become WRONG: timing only);
 * bit 3: plain instead of write-through (sc1) output stores; bit 4: unstaged QKV epilogue (bits 3, 4 keep results). */
void gtav_op_gemm_set_debug(int32_t bits);
/* Per-block timeline of the following GEMM launches: 8 x uint64 per block (csrc/gemm.h gemm_set_stamps); NULL = off. */
void gtav_op_gemm_set_stamps(void* buf_dev, int32_t max_blocks);
#ifdef __cplusplus
}
#endif
