/* gtav_amd — test hooks of libgtav_amd.so.  NOT part of the product interface (include/gtav_amd.h): nothing in a binding of the
 * reference's API needs them.  tests/ use them to run every GEMM block shape the launch heuristic can pick through every epilogue.
 *
 * Both settings are PER THREAD (thread_local in csrc/gemm.hip), like the handles: forcing a shape on one thread never changes what
 * another thread's handle launches.  Every choice computes the same result.  0 restores the heuristic.
 *   gtav_op_gemm_set_stages   LDS ring depth of the 128 x 128 tiles: 2 or 4
 *   gtav_op_gemm_set_wm       block shape: 2, 3 (128 x 128, 4 / 8 waves), 7 (256 x 256), 11 (64 x 48), 12 (128 x 192), 13 (128 x 96, 4 waves), 14 (64 x 96),
 *                             20 / 24 / 26 (loader-wave kernel, 128 x 96 / 64 x 48 / 64 x 96 tiles), 31 (persistent loader-wave kernel, 128 x 192 tiles, 3-stage ring; its 4-stage / 256 x 128 / 128 x 256 forms 30 / 32 / 33 exist in the experiments build only)
 *                             — csrc/gemm.hip launch_epi
 * (Timing experiments that change results, and the block shapes that measured slower than these, exist only in the separate
 * -DGTAV_EXPERIMENTS build: csrc/build.sh exp -> libgtav_amd_exp.so, loaded by tools/ only.) */
#ifndef GTAV_AMD_TESTING_H
#define GTAV_AMD_TESTING_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

void gtav_op_gemm_set_stages(int32_t ns);
void gtav_op_gemm_set_wm(int32_t wm);

#ifdef __cplusplus
}
#endif
#endif /* GTAV_AMD_TESTING_H */
