/* gctplus_diag.h -- diagnostics of the box a run lands on (libgctplus_diag.so, built from csrc/graphprobe.hip).
 * NOT part of the operator boundary (include/gctplus_hip.h): nothing here computes anything of the model; the decode
 * path works without this library (gct_plus_amd/graphdiag.py degrades to "no diagnosis" when it is missing).
 * Error convention as in gctplus_hip.h: 0 = ok, negative = error, text through gct_diag_last_error(). */
#ifndef GCTPLUS_DIAG_H
#define GCTPLUS_DIAG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* gct_diag_last_error(void);

/* ------------------------------------------------------------------ hipGraph replay diagnostics (csrc/graphprobe.hip)
 * BASELINE configs[4] asks for a hipGraph-captured decode step (reference loop: Inference/sampling_tool.py:140-184).
 * Some boxes replay graphs far slower than they launch the same kernels one by one; these calls tell a caller what
 * the box does before it trusts a replay.
 * gct_graph_probe: `nodes` launches of a do-nothing kernel chained on a private stream, timed as `reps` eager passes
 *   and as `reps` replays of the captured chain (ms per pass).  variant 0: 8-byte kernarg, one workgroup; 1: a 320-byte
 *   by-value argument block read by every wave of a 2048 x 512 grid; 2: as 1 with 144 KB of dynamic LDS; 3: as 1 with
 *   the block behind one pointer into device memory.  Synchronises its own stream only.
 * gct_device_facts: runtime / driver version, large-BAR and host-access attributes ... as one JSON object.
 * gct_graph_census: node counts of a captured hipGraph_t; out8 = {nodes, kernels, memcpys, memsets, others,
 *   max dynamic LDS bytes, max grid blocks, kernels with more than 64 KB of LDS}. */
int gct_graph_probe(int variant, int nodes, int reps, float* eager_ms, float* graph_ms, int32_t* graph_nodes);
int gct_device_facts(char* buf, int cap);
int gct_graph_census(void* hip_graph, int64_t* out8);

/* Box calibration: the bf16 MFMA rate (TFLOP/s, dense peak ~2 500) this device sustains on a register-only loop of
 * v_mfma_f32_16x16x32_bf16 (two waves per SIMD on every CU, pseudo-random operands, `iters` x 8 MFMAs per wave; best of
 * three timed launches on a private stream, ms_out nullable).  Devices of one pool differ by ~10 % here, and so does
 * the matrix-bound training step: a bench line quotes this figure so that two boxes can be told from two code versions. */
int gct_mfma_clock_probe(int iters, float* tflops, float* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* GCTPLUS_DIAG_H */
