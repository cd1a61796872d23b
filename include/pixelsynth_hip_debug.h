/*
 * pixelsynth_hip_debug.h -- measurement, tuning and debugging entry points of libpixelsynth_hip.so.
 *
 * NOT part of the drop-in boundary (include/pixelsynth_hip.h): nothing here has a counterpart in the reference, and no
 * product path calls it.  Users: tests/ (switching launch forms inside one process, reading the handle's caches), bench.py
 * (HIP-event timing of the column launches for the `roofline` object) and tools/.
 */
#ifndef PIXELSYNTH_HIP_DEBUG_H
#define PIXELSYNTH_HIP_DEBUG_H

#include "pixelsynth_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Tuning values of a handle, by name (pixelsynth_amd/csrc/lmconv_handle.h: struct Tuning; lmconv.hip: tuning_table): which launch
 * form the whole-grid pass takes from which size on (gemm_merge_min, gemm_wg_min, gemm_ws_min, gemm_ws, wg_ti_out / _in / _dil), whether the prefix
 * pass skips the items nobody reads (prefix_full, prefix_cone_force), the look-ahead depths of the column launches (tp_ahead,
 * col_ahead: only before the handle's first column launch), the form and placement of a column launch (tp_min_cols, tp_xcds,
 * tp_fill, col_cap, chain_xcds, nbr_groups).  No value changes results: every form is bit-identical (tests/test_lmconv_gpu.py).
 * ps_pixelcnn_create reads the same names ONCE from the environment as PS_<NAME IN UPPER CASE>. */
int ps_pixelcnn_set_tuning(ps_pixelcnn *h, const char *key, int value);
int ps_pixelcnn_get_tuning(ps_pixelcnn *h, const char *key, int *value);

/* bench.py aid: ps_pixelcnn_ar_run_waves (uniforms, no logits) with a HIP event pair around every column launch on
 * the caller's stream; synchronises.  launches / total_ms: the k_column launches of the run and their summed
 * duration; flops_per_column: dense flops of one column (11.163 MFLOP). */
int ps_pixelcnn_time_ar_run_waves(ps_pixelcnn *h, int32_t *codes, const int32_t *order,
                                  const uint8_t *sample_region, const float *mask_init,
                                  const float *mask_undilated, const float *mask_dilated,
                                  const float *uniforms, float temperature, int F, int first_step,
                                  const int32_t *wave_cols, const int32_t *wave_start, int n_waves,
                                  int *launches, float *total_ms, double *flops_per_column, void *stream);
/* ... counting only the launches of the wavefronts [wave_from, wave_to) of the schedule (bench.py: one steady-state step of the
 * pipelined form -- the last wavefronts of one batch inside the launches of the next batch's first ones -- within a two-batch run).
 * first_steps (device, (F)) / max_first_step: per-frame prefixes as for ps_pixelcnn_ar_prefix_frames, or NULL / -1. */
int ps_pixelcnn_time_ar_run_waves_range(ps_pixelcnn *h, int32_t *codes, const int32_t *order,
                                        const uint8_t *sample_region, const float *mask_init,
                                        const float *mask_undilated, const float *mask_dilated,
                                        const float *uniforms, float temperature, int F, int first_step,
                                        const int32_t *wave_cols, const int32_t *wave_start, int n_waves,
                                        int wave_from, int wave_to, const int32_t *first_steps, int max_first_step,
                                        int *launches, float *total_ms, double *flops_per_column, void *stream);

/* Which kernels carried the matrix work (tests: "the forms the headline takes really ran"; bench.py: `roofline.kernels`).
 * Kinds 0 .. ps_pixelcnn_launch_kinds() - 1, named by ps_pixelcnn_launch_kind_name: k_column, k_column_la, k_column_tp, k_column_tp8
 * (column launches: one-position walk, latency form, throughput form with chain tiles of 16 / 8 columns), k_gemm, k_gemm_wg,
 * k_gemm_ws<0> / <1> / <2> (whole-grid products: one wave per tile; rows through LDS; weights through LDS for conv_out /
 * conv_input / dilated).  ps_pixelcnn_launch_counts: cumulative launches of the handle since its creation, by kind (host counters,
 * no synchronisation).  ps_pixelcnn_profile_begin / _end: between the two calls every such launch -- whichever entry point
 * enqueues it, on whichever stream -- is bracketed by a HIP event pair on ITS stream; _end synchronises the device and returns
 * launches and summed kernel time by kind (launches[n], total_ms[n]).  The kernels of two streams overlap: the sums can exceed
 * the wall time. */
int ps_pixelcnn_launch_kinds(void);
const char *ps_pixelcnn_launch_kind_name(int kind);
int ps_pixelcnn_launch_counts(ps_pixelcnn *h, long long *counts, int n);
int ps_pixelcnn_profile_begin(ps_pixelcnn *h);
int ps_pixelcnn_profile_end(ps_pixelcnn *h, int n, int *launches, float *total_ms);

/* Debugging aid (tools/tp_debug.py): device address of one of the handle's
 * activation caches -- what 0: raw u of node idx (19 nodes, row stride 96 floats), 1: concat_elu(u) of node idx (160),
 * 2: the activation inside gated resnet idx (14 blocks, 160); rows are frame * L + location; 5: the (33, F) int32 table of
 * the last AR run's prefix pass -- first order rank evaluated per stage and frame (oracle/prefix_cone_oracle.py).  NULL if
 * out of range. */
void *ps_pixelcnn_debug_cache(ps_pixelcnn *h, int what, int idx);

/* Measurement aid for bench.py (not part of the reference surface): evaluates `reps` order positions
 * eagerly on `stream` (at position `step`, without drawing) with a HIP event pair around every kernel
 * launch and returns the number of launches and their summed duration in ms:
 *   [0] unused (0 launches; the neighbour taps had their own kernel before the single-launch column step)
 *   [1] k_column (one launch per order position here: neighbour-tap slots of all 32 masked convs on MFMA +
 *       the per-frame centre-tap chains, post ops and draw)
 * flops_per_launch / weight_bytes_per_launch [2]: dense algorithmic work of one launch, split as
 *   [0] neighbour taps, [1] centre-tap chain (2*Co*Cin per tap and frame; fp32 weight bytes streamed once,
 *   per frame for [1]).  Synchronises the stream. */
#define PS_PROF_NTAGS 2
int ps_pixelcnn_time_column_step(ps_pixelcnn *h, const int32_t *codes, const int32_t *order,
                                 const float *mask_init, const float *mask_undilated,
                                 const float *mask_dilated, int F, int step, int reps, int *launches,
                                 float *total_ms, double *flops_per_launch,
                                 double *weight_bytes_per_launch, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PIXELSYNTH_HIP_DEBUG_H */
