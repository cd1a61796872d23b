/*
 * pixelsynth_hip.h -- C ABI of libpixelsynth_hip.so, the MI355X (gfx950) implementation of the
 * PixelSynth novel-view inference hot path.
 *
 * Boundary contract:
 *   - plain pointers and sizes only; every device buffer is allocated and owned by the CALLER
 *     (PyTorch-ROCm: tensor.data_ptr()); the library never frees or retains caller memory;
 *   - contiguous row-major layouts exactly as documented per function;
 *   - `stream` is the caller's hipStream_t (torch.cuda.current_stream().cuda_stream); all device
 *     work is asynchronous on it; no hidden hipDeviceSynchronize / hipMalloc on the data path;
 *   - return value 0 = ok, negative = error; the message is in ps_last_error() (thread local);
 *     no C++ exception crosses the ABI;
 *   - re-entrant: no global mutable state besides opaque handles the caller created.
 *
 * Every entry point names the reference interface (crockwell/pixelsynth file:line) it replaces.
 */
#ifndef PIXELSYNTH_HIP_H
#define PIXELSYNTH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PS_ABI_VERSION 2

enum { PS_OK = 0, PS_ERR_ARG = -1, PS_ERR_HIP = -2, PS_ERR_WORKSPACE = -3, PS_ERR_STATE = -4 };
enum { PS_ACC_ALPHACOMPOSITE = 0, PS_ACC_WSUM = 1, PS_ACC_WSUMNORM = 2 }; /* opts.accumulation */
/* Bits of a caller-owned device status word (`int32_t *status`, zero-initialised by the caller, may be NULL): asynchronous
 * entry points that can only detect bad DATA on the device raise them there; ps_read_status turns them into an error. */
enum { PS_STATUS_BAD_ORDER = 1, PS_STATUS_BAD_PIXEL = 2 };

int ps_abi_version(void);
/* What the library was built from: ABI version, target, flags, and whether any tuning / trace / experiment macro was passed to the
 * build ("product build" or "NON-PRODUCT build, extra flags: ..."; pixelsynth_amd/build.py).  A static string. */
const char *ps_build_info(void);
const char *ps_last_error(void);
/* Synchronises `stream`, reads the caller's status word and clears it: PS_OK when no asynchronous call that was handed this
 * word has raised a bit since the last read, otherwise PS_ERR_STATE with the decoded bits in ps_last_error().  (The library
 * itself keeps no flag: no global mutable state.) */
int ps_read_status(int32_t *status, void *stream);

/* ------------------------------------------------------------------------------------------
 * Reprojection + soft z-buffer splat
 * ---------------------------------------------------------------------------------------- */

/* PtsManipulator.project_pts  (models/projection/z_buffer_manipulator.py:50-83).
 *   depth (B,1,N=W*W) f32, K/Kinv/RT1inv/RT2 (B,4,4) f32  ->  sampler (B,3,N) f32.
 *   (RT_cam1 and RTinv_cam2 of the reference signature are unused by the reference itself.) */
int ps_project_pts_f32(const float *depth, const float *K, const float *Kinv, const float *RT1inv,
                       const float *RT2, int B, int W, float *sampler, void *stream);

/* PtsManipulator.project_pts_cumulative  (models/projection/z_buffer_manipulator.py:221-266).
 *   depth_new (B,1,n_new); new_index (B,n_new) int32 = row-major grid indices kept by
 *   last_background_mask (NULL: identity, n_new == W*W); prior (B,4,n_prior) or NULL;
 *   RT3inv (B,4,4) or NULL  ->  sampler (B,3,n_new+n_prior), cloud (B,4,n_new+n_prior)
 *   (cloud carries the in-place EPS write of :253-254, it is the next prior_point_cloud). */
int ps_project_pts_cumulative_f32(const float *depth_new, const int32_t *new_index,
                                  const float *prior, const float *K, const float *Kinv,
                                  const float *RT1inv, const float *RT2, const float *RT3inv, int B,
                                  int W, int n_new, int n_prior, float *sampler, float *cloud,
                                  void *stream);

/* Bytes of scratch ps_splat_f32 / ps_project_splat_f32 need for (B clouds of N points, S x S image,
 * radius in pixels).  Host-only arithmetic. */
size_t ps_splat_workspace_bytes(int B, int N, int S, double radius_px);

/* RasterizePointsXYsBlending.forward  (models/layers/z_buffer_layers.py:55-131) together with the
 * PyTorch3D calls it makes (rasterize_points :82-84; compositing.* :112-129):
 *   pts  (B,N,3) f32  -- x and y are negated IN PLACE exactly like the reference (:71-72)
 *   feat (B,C,N) f32
 *   out_feat (B,C,S,S) f32 ; out_bg (B,S,S) uint8 (bool)  = dilated "no point hit" mask (:100-110)
 *   optional debug outputs (NULL to skip), PyTorch3D rasterize_points layout (B,S,S,K):
 *     out_idx int32 (packed index b*N+n, -1 padded), out_zbuf f32, out_dist f32 (dist^2, NDC)
 *   radius_px = opts.radius, K = opts.pp_pixel, tau, rad_pow, accumulation (PS_ACC_*),
 *   bg_ksize = opts.background_smoothing_kernel_size (odd).
 *   workspace: >= ps_splat_workspace_bytes(B,N,S,radius_px) bytes of device memory.
 *   Numerics: out_bg and the debug outputs are exact (the K nearest hits per pixel in (z, index) order).  out_feat under
 *   PS_ACC_ALPHACOMPOSITE WITHOUT debug outputs: a pixel's front-to-back walk stops once its transmittance prod(1 - alpha) is below
 *   2^-23 -- the hits behind can add at most that times max |feature| (below one ulp of a unit-magnitude result; the parity tests
 *   state 1e-6) --, a hit's alpha comes from the hardware's 1-ulp square root and the sum cum * alpha * feature is one fused multiply-add
 *   (together <= 3e-7 x max |feature| from the exact walk, measured); with debug outputs, and in the other accumulation modes, every one of
 *   the K hits is walked with the correctly rounded root and separately rounded product and sum (the bit-exact route). */
int ps_splat_f32(float *pts, const float *feat, int B, int N, int C, int S, double radius_px, int K,
                 float tau, int rad_pow, int accumulation, int bg_ksize, float *out_feat,
                 uint8_t *out_bg, int32_t *out_idx, float *out_zbuf, float *out_dist,
                 void *workspace, size_t workspace_bytes, void *stream);

/* PtsManipulator.forward_justpts  (models/projection/z_buffer_manipulator.py:85-107): project_pts
 * fused with the splat; the (B,N,3) point cloud lives only in the workspace.
 *   depth (B,1,S,S), feat (B,C,S,S)  ->  out_feat (B,C,S,S), out_bg (B,S,S) uint8. */
int ps_project_splat_f32(const float *depth, const float *feat, const float *K, const float *Kinv,
                         const float *RT1inv, const float *RT2, int B, int C, int S,
                         double radius_px, int Kpp, float tau, int rad_pow, int accumulation,
                         int bg_ksize, float *out_feat, uint8_t *out_bg, void *workspace,
                         size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Generation order + kernel masks (host side, integer; models/z_buffermodel.py:641-701)
 * ---------------------------------------------------------------------------------------- */

/* One image of ZbufferModelPts.get_masks_for_batch up to the order:
 *   bg (S,S) uint8 -> AvgPool2d(S/G) + uint8 truncation (:646-647,668-669) -> 5x5 chamfer L2
 *   distance transforms (cv2.distanceTransform :673-674, portable fixed-point definition) ->
 *   D = int(fg_dist - bg_dist) (:675) -> custom_idx (models/lmconv/get_custom_order.pyx:4-124).
 *   out: order (G*G,2) int32 (r,c); bg_blocks (G,G) uint8 (1 = block entirely background, i.e. the
 *   sample region of models/lmconv/sample.py:24-41); distances (G,G) int64 = D (NULL to skip). */
int ps_generation_order(const uint8_t *bg, int S, int G, int32_t *order, uint8_t *bg_blocks,
                        int64_t *distances);

/* get_custom_order.custom_idx alone (models/lmconv/get_custom_order.pyx:4-124):
 *   distances (rows,cols) int64, multiplied by 10000 IN PLACE like the reference (:26). */
int ps_custom_order(int rows, int cols, int64_t *distances, int32_t *order);

/* masking.get_unfolded_masks / kernel_masks (models/lmconv/masking.py:287-349), observed_idx=None:
 *   order (L,2) int32 -> masks (k*k, nrows*ncols) f32 (the (1,9,L) tensor of the reference).
 *   mask_type_b: 0 = type A (centre 0), 1 = type B (centre 1). */
int ps_kernel_masks_f32(const int32_t *order, int L, int nrows, int ncols, int k, int dilation,
                        int mask_type_b, float *masks);

/* Batched host glue of ZbufferModelPts.get_masks_for_batch (models/z_buffermodel.py:641-701) in the
 * compact form the HIP sampler consumes: for each of the B background masks bg (B,S,S) uint8:
 *   order_loc (B,L) int32: row-major location visited at each order position (L = G*G);
 *   region (B,L) uint8 by location: 1 = block entirely background = sampled (sample.py:24-41);
 *   mask_init / mask_undilated / mask_dilated (B,9,L) f32: type A dil 1, type B dil 1, type B dil 2
 *   (one copy per image instead of the reference's x513 / x160 / x80 channel repeats, :697-699);
 *   first_step: smallest order position that is sampled in any image (L if none). */
int ps_ar_plan(const uint8_t *bg, int B, int S, int G, int32_t *order_loc, uint8_t *region,
               float *mask_init, float *mask_undilated, float *mask_dilated, int32_t *first_step);
/* (the three mask pointers may all be NULL: callers that keep the plan on the device build the masks there, below) */

/* The three kernel masks of ps_ar_plan / ps_kernel_masks_f32 on the DEVICE, from generation orders that are already there:
 * order_loc (F,L) int32 device -> mask_init / mask_undilated / mask_dilated (F,9,L) f32 device (type A dil 1, type B dil 1,
 * type B dil 2; masking.py:287-370).  Saves the 27 floats per location a host-built plan sends over PCIe (14 MB at 128
 * views); asynchronous on `stream`.  An order that is no permutation of the H*W locations raises PS_STATUS_BAD_ORDER in
 * `status` (its out-of-range entries are skipped). */
int ps_order_masks_f32(const int32_t *order_loc, int F, int H, int W, float *mask_init, float *mask_undilated,
                       float *mask_dilated, int32_t *status, void *stream);

/* Wavefront schedule of an AR run (host).  In the exact incremental form of sample() (models/lmconv/sample.py:24-66)
 * the column of order position i of a frame reads only the finished columns of locations that are a 3x3 tap neighbour
 * (dilation 1 or 2) of its own location AND earlier in the order -- the open taps of the three kernel masks
 * (masking.py:287-370).  The columns of a frame therefore form a DAG; all columns of one DAG level ("wave") can be
 * evaluated and sampled together, with results identical to the position-by-position walk.
 *   order_loc (B,L) int32 as from ps_ar_plan, L = H*W; first_step as from ps_ar_plan (columns before it are done by
 *   the whole-grid pass);
 *   cols (B*(L-first_step), 2) int32 OUT: (frame, order position) of every walked column, wave by wave;
 *   wave_start (L-first_step+1) int32 OUT: wave w = cols[wave_start[w] .. wave_start[w+1]);  *n_waves OUT. */
int ps_ar_wavefronts(const int32_t *order_loc, int B, int H, int W, int first_step, int32_t *cols,
                     int32_t *wave_start, int32_t *n_waves);

/* The same with at most max_cols columns per wave (what one launch of ps_pixelcnn_ar_run_waves takes: 128): list
 * scheduling over the ready columns, longest chain of dependants first, so that what does not fit in a wave shares the
 * next one with the columns that have become ready meanwhile instead of costing a launch of its own.  max_cols = 0:
 * the pure levels of ps_ar_wavefronts.  wave_start needs (L-first_step) + ceil(B*(L-first_step)/max_cols) + 1 entries. */
int ps_ar_wavefronts_capped(const int32_t *order_loc, int B, int H, int W, int first_step, int max_cols,
                            int32_t *cols, int32_t *wave_start, int32_t *n_waves);
/* The same with a first walked position PER FRAME (first_steps (B) int32, host): frame b's columns are its positions from
 * first_steps[b] on -- everything in front of a frame's own first sampled location belongs to its whole-grid pass
 * (ps_pixelcnn_ar_prefix_frames).  cols holds sum_b (L - first_steps[b]) columns. */
int ps_ar_wavefronts_frames(const int32_t *order_loc, int B, int H, int W, const int32_t *first_steps, int max_cols,
                            int32_t *cols, int32_t *wave_start, int32_t *n_waves);

/* ------------------------------------------------------------------------------------------
 * Locally masked convolution / PixelCNN (models/lmconv)
 * ---------------------------------------------------------------------------------------- */

/* Scratch bytes ps_lmconv_forward_f32 needs (device memory, caller-owned). */
size_t ps_lmconv_workspace_bytes(int B, int Ci, int Co, int H, int W);

/* _locally_masked_conv2d.forward  (models/lmconv/locally_masked_convolution.py:11-50), 3x3:
 *   x (B,Ci,H,W) f32 ; mask (B,9,H*W) f32 (the reference passes it repeated Ci times, (B*Ci,9,L),
 *   identical across channels -- the caller hands over one copy per image; mask_batch_stride = 0
 *   broadcasts one mask to the whole batch, otherwise 9*H*W) ; weight (Co,Ci,3,3) ; bias (Co) or NULL
 *   -> y (B,Co,H,W).  padding = dilation (:118-120). */
int ps_lmconv_forward_f32(const float *x, const float *mask, size_t mask_batch_stride,
                          const float *weight, const float *bias, int B, int Ci, int Co, int H,
                          int W, int dilation, float *y, void *workspace, size_t workspace_bytes,
                          void *stream);

/* ---- OurPixelCNN with PixelSynth's configuration (models/z_buffermodel.py:62-74):
 *      nr_resnet=2, nr_filters=80, input_channels=512, 3x3 kernels, max_dilation=2, PONO norms,
 *      weight_norm=False on the convs, weight-normed nin's, dropout 0.
 * The handle owns device copies of the weights (re-packed for MFMA), the per-location activation
 * caches and its scratch; it is created once and destroyed explicitly. */
typedef struct ps_pixelcnn ps_pixelcnn;

#define PS_PIXELCNN_NUM_PARAMS 93
/* params: 93 HOST pointers to f32 tensors in the reference state_dict order of OurPixelCNN
 * (models/lmconv/model.py:61-108; the order is spelled out in pixelsynth_amd/lmconv/model.py:PARAM_KEYS):
 * down_layers.{0..2}.u_stream.*.{conv_input.weight,conv_input.bias,nin_skip.lin_a.bias,
 * nin_skip.lin_a.weight_g,nin_skip.lin_a.weight_v,conv_out.weight,conv_out.bias},
 * up_layers.{0..2}.u_stream.{0,1}.{conv_input.weight,.bias,conv_out.weight,.bias}, u_init.{weight,bias},
 * downsize_u_stream.{0,1}.{weight,bias}, upsize_u_stream.{0,1}.{weight,bias},
 * nin_out.lin_a.{bias,weight_g,weight_v}.   H x W = code grid (32 x 32), max_frames = most images
 * evaluated at once. */
int ps_pixelcnn_create(const float *const *params, int n_params, int H, int W, int max_frames,
                       ps_pixelcnn **out);
void ps_pixelcnn_destroy(ps_pixelcnn *h);

/* OurPixelCNN.forward(sample=True) on one-hot input (models/lmconv/model.py:110-155):
 *   codes (F,H*W) int32: the class of the one-hot input at each location, -1 = all-zero input
 *   (a not-yet-sampled location, models/lmconv/sample.py:47);
 *   mask_init / mask_undilated / mask_dilated (F,9,H*W) f32 (one copy per image, see above)
 *   -> logits (F,512,H,W) f32. */
int ps_pixelcnn_forward_f32(ps_pixelcnn *h, const int32_t *codes, const float *mask_init,
                            const float *mask_undilated, const float *mask_dilated, int F,
                            float *logits, void *stream);

/* The autoregressive loop of sample() (models/lmconv/sample.py:8-73), exact incremental form: every
 * location is evaluated once, in generation order, as a single column through the network against
 * cached activations (valid because every mask only admits locations that precede in the order).
 *   codes (F,L) int32 in/out: observed codes; entries of the sample region are overwritten;
 *   order (F,L) int32: row-major location index visited at each order position;
 *   sample_region (F,L) uint8 by LOCATION: 1 = sample this location (sample.py:24-41);
 *   forced (F,L) int32 by location or NULL: teacher-forced codes (parity tests);
 *   uniforms (F,L) f32 by location or NULL: u in [0,1) for the inverse-CDF draw from
 *     softmax(logits / temperature) (sample.py:60-63; RNG streams differ from torch.multinomial);
 *   out_logits (F,L,512) f32 by location or NULL: the logits each location was decided from.
 * Exactly one of forced / uniforms must be given.
 *   first_step: order positions < first_step are not walked one by one: they must all be observed
 *   (not in the sample region) in every image, and are covered by one whole-grid pass
 *   (0 is always valid; the caller knows the orders, it built them on the host).
 * Asynchronous on the caller's stream (one launch per order position, enqueued eagerly; see
 * ps_pixelcnn_ar_run_waves for the schedule that needs far fewer dependent launches). */
int ps_pixelcnn_ar_run(ps_pixelcnn *h, int32_t *codes, const int32_t *order,
                       const uint8_t *sample_region, const float *mask_init,
                       const float *mask_undilated, const float *mask_dilated, const int32_t *forced,
                       const float *uniforms, float temperature, int F, int first_step,
                       float *out_logits, void *stream);

/* The same run, wavefront by wavefront: wave_cols (ncols,2) int32 on the DEVICE and wave_start (n_waves+1) int32 on
 * the HOST are the schedule of ps_ar_wavefronts for these orders and this first_step (ncols = F * (L - first_step)).
 * One launch per wave (waves of more than 128 columns are split) instead of one per order position -- 60-110
 * dependent launches instead of 400-700 for PixelSynth's orders -- and bit-identical codes and logits. */
int ps_pixelcnn_ar_run_waves(ps_pixelcnn *h, int32_t *codes, const int32_t *order,
                             const uint8_t *sample_region, const float *mask_init,
                             const float *mask_undilated, const float *mask_dilated,
                             const int32_t *forced, const float *uniforms, float temperature, int F,
                             int first_step, const int32_t *wave_cols, const int32_t *wave_start,
                             int n_waves, float *out_logits, void *stream);

/* The two halves of ps_pixelcnn_ar_run_waves as calls of their own, for callers that overlap them across batches (the column
 * launches of one batch are bound by the latency of their 33 dependent stages and leave about a third of the chip idle; the
 * whole-grid pass of the NEXT batch fills it when it runs beside them, on another handle and another stream):
 *   ps_pixelcnn_ar_prefix   frames [frame_begin, frame_end) of the F: their sampled codes are masked out and the whole-grid pass
 *                           over their observed prefix (order positions < first_step) fills the handle's activation caches.
 *                           Disjoint frame ranges are independent -- they may be issued on different streams;
 *   ps_pixelcnn_ar_columns  the column launches of all F frames (schedule as for ps_pixelcnn_ar_run_waves), after every frame's
 *                           prefix pass has completed (the caller orders the streams).
 * Together they do exactly what ps_pixelcnn_ar_run_waves(..., out_logits = NULL, ...) does.
 * ps_pixelcnn_ar_columns also takes a schedule that holds only PART of the run's columns (any (frame, position) pairs with position >=
 * first_step, wave by wave): the caller then answers for the dependencies -- every column a column reads (earlier positions of its
 * frame) was walked by an earlier wave or call.  Callers use it to run the narrow last wavefronts of one batch of frames inside the
 * launches of the next batch's first wavefronts (both batches resident in one handle: pixelsynth_amd/z_buffermodel.py,
 * outpaint_pipelined). */
int ps_pixelcnn_ar_prefix(ps_pixelcnn *h, int32_t *codes, const int32_t *order, const uint8_t *sample_region,
                          const float *mask_init, const float *mask_undilated, const float *mask_dilated, int F,
                          int first_step, int frame_begin, int frame_end, void *stream);
/* ps_pixelcnn_ar_prefix with PER-FRAME prefixes: frame f's whole-grid pass covers its order positions [0, first_steps[f]) -- the
 * observed locations in front of ITS first sampled one, not only those in front of the batch's (first_steps: device, (F) int32,
 * min_first_step <= first_steps[f] <= max_first_step; both bounds from the host).  The whole-grid pass evaluates a location for about
 * half of what a column costs and every form produces the same bits, so nothing changes but the time.  The column
 * schedule then holds frame f's positions from first_steps[f] on (ps_ar_wavefronts_frames) and goes to ps_pixelcnn_ar_columns with
 * first_step = min_first_step. */
int ps_pixelcnn_ar_prefix_frames(ps_pixelcnn *h, int32_t *codes, const int32_t *order, const uint8_t *sample_region,
                                 const float *mask_init, const float *mask_undilated, const float *mask_dilated, int F,
                                 const int32_t *first_steps, int min_first_step, int max_first_step, int frame_begin,
                                 int frame_end, void *stream);
int ps_pixelcnn_ar_columns(ps_pixelcnn *h, int32_t *codes, const int32_t *order, const uint8_t *sample_region,
                           const float *mask_init, const float *mask_undilated, const float *mask_dilated,
                           const int32_t *forced, const float *uniforms, float temperature, int F, int first_step,
                           const int32_t *wave_cols, const int32_t *wave_start, int n_waves, void *stream);
/* Compute units the stream of this handle's column launches can use (a multiple of 8; 0 = all of the device): a column launch
 * keeps one workgroup per compute unit resident, so a caller that confines the stream to part of the chip
 * (ps_stream_create_cu_range) says so here. */
int ps_pixelcnn_set_compute_units(ps_pixelcnn *h, int n_cus);
/* A stream whose kernels run on compute units [first_cu, first_cu + n_cus) only, in the numbering of hipExtStreamCreateWithCUMask
 * (contiguous ranges are spread evenly over the XCDs: measured on MI355X, tools/cumask_probe.hip).  Two such streams over
 * disjoint ranges share no compute unit.  The handle is a hipStream_t; destroy it with ps_stream_destroy. */
int ps_stream_create_cu_range(int first_cu, int n_cus, void **stream);
int ps_stream_destroy(void *stream);

/* One order position of the loop above for callers that draw the sample themselves (the drop-in
 * sample() keeps torch.multinomial): evaluates the column of location order[f][step] for every
 * image and writes its logits (F,512).  step == first_step additionally (re)builds the caches with
 * one whole-grid pass on the current codes (sample-region entries must be -1 at that point).  The
 * caller writes the chosen code into codes before the next call. */
int ps_pixelcnn_ar_step(ps_pixelcnn *h, const int32_t *codes, const int32_t *order,
                        const float *mask_init, const float *mask_undilated,
                        const float *mask_dilated, int F, int step, int first_step, float *logits,
                        void *stream);

/* Inside a column launch the per-frame chains consume results of other workgroups of the same launch; every such
 * wait is bounded, and one that runs out raises a flag in the handle instead of hanging the GPU.  This call
 * synchronises `stream` and returns PS_OK, or an error if any launch since the handle's creation hit that
 * limit (its results are then invalid).  Tests, smoke() and bench.py call it after their runs. */
int ps_pixelcnn_status(ps_pixelcnn *h, void *stream);

/* Hard z-buffer scatter of DepthManipulator.project_zbuffer (models/projection/depth_manipulator.py:66-104):
 * the reference sorts the points by z and assigns  out[b, 0|1, ys, xs] = v0|v1  for all of them at once; with the sequential
 * semantics of torch's CPU index_put_ the LAST point in sorted order that lands on a pixel stays.  ys / xs (B,N) int32: pixel of
 * the n-th point in sorted order (inside the image: the reference clamps first), v0 / v1 (B,N) f32 its values; out (B,2,H,W) f32
 * pre-filled by the caller (the reference's -2); winner (B,H,W) int32 workspace.  z-test = atomic max of the sorted position.
 * A pixel outside the image raises PS_STATUS_BAD_PIXEL in `status` and the point is dropped (the reference's indexed
 * assignment would raise). */
int ps_zbuffer_scatter_f32(const int32_t *ys, const int32_t *xs, const float *v0, const float *v1, int B, int N,
                           int H, int W, float *out, int32_t *winner, int32_t *status, void *stream);

/* The projection in front of it, DepthManipulator.project_zbuffer :43-66 and :86-87, one thread per source pixel:
 *   depth (B,1,W,W) f32, grid (4,W*W) f32 = the module's `grid` buffer (x, y, -1, 1 rows; :20-26), cameras (B,4,4)  ->
 *   zproj (B,N) f32 = third row of K (RT2 RT1inv) Kinv (grid * depth) (the sort key; the returned depth is -zproj),
 *   ys / xs (B,N) int32 = ((sampler + 1) * 128).long().clamp(0, 255) with the reference's literals, flag (B,N) f32 = 4 where
 *   the unclamped coordinate left [0, 255], else 0 -- all by ORIGINAL point position.  W must be 256 for the literals to
 *   mean what the reference means (the Python mirror refuses anything else). */
int ps_zbuffer_project_f32(const float *depth, const float *grid, const float *K, const float *Kinv, const float *RT1inv,
                           const float *RT2, int B, int W, float *zproj, int32_t *ys, int32_t *xs, float *flag, void *stream);

/* ps_zbuffer_scatter_f32 on the outputs of ps_zbuffer_project_f32 and the caller's argsort of zproj (descending, stable):
 * order (B,N) int64 = point at each sorted position; sorted position n writes  grid_x[order[n]] + flag[n]  and
 * -grid_y[order[n]] + flag[n]  at pixel (ys, xs)[order[n]] -- the flag by ORIGINAL position on values in SORTED order, as the
 * reference does (:88-97). */
int ps_zbuffer_scatter_sorted_f32(const int64_t *order, const int32_t *ys, const int32_t *xs, const float *grid,
                                  const float *flag, int B, int N, int H, int W, float *out, int32_t *winner,
                                  int32_t *status, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Vector quantisation around the AR loop (VQ-VAE-2 top level, models/vqvae2/vqvae.py) -- SURVEY 8f row 1
 * ------------------------------------------------------------------------------------------- */

/* Quantize.forward, inference part (models/vqvae2/vqvae.py:41-51): for every latent vector the index of the
 * nearest codebook column.  z: layout 0 = (N,D) row-major (the reference's `flatten`), layout 1 = (B,D,HW) as the
 * 1x1 conv in front of it leaves it (row n = b*HW + p; HW given, N = B*HW); embed (D,K) f32 (the reference's
 * `embed` buffer); idx (N) int32; mindist (N) f32 or NULL.  D <= 64.
 *   dist[n][k] = (|z_n|^2 - 2 z_n.e_k) + |e_k|^2, sums in ascending d (fused multiply-adds); ties -> smallest k. */
int ps_vq_nearest_f32(const float *z, int layout, const float *embed, int N, int D, int K, int HW,
                      int32_t *idx, float *mindist, void *stream);

/* Quantize.embed_code + permute(0,3,1,2) (models/vqvae2/vqvae.py:77-78,306-307): idx (B,HW) int32 -> out (B,D,HW)
 * f32, out[b][d][p] = embed[d][idx[b][p]] (zeros for an index outside [0,K)). */
int ps_vq_embed_f32(const int32_t *idx, const float *embed, int B, int HW, int D, int K, float *out,
                    void *stream);

/* ---- refinement decoder, elementwise passes of a ResNet_Block (models/layers/blocks.py:34-73), channels-last --------
 * All tensors (B, H, W, C) f32 contiguous (torch channels_last storage), C a multiple of 4.
 * ps_affine_relu_nhwc_f32: LinearNoiseLayer + stored-statistics batch norm + ReLU (models/layers/normalization.py:21-47,
 *   :170-184; blocks.py:41-47) as one pass, y = max(x * scale[b][c] - shift[b][c], 0); scale, shift (B, C). */
int ps_affine_relu_nhwc_f32(const float *x, const float *scale, const float *shift, int B, int HW, int C,
                            float *y, void *stream);
/* ps_pool_add_nhwc_f32: blocks.py:61-73 with 'Down': out (B, H/2, W/2, C) = avg_pool2d(a, 3, 2, 1) + avg_pool2d(b, 3, 2, 1)
 *   (zero padding counted in the divisor, torch's default); b may be NULL.  bias (C) or NULL: a per-channel constant still
 *   to be added to every INPUT pixel (the bias of the convolutions that produced a and b, which ran without it). */
int ps_pool_add_nhwc_f32(const float *a, const float *b, const float *bias, int B, int H, int W, int C, float *out,
                         void *stream);
/* ps_pool_add_post_nhwc_f32: ps_pool_add_nhwc_f32 plus `post` (B, H/2, W/2, C) or NULL, added AFTER the pooling: the projected
 * branch of a down-sampling block whose 1 x 1 convolution ran on the pooled input (avg_pool2d and a 1 x 1 convolution commute). */
int ps_pool_add_post_nhwc_f32(const float *a, const float *b, const float *bias, const float *post, int B, int H, int W, int C,
                              float *out, void *stream);
/* ps_upsample_add_nhwc_f32: blocks.py:61-73 with 'Up': out (B, 2H, 2W, C) = bilinear x2 (align_corners = False) of a,
 *   plus the same of b unless b is NULL; bias as above. */
int ps_upsample_add_nhwc_f32(const float *a, const float *b, const float *bias, int B, int H, int W, int C, float *out,
                             void *stream);
/* ps_add_bias_nhwc_f32: the residual sum of a block without resampling, out = a + b + bias[c] (bias may be NULL). */
int ps_add_bias_nhwc_f32(const float *a, const float *b, const float *bias, int B, int HW, int C, float *out,
                         void *stream);

/* ps_cat_mask_nhwc_f32: the refinement decoder's input (models/networks/architectures.py:153-156: torch.cat((x, ~background_mask), 1)):
 * x (B, 3, H, W) fp32 NCHW, background_mask (B, H, W) bytes (non-zero = background) -> out (B, H, W, 4) fp32 NHWC, channel 3 =
 * 1 - background. */
int ps_cat_mask_nhwc_f32(const float *x, const unsigned char *background_mask, int B, int H, int W, float *out, void *stream);

/* ps_noise_affine_f32: the per-(sample, channel) affine of LinearNoiseLayer + stored-statistics batch norm
 * (models/layers/normalization.py:21-47, :170-184): scale[b][c] = rsqrt(var[c] + eps) * (1 + <noise[b], Wg[c]>),
 * shift[b][c] = mean[c] * scale[b][c] - <noise[b], Wb[c]> - pend[c] * scale[b][c].  noise (B, K); Wg, Wb (C, K): the (spectral-
 * normalised) weights of the layer's gain / bias Linear; mean, var (C); pend (C) or NULL: a convolution bias still missing from
 * the tensor to be normalised.  scale, shift (B, C): what ps_affine_relu_nhwc_f32 and the fused convolutions take. */
int ps_noise_affine_f32(const float *noise, const float *wg, const float *wb, const float *mean, const float *var, const float *pend,
                        float eps, int B, int C, int K, float *scale, float *shift, void *stream);

/* ---- 3 x 3 convolutions of the refinement decoder on the fp16 matrix pipe (csrc/conv_f16x3.hip) ------------------------------
 * Replaces, for the decoder's wide layers, torch.nn.Conv2d(Ci, Co, 3, 1, 1) as ResNet_Block calls it (models/layers/blocks.py:34-73,
 * models/networks/architectures.py:126-167), WITHOUT the bias (the caller folds it into the next pass, as for the MIOpen path).
 * fp32 in, fp32 out; every product is three fp16 MFMAs on split operands (v = hi + lo), sums in fp32: see the unit's header.
 *
 * ps_conv3x3_f16x3_packed_bytes: size of the packed weights of a (Co, Ci, 3, 3) convolution.
 * ps_conv3x3_f16x3_pack: w (Co, 3, 3, Ci) fp32 contiguous -- torch's channels_last storage of a (Co, Ci, 3, 3) weight -- into
 *   `packed` (device, ps_conv3x3_f16x3_packed_bytes).  Co a multiple of 64 (128 is the kernel's block: 64 runs half empty), Ci a
 *   multiple of 32.
 * ps_conv3x3_f16x3_nhwc: y (B, H, W, Co) = conv3x3(act(x), w), zero padding 1, stride 1; x (B, H, W, Ci); H, W multiples of 16.
 *   scale, shift: (B, Ci) or both NULL: act(x) = max(x * scale[b][c] - shift[b][c], 0) -- the LinearNoiseLayer + ReLU in front of the
 *   convolution (models/layers/normalization.py:21-47), applied on the way in; NULL: act(x) = x.
 *   bias (Co) and res (B, H, W, Co), each may be NULL, are added to the results on the way out: y = conv + bias + res -- the sum a
 *   ResNet_Block forms of its two branches (blocks.py:61-73), without a pass of its own.
 *   overflow: device int the kernel sets to 1 when an activation lies beyond fp16's range (|v| > 65000) or is not a number;
 *   never cleared by the library. */
size_t ps_conv3x3_f16x3_packed_bytes(int Co, int Ci);
int ps_conv3x3_f16x3_pack(const float *w, int Co, int Ci, void *packed, void *stream);
int ps_conv3x3_f16x3_nhwc(const float *x, const float *scale, const float *shift, const void *packed, const float *bias,
                          const float *res, int B, int H, int W, int Ci, int Co, float *y, int *overflow, void *stream);
/* ps_conv3x3_f16x3_ex_nhwc: the same with a permutation folded into either end (the VQ-VAE's stride-2 layers, models/vqvae2/vqvae.py:107-161,
 *   run as 3 x 3 convolutions over 2 x 2 blocks):  in_s2d != 0: x is (B, 2 H, 2 W, Ci / 4) and is read as its space-to-depth form
 *   (B, H, W, Ci), channel (sy * 2 + sx) * Ci / 4 + c = pixel (2 y + sy, 2 x + sx) -- Ci / 4 a multiple of 32 (scale / shift stay (B, Ci));
 *   out_d2s != 0: y is (B, 2 H, 2 W, Co / 4), written as the depth-to-space form of the (B, H, W, Co) result, channel
 *   (py * 2 + px) * Co / 4 + c -> pixel (2 y + py, 2 x + px) -- Co / 4 a multiple of 64, res NULL.
 *   co_live (0 = Co): the caller's word that output channels [co_live, Co) carry all-zero weights (a 32-channel layer packed as 64): they
 *   are not multiplied (speed only: they come out as bias + res either way). */
int ps_conv3x3_f16x3_ex_nhwc(const float *x, const float *scale, const float *shift, const void *packed, const float *bias,
                             const float *res, int B, int H, int W, int Ci, int Co, int co_live, int in_s2d, int out_d2s, float *y,
                             int *overflow, void *stream);

/* ---- the decoder's two thin 3 x 3 convolutions (csrc/conv_thin.hip), fp32 FMAs, same contract as above (no bias; optional
 * act(x) = max(x * scale[b][c] - shift[b][c], 0) on the way in; zero padding 1, stride 1; NHWC):
 * ps_conv3x3_thin_in_nhwc_f32:  x (B, H, W, 4) -> y (B, H, W, Co), Co a multiple of 4; w [3][3][4][Co] (= weight.permute(2, 3, 1, 0)).
 * ps_conv3x3_thin_out_nhwc_f32: x (B, H, W, Ci) -> y (B, H, W, Co), Ci a multiple of 32, 1 <= Co <= 4; w [3][3][Ci][Co].
 * H a multiple of 8; W a multiple of 64 (thin_in; Co <= 256) / 32 (thin_out).  Replace torch.nn.Conv2d(4, 64, 3, 1, 1) / Conv2d(128, 3, 3, 1, 1) of the decoder's first and last
 * blocks (models/networks/architectures.py:126-167). */
int ps_conv3x3_thin_in_nhwc_f32(const float *x, const float *scale, const float *shift, const float *w, int B, int H, int W,
                                int Co, float *y, void *stream);
int ps_conv3x3_thin_out_nhwc_f32(const float *x, const float *scale, const float *shift, const float *w, int B, int H, int W,
                                 int Ci, int Co, float *y, void *stream);
/* ps_conv3x3_thin_in_f16x3_nhwc: the 4 -> 64 layer on the fp16 matrix pipe (split operands, three MFMAs per product, as
 * ps_conv3x3_f16x3_nhwc; `overflow` likewise).  Co = 64, W a multiple of 16; w [3][3][4][64]. */
int ps_conv3x3_thin_in_f16x3_nhwc(const float *x, const float *scale, const float *shift, const float *w, int B, int H, int W,
                                  int Co, float *y, int *overflow, void *stream);

/* ---- 1 x 1 convolutions (csrc/conv1x1.hip) on the fp32 matrix pipe: exact fp32 products, fp32 accumulation.
 * ps_conv1x1_nhwc_f32: y (npix, Co) = x (npix, Ci) . w^T, w (Co, Ci) contiguous -- torch.nn.Conv2d(Ci, Co, 1) WITHOUT its bias on
 *   channels-last activations, npix = B H W.  Ci in {4, 32, 64, 128, 256}, Co >= 1, ceil16(Co) * Ci <= 32768 (the weights stay in LDS);
 *   buffers 16-byte aligned.  Replaces the projection branch of a ResNet_Block (models/layers/blocks.py:46-57) and the ResBlock /
 *   quantize_conv_t projections of the VQ-VAE (models/vqvae2/vqvae.py:81-97, :262).
 * ps_conv1x1_ex_nhwc_f32: the same with the passes around it folded in: x rows `ldx` floats apart (>= Ci, a multiple of 4: the first Ci
 *   channels of a wider activation); flags bit 0: max(x, 0) on the way in; bias (Co) and res (npix, Co), each may be NULL, added on the
 *   way out -- max(res, 0) instead of res with flags bit 1: y = act(x) . w^T + bias + res.  The tail of the VQ-VAE's ResBlock
 *   (vqvae.py:81-97: ReLU, 1 x 1, + the in-place-ReLU'd input) in one launch.
 * ps_conv1x1_takes: 1 when these take a (Co, Ci) weight, else 0. */
int ps_conv1x1_takes(int Ci, int Co);
int ps_conv1x1_nhwc_f32(const float *x, const float *w, size_t npix, int Ci, int Co, float *y, void *stream);
int ps_conv1x1_ex_nhwc_f32(const float *x, int ldx, const float *w, const float *bias, const float *res, int flags, size_t npix, int Ci,
                           int Co, float *y, void *stream);

/* ---- the 3-channel ends of the VQ-VAE (csrc/vq_ends.hip; models/vqvae2/vqvae.py:107, :150), fp32 matrix pipe, exact fp32 products.
 * ps_vq_stem_s2d_f32: Conv2d(3, 64, 4, stride 2, padding 1) of an image x (B, 3, H, W) NCHW, bias included, written in the
 *   space-to-depth form the next layer reads: y (B, H / 4, W / 4, 256), channel (sy * 2 + sx) * 64 + co = output pixel (2 y + sy, 2 x + sx).
 *   w (64, 3, 4, 4), bias (64); H a multiple of 4, W of 16.
 * ps_vq_head_f32: image y (B, 3, 2 Hh, 2 Wh) NCHW = ConvTranspose2d(64, 3, 4, stride 2, padding 1)(relu(h)) + bias, h (B, Hh, Wh, 64)
 *   channels-last; wt (64, 3, 4, 4) (torch's (in, out, kh, kw)), bias (3); Wh a multiple of 16. */
int ps_vq_stem_s2d_f32(const float *x, const float *w, const float *bias, int B, int H, int W, float *y, void *stream);
int ps_vq_head_f32(const float *h, const float *wt, const float *bias, int B, int Hh, int Wh, float *y, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PIXELSYNTH_HIP_H */
