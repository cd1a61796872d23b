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

#define PS_ABI_VERSION 1

enum { PS_OK = 0, PS_ERR_ARG = -1, PS_ERR_HIP = -2, PS_ERR_WORKSPACE = -3, PS_ERR_STATE = -4 };
enum { PS_ACC_ALPHACOMPOSITE = 0, PS_ACC_WSUM = 1, PS_ACC_WSUMNORM = 2 }; /* opts.accumulation */

int ps_abi_version(void);
const char *ps_last_error(void);

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
 *   workspace: >= ps_splat_workspace_bytes(B,N,S,radius_px) bytes of device memory. */
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

/* ------------------------------------------------------------------------------------------
 * Locally masked convolution / PixelCNN (models/lmconv)
 * ---------------------------------------------------------------------------------------- */

/* _locally_masked_conv2d.forward  (models/lmconv/locally_masked_convolution.py:11-50), 3x3:
 *   x (B,Ci,H,W) f32 ; mask (B,9,H*W) f32 (the reference passes it repeated Ci times, (B*Ci,9,L),
 *   identical across channels -- the caller hands over one copy per image; mask_batch_stride = 0
 *   broadcasts one mask to the whole batch) ; weight (Co,Ci,3,3) ; bias (Co) or NULL
 *   -> y (B,Co,H,W).  padding = dilation (:118-120). */
int ps_lmconv_forward_f32(const float *x, const float *mask, size_t mask_batch_stride,
                          const float *weight, const float *bias, int B, int Ci, int Co, int H,
                          int W, int dilation, float *y, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PIXELSYNTH_HIP_H */
