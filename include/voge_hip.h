/*
 * voge_hip.h -- C ABI of libvoge_hip.so, the MI355X (gfx950) implementation of the VoGE
 * ray-trace + aggregation hot path.
 *
 * This is the drop-in boundary: it replaces the on-path part of the reference's pybind11
 * module `VoGE._C` (VoGE/csrc/ext.cpp:7-17) and the tensor programs of
 * VoGE/Aggregation.py that sit directly behind it.  Plain C: raw DEVICE pointers, sizes,
 * and a HIP stream handle -- no torch / ATen types.
 *
 * Conventions (all functions):
 *   - every pointer is a device pointer on the current HIP device, caller-allocated and
 *     caller-owned; the library never allocates, frees or keeps state between calls;
 *   - float = IEEE fp32, idx = int32, valid_num = int64 (Aggregation.py:104 returns int64);
 *   - tensors are contiguous, row-major, layouts as in the reference:
 *       mus [P,3], isigmas [P,3,3], rays [B,H,W,3], per-slot arrays [B,H,W,K]
 *       (ray index = b*H*W + y*W + x, ray_trace_voge.cu:176);
 *   - work is enqueued asynchronously on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream), no host synchronisation, re-entrant across streams;
 *   - return value: 0 on success, a positive hipError_t value if the HIP runtime reported an
 *     error (the reference raises via AT_CUDA_CHECK, ray_trace_voge.cu:278,:377), or a
 *     negative VOGE_ERR_* code for argument errors.  Never throws.
 */
#ifndef VOGE_HIP_H
#define VOGE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VOGE_ABI_VERSION 7

#define VOGE_ERR_BAD_ARG (-1)        /* null pointer / non-positive size */
#define VOGE_ERR_WORKSPACE (-2)      /* workspace smaller than one view's voge_trace_workspace_bytes(1, ...) */
#define VOGE_ERR_K_TOO_LARGE (-3)    /* K above VOGE_MAX_K (top-K lists live in LDS) */

#define VOGE_MAX_K 256

typedef void *voge_stream_t; /* hipStream_t */

/* ABI version of the loaded library (== VOGE_ABI_VERSION it was built with). */
int voge_abi_version(void);

/* Human-readable text for a return code of any function below. */
const char *voge_error_string(int code);

/*
 * Bytes of scratch the forward trace needs for B batch elements of N Gaussians and an HxW
 * image: per-Gaussian derived records (cull sphere + quadratic-form coefficients, 112 B each),
 * the binning's segments (per 32x32-px super-tile and Gaussian slice) with their extension arenas,
 * the per-tile (8x8 px) depth-ordered candidate lists, and a pool of list entries (32 per Gaussian,
 * at least 2^20) for image quads with more candidates than the in-LDS sort takes.  165 MB per view at
 * 50k Gaussians / 512^2, 774 MB at 200k / 1024^2.  Views are independent: the entry points walk a
 * batch in CHUNKS of as many views as the scratch holds (stream-ordered launches on the same
 * buffers), and this function asks for the largest chunk under 1 GiB -- never less than one
 * view, never more than the batch.  Any size >= one view's (voge_trace_workspace_bytes(1, ...))
 * is accepted and used in full; pass more to keep a big batch in one chunk.  Caller allocates,
 * 256-byte aligned (any torch allocation is); the contents need no initialisation and nothing
 * in it outlives the call except what voge_trace_pool_usage reads (every chunk's pool counter, in the scratch's last 512 bytes).
 */
size_t voge_trace_workspace_bytes(int B, int N, int H, int W);

/*
 * Diagnostic: how much of the workspace's candidate-list POOL the last forward trace that ran on `workspace` (same
 * B, N, H, W) used.  The pool holds the lists of image quads (16x16 px) with more candidates than the in-LDS sort
 * takes (a small object behind a few dozen pixels); *used > *capacity means it ran out and those quads' tiles fell
 * back to streaming every Gaussian (slow, still exact).  workspace_bytes = what the trace was given: the chunks it walked
 * follow from it, and *used is the LARGEST use of any chunk (ABI 7; ABI 6 derived the layout from the default size and saw
 * the last chunk only).  Synchronous (copies 512 bytes from the device).  No reference
 * counterpart: the reference's coarse stage drops points when a bin overflows (rasterize_coarse.cu).
 */
int voge_trace_pool_usage(const void *workspace, size_t workspace_bytes, int B, int N, int H, int W, int *used,
                          int *capacity);

/*
 * (Not part of this ABI: `voge_debug_sweep_variant(int)` exists in -DVOGE_AB builds only -- voge_amd/libvoge_hip_ab.so, a
 * test artefact that also carries round 3's scalar-sigma sweep for the bit-for-bit comparison in
 * tests/test_gpu_configs.py::test_rebuilt_sweep_equals_round_3_sweep_bit_for_bit.  The product library is stateless.)
 */

/*
 * Fine ray trace forward, "all Gaussians are candidates" form.
 * Replaces: VoGE._C.ray_trace_voge_fine (ray_trace_voge.h:7-15, ray_trace_voge.cu:219-280)
 * called with the bin list RayTracing.py:22-26 builds for max_points_per_bin == -1
 * (every pixel of batch b sees Gaussians b*N .. b*N+N-1).  That P-long list is never
 * materialised here.
 *
 * For each pixel ray d and Gaussian (mu, A): dsd = d^T A d, len = mu^T A d / dsd,
 * act = mu^T A mu - (mu^T A d)^2 / dsd; keep if act < thr_act (and len < 1e10, the
 * sentinel); output the K kept candidates with the smallest (len, index), ascending;
 * unused slots hold idx=-1, len=1e10, act=1e10, dsd=0 (ray_trace_voge.cu:184-214,:244-247).
 * idx holds the global index b*N+i.  Outputs need no pre-fill.  cnt: NULL, or [B,H,W] int32
 * receiving the number of hits kept per pixel (the filled prefix of the K slots).
 *
 * cam_fwd: NULL, or [B,3] unit view axis in the rays' frame: Gaussians with mu.fwd < 0
 * are skipped, which is the candidate rule of the reference's coarse stage
 * (rasterize_coarse.cu:35, "skip z<0") used when max_points_per_bin != -1.
 * cones: NULL, or the bounding-cone hierarchy of the rays (per 32x32-pixel super-tile: its own cone, its four
 * 16x16 quads', its sixteen 8x8 tiles') that voge_rays_fwd / voge_ray_cones produced for exactly this
 * `rays` tensor (voge_cones_floats(B,H,W) floats, an opaque blob); they only steer the conservative
 * candidate culling.  NULL costs one more launch that derives them.
 */
int voge_trace_topk_fwd(const float *mus, const float *isigmas, const float *rays,
                        const float *cam_fwd, const float *cones, int B, int N, int H, int W, int K,
                        float thr_act, void *workspace, size_t workspace_bytes,
                        int32_t *idx, float *len, float *act, float *dsd, int32_t *cnt,
                        voge_stream_t stream);

/*
 * Fine ray trace forward with caller-supplied candidate lists.
 * Replaces: VoGE._C.ray_trace_voge_fine for an explicit bin_points tensor
 * [B,BH,BW,M] int32, -1 = empty (ray_trace_voge.cu:135-217): pixel (y,x) of batch b
 * scans bin (y/bin_size, x/bin_size).  P = total Gaussians (list entries index [0,P)).
 * Same outputs as above (cnt: NULL or [B,H,W] int32, the hits kept per pixel).  Exact ties in
 * len are ordered by index (the reference orders them by list position, which is not
 * deterministic for coarse-rasterised lists).
 */
int voge_trace_topk_list_fwd(const float *mus, const float *isigmas, const float *rays,
                             const int32_t *bin_points, int B, int P, int H, int W, int K,
                             int BH, int BW, int M, int bin_size, float thr_act,
                             int32_t *idx, float *len, float *act, float *dsd, int32_t *cnt,
                             voge_stream_t stream);

/*
 * Fine ray trace backward.
 * Replaces: VoGE._C.ray_trace_voge_fine_backward (ray_trace_voge.h:17-25,
 * ray_trace_voge.cu:283-379).  The pixel grid is given as nrows = B*H rows of W pixels (the
 * kernel works on 16x16 pixel tiles of that grid).  For every slot with idx >= 0 applies the
 * chain rule of ray_trace_voge.cu:324-326 and scatters into
 *   g_ray [nrows*W,3], g_mus [P,3], g_isg [P,3,3]  (raw outer products, not symmetrised).
 * g_mus / g_isg are fully written by this call (the reference allocates zeros, :354-356);
 * g_ray is fully written, or may be NULL when the ray gradient is not needed.
 * cnt: NULL, or [nrows*W] int32 = number of leading slots of each pixel that may hold a hit
 * (the forward's out_cnt): slots k >= cnt[pixel] are not even loaded.
 * workspace: >= voge_trace_bwd_workspace_bytes(P) bytes, 256-byte aligned.
 */
int voge_trace_bwd(const float *mus, const float *isigmas, const float *rays,
                   const int32_t *idx, const int32_t *cnt, const float *g_len, const float *g_act,
                   const float *g_dsd, int P, long nrows, int W, int K, void *workspace,
                   size_t workspace_bytes, float *g_ray, float *g_mus, float *g_isg,
                   voge_stream_t stream);

/* Scratch bytes voge_trace_bwd needs for P Gaussians (packed records + padded accumulator). */
size_t voge_trace_bwd_workspace_bytes(int P);

/*
 * Isotropic variants of the two calls above: every Gaussian is A = a I with ONE scalar a [P]
 * (the reference's (N,) sigma form: expend_sigma, Aggregation.py:155-157, then 2*sigma,
 * Renderer.py:133).  Same outputs as voge_trace_topk_fwd; the backward returns g_a [P], the
 * gradient of that scalar (= trace of the general call's g_isg), and g_mus [P,3].  Four sums per
 * Gaussian instead of twelve, and the caller's scalar -> 3x3 expansion (and its backward)
 * disappears.  workspace of the backward: >= voge_trace_bwd_iso_workspace_bytes(P).
 */
int voge_trace_topk_fwd_iso(const float *mus, const float *a, const float *rays,
                            const float *cam_fwd, const float *cones, int B, int N, int H, int W, int K,
                            float thr_act, void *workspace, size_t workspace_bytes,
                            int32_t *idx, float *len, float *act, float *dsd, int32_t *cnt,
                            voge_stream_t stream);
int voge_trace_bwd_iso(const float *mus, const float *a, const float *rays, const int32_t *idx,
                       const int32_t *cnt, const float *g_len, const float *g_act, const float *g_dsd,
                       int P, long nrows, int W, int K, void *workspace, size_t workspace_bytes,
                       float *g_ray, float *g_mus, float *g_a, voge_stream_t stream);
size_t voge_trace_bwd_iso_workspace_bytes(int P);

/*
 * The same two calls with the renderer's elementwise preamble folded in (VoGE/Renderer.py:130-137:
 * `verts - origin` per view and `2 * sigmas` / `2 / sigmas` (inverse_sigma)): the per-Gaussian pass that
 * reads the inputs anyway applies them, and the backward's per-Gaussian pass applies their chain rule,
 * so the caller launches no elementwise kernels around the trace.
 *   verts   [N,3] (shared != 0: one set seen by all B views) or [B*N,3];  sigmas [N] or [B*N]
 *   origin  [B,3] camera centres, or NULL (no centring);  sigma_mode 0: a = sigma, 1: a = 2 sigma, 2: a = 2/sigma
 * Forward outputs as voge_trace_topk_fwd (indices are b*N + n).  Backward: g_verts / g_sigmas have the
 * shapes of verts / sigmas (summed over the views when shared); no gradient is produced for origin --
 * a caller that needs it (camera pose optimisation) uses the plain calls.  Workspaces as above with P = B*N.
 */
int voge_trace_topk_fwd_iso_view(const float *verts, const float *sigmas, const float *origin, int shared,
                                 int sigma_mode, const float *rays, const float *cam_fwd, const float *cones,
                                 int B, int N, int H, int W, int K, float thr_act, void *workspace,
                                 size_t workspace_bytes, int32_t *idx, float *len, float *act, float *dsd,
                                 int32_t *cnt, voge_stream_t stream);
int voge_trace_bwd_iso_view(const float *verts, const float *sigmas, const float *origin, int shared,
                            int sigma_mode, const float *rays, const int32_t *idx, const int32_t *cnt,
                            const float *g_len, const float *g_act, const float *g_dsd, int B, int N, long nrows,
                            int W, int K, void *workspace, size_t workspace_bytes, float *g_ray,
                            float *g_verts, float *g_sigmas, voge_stream_t stream);

/*
 * The reference's coarse stage, for callers that want ITS candidate lists (the default render path does
 * not need them: see voge_trace_topk_fwd).  Replaces: VoGE._C.rasterize_points_coarse
 * (rasterize_coarse.h:18-25; rasterize_coarse.cu:20-42,44-188,254-305): points [P,3] f32 = (x_ndc, y_ndc,
 * view z), radius [P,2] f32 (half extents in NDC units), cloud_to_packed_first_idx / num_points_per_cloud
 * [B] int64 -> bin_elems [B,BH,BW,M] int32, -1 padded (BH = 1 + (H-1)/bin_size, BW likewise; both < 66).
 * A point is listed in every bin its bbox overlaps (half-pixel pad), unless z < 0; points are taken in
 * chunks of 512 and a chunk that no longer fits a bin's M slots is dropped.  Unlike the reference the
 * chunks are taken in ascending order: lists are ascending in index and deterministic.
 */
int voge_bin_gaussians(const float *points, const int64_t *cloud_to_packed_first_idx,
                       const int64_t *num_points_per_cloud, int B, int P, int H, int W, const float *radius,
                       int bin_size, int max_points_per_bin, int32_t *bin_elems, voge_stream_t stream);

/*
 * Fine trace + composite in ONE call: what GaussianRenderer.forward asks for (VoGE/Renderer.py:139-150:
 * ray_tracing, then aggregation -> Fragments).  Replaces the pair
 *   VoGE._C.ray_trace_voge_fine (ray_trace_voge.cu:135-280)  +  `aggregation` (VoGE/Aggregation.py:82-107)
 * for the all-candidates list.  Arguments as voge_trace_topk_fwd / _iso / _iso_view, plus occ (the
 * renderer's absorptivity) and the fragment outputs weight [B,H,W,K] f32 and valid_num [B,H,W] i64.
 * records (iso forms): NULL, or [B*N,4] floats receiving the per-Gaussian (centred mean, a) pairs the trace
 * derives anyway -- the fused backward (voge_fragment_shade_bwd_iso) reads them instead of packing its own.
 * idx / len / weight / valid_num are the fragments (sentinels -1 / 1e10 / 0 in empty slots); act, dsd
 * and cnt [B,H,W] (required here) are what the backward needs -- in pixels WITHOUT any hit act / dsd are
 * may be left unwritten (nothing reads them: every consumer goes by cnt).  Results are bit-identical
 * to calling the two entry points one after the other (which is what happens on the device: compositing
 * inside the sweep's epilogue was measured slower, see trace_fwd.hip); the call saves the host side one
 * autograd node and a set of allocations per frame.
 */
/* Scalar-sigma forms (_iso, _iso_view) only: act and dsd may BOTH be NULL.  The sweep then writes index and len alone
 * (no per-slot gather in its epilogue), the composite derives act / dsd from the (mean, a) records with the same
 * operations (voge_composite_fwd_iso), and so does voge_fragment_shade_bwd_iso when handed NULL for them: 168 MB per
 * frame at 50k Gaussians / 512^2 / K = 40 that are neither written nor read.  Weights are bit-identical either way.
 * voge_fragment_act_dsd_iso materialises them afterwards for consumers that want the arrays.
 * TRACE ONLY (scalar-sigma forms): weight = valid_num = act = dsd = NULL with records given -- the call stops behind the
 * sweep (idx, len, cnt, records are written) and the caller composites when it knows what it wants: voge_composite_fwd_iso
 * for the weights alone, or voge_composite_shade_fwd_iso for the weights AND the image of to_colored_background in one
 * pass (GaussianRenderer returns its Fragments before the colours are known: the Python side defers the composite
 * until the fragments' weights are first asked for, voge_amd/Renderer.py). */
/* TRACE ONLY for the general forms (mus [P,3], isigmas [P,3,3]): idx, len, cnt and `records` -- the packed (mu, A),
 * [B*N][12] floats -- are written; voge_composite_fwd_rec / voge_composite_shade_fwd_rec composite later from them, and
 * voge_fragment_shade_bwd / voge_fragment_bwd / voge_fragment_merge_bwd take act = dsd = NULL for such fragments.
 * Replaces the same reference code as voge_fragments_fwd (RayTracing.py:17-59 + ray_trace_voge.cu:219-280), deferred. */
int voge_trace_lean_fwd(const float *mus, const float *isigmas, const float *rays, const float *cam_fwd,
                        const float *cones, int B, int N, int H, int W, int K, float thr_act, void *workspace,
                        size_t workspace_bytes, int32_t *idx, float *len, int32_t *cnt, float *records,
                        voge_stream_t stream);
int voge_fragments_fwd(const float *mus, const float *isigmas, const float *rays, const float *cam_fwd,
                       const float *cones, int B, int N, int H, int W, int K, float thr_act, float occ,
                       void *workspace, size_t workspace_bytes, int32_t *idx, float *len, float *act, float *dsd,
                       int32_t *cnt, float *weight, int64_t *valid_num, voge_stream_t stream);
int voge_fragments_fwd_iso(const float *mus, const float *a, const float *rays, const float *cam_fwd,
                           const float *cones, int B, int N, int H, int W, int K, float thr_act, float occ,
                           void *workspace, size_t workspace_bytes, int32_t *idx, float *len, float *act,
                           float *dsd, int32_t *cnt, float *weight, int64_t *valid_num, float *records,
                           voge_stream_t stream);
int voge_fragments_fwd_iso_view(const float *verts, const float *sigmas, const float *origin, int shared,
                                int sigma_mode, const float *rays, const float *cam_fwd, const float *cones,
                                int B, int N, int H, int W, int K, float thr_act, float occ, void *workspace,
                                size_t workspace_bytes, int32_t *idx, float *len, float *act, float *dsd,
                                int32_t *cnt, float *weight, int64_t *valid_num, float *records, voge_stream_t stream);

/*
 * Fused backward of the fragment pipeline for isotropic Gaussians: shade (merge_final + get_silhouette +
 * to_colored_background, VoGE/Aggregation.py:111-141, VoGE/Renderer.py:157-171) -> aggregation
 * (VoGE/Aggregation.py:30-107) -> fine trace (ray_trace_voge.cu:283-332), in ONE pass over the fragments:
 * what voge_shade_bwd, voge_composite_bwd and voge_trace_bwd_iso(_view) compute one after the other, without
 * their exchanges through memory (g_weight; g_len / g_act / g_dsd) and with one accumulation table per wave.
 * records [B*N,4] = the (centred mean, a) pairs the forward kept (the `records` argument of
 * voge_fragments_fwd_iso / _iso_view); sigmas / shared / sigma_mode as in voge_trace_bwd_iso_view (pass the a
 * array, 0, 0 for plain (mus, a) inputs); rgb / wsum = voge_shade_fwd's out_rgb / out_wsum; g_img = the gradient of
 * the image, element (pixel p, channel c) at g_img[p * g_stride_pix + c * g_stride_c]: (C, 1) for a contiguous
 * [nrows*W,C] array, (0, 0) for the one broadcast scalar autograd hands back for sum() / mean() losses (no
 * materialised copy of it is needed).  Writes g_verts, g_sigmas (both or neither) and g_colors [Nattr,C].  K <= 128;
 * C <= 4; cnt required.  workspace: >= voge_fragment_bwd_workspace_bytes(B*N) bytes.
 */
size_t voge_fragment_bwd_workspace_bytes(int P);
/* act / dsd [npix,K] of scalar-sigma fragments that were traced without them (see voge_fragments_fwd_iso): records
 * [P,4] = the (centred mean, a) pairs of that call, idx / len / cnt its outputs.  Sentinels (1e10, 0) in empty slots. */
int voge_fragment_act_dsd_iso(const float *records, const float *rays, const int32_t *idx, const float *len,
                              const int32_t *cnt, long npix, int K, int P, float *act, float *dsd, voge_stream_t stream);
/* Composite forward from (idx, len) and the records instead of (act, len, dsd): what voge_fragments_fwd_iso* runs behind
 * its sweep when act / dsd are omitted.  cnt required. */
int voge_composite_fwd_iso(const int32_t *idx, const int32_t *cnt, const float *len, const float *records,
                           const float *rays, float occ, long npix, int K, float *weight, int64_t *valid_num,
                           voge_stream_t stream);
/* Composite forward (as voge_composite_fwd_iso) with the SHADE stage in the same pass: merge_final + get_silhouette +
 * to_colored_background (VoGE/Aggregation.py:111-141, VoGE/Renderer.py:157-171) -- what voge_composite_fwd_iso followed
 * by voge_shade_fwd computes, without reading idx / weight back (8 bytes per slot) and without the second launch.
 * colors [Nattr,C], C = 3 | 4; bg [C]; thr as voge_shade_fwd.  Writes weight [npix,K], valid_num [npix], rgb / img [npix,C],
 * wsum [npix], and rewrites the empty slots of idx in place (-1 -> 0, Aggregation.py:131).  Any K <= VOGE_MAX_K; cnt required.
 * img = NULL (bg unused): merge_final and the weight sum only -- interpolate_attr / get_silhouette, no background. */
int voge_composite_shade_fwd_iso(int32_t *idx, const int32_t *cnt, const float *len, const float *records,
                                 const float *rays, float occ, const float *colors, const float *bg, float thr,
                                 long npix, int K, int C, long Nattr, float *weight, int64_t *valid_num, float *rgb,
                                 float *img, float *wsum, voge_stream_t stream);
/* The two composite entry points above for the GENERAL path: records = the packed (mu, A) [B*N][12] of voge_trace_lean_fwd
 * (act / dsd re-derived with make_eval + pair_eval, the operations of the sweep's own epilogue: bit-identical weights).
 * act / dsd [npix,K] (both or NULL): written for the live slots when given -- the fused backward reads 8 bytes per slot
 * rather than gathering 48 and evaluating again. */
int voge_composite_fwd_rec(int32_t *idx, const int32_t *cnt, const float *len, const float *records, const float *rays,
                           float occ, long npix, int K, float *weight, int64_t *valid_num, float *act, float *dsd,
                           voge_stream_t stream);
int voge_composite_shade_fwd_rec(int32_t *idx, const int32_t *cnt, const float *len, const float *records,
                                 const float *rays, float occ, const float *colors, const float *bg, float thr,
                                 long npix, int K, int C, long Nattr, float *weight, int64_t *valid_num, float *rgb,
                                 float *img, float *wsum, float *act, float *dsd, voge_stream_t stream);
/* The same for full 3x3 forms (mus [P,3], isigmas [P,3,3] as given to voge_fragments_fwd; P = B*N): writes g_mus
 * [P,3], g_isigmas [P,3,3] (the raw, unsymmetrised outer-product sums of ray_trace_voge.cu:324-326, as voge_trace_bwd)
 * -- both or neither -- and g_colors [Nattr,C].  Same constraints and workspace as the isotropic form. */
int voge_fragment_shade_bwd(const float *mus, const float *isigmas, const float *rays, const float *colors,
                            const int32_t *idx, const int32_t *cnt, const float *weight, const float *act,
                            const float *len, const float *dsd, const float *rgb, const float *wsum, const float *bg,
                            float thr, const float *g_img, long g_stride_pix, long g_stride_c, float occ, int P,
                            long nrows, int W, int K, int C, long Nattr, void *workspace, size_t workspace_bytes,
                            float *g_mus, float *g_isigmas, float *g_colors, voge_stream_t stream);
int voge_fragment_shade_bwd_iso(const float *records, const float *sigmas, int shared, int sigma_mode,
                                const float *rays, const float *colors, const int32_t *idx, const int32_t *cnt,
                                const float *weight, const float *act, const float *len, const float *dsd,
                                const float *rgb, const float *wsum, const float *bg, float thr,
                                const float *g_img, long g_stride_pix, long g_stride_c, float occ, int B, int N,
                                long nrows, int W, int K, int C, long Nattr, void *workspace, size_t workspace_bytes, float *g_verts,
                                float *g_sigmas, float *g_colors, voge_stream_t stream);

/*
 * ABI 7 -- ONE FRAME of the renderer as six launches (binA, binB, sweep | composite | fused backward, finish): the camera goes
 * in, no ray-generation launch and no fill launch run.
 *
 * voge_frame_trace_fwd_iso   GaussianRenderer.forward (VoGE/Renderer.py:102-150) up to and including ray_tracing, for scalar
 *   sigmas and one Gaussian set per view (`shared` != 0: [N,3] / [N] seen by all B views) or per batch element ([B,N,..]):
 *   the ray bundle of Renderer.py:124-128, the centring of :130, the sigma rule of :133-137 (sigma_mode 0: a = sigma,
 *   1: a = 2 sigma, 2: a = 2 / sigma), RayTracing.py:12-30 and ray_trace_voge.cu:135-217 -- as binA + binB + sweep, with
 *   NO ray-generation launch: every kernel derives the rays / bounding cones / camera centre / view axis it needs from
 *   R [B,3,3], T [B,3], focal [B,2], pp [B,2] with voge_rays_fwd's own operations (bit-identical rays; the cones are the
 *   analytic corner-ray cones of csrc/voge_common.h).  The band rendered is h stacked rows of W pixels; stacked row i is image
 *   row row0 + (i / stripe_h) * pitch + i % stripe_h (one contiguous band: stripe_h >= h, pitch = 0).  behind != 0: Gaussians
 *   behind the camera plane are no candidates (rasterize_coarse.cu:35; what cam_fwd = R[:, :, 2] selects in the other entries).
 *   Writes idx, len [B,h,W,K], cnt [B,h,W], records [B*N,4] (centred mean, a), rays [B,h,W,3] (by the sweep) and origin [B,3]
 *   (NULL: not wanted).
 *   workspace: voge_trace_workspace_bytes(B, N, h, W).  No act / dsd (the fragments' consumers re-derive them).
 * voge_frame_shade_fwd_iso   = voge_composite_shade_fwd_iso (aggregation + merge_final + get_silhouette +
 *   to_colored_background, Aggregation.py:82-141, Renderer.py:157-171; img = bg = NULL: interpolate_attr + the weight sum)
 *   that ALSO zeroes bwd_acc [bwd_acc_bytes, a multiple of 16, 16-byte aligned; NULL: nothing] on its way: the accumulator
 *   of the backward below (voge_frame_bwd_acc_bytes(B * N)), so that no fill launch stands in front of it -- and writes
 *   sil [npix] = min(sum_k w_k, 1) (get_silhouette, Renderer.py:157-159; NULL: not wanted) with the sum it has in hand: the
 *   reference's training pattern (interpolate_attr + get_silhouette, demo/ShapeFitting.py:217-222) is this ONE launch, and its
 *   backward ONE call: voge_frame_merge_bwd_iso with wsum_fwd = the forward's wsum takes g_wsum as the SILHOUETTE's gradient
 *   (passed on like torch.minimum's: all below 1, half at a tie, none above; wsum_fwd = NULL: g_wsum is the sum's own gradient).
 * voge_frame_shade_bwd_iso / voge_frame_merge_bwd_iso   = voge_fragment_shade_bwd_iso / voge_fragment_merge_bwd_iso for
 *   fragments that keep no act / dsd, taking that accumulator -- ZEROED, good for one call -- in place of a scratch they would
 *   have to fill first: the fused kernel + the finishing pass.  (Adding the per-Gaussian sums straight into the gradient
 *   arrays was built and measured: one launch less, 30 us slower -- three atomic requests per table entry instead of one.)
 */
size_t voge_frame_bwd_acc_bytes(int P);
int voge_frame_trace_fwd_iso(const float *verts, const float *sigmas, int shared, int sigma_mode, const float *R,
                             const float *T, const float *focal, const float *pp, int row0, int stripe_h, int pitch,
                             int behind, int B, int N, int h, int W, int K, float thr_act, void *workspace,
                             size_t workspace_bytes, int32_t *idx, float *len, int32_t *cnt, float *records,
                             float *rays, float *origin, voge_stream_t stream);
int voge_frame_shade_fwd_iso(int32_t *idx, const int32_t *cnt, const float *len, const float *records,
                             const float *rays, float occ, const float *colors, const float *bg, float thr,
                             long npix, int K, int C, long Nattr, float *weight, int64_t *valid_num,
                             float *rgb, float *img, float *wsum, float *sil, void *bwd_acc, size_t bwd_acc_bytes,
                             voge_stream_t stream);
int voge_frame_shade_bwd_iso(const float *records, const float *sigmas, int shared, int sigma_mode,
                             const float *rays, const float *colors, const int32_t *idx, const int32_t *cnt,
                             const float *weight, const float *len, const float *rgb, const float *wsum,
                             const float *bg, float thr, const float *g_img, long g_stride_pix, long g_stride_c,
                             float occ, int B, int N, long nrows, int W, int K, int C, long Nattr, void *acc_zeroed,
                             size_t acc_bytes, float *g_verts, float *g_sigmas, float *g_colors, voge_stream_t stream);
int voge_frame_merge_bwd_iso(const float *records, const float *sigmas, int shared, int sigma_mode,
                             const float *rays, const float *attr, const int32_t *idx, const int32_t *cnt,
                             const float *weight, const float *len, const float *g_rgb, long g_stride_pix,
                             long g_stride_c, const float *g_wsum, const float *wsum_fwd, float occ, int B, int N, long nrows,
                             int W, int K, int C, long Nattr, void *acc_zeroed, size_t acc_bytes, float *g_verts,
                             float *g_sigmas, float *g_attr, voge_stream_t stream);

/*
 * ABI 7, the frame path for (N,3) / (N,3,3) sigmas (Renderer.py:130-137 with Aggregation.py:144-175: A = 2 expend_sigma(sigmas);
 * no inverse_sigma).  voge_frame_trace_fwd_gen: voge_frame_trace_fwd_iso's contract with the USER's arrays as inputs -- verts
 * [N | B*N][3] (shared_verts: one set for all views), sigmas [N | B*N][3] (kind 1: per-axis) or [N | B*N][3][3] (kind 2) -- the record pass
 * centres and expands them (the preamble's own fp32 operations), so neither a ray launch nor a preamble launch stands in front
 * of the frame; records = kind 2: the packed (centred mu, A) [B*N][12]; kind 1: the compact (centred mu, a0, a1, a2, 0, 0) [B*N][8], which
 * the composite and the backward evaluate with the three coefficients alone (the general chain's bits on such a form).
 * voge_frame_shade_fwd_rec = voge_composite_shade_fwd_rec (C = 0: voge_composite_fwd_rec) on records of `kind` that also zeroes
 * bwd_acc (voge_frame_bwd_gen_acc_bytes); kind 1 keeps no act / dsd (pass NULL).  voge_frame_bwd_gen: EVERY backward route of such fragments -- form 0: the image's
 * gradient (voge_fragment_shade_bwd), 1: merge_final's (voge_fragment_merge_bwd; wsum = g_wsum | NULL), 2: the weights' own
 * (voge_fragment_bwd: g = g_weight with strides, g_hitlen | NULL; attr / rgb / bg unused, C = 0) -- reading the forward's records (no
 * pack launch), with acc zeroed already (acc_is_zero != 0) or filled here, and writing the gradients of verts and sigmas AS THE USER
 * HOLDS THEM (the sum over the views of a shared set, d A / d sigma = 2: no preamble-backward launch).  K <= 128 (form 2: 256).
 */
size_t voge_frame_bwd_gen_acc_bytes(int P);
int voge_frame_trace_fwd_gen(const float *verts, const float *sigmas, int shared_verts, int shared_sigmas, int kind,
                             const float *R, const float *T, const float *focal, const float *pp, int row0, int stripe_h,
                             int pitch, int behind, int B, int N, int h, int W, int K, float thr_act, void *workspace,
                             size_t workspace_bytes, int32_t *idx, float *len, int32_t *cnt, float *records, float *rays,
                             float *origin, voge_stream_t stream);
int voge_frame_shade_fwd_rec(int kind, int32_t *idx, const int32_t *cnt, const float *len, const float *records,
                             const float *rays, float occ, const float *colors, const float *bg, float thr,
                             long npix, int K, int C, long Nattr, float *weight, int64_t *valid_num,
                             float *rgb, float *img, float *wsum, float *sil, float *act, float *dsd, void *bwd_acc,
                             size_t bwd_acc_bytes, voge_stream_t stream);
int voge_frame_bwd_gen(int form, const float *records, int shared_verts, int shared_sigmas, int kind, const float *rays,
                       const float *attr, const int32_t *idx, const int32_t *cnt, const float *weight, const float *act,
                       const float *len, const float *dsd, const float *rgb, const float *wsum, const float *bg, float thr,
                       const float *g, long g_stride0, long g_stride1, const float *g_hitlen, float occ, int B, int N,
                       long nrows, int W, int K, int C, long Nattr, void *acc, size_t acc_bytes, int acc_is_zero,
                       float *g_verts, float *g_sigmas, float *g_attr, voge_stream_t stream);

/* interpolate_attr (+ get_silhouette) on fragments of this renderer, backward: merge_final's own backward
 * (VoGE/Aggregation.py:111-141; g_rgb = the gradient of the merged attributes [nrows*W,C], strides as g_img above), plus
 * g_wsum [nrows*W] or NULL = the gradient of the per-pixel weight sum (get_silhouette = min(sum, 1) of the same fragments),
 * then composite and trace as above -- the reference's training pattern (demo/ShapeFitting.py:217,295) as ONE kernel.
 * attr [Nattr,C], C <= 4, K <= 128.  Writes g_verts / g_sigmas (both or neither) and g_attr [Nattr,C]. */
int voge_fragment_merge_bwd(const float *mus, const float *isigmas, const float *rays, const float *attr,
                            const int32_t *idx, const int32_t *cnt, const float *weight, const float *act,
                            const float *len, const float *dsd, const float *g_rgb, long g_stride_pix,
                            long g_stride_c, const float *g_wsum, float occ, int P, long nrows, int W, int K, int C,
                            long Nattr, void *workspace, size_t workspace_bytes, float *g_mus, float *g_isigmas,
                            float *g_attr, voge_stream_t stream);      /* (the general 3x3 form of the entry point below) */
int voge_fragment_merge_bwd_iso(const float *records, const float *sigmas, int shared, int sigma_mode,
                                const float *rays, const float *attr, const int32_t *idx, const int32_t *cnt,
                                const float *weight, const float *act, const float *len, const float *dsd,
                                const float *g_rgb, long g_stride_pix, long g_stride_c, const float *g_wsum,
                                float occ, int B, int N, long nrows, int W, int K, int C, long Nattr,
                                void *workspace, size_t workspace_bytes, float *g_verts, float *g_sigmas,
                                float *g_attr, voge_stream_t stream);

/*
 * The same single pass driven by the gradient of the WEIGHTS, whoever produced it: the backward of
 * `aggregation` (VoGE/Aggregation.py:30-107, autograd in the reference) followed by the fine trace's
 * (ray_trace_voge.cu:283-332) for fragments made by voge_fragments_fwd*.  This is what the reference's own training
 * loops need -- they differentiate through interpolate_attr (merge_final) and get_silhouette, not through
 * to_colored_background (demo/ShapeFitting.py:217,295; demo/ReasonOcclusion.py:106-109;
 * demo/EfficientCuboidViaOptimization.py:109-112).  Replaces voge_composite_bwd + voge_trace_bwd*(_iso, _iso_view) and
 * their exchange of g_len / g_act / g_dsd through memory; act / dsd may be NULL for scalar-sigma fragments that were
 * traced without them.
 *   g_weight: element (pixel p, slot k) at g_weight[p * gw_stride_pix + k * gw_stride_k] -- (K, 1) for a contiguous
 *             [nrows*W,K] array, (1, 0) for a per-pixel value broadcast over the slots (the gradient a silhouette
 *             loss alone hands back); NULL = zero.
 *   g_hitlen: NULL, or [nrows*W,K] contiguous: the gradient of vert_hit_length (= the trace's len, Aggregation.py:107).
 * Any K <= VOGE_MAX_K (odd K too; lists of more than 128 slots put four slots on a lane).  cnt required.
 * idx must be what the trace wrote: the first cnt[p] slots of a pixel hold DISTINCT Gaussians (the per-Gaussian
 * accumulation takes a pixel's lanes through its table together, without arbitration between them).
 * Writes g_verts / g_sigmas (iso: through the view's chain rule, as voge_trace_bwd_iso_view) or g_mus [P,3] /
 * g_isigmas [P,3,3] (raw outer-product sums, as voge_trace_bwd).  No ray gradient: callers that optimise the rays
 * themselves use voge_composite_bwd + voge_trace_bwd.  workspace: >= voge_fragment_bwd_workspace_bytes(B*N) bytes.
 */
int voge_fragment_bwd_iso(const float *records, const float *sigmas, int shared, int sigma_mode, const float *rays,
                          const int32_t *idx, const int32_t *cnt, const float *weight, const float *act,
                          const float *len, const float *dsd, const float *g_weight, long gw_stride_pix,
                          long gw_stride_k, const float *g_hitlen, float occ, int B, int N, long nrows, int W, int K,
                          void *workspace, size_t workspace_bytes, float *g_verts, float *g_sigmas,
                          voge_stream_t stream);
int voge_fragment_bwd(const float *mus, const float *isigmas, const float *rays, const int32_t *idx,
                      const int32_t *cnt, const float *weight, const float *act, const float *len,
                      const float *dsd, const float *g_weight, long gw_stride_pix, long gw_stride_k,
                      const float *g_hitlen, float occ, int P, long nrows, int W, int K, void *workspace,
                      size_t workspace_bytes, float *g_mus, float *g_isigmas, voge_stream_t stream);

/*
 * Composite forward.  Replaces: VoGE/Aggregation.py:82-107 `aggregation`
 * (get_cross_activation :30-51 + assign2weight :54-79), without any [npix,K,K] temporary.
 *   w_m = exp(-occ * sum_k exp(-act_k) * (erf((len_m-len_k)*sqrt(dsd_k+1e-10)) + 1)/2)
 *         * exp(-act_m) / exp(-0.5);   valid_num = #(idx >= 0)
 * cnt: NULL, or [npix] int32 = the trace forward's out_cnt for the same lists (slots k >= cnt are
 * its sentinels): valid_num = cnt, idx may be NULL and is not read, slots beyond cnt are not
 * loaded at all and workgroups whose pixels are all empty only write zeros.
 */
int voge_composite_fwd(const int32_t *idx, const int32_t *cnt, const float *act, const float *len,
                       const float *dsd, float occ, long npix, int K, float *weight,
                       int64_t *valid_num, voge_stream_t stream);

/*
 * Composite backward (the reference relies on autograd through Aggregation.py:49,70,74,77).
 * g_weight [npix,K] -> g_act, g_len, g_dsd [npix,K] (fully written).  weight = the forward's
 * output for the same inputs (saves the backward recomputing it); NULL -> recomputed.
 * cnt (NULL allowed) = the trace forward's out_cnt, as in voge_composite_fwd: slots k >= cnt are
 * the trace's sentinels and are not loaded; workgroups whose pixels are all empty write zeros.
 */
int voge_composite_bwd(const float *act, const float *len, const float *dsd, const float *weight,
                       const int32_t *cnt, const float *g_weight, float occ, long npix, int K, float *g_act,
                       float *g_len, float *g_dsd, voge_stream_t stream);

/*
 * Attribute merge forward.  Replaces: VoGE/Aggregation.py:111-141 `merge_final`
 * (reached through Renderer.py:153 interpolate_attr).
 *   out[pix,c] = sum_{k < valid_num[pix]} attr[max(idx_k,0)... , c] * weight[pix,k]
 * attr [Nattr,C].  If fix_negative_idx != 0, idx is updated in place the way the reference
 * does (`vert_assign += (vert_assign < 0)`, Aggregation.py:131: -1 becomes 0).
 */
int voge_merge_fwd(const float *attr, int32_t *idx, const float *weight,
                   const int64_t *valid_num, long npix, int K, int C, long Nattr,
                   int fix_negative_idx, float *out, voge_stream_t stream);

/*
 * Attribute merge backward (autograd of merge_final): g_out [nrows*W,C] ->
 * g_attr [Nattr,C] (zero-filled here, then scatter-added) and g_weight [nrows*W,K].
 * The pixel grid is nrows rows of W pixels (any factorisation of the pixel count is valid;
 * the true image width gives the best locality).
 * Either output pointer may be NULL to skip it.
 */
int voge_merge_bwd(const float *attr, const int32_t *idx, const float *weight,
                   const int64_t *valid_num, const float *g_out, long nrows, int W, int K,
                   int C, long Nattr, float *g_attr, float *g_weight, voge_stream_t stream);

/*
 * Background blend forward.  Replaces: VoGE/Renderer.py:157-171
 * (get_silhouette + to_colored_background):
 *   sil = min(sum_k w_k, 1); mask = thr > 0 ? (sil > thr) : sil;
 *   out = min(rgb + (1 - mask) * bg, 1)
 * rgb,out [npix,C], bg [C], sil_out [npix] (may be NULL).
 */
int voge_blend_fwd(const float *rgb, const float *weight, const float *bg, float thr,
                   long npix, int K, int C, float *out, float *sil_out,
                   voge_stream_t stream);

/*
 * get_silhouette alone (VoGE/Renderer.py:157-159): sil [npix] = min(sum_k weight, 1); wsum [npix] (may be NULL) = the
 * unclamped sum, which the backward needs.  Backward: g_pix [npix] = g_sil * [wsum < 1] (1/2 at wsum == 1, as
 * torch.min splits ties) -- the gradient of EVERY slot of the pixel, so the [npix,K] gradient of the weights is this
 * array viewed with stride 0 along K (voge_fragment_bwd* reads it that way: gw_stride_pix = 1, gw_stride_k = 0).
 */
int voge_silhouette_fwd(const float *weight, long npix, int K, float *sil, float *wsum, voge_stream_t stream);
int voge_silhouette_bwd(const float *wsum, const float *g_sil, long npix, float *g_pix, voge_stream_t stream);

/*
 * Background blend backward: g_out [npix,C] -> g_rgb [npix,C] and the additive term
 * g_weight_add [npix,K] (d out / d weight through the silhouette; zero where thr > 0 or
 * where the silhouette is clamped).  Recomputes the clamps from rgb / weight / bg.
 */
int voge_blend_bwd(const float *rgb, const float *weight, const float *bg, float thr,
                   const float *g_out, long npix, int K, int C, float *g_rgb,
                   float *g_weight_add, voge_stream_t stream);

/*
 * Fused merge + silhouette + blend ("shade").  Replaces, in one pass: interpolate_attr ->
 * merge_final (Aggregation.py:111-141), get_silhouette (Renderer.py:157-159) and
 * to_colored_background (Renderer.py:162-171):
 *   rgb = sum_{k < valid_num} attr[idx_k] w_k ;  sil = min(sum_k w_k, 1) ;
 *   img = min(rgb + (1 - (thr > 0 ? sil > thr : sil)) * bg, 1)
 * Outputs (each may be NULL): out_rgb [npix,C] (needed again by voge_shade_bwd), out_img
 * [npix,C] (requires bg [C]), out_sil [npix], out_wsum [npix] (sum_k w_k before the clamp; saves
 * voge_shade_bwd a pass over weight).  fix_negative_idx as in voge_merge_fwd.
 */
int voge_shade_fwd(const float *attr, int32_t *idx, const float *weight, const int64_t *valid_num,
                   const float *bg, float thr, long npix, int K, int C, long Nattr,
                   int fix_negative_idx, float *out_rgb, float *out_img, float *out_sil,
                   float *out_wsum, voge_stream_t stream);

/*
 * Backward of voge_shade_fwd for C <= 4.  g_up [nrows*W,C] is the gradient of img (bg != NULL;
 * rgb = the forward's out_rgb) or of rgb itself (bg == NULL, rgb ignored).  Writes g_weight
 * [nrows*W,K] (may be NULL) and g_attr [Nattr,C] (zero-filled here then accumulated; may be NULL).
 * wsum [nrows*W] = the forward's out_wsum (unclamped sum_k w_k); NULL -> recomputed from weight.
 */
int voge_shade_bwd(const float *attr, const int32_t *idx, const float *weight,
                   const int64_t *valid_num, const float *rgb, const float *wsum, const float *bg,
                   float thr, const float *g_up, long nrows, int W, int K, int C, long Nattr,
                   float *g_attr, float *g_weight, voge_stream_t stream);

/*
 * Pixel-ray generation.  Replaces: the PyTorch3D call in VoGE/Renderer.py:124-130
 * (NDCMultinomialRaysampler(unit_directions=True) on screen-space PerspectiveCameras):
 *   d_view(i,j) = [(px-j-0.5)/fx, (py-i-0.5)/fy, 1],  rays = normalise(d_view @ R^-1),
 *   origin = -T @ R^-1          (row vectors, X_view = X_world @ R + T).
 * R [B,3,3], T [B,3], focal [B,2], pp [B,2] (principal point, pixels).  Renders image rows
 * row0 .. row0+h-1 (a pixel-row band): rays [B,h,W,3], origin [B,3].
 * cones: NULL, or voge_cones_floats(B,h,W) floats receiving the bounding cones (axis, cos, sin of the
 * half angle, flags) of every 32x32-pixel super-tile of the band, of its quads and of its 8x8 tiles:
 * what voge_trace_topk_fwd*'s `cones` argument takes.  They come out of the same arithmetic as the
 * rays, at no extra launch.
 */
int voge_rays_fwd(const float *R, const float *T, const float *focal, const float *pp, int B,
                  int row0, int h, int W, float *rays, float *origin, float *cones, voge_stream_t stream);

/*
 * The same for a STRIPED row set: output row i is image row row0 + (i / stripe_h) * pitch + i % stripe_h -- every
 * pitch-th stripe of stripe_h rows from row0 on, h rows in total (only the last stripe may be cut short by h).  What a
 * rank renders when one frame is dealt to the ranks of a node in interleaved stripes (voge_amd/distributed.py: a
 * contiguous band per rank leaves the centre band 4x as expensive as the rim bands).  The stacked rows are an image like
 * any other to every later stage (pixels are independent; the cones bound the rays actually present).  stripe_h >= h is
 * voge_rays_fwd.  voge_rays_striped_bwd is its backward (voge_rays_bwd's arguments plus the stripe geometry).
 * No counterpart in the reference (single device).
 */
int voge_rays_striped_fwd(const float *R, const float *T, const float *focal, const float *pp, int B,
                          int row0, int h, int stripe_h, int pitch, int W, float *rays, float *origin, float *cones,
                          voge_stream_t stream);
int voge_rays_striped_bwd(const float *R, const float *T, const float *focal, const float *pp,
                          const float *g_rays, const float *g_origin, int B, int row0, int h, int stripe_h, int pitch, int W,
                          float *scratch, float *g_R, float *g_T, float *g_focal, float *g_pp,
                          voge_stream_t stream);

/* The same cones from any rays [B,H,W,3] tensor (no counterpart in the reference: culling aid). */
size_t voge_cones_floats(int B, int H, int W);
int voge_ray_cones(const float *rays, int B, int H, int W, float *cones, voge_stream_t stream);

/*
 * The renderer's elementwise preamble for (N,3) / (N,3,3) sigmas in one launch each way.
 * Replaces: VoGE/Renderer.py:130-137 with VoGE/Aggregation.py:144-175 (centred = verts - origin[b];
 * isigma = 2 * expend_sigma(sigmas)) and their autograd.  kind 1: sigmas [.., N, 3] (A = 2 diag(s)); kind 2: [.., N, 3, 3]
 * (A = 2 S).  shared_verts / shared_sigmas: one [N, ...] set for every view, or [B, N, ...].  Writes mus [B*N,3],
 * isigmas [B*N,3,3]; the backward writes g_verts / g_sigmas in the parameters' own shapes (either may be NULL), a shared
 * set summing its views in a fixed order.
 */
int voge_general_preamble_fwd(const float *verts, const float *sigmas, const float *origin, int B, int N,
                              int shared_verts, int shared_sigmas, int kind, float *mus, float *isigmas,
                              voge_stream_t stream);
int voge_general_preamble_bwd(const float *g_mus, const float *g_isigmas, int B, int N, int shared_verts,
                              int shared_sigmas, int kind, float *g_verts, float *g_sigmas, voge_stream_t stream);

/*
 * Backward of voge_rays_fwd: g_rays [B,h,W,3] (may be NULL) and g_origin [B,3] (may be NULL) ->
 * g_R [B,3,3], g_T [B,3], g_focal [B,2], g_pp [B,2] (each may be NULL).  scratch: B*16 floats.
 */
int voge_rays_bwd(const float *R, const float *T, const float *focal, const float *pp,
                  const float *g_rays, const float *g_origin, int B, int row0, int h, int W,
                  float *scratch, float *g_R, float *g_T, float *g_focal, float *g_pp,
                  voge_stream_t stream);

/* ---- rows "next" (SURVEY.md §8f): dense ray API and the sampler's helpers ------------------ */

/*
 * Dense trace: every (ray n, Gaussian m) pair.  Replaces: VoGE._C.ray_trace_voge_ray
 * (voge_ray_tracing_ray.cu:114-143, :242-283).  mus [M,3], isigmas [M,3,3], rays [N,3] ->
 * len, act, dsd [N,M] (same arithmetic as the fine trace).
 */
int voge_ray_dense_fwd(const float *mus, const float *isigmas, const float *rays, int M, long N,
                       float *len, float *act, float *dsd, voge_stream_t stream);

/*
 * Its backward.  Replaces: VoGE._C.ray_trace_voge_ray_backward (voge_ray_tracing_ray.cu:147-188).
 * g_len, g_act, g_dsd [N,M] -> g_ray [N,3], g_mus [M,3], g_isg [M,3,3] (all written).
 */
int voge_ray_dense_bwd(const float *mus, const float *isigmas, const float *rays, const float *g_len,
                       const float *g_act, const float *g_dsd, int M, long N, float *g_ray,
                       float *g_mus, float *g_isg, voge_stream_t stream);

/*
 * Top-K over dense rows.  Replaces: VoGE._C.find_nearest_k (voge_ray_tracing_ray.cu:191-239,
 * :328-375): per ray the K entries with act < thr_act and the smallest (len, m), ascending;
 * unused slots: idx -1, len 1e10, act 0, dsd 0.
 */
int voge_find_nearest_k(const float *len_in, const float *act_in, const float *dsd_in, float thr_act,
                        int M, int K, long N, int32_t *idx, float *len, float *act, float *dsd,
                        voge_stream_t stream);

/*
 * Gradient of that selection (the reference scatters in Python, RayTracing.py:230-241):
 * gi_*[n, idx[n,k]] = g_*[n,k] for idx >= 0, zero elsewhere.  gi_* [N,M] are fully written.
 */
int voge_find_nearest_k_bwd(const int32_t *idx, const float *g_len, const float *g_act, const float *g_dsd,
                            int M, int K, long N, float *gi_len, float *gi_act, float *gi_dsd,
                            voge_stream_t stream);

/*
 * Per-Gaussian maximum weight.  Replaces: VoGE._C.scatter_max (sample_voge.cu:69-92,:135-170):
 * out[idx] = max over the n slots with that index of weight (weights are >= 0); out [Nv] zero-filled.
 * (sample_voge / sample_voge_backward, sample_voge.cu:35-66,:173-252, are the transpose of
 * merge_final: use voge_merge_bwd with g_out = [image | 1] for the forward and voge_merge_fwd /
 * voge_merge_bwd for the backward -- voge_amd/Sampler.py.)
 */
int voge_scatter_max(const float *weight, const int32_t *idx, long n, long Nv, float *out,
                     voge_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VOGE_HIP_H */
