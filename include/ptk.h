/*
 * ptk.h -- C ABI of libptk, the MI355X (gfx950) implementation of the
 * Path-Tracer per-frame hot path: path trace + accumulate, exposure reduce,
 * denoiser inference (feature statistics -> MLP -> cross-bilateral filter).
 *
 * The reference has no FFI; its GPU seam is backend slot 2 of four switch
 * dispatchers (headers/Renderers.h:31-60,72-101, Denoiser.cpp:9-35,
 * headers/DenoiserNN.h:103-131), which forward to void(void) functions in
 * three namespaces (headers/CUDAHeader.h:5-52) that exchange data through
 * globals (headers/GLOBALS.h:15-76).  Each entry point below names the
 * reference function(s) it replaces; the data that crossed through globals
 * crosses here through explicit pointers.
 *
 * Conventions: every function returns PTK_OK (0) or a negative PTK_ERR_* code
 * and leaves a message for ptk_last_error() (thread-local).  Pointers are host
 * memory unless the name ends in _d.  A context is bound to one GPU and is not
 * thread-safe; calls on one context execute in issue order.  Different contexts
 * may be used from different threads at the same time, on one GPU or several
 * (the library keeps no state that couples them).  Images are
 * row-major with row 0 at the BOTTOM (the reference's convention,
 * Renderers.cpp:67, PT.cpp:487).
 */
#ifndef PTK_H
#define PTK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTK_OK 0
#define PTK_ERR_ARG (-1)    /* bad argument (null pointer, size, range) */
#define PTK_ERR_HIP (-2)    /* a HIP runtime call failed */
#define PTK_ERR_STATE (-3)  /* call order violated (e.g. render before set_constants) */
#define PTK_ERR_NOMEM (-4)  /* host or device allocation failed */

/* Scene POD handed to ptk_set_constants: byte-for-byte the reference's
 * struct Constants (headers/GLOBALS.h:80-94), so a host that already fills
 * that struct (Renderers.cpp:1051-1149) passes its address unchanged. */
#define PTK_MAX_SHAPES 50
#define PTK_MAX_ATTRS 450
#define PTK_MAX_MATS 50
#define PTK_MAX_IMPORTANT 10
#define PTK_MAX_DEPTH 12
typedef struct ptk_constants {
    float cam_pos[3], cam_forward[3], cam_right[3], cam_up[3];
    float max_angle, rand_samp;
    int32_t num_shapes, max_depth;
    int32_t res_v, res_h;
    float max_angle_v, max_angle_h, focal_length;
    int32_t shapes[PTK_MAX_SHAPES][3];   /* {type 0 sphere / 1 box, material, attr offset} */
    float obj_attributes[PTK_MAX_ATTRS]; /* sphere: centre,r; box: centre, lo, hi, quat w,x,y,z */
    float mat_list[PTK_MAX_MATS][6];     /* r,g,b, blur, refractive index, type 0..3 */
    uint32_t important_shapes[PTK_MAX_IMPORTANT];
    uint32_t num_important_shapes;
    uint32_t get_denoiser_inf;
} ptk_constants;
#define PTK_CONSTANTS_BYTES 3732

/* Accumulator planes held in HBM, plane-major: plane p of pixel i is at
 * base[p * P + i], P = width * height. */
enum {
    PTK_PL_COL = 0,     /* 3: sum of radiance                      (preScreen)   */
    PTK_PL_NORMAL = 3,  /* 3: sum of first-hit normals             (normal)      */
    PTK_PL_ALB1 = 6,    /* 3: sum of first-hit albedo              (albedo1)     */
    PTK_PL_ALB2 = 9,    /* 3: sum of second-hit albedo             (albedo2)     */
    PTK_PL_WPOS = 12,   /* 3: sum of first-hit positions           (worldPos)    */
    PTK_PL_DL = 15,     /* 1: sum of direct-light fractions        (directLight) */
    PTK_PL_SDV = 16,    /* 16: running squared deviations (stdDevVecs, 5x3 + 1)  */
    PTK_PL_SD = 32,     /* 6: stdDev[6] as of the last sample                    */
    PTK_N_PLANES = 38
};

/* seed_mode of ptk_render */
#define PTK_SEED_REFERENCE 0 /* host-sequential global seed stream (Renderers.cpp:832-838) */
#define PTK_SEED_COUNTER 1   /* device counter-based seeds: hash(pixel, sample, stream) */

/* math mode of ptk_set_math_mode */
#define PTK_MATH_EXACT 0 /* IEEE divide/sqrt, no FMA contraction, reference operation order */
#define PTK_MATH_FAST 1  /* reciprocal-based slabs, FMA contraction; statistically equivalent */

/* formula of ptk_auto_exposure */
#define PTK_EXPOSURE_CPU 0   /* sum * 9.6f / xRes * yRes * spp, as written (Renderers.cpp:1158) */
#define PTK_EXPOSURE_SKEPU 1 /* 9.6f * sum / P * spp, as written (Renderers.cpp:1192) */
#define PTK_EXPOSURE_MEAN 2  /* 9.6 * mean luminance per sample */

typedef struct ptk_ctx ptk_ctx;

/* Device time of the most recent call of each kind, from HIP events on the
 * context's stream (replaces the chrono globals renderTime / exposureTime /
 * denoiseTime, headers/Renderers.h:33,59,74,100, Denoiser.cpp:5,37). */
typedef struct ptk_times {
    double render_ms;        /* path-trace kernels of the last ptk_render            */
    double seed_ms;          /* host seed generation + upload (seed mode 0 only)     */
    double exposure_ms;      /* reduce kernel of the last ptk_auto_exposure          */
    double features_ms;      /* feature + MLP kernel of the last ptk_denoise         */
    double filter_ms;        /* filter kernel of the last ptk_denoise                */
    double denoise_ms;       /* features_ms + filter_ms                              */
    uint64_t render_samples; /* width*height*spp of the last ptk_render              */
    uint64_t render_rays;    /* bounce rays traced by the last ptk_render            */
    double post_ms;          /* kernel of the last ptk_post_process                  */
    double train_ms;         /* reduction kernel of the last ptk_train_step          */
} ptk_times;

/* Replaces CUDADenoiser::InitBuffers / CUDADenoiserNN::InitBuffers
 * (CUDADenoiser.cu:184-189, CUDADenoiserNN.cu:614-632) and the allocation
 * half of CUDARender::UpdateConstants (CUDARender.cu:933-957).  Binds the
 * context to GPU `device`; buffers are sized on ptk_set_constants. */
int ptk_create(int device, ptk_ctx** out);

/* Replaces the three FreeBuffers (CUDADenoiser.cu:190-195,
 * CUDADenoiserNN.cu:634-646). */
int ptk_destroy(ptk_ctx* ctx);

/* Message of the calling thread's most recent failed call ("" if none). */
const char* ptk_last_error(void);

/* Replaces CUDARender::UpdateConstants (CUDARender.cu:933-957): takes the
 * 3732-byte scene POD, validates it (the reference does not: shape, material
 * and attribute indices are range-checked here and rejected with
 * PTK_ERR_ARG), compiles it for the device and (re)allocates the planes when
 * the resolution changed.  Resets the accumulators, as the reference's
 * RefreshScreen does (PT.cpp:159-169).  want_denoise_features overrides the
 * POD's get_denoiser_inf when >= 0. */
int ptk_set_constants(ptk_ctx* ctx, const void* constants, size_t nbytes,
                      int want_denoise_features);

/* Scene description without the POD's fixed limits (the reference caps a scene
 * at 50 shapes / 450 attribute floats / 50 materials / 10 important shapes,
 * headers/GLOBALS.h:80-94).  Same fields and meanings as ptk_constants, with
 * the per-shape tables as arrays of any length; every shape owns 13 attribute
 * floats in the POD's order (sphere: centre, r, 9 unused; box: centre, lo, hi,
 * quaternion w,x,y,z). */
typedef struct ptk_scene_desc {
    float cam_pos[3], cam_forward[3], cam_right[3], cam_up[3];
    float rand_samp;
    int32_t max_depth;
    int32_t res_v, res_h;
    float max_angle_v, max_angle_h, focal_length;
    int32_t num_shapes;            /* 0 .. PTK_SCENE_MAX_SHAPES */
    int32_t num_mats;              /* 1 .. PTK_SCENE_MAX_MATS */
    int32_t num_important_shapes;  /* 0 .. num_shapes */
    int32_t flags;                 /* PTK_SCENE_* */
    const int32_t* shape_type;     /* [num_shapes] 0 sphere / 1 box */
    const int32_t* shape_mat;      /* [num_shapes] material index */
    const float* shape_attrs;      /* [num_shapes][13] */
    const float* mat_list;         /* [num_mats][6] r,g,b, blur, refractive index, type 0..3 */
    const uint32_t* important_shapes; /* [num_important_shapes] */
} ptk_scene_desc;
#define PTK_SCENE_MAX_SHAPES (1 << 24)
#define PTK_SCENE_MAX_MATS 32767
#define PTK_SCENE_ATTRS 13
/* Give every node of the hierarchy unbounded extents, so each ray tests every
 * shape: the same kernels with nothing culled (a checking aid). */
#define PTK_SCENE_VISIT_ALL 1

typedef struct ptk_scene_stats {
    int32_t num_shapes, nodes, leaves, depth;
    double build_ms;      /* host time spent compiling the shapes and building the hierarchy */
    uint64_t device_bytes;
} ptk_scene_stats;

/* The scalable counterpart of ptk_set_constants (no counterpart in the
 * reference, whose Renderers::UpdateConstants, Renderers.cpp:1051-1149, can
 * only fill the fixed POD): the shapes are kept in HBM behind a bounding-volume
 * hierarchy built here on the host, and ptk_render then traces through it.
 * Every per-sample result is what the reference's index-order scans over the
 * same shapes would give (closest hit with the lowest index among equal
 * distances, Renderers.cpp:117-213; shadow rays stopped by the lowest-index
 * opaque occluder and bent by refracting ones in index order, 490-668): the
 * shapes are tested by the same routines and the hierarchy only skips shapes a
 * ray cannot reach -- including the unrotated boxes the reference's slab test
 * accepts from anywhere when a plane distance is 0 / 0 (a direction component
 * of exactly +0, the origin exactly in a face's plane; Renderers.cpp:157-196):
 * those are looked up by coordinate.  Box quaternions must be of unit length
 * (within 1e-4).
 * Everything else (planes, seeds, denoiser, post-process) is as after
 * ptk_set_constants.  stats is optional. */
int ptk_set_scene(ptk_ctx* ctx, const ptk_scene_desc* scene, int want_denoise_features,
                  ptk_scene_stats* stats);

/* The host half of ptk_set_scene without a context or a GPU: validates the
 * description, compiles the shapes, builds the hierarchy and verifies it (every
 * shape in exactly one leaf, every node's bounds holding the padded bounds of
 * what lies beneath, depth within the kernels' traversal stack).  stats is
 * optional (device_bytes is 0). */
int ptk_check_scene(const ptk_scene_desc* scene, ptk_scene_stats* stats);

/* Replaces the camera half of Renderers::UpdateCam (Renderers.cpp:1132-1149) for
 * the scene already set by ptk_set_constants or ptk_set_scene: new camera
 * position and axes, everything else (shapes, materials, resolution, and for
 * ptk_set_scene the hierarchy) kept; the accumulators are reset.  A
 * ptk_set_scene scene is built for cameras within the bounds of its shapes and
 * of the camera it was set with (grown by 2 % of their diagonal: the spheres'
 * bounds are padded for the longest ray that fits); a position outside that
 * box is PTK_ERR_ARG (set the scene again). */
int ptk_set_camera(ptk_ctx* ctx, const float cam_pos[3], const float cam_forward[3],
                   const float cam_right[3], const float cam_up[3]);

/* The zeroing half of PT::RefreshScreen (PT.cpp:159-169): sampleCount = 0 and
 * all planes cleared. */
int ptk_reset_accumulators(ptk_ctx* ctx);

/* PTK_MATH_EXACT (default) or PTK_MATH_FAST for subsequent ptk_render calls. */
int ptk_set_math_mode(ptk_ctx* ctx, int mode);

/* Stream id mixed into PTK_SEED_COUNTER seeds (one per GPU when samples are
 * split across ranks).  Default 0. */
int ptk_set_stream_id(ptk_ctx* ctx, uint64_t stream_id);

/* Replaces `spp` calls of CUDARender::render (CUDARender.cu:857-931), with
 * the host accumulate loop (886-929, identical to CPURender's,
 * Renderers.cpp:845-888) fused into the kernel: each pass increments
 * sampleCount, traces one path per pixel and updates all planes.
 * seed_mode PTK_SEED_REFERENCE advances glo_seed exactly spp*width*height
 * steps of the reference's recurrence (in-out, must be non-null); the
 * seeds are generated on the host and staged through at most 256 MiB of
 * device and of pinned host memory (PTK_ERR_NOMEM if that cannot be had);
 * PTK_SEED_COUNTER ignores glo_seed.  rays_out (optional) receives the bounce
 * rays of the whole call (the reference's rayCount for spp == 1). */
int ptk_render(ptk_ctx* ctx, int spp, int seed_mode, uint64_t glo_seed[2], uint64_t* rays_out);

/* Hands the accumulators back in the layout of the reference's globals
 * (headers/GLOBALS.h:45-76) so a host can refill them: interleaved xyz per
 * pixel for the five vec3 screens, one float per pixel for directLight's x,
 * six floats per pixel for DenoisingInf::stdDev.  Any pointer may be null. */
int ptk_read_accumulators(ptk_ctx* ctx, float* pre_screen_3p, float* normal_3p,
                          float* albedo1_3p, float* albedo2_3p, float* direct_light_1p,
                          float* world_pos_3p, float* std_dev_6p, int* sample_count);

/* Raw plane access (plane-major floats, n_planes * P values): used to seed
 * synthetic accumulators, to checkpoint, and by multi-GPU reduction. */
int ptk_read_planes(ptk_ctx* ctx, int first_plane, int n_planes, float* out);
int ptk_write_planes(ptk_ctx* ctx, int first_plane, int n_planes, const float* in,
                     int sample_count);

/* Device address of plane 0 and the per-plane pixel count, for collectives
 * that reduce the planes in place (the planes are one contiguous allocation
 * of PTK_N_PLANES * P floats). */
int ptk_planes_device(ptk_ctx* ctx, float** base_d, size_t* pixels_per_plane);
int ptk_get_sample_count(ptk_ctx* ctx, int* sample_count);
/* Declares the planes valid for `sample_count` samples (after the caller has written them
 * through the device address): on a context configured with want_denoise_features this
 * covers the feature planes, which ptk_denoise then accepts. */
int ptk_set_sample_count(ptk_ctx* ctx, int sample_count);

/* Recomputes planes PTK_PL_SD from sum and sum-of-squares style planes after a
 * cross-GPU reduction: stdDev[k] = sum_c sdv[k][c] / sample_count. */
int ptk_refresh_std_dev(ptk_ctx* ctx);

/* Replaces CUDARender::CUDAAutoExp (CUDARender.cu:1068-1094) and mirrors
 * Renderers::CPUAutoExp / SkePUAutoExp (Renderers.cpp:1151-1159, 1174-1193):
 * luminance map + sum on the device.  raw_sum (optional) receives the
 * unscaled sum of per-pixel luminance in double. */
int ptk_auto_exposure(ptk_ctx* ctx, int formula, float* exposure, double* raw_sum);

/* The exposure reduce for callers that cannot allocate (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  It has no
 * counterpart in the reference.  Same kernels and arithmetic as
 * ptk_auto_exposure, but context-free: the input is the device address of three
 * colour planes (plane-major, `pixels` floats each, width * height == pixels; e.g.
 * plane 0 of ptk_planes_device), the outputs are DEVICE pointers (either may be
 * null, not both) written by the last kernel, everything is enqueued on
 * `hip_stream` (a hipStream_t; null = the default stream) of the CURRENT device
 * and nothing is synchronised.  The partial sums live in a 4 KiB array inside the
 * library's code object, one copy per GPU, held until the process ends; the call
 * performs no hipMalloc / hipFree and no host<->device copy.  pixels above
 * PTK_STATIC_MAX_PIXELS is PTK_ERR_ARG.  All calls on one GPU share that array:
 * calls that overlap on one GPU (other streams or threads) race; issue them on
 * one stream.  colour_planes_d needs 4-byte alignment only (16-byte aligned planes take
 * the 16-byte loads, others the one-pixel loop).  The FIRST call on each GPU looks the
 * array's address up (a mutex, and possibly the loading of the library's code object onto
 * that GPU, which allocates): make it outside graph capture, or after a ptk_create on that
 * GPU, which performs the same look-up. */
#define PTK_STATIC_MAX_PIXELS ((size_t)1 << 28)
int ptk_auto_exposure_static(const float* colour_planes_d, size_t pixels, int width, int height,
                             int sample_count, int formula, float* exposure_d, double* raw_sum_d,
                             void* hip_stream);

/* Replaces CUDADenoiserNN::ForwardProp (CUDADenoiserNN.cu:297-368) followed by
 * CUDADenoiser::denoise (CUDADenoiser.cu:123-182), with the arithmetic of the
 * sequential copies (DenoiserNN.cpp:12-301, Denoiser.cpp:222-355): 36
 * neighbourhood features -> 36-10-10-7 MLP (weights in the file order of
 * DenoiserNN.cpp:1625-1663: w12[36*node+input], w23[10*node+input],
 * w34[10*node+input]) -> (2N+1)^2 clamp-to-edge cross-bilateral filter.
 * variant 0 is the denoiser (a feature + MLP kernel, then a filter kernel: for
 * N = 1 on a register window of rows, for wider windows on LDS tiles).  variant 1
 * forces the LDS-tile filter at N = 1 too: same arithmetic, same bits, slower; it
 * exists so that the two forms can be checked against each other.  variant 2 runs
 * the feature pass in its tolerance-held form: primaries, per-pixel deviations, the
 * "median getter" statistic and 1/spp as in variant 0, bit for bit; the 7x7 mean
 * and deviation (from per-column statistics shared between the seven pixels of a
 * row that use them), the Sobel magnitude and the square roots formed a cheaper
 * way, within a few ulp of the channel's magnitude of variant 0's; bandwidths,
 * filtered colour and weight sum within 2e-5 relative.  A quarter faster.
 * Anything else is PTK_ERR_ARG.
 * Outputs (all optional): denoised rgb interleaved (3P), wcSum (P),
 * variances (7P, pixel-major), features (36P, pixel-major). */
int ptk_denoise(ptk_ctx* ctx, const float w12[360], const float w23[100], const float w34[70],
                int n, int variant, float* denoised_3p, float* wc_sum_p, float* variances_7p,
                float* features_36p);

/* Row-band form of ptk_denoise for sharding the denoiser across GPUs: produces
 * rows [row_begin, row_end) only and leaves the results in device memory
 * (plane-major: 3 colour planes, 1 weight-sum plane, 7 bandwidth planes; see
 * ptk_denoise_outputs_device).  Reads the accumulator planes of up to
 * max(3, n) rows on either side of the band, which must therefore be valid on
 * this GPU.  The reference has no counterpart (its denoiser is single-GPU,
 * Denoiser.cpp:3-38). */
int ptk_denoise_rows(ptk_ctx* ctx, const float w12[360], const float w23[100],
                     const float w34[70], int n, int variant, int row_begin, int row_end);

/* Multi-GPU accumulation exchange for the row-band denoiser; the reference renders and
 * denoises on one GPU (Denoiser.cpp:3-38) and has no counterpart.  GPU `rank` of `world`
 * denoises rows [y0, y1) of the frame (contiguous bands whose sizes differ by at most one
 * row, the first height % world ranks taking the extra one) and for that must hold the
 * accumulators of rows [row_lo, row_hi) = the band plus `halo` rows either side (3 for the
 * feature statistics, N for a filter of half-width N > 3), merged over all GPUs.
 * ptk_band_layout returns that span and rows_max, the longest span of any rank (any
 * pointer may be null). */
int ptk_band_layout(int height, int world, int rank, int halo, int* row_lo, int* row_hi,
                    int* rows_max);

/* Lays this GPU's 16 sum planes and 16 squared-deviation planes out for one all-to-all
 * with equal blocks: send_d[d][plane 0..31][row 0..rows_max)[x], rows [0, span length of
 * rank d) of each plane holding rank d's span (the rest of a block is left as it is and is
 * not read by the receiver).  send_floats >= world * 32 * rows_max * width.  The caller
 * then exchanges block d with GPU d (ncclSend/ncclRecv pairs, all_to_all_single, ...) into
 * a receive buffer of the same size.  Returns when the buffer is filled.  The copies run
 * on the context's own stream: whatever the caller queued on send_d on another stream (a
 * fill, say) must have finished before the call. */
int ptk_pack_bands(ptk_ctx* ctx, int world, int halo, float* send_d, size_t send_floats);

/* Merges what the all-to-all delivered (recv_d[s]: rank s's block for this rank) into the
 * context's planes, rows [row_lo, row_hi) only, in one kernel, ranks added in rank order:
 * sums add; squared deviations add with the between-rank term of the parallel-variance
 * identity, n_s (mean_s - mean)^2, which is exact for Welford's statistic and within
 * O(ln n / n) of the reference's order-dependent one (Renderers.cpp:860-887); stdDev =
 * squared deviations / n.  samples_per_rank[world] are the samples each rank rendered; the
 * context's sample count becomes their total.  Rows outside the span keep this rank's own
 * partial sums and must not be read afterwards.  The kernel runs on the context's own
 * stream: the exchange into recv_d must have completed before the call. */
int ptk_merge_received(ptk_ctx* ctx, int world, int rank, int halo, const float* recv_d,
                       size_t recv_floats, const int* samples_per_rank);

/* Device addresses of the denoiser outputs (any pointer may be null). */
int ptk_denoise_outputs_device(ptk_ctx* ctx, float** denoised_d, float** wc_sum_d,
                               float** variances_d);

/* Replaces CUDARender::PostProcess (CUDARender.cu:959-1066): the display
 * transform of the frame held by the context.  display_metric selects the source
 * as the reference's displayMetric does: 0 colour / spp, 1 the last ptk_denoise
 * output, 2 (normal / spp + 1) / 2, 3 albedo1 / spp, 4 albedo2 / spp,
 * 5 directLight / spp, 6 1 / (worldPos / spp), 7 the training target set by
 * ptk_save_target / ptk_set_target (CUDARender.cu:1006-1010, 1050-1052;
 * PTK_ERR_STATE before one is set).  Then / exposure and pow(., 1 / g)
 * (within 1e-6 relative of powf).
 * post_screen (optional) receives the reference's postScreen: interleaved rgb,
 * 3 * P floats.  rgb8 (optional) receives what PT::SaveImage writes
 * (PT.cpp:484-501): int(255.999 * c) per channel, top row first, 3 * P bytes;
 * NaN and out-of-range values, undefined in the reference's cast, clamp to
 * [0, 255]. */
int ptk_post_process(ptk_ctx* ctx, int display_metric, float exposure, float g,
                     float* post_screen, uint8_t* rgb8);

/* Replaces Denoiser::saveTargetCol (Denoiser.cpp:148-155): the training target
 * becomes the current colour sums / sample count, on the device. */
int ptk_save_target(ptk_ctx* ctx);

/* Sets the training target from host memory: interleaved rgb, 3 * P floats (the
 * reference's targetCol). */
int ptk_set_target(ptk_ctx* ctx, const float* target);

/* One training epoch's arithmetic after the render (DenoiserNN::TrainNN,
 * DenoiserNN.cpp:1520-1563): denoiser.denoise() with the given weights (forward
 * pass keeping the 36 features and both hidden layers, then the filter, window
 * n), GenRelMSE (DenoiserNN.cpp:1461-1474) into rel_mse (optional), and BackProp
 * in its sequential CPU form (CPUBackProp / CPUFilterDerivative,
 * DenoiserNN.cpp:959-1103, followed as written), which updates the three weight
 * arrays in place.  The reference accumulates the update pixel by pixel in fp32;
 * here the 56 sums are reduced in double on the device and added once.  Images
 * taller than wide are PTK_ERR_ARG: the reference's filter derivative indexes
 * pixel (j, i) as j * yRes + i (DenoiserNN.cpp:964-967), which leaves the image. */
int ptk_train_step(ptk_ctx* ctx, float w12[360], float w23[100], float w34[70], int n,
                   float learning_rate, float* rel_mse);

/* Replaces the chrono globals; see ptk_times. */
int ptk_timing(ptk_ctx* ctx, ptk_times* out);

/* Blocks until all work issued on the context has finished. */
int ptk_synchronize(ptk_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* PTK_H */
