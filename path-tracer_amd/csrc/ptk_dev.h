// ptk_dev.h -- device-side data layout and kernel launch entry points shared by
// the host API (ptk_api.cpp) and the HIP kernels.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <hip/hip_runtime_api.h>

#include "../../include/ptk.h"

// One shape of the compiled scene.  The closest-hit and shadow scans index
// this array with the (wave-uniform) loop counter, so its fields arrive in
// SGPRs through scalar loads; the copy in LDS serves per-lane lookups of the
// shape a lane actually hit.  Stride is 25 dwords so per-lane LDS reads of
// different shapes spread over the banks.
struct DevShape {
    float c[3];    // centre (sphere centre / box position)
    float lo[3];   // box bounds, world space (unused for spheres)
    float hi[3];
    float q[4];    // box quaternion w,x,y,z
    float r;       // sphere radius
    float rsq;     // r*r, rounded once on the host exactly as the kernel would
    float area;    // box: 2(xy+zy+xz) of the extents, reference operation order
    int32_t type;      // 0 sphere, 1 box
    int32_t mat;       // material index
    int32_t mat_type;  // (int)mat_list[mat][5]
    int32_t rotated;   // q.x + q.y + q.z > 1e-5f
    int32_t flat_axis; // first axis with |lo-hi| < 1e-5f, or -1
    int32_t same_mask; // bit k set when lo[k] == hi[k] exactly (not read by the kernels)
    int32_t pad[3];
};
static_assert(sizeof(DevShape) == 25 * 4, "DevShape must be 25 dwords");

// What a scan over the shapes reads of each shape, as one 64-byte record so a
// single 16-dword scalar load fetches it (and the next record can be requested
// while the current one is tested).
struct alignas(64) ScanRec {
    float c[3], lo[3], hi[3];
    float rsq;
    int32_t type, rotated, same_mask, mat_type, mat;
    int32_t index;   // the shape's index (DevScene::scan_kind and the hierarchy walks' copies); 0 in DevScene::scan
};
static_assert(sizeof(ScanRec) == 64, "ScanRec must be 16 dwords");

struct DevScene {
    float cam_pos[3], cam_fwd[3], cam_right[3], cam_up[3];
    float rand_samp, max_angle_v, max_angle_h, focal;
    int32_t num_shapes, max_depth, res_v, res_h;
    int32_t num_imp, want_features;
    int32_t imp[PTK_MAX_IMPORTANT];
    float inv_depth;   // 1.0f / (float)max_depth
    float imp_boost;   // (float)num_imp when num_imp > 1, else 1
    int32_t ordered_boxes;   // every box has lo <= hi on every axis (the merged slab test needs it)
    DevShape shape[PTK_MAX_SHAPES];
    float mats[PTK_MAX_MATS][6];
    // Per material, what the dielectric branch derives from the refractive index alone, formed
    // on the host by the kernel's fp32 operations: eta entering (1 / index), eta leaving
    // (1 / that), and the Schlick term ((1 - eta) / (1 + eta))^2 of each.
    float mat_glass[PTK_MAX_MATS][4];
    // Per shape, the sixteen sums of quaternion components that the two
    // quaternion products of a world-to-box rotation form from the box's
    // quaternion alone (same fp32 additions, done once on the host): [0..7] for
    // (0,v) * q, [8..15] for conj(q) * (that product).  Read only for rotated boxes.
    float rot[PTK_MAX_SHAPES][16];
    // Per box, the world-space normal of each face code (0 none, 1 -x, 2 +x,
    // 3 -y, 4 +y, 5 -z, 6 +z): the box-space unit normal rotated by conj(q) and
    // normalised, evaluated on the host with the kernel's fp32 operations in
    // the kernel's order (Renderers.cpp:25-47, 262-267).
    float face_normal[PTK_MAX_SHAPES][7][3];
    ScanRec scan[PTK_MAX_SHAPES + 1];   // one past the end so the look-ahead load is always valid
    // The same records grouped by kind -- unrotated boxes, rotated boxes, spheres, each group in
    // index order and carrying its index -- so that a scan can run one straight loop per kind.
    // The closest-hit scan always may (ties go to the lower index either way); a shadow scan
    // may when no shape refracts (any_dielectric == 0), because then its order does not matter.
    ScanRec scan_kind[PTK_MAX_SHAPES + 1];
    int32_t n_plain, n_rotated, n_spheres, any_dielectric;
    // The records of scan_kind once more, for rays that leave the camera: every such ray has
    // the same origin, so what a shape test forms from the origin and the shape alone is the
    // same for every lane and every sample, and the host forms it once with the kernel's fp32
    // operations in the kernel's order.  Boxes: c = the origin in the box's frame ((o - c) + c,
    // rotated first for a rotated box), lo / hi = the slab numerators lo - c' and hi - c'.
    // Spheres: c = centre - origin, lo[0] = its squared length.  Everything else as scan_kind.
    ScanRec cam_kind[PTK_MAX_SHAPES + 1];
    // For shadow scans of scenes WITH refracting shapes: a refracting shape bends the ray for
    // the shapes after it, so the order is free only between two of them.  scan_seg holds, for
    // each stretch of the index order that ends at a refracting shape (or at the last shape),
    // the stretch's other shapes grouped by kind and then that refracting shape; seg[i] says
    // how many of each there are.
    ScanRec scan_seg[PTK_MAX_SHAPES + 1];
    struct Segment { int32_t n_plain, n_rotated, n_spheres, ends_refracting; } seg[PTK_MAX_SHAPES + 1];
    int32_t n_segments, pad_seg[3];
};

// ---- scenes beyond the POD's limits (ptk_set_scene): the shapes live in HBM
// and rays find them through a bounding-volume hierarchy.
//
// A node holds the (padded) bounds of both children, so one 32-byte fetch
// decides both: twelve 16-bit coordinates on a grid spanning the scene
// (BigScene::grid_base + q * grid_scale), lower bounds rounded down and upper
// bounds up.  child >= 0: node index.  child < 0: a leaf of one shape,
// ~child = shape * 4 + kind (0 sphere, 1 box, 2 rotated box, 3 nothing).
struct alignas(32) BvhNode {
    uint16_t q[12];      // child 0 lo xyz, hi xyz; child 1 lo xyz, hi xyz
    int32_t child[2];
};
static_assert(sizeof(BvhNode) == 32, "BvhNode must be 8 dwords");
#define PTK_BVH_STACK 64      // bound on the binary tree's depth (+2); the builder keeps trees shallower

// What the kernels walk: the binary tree with every other level folded away.  A node holds
// up to four children -- the children of the binary node's two children; a child that is a
// leaf in the binary tree stays one slot -- as four boxes of six 16-bit grid coordinates
// (three dwords per child: lo.x | lo.y << 16, lo.z | hi.x << 16, hi.y | hi.z << 16) and four
// child codes (node index, or ~(shape * 4 + kind) as in BvhNode; kind 3 = empty slot).
// Slots 0, 1 come from the binary node's first child and 2, 3 from its second, so the walk
// can keep the binary tree's order (nearer subtree first, nearer child first within it)
// with two comparisons.  One 64-byte fetch decides four boxes and a walk makes half as many
// dependent fetches as in the binary tree.
struct alignas(64) BvhNode4 {
    uint32_t box[4][3];
    int32_t child[4];
};
static_assert(sizeof(BvhNode4) == 64, "BvhNode4 must be 16 dwords");
#define PTK_BVH4_STACK 128    // traversal stack entries per lane: three per level of at most 31 levels

// What a shape test reads, by shape index; a sphere needs the first 16 bytes only.
struct alignas(16) LeafRec {
    float c[3], rsq;
    float lo[3], hi_x;
    float hi_y, hi_z;
    int32_t same_mask, pad;
};
static_assert(sizeof(LeafRec) == 48, "LeafRec must be 12 dwords");

// One face of an unrotated box on one axis: the plane's coordinate and the box's leaf code.
// Sorted by coordinate per axis (see bvh_walk in bvh_traverse.inc for what they are for).
struct FacePlane {
    float coord;
    int32_t code;
};

struct BigScene {
    const BvhNode* nodes;        // the binary tree; node 0 is the root
    const BvhNode4* nodes4;      // the same tree, four children per node; node 0 is the root
    const LeafRec* recs;         // [shape]
    const float* rot;            // [shape][16], as DevScene::rot
    const float* face_normal;    // [shape][7][3], as DevScene::face_normal
    const float* shapes;         // [shape] DevShape
    const float* mats;           // [material][6]
    const int32_t* imp;          // important shapes
    const FacePlane* face_planes; // three sorted runs: x faces, y faces, z faces
    int32_t face_plane_first[3], face_plane_count[3];
    float face_plane_tol;        // how far a ray origin's coordinate may lie from a plane it counts as being in
    float grid_base[3], grid_scale[3];
    int32_t any_dielectric;      // some shape's material is of type 3
    int32_t visit_all;           // every node test passes: each ray tests every shape
};

struct RenderArgs {
    const DevScene* scene;   // device memory
    float* planes;           // PTK_N_PLANES * P floats
    const int64_t* seeds;    // seed mode 0: [pass][pixel - seed_pixel0][2]; null in counter mode
    unsigned long long* rays; // device counter, incremented by the call's bounce rays
    uint64_t stream_id;      // counter mode
    int32_t width, height;
    int32_t spp;             // passes in this launch
    int32_t base_count;      // sampleCount before this launch
    int32_t has_blur;        // some shape's material has blur > 0
    // Rows [row_begin, row_end) are rendered (the image, or a band of it when one pass of
    // reference seeds is staged in pieces); row_begin is a multiple of 16.  The seed of pass s
    // for pixel p is at seeds[(s * seed_stride + p - seed_pixel0) * 2].
    int32_t row_begin, row_end;
    int64_t seed_pixel0, seed_stride;
    // Work items of a launch (filled by the launcher, except sched and chunk_spp): the band's
    // 16x16-pixel tiles times the chunks of chunk_spp samples the launch's spp are cut into.
    // sched: device memory, 1 + tiles words: [0] the next item to hand out, [1 + t] the chunks
    // of tile t that are finished; zeroed by the launcher.  chunk_spp on entry: 0 = the
    // launcher chooses, otherwise the chunk length to use.
    unsigned* sched;
    int32_t chunk_spp, n_chunks, n_items, tiles_x, tiles_y;
    BigScene big;            // ptk_set_scene scenes only (render_bvh_*.hip)
};

// Launchers (defined next to their kernels).  All enqueue on `stream` and
// return the launch status; none synchronises or allocates.
hipError_t ptk_launch_render_exact(const RenderArgs& a, int max_depth, bool features,
                                   hipStream_t stream);
hipError_t ptk_launch_render_fast(const RenderArgs& a, int max_depth, bool features,
                                  hipStream_t stream);
// The same for scenes in which no shape refracts (DevScene::any_dielectric == 0): scans by kind of shape.
hipError_t ptk_launch_render_kind_exact(const RenderArgs& a, int max_depth, bool features,
                                        hipStream_t stream);
hipError_t ptk_launch_render_kind_fast(const RenderArgs& a, int max_depth, bool features,
                                       hipStream_t stream);
hipError_t ptk_launch_render_bvh_exact(const RenderArgs& a, int max_depth, bool features,
                                       hipStream_t stream);
hipError_t ptk_launch_render_bvh_fast(const RenderArgs& a, int max_depth, bool features,
                                      hipStream_t stream);

// Luminance map + sum of planes 0..2.  sum_d holds PTK_EXPOSURE_SHARDS partial sums
// PTK_EXPOSURE_SHARD_STRIDE doubles apart (all zero on entry); the total is their sum.
#define PTK_EXPOSURE_SHARDS 32
#define PTK_EXPOSURE_SHARD_STRIDE 16
#define PTK_EXPOSURE_SUM_DOUBLES (PTK_EXPOSURE_SHARDS * PTK_EXPOSURE_SHARD_STRIDE)
hipError_t ptk_launch_exposure(const float* planes, size_t pixels, int formula, double* sum_d,
                               hipStream_t stream);

// The same reduce with the shard sums in a file-scope device array and the result
// finished on the device (exposure_d / raw_sum_d: device memory, either may be
// null): enqueues a memset and two kernels, allocates nothing.
// ptk_exposure_static_setup looks up that array's address on the current device
// (once per device) and is also called by the launcher.
hipError_t ptk_exposure_static_setup(double** ptr_out);
hipError_t ptk_launch_exposure_static(const float* planes, size_t pixels, int width, int height,
                                      int sample_count, int formula, float* exposure_d,
                                      double* raw_sum_d, hipStream_t stream);

// stdDev[k] = (sdv[3k] + sdv[3k+1] + sdv[3k+2]) / n for k < 5, sdv[15] / n for k = 5.
hipError_t ptk_launch_refresh_std_dev(float* planes, size_t pixels, int sample_count,
                                      hipStream_t stream);

struct PostArgs {
    const float* planes;     // accumulators
    const float* denoised;   // 3 planes (metric 1)
    const float* target;     // 3 planes (metric 7: the training target)
    float* post;             // 3 planes out
    uint8_t* rgb8;           // 3 bytes per pixel, top row first; may be null
    size_t pixels;
    int32_t width, height, sample_count, metric;
    float exposure, inv_gamma;
};
hipError_t ptk_launch_post_process(const PostArgs& a, hipStream_t stream);

// the denoiser's weights as uploaded: first layer 360, second 100, third 10 rows of 8
constexpr int PTK_DEV_WEIGHTS = 540;

struct DenoiseArgs {
    const float* planes;
    const float* weights;   // PTK_DEV_WEIGHTS floats in the kernels' layout (denoise.hip), 8-byte aligned
    float* variances;       // 7 planes, plane-major (device)
    float* features;        // 36 planes, plane-major, or null
    float* layers;          // 20 planes (second- and third-layer activations), or null
    float* denoised;        // 3 planes, plane-major
    float* wc_sum;          // 1 plane
    int32_t width, height;
    int32_t sample_count;
    int32_t n;              // filter half-width
    int32_t row_begin, row_end;  // rows [row_begin, row_end) are produced (a band, or the image)
};
// fast: the tolerance-held form of the pass (smooth statistics formed a cheaper way; the
// discrete ones unchanged).
hipError_t ptk_launch_features_mlp(const DenoiseArgs& a, hipStream_t stream, bool fast = false);

// One training step's reductions (DenoiserNN::GenRelMSE and CPUBackProp,
// DenoiserNN.cpp:959-1103, 1461-1474).  sums (device, PTK_TRAIN_SUM_DOUBLES
// doubles, zeroed by the caller) receives, sharded over PTK_TRAIN_SHARDS
// copies that the host adds up: [0] the relMSE sum, [1..36] sum of G * feature,
// [37..46] sum of G * layer-two value, [47..56] sum of G * layer-three value,
// where G is the pixel's sum over the seven filter parameters of
// dot(error over colour, colour over parameter).
#define PTK_TRAIN_TERMS 57
#define PTK_TRAIN_SHARDS 16
#define PTK_TRAIN_SUM_DOUBLES (PTK_TRAIN_TERMS * PTK_TRAIN_SHARDS)
struct TrainArgs {
    const float* planes;     // accumulators
    const float* denoised;   // 3 planes
    const float* wc_sum;     // 1 plane
    const float* variances;  // 7 planes
    const float* features;   // 36 planes
    const float* layers;     // 20 planes
    const float* target;     // 3 planes
    double* sums;
    int32_t width, height, sample_count, n;
};
hipError_t ptk_launch_train(const TrainArgs& a, hipStream_t stream);
// target = colour sums / sample count (Denoiser::saveTargetCol, Denoiser.cpp:148-155)
hipError_t ptk_launch_save_target(const float* planes, float* target, size_t pixels,
                                  int sample_count, hipStream_t stream);
// lds_tiles: N = 1 also takes the LDS-tile kernel that serves the wider windows (same
// arithmetic, kept callable so the two forms can be compared bit for bit).
hipError_t ptk_launch_filter(const DenoiseArgs& a, hipStream_t stream, bool lds_tiles = false);

// Merge of the accumulators received from every rank for this rank's row span (bands.hip).
// recv: [world][32 planes: 16 sums, 16 squared deviations][rows_max][width]; rows
// [0, row_hi - row_lo) of each plane block are this rank's span as source r rendered it.
#define PTK_MAX_RANKS 64
struct BandMergeArgs {
    const float* recv;
    float* planes;          // PTK_N_PLANES * P floats, rows [row_lo, row_hi) are written
    int32_t width, height, row_lo, row_hi, rows_max, world;
    float n_total;
    float n_rank[PTK_MAX_RANKS];
};
hipError_t ptk_launch_merge_received(const BandMergeArgs& a, hipStream_t stream);
