// ptk_api.cpp -- host side of libptk: context, HBM buffers, scene compilation,
// seed generation, launches.  Everything here runs on the caller's thread;
// device work is enqueued on one stream per context.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "bvh_build.h"
#include "ptk_dev.h"

static_assert(sizeof(ptk_constants) == PTK_CONSTANTS_BYTES, "ptk_constants must match the reference POD");

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return fail(PTK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                             \
    } while (0)

// Reference seeds (16 bytes per pixel per pass: 133 MB at 4K, 531 MB at 8K) are staged in
// pieces of at most this many bytes, device and pinned host each: several whole passes, or
// one pass in bands of rows when a pass is larger.  The environment variable
// PTK_SEED_CHUNK_BYTES, read when a context is created, overrides it (the tests use it to
// drive the band path at small sizes).
constexpr size_t SEED_CHUNK_BYTES = size_t(256) << 20;

}  // namespace

struct ptk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};   // ev[2]: the denoiser's second launch

    bool have_scene = false;
    DevScene host_scene;
    DevScene* scene_d = nullptr;
    int width = 0, height = 0;
    size_t pixels = 0;
    int max_depth = 4;
    bool features = false;
    int math_mode = PTK_MATH_EXACT;
    uint64_t stream_id = 0;
    int sample_count = 0;

    // scenes set through ptk_set_scene (device arrays, owned)
    bool big_scene = false;
    BigScene big = {};
    void* big_alloc[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool has_blur = false;
    bool ordered_boxes = true;        // every box has lo <= hi on every axis
    double big_extent = 0.0;          // what a ptk_set_scene scene's bounds were padded for
    double big_cam_lo[3] = {0, 0, 0}, big_cam_hi[3] = {0, 0, 0};   // where its camera may move

    float* planes_d = nullptr;        // PTK_N_PLANES * pixels
    int64_t* seeds_d = nullptr;       // seed chunk
    int64_t* seeds_h = nullptr;       // pinned staging for the chunk
    size_t seed_capacity = 0;         // in seeds (16 bytes each)
    size_t seed_chunk_bytes = SEED_CHUNK_BYTES;
    bool segment_scans_only = false;  // PTK_SEGMENT_SCANS=1: the kernels with the bending code for every scene (tests)
    unsigned long long* rays_d = nullptr;
    unsigned* sched_d = nullptr;      // path-trace work items: 1 + (16x16-pixel tiles of the image) words
    int render_chunk_spp = 0;         // PTK_RENDER_CHUNK_SPP: samples per work item (0: the launcher chooses)
    double* sum_d = nullptr;

    float* weights_d = nullptr;       // PTK_DEV_WEIGHTS floats
    float weights_up[PTK_DEV_WEIGHTS];   // what weights_d holds, in the kernels' layout
    bool weights_valid = false;
    float* variances_d = nullptr;     // 7 planes
    float* features_d = nullptr;      // 36 planes (allocated on demand)
    float* denoised_d = nullptr;      // 3 planes
    float* wcsum_d = nullptr;         // 1 plane
    float* layers_d = nullptr;        // 20 planes (allocated on the first ptk_train_step)
    float* target_d = nullptr;        // 3 planes (allocated when a target is set)
    double* train_sums_d = nullptr;   // PTK_TRAIN_SUM_DOUBLES
    bool have_target = false;
    float* post_d = nullptr;          // 3 planes (allocated on the first ptk_post_process)
    uint8_t* rgb8_d = nullptr;        // 3 bytes per pixel (likewise)
    bool have_denoised = false;
    bool have_features = false;       // planes 3..37 hold feature sums and statistics
    float* staging_h = nullptr;       // pinned, for layout conversion on read-back
    size_t staging_floats = 0;

    ptk_times times;
};

namespace {

int bind(ptk_ctx* c)
{
    HIP_TRY(hipSetDevice(c->device));
    return PTK_OK;
}

void free_image_buffers(ptk_ctx* c)
{
    (void)hipFree(c->planes_d); c->planes_d = nullptr;
    (void)hipFree(c->sched_d); c->sched_d = nullptr;
    (void)hipFree(c->seeds_d); c->seeds_d = nullptr;
    if (c->seeds_h) (void)hipHostFree(c->seeds_h);
    c->seeds_h = nullptr;
    c->seed_capacity = 0;
    (void)hipFree(c->variances_d); c->variances_d = nullptr;
    (void)hipFree(c->features_d); c->features_d = nullptr;
    (void)hipFree(c->denoised_d); c->denoised_d = nullptr;
    (void)hipFree(c->wcsum_d); c->wcsum_d = nullptr;
    (void)hipFree(c->post_d); c->post_d = nullptr;
    (void)hipFree(c->layers_d); c->layers_d = nullptr;
    (void)hipFree(c->target_d); c->target_d = nullptr;
    c->have_target = false;
    (void)hipFree(c->rgb8_d); c->rgb8_d = nullptr;
    c->have_denoised = false;
    if (c->staging_h) (void)hipHostFree(c->staging_h);
    c->staging_h = nullptr;
    c->staging_floats = 0;
}

void free_big_scene(ptk_ctx* c)
{
    for (void*& p : c->big_alloc) {
        (void)hipFree(p);
        p = nullptr;
    }
    c->big = BigScene{};
    c->big_scene = false;
}

// (Re)allocates the per-image buffers when the resolution changed.
int ensure_image(ptk_ctx* c, int width, int height)
{
    if (width == c->width && height == c->height && c->planes_d) return PTK_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_image_buffers(c);
    c->width = width;
    c->height = height;
    c->pixels = (size_t)width * (size_t)height;
    const size_t P = c->pixels;
    hipError_t e = hipMalloc((void**)&c->planes_d, PTK_N_PLANES * P * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&c->variances_d, 7 * P * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&c->denoised_d, 3 * P * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&c->wcsum_d, P * sizeof(float));
    const size_t tiles = (size_t)((width + 15) / 16) * (size_t)((height + 15) / 16);
    if (e == hipSuccess) e = hipMalloc((void**)&c->sched_d, (1 + tiles) * sizeof(unsigned));
    if (e != hipSuccess) {
        free_image_buffers(c);
        c->width = c->height = 0;
        c->pixels = 0;
        c->have_scene = false;
        return fail(PTK_ERR_NOMEM, "allocating %dx%d image planes: %s", width, height,
                    hipGetErrorString(e));
    }
    return PTK_OK;
}

int ensure_staging(ptk_ctx* c, size_t floats)
{
    if (c->staging_floats >= floats) return PTK_OK;
    if (c->staging_h) (void)hipHostFree(c->staging_h);
    c->staging_h = nullptr;
    c->staging_floats = 0;
    HIP_TRY(hipHostMalloc((void**)&c->staging_h, floats * sizeof(float), hipHostMallocDefault));
    c->staging_floats = floats;
    return PTK_OK;
}

// Scene POD -> device layout.  Derived values are computed here in fp32 with
// the reference's operation order so the kernel reads what it would have
// computed itself (this translation unit is built without FMA contraction).
// The reference's quaternion product and rotation (Renderers.cpp:25-47) and its
// normalisation, in fp32 with no contraction (this file is built with
// -ffp-contract=off): the same operations, in the same order, as the kernel's.
struct Quat { float w, x, y, z; };
Quat quat_mul_host(Quat l, Quat r)
{
    float a1 = (l.z + l.x) * (r.x + r.y);
    float a3 = (l.w - l.y) * (r.w + r.z);
    float a4 = (l.w + l.y) * (r.w - r.z);
    float a2 = a1 + a3 + a4;
    float a5 = (l.z - l.x) * (r.x - r.y);
    a5 = (a5 + a2) / 2.0f;
    Quat o;
    o.w = a5 - a1 + (l.z - l.y) * (r.y - r.z);
    o.x = a5 - a2 + (l.x + l.w) * (r.x + r.w);
    o.y = a5 - a3 + (l.w - l.x) * (r.y + r.z);
    o.z = a5 - a4 + (l.z + l.y) * (r.w - r.x);
    return o;
}
void face_normal_host(const float q[4], const float n[3], float out[3])
{
    Quat p = {0.0f, n[0], n[1], n[2]};
    Quat qc = {q[0], -q[1], -q[2], -q[3]};     // the normal is rotated back by the conjugate
    Quat qcc = {qc.w, -qc.x, -qc.y, -qc.z};
    p = quat_mul_host(p, qc);
    qcc = quat_mul_host(qcc, p);
    volatile float len2 = qcc.x * qcc.x + qcc.y * qcc.y + qcc.z * qcc.z;
    float len = std::sqrt((float)len2);
    out[0] = qcc.x / len;
    out[1] = qcc.y / len;
    out[2] = qcc.z / len;
}

// One shape: the device record, the quaternion-only sums, the face normals and
// the scan record (all zero on entry).  `at` is the shape's attribute run in the
// POD's order: sphere centre, r; box centre, lo, hi, quaternion w,x,y,z.
void compile_shape(int type, int mat, int mat_type, const float* at, DevShape& o, float rot[16],
                   float face_normal[7][3], ScanRec& r)
{
    const float E = 0.00001f;
    o.type = type;
    o.mat = mat;
    o.mat_type = mat_type;
    o.flat_axis = -1;
    o.c[0] = at[0]; o.c[1] = at[1]; o.c[2] = at[2];
    if (type == 0) {
        o.r = at[3];
        o.rsq = at[3] * at[3];
        o.q[0] = 1.0f;
    } else {
        for (int i = 0; i < 3; i++) { o.lo[i] = at[3 + i]; o.hi[i] = at[6 + i]; }
        for (int i = 0; i < 4; i++) o.q[i] = at[9 + i];
        o.rotated = (at[10] + at[11] + at[12] > E) ? 1 : 0;
        {
            // r = q in the first product, l = conj(q) in the second
            // (Renderers.cpp:25-47); every entry is one fp32 operation
            const float rw = o.q[0], rx = o.q[1], ry = o.q[2], rz = o.q[3];
            const float lw = o.q[0], lx = -o.q[1], ly = -o.q[2], lz = -o.q[3];
            float* t = rot;
            t[0] = rx + ry; t[1] = rw + rz; t[2] = rw - rz; t[3] = rx - ry;
            t[4] = ry - rz; t[5] = rx + rw; t[6] = ry + rz; t[7] = rw - rx;
            t[8] = lz + lx; t[9] = lw - ly; t[10] = lw + ly; t[11] = lz - lx;
            t[12] = lz - ly; t[13] = lx + lw; t[14] = lw - lx; t[15] = lz + ly;
        }
        for (int face = 0; face < 7; face++) {
            float n[3] = {0.0f, 0.0f, 0.0f};
            if (face > 0) n[(face - 1) / 2] = (face & 1) ? -1.0f : 1.0f;
            face_normal_host(o.q, n, face_normal[face]);
        }
        for (int i = 2; i >= 0; i--)
            if (std::fabs(o.lo[i] - o.hi[i]) < E) o.flat_axis = i;
        for (int i = 0; i < 3; i++)
            if (o.lo[i] == o.hi[i]) o.same_mask |= 1 << i;
        // importance pdf area (Renderers.cpp:691-697)
        float dx = at[3] - at[6], dy = at[4] - at[7], dz = at[5] - at[8];
        float area = 0;
        area += dx * dy * 2.0f;
        area += dz * dy * 2.0f;
        area += dx * dz * 2.0f;
        o.area = area;
    }
    for (int i = 0; i < 3; i++) { r.c[i] = o.c[i]; r.lo[i] = o.lo[i]; r.hi[i] = o.hi[i]; }
    r.rsq = o.rsq;
    r.type = o.type;
    r.rotated = o.rotated;
    r.same_mask = o.same_mask;
    r.mat_type = o.mat_type;
    r.mat = o.mat;
}

// quat_rotate(v, q) through the quaternion-only sums of compile_shape, as the kernel's
// quat_rotate_sums forms it: the same fp32 operations in the same order.
void quat_rotate_sums_host(const float v[3], const float t[16], float out[3])
{
    const float lw = 0.0f;
    const float vx = v[0], vy = v[1], vz = v[2];
    float a1 = (vz + vx) * t[0];
    float a3 = (lw - vy) * t[1];
    float a4 = (lw + vy) * t[2];
    float a2 = a1 + a3 + a4;
    float a5 = (vz - vx) * t[3];
    a5 = (a5 + a2) / 2.0f;
    const float pw = a5 - a1 + (vz - vy) * t[4];
    const float px = a5 - a2 + (vx + lw) * t[5];
    const float py = a5 - a3 + (lw - vx) * t[6];
    const float pz = a5 - a4 + (vz + vy) * t[7];
    float b1 = t[8] * (px + py);
    float b3 = t[9] * (pw + pz);
    float b4 = t[10] * (pw - pz);
    float b2 = b1 + b3 + b4;
    float b5 = t[11] * (px - py);
    b5 = (b5 + b2) / 2.0f;
    out[0] = b5 - b2 + t[13] * (px + pw);
    out[1] = b5 - b3 + t[14] * (py + pz);
    out[2] = b5 - b4 + t[15] * (pw - px);
}

// DevScene::cam_kind: the scan records as a ray from the camera position meets them.
void compile_camera_records(DevScene& d)
{
    const float* o = d.cam_pos;
    for (int k = 0; k < d.num_shapes; k++) {
        const ScanRec& r = d.scan_kind[k];
        ScanRec& q = d.cam_kind[k];
        q = r;
        if (r.type == 1) {
            float rp[3];
            if (r.rotated) {
                const float rel[3] = {o[0] - r.c[0], o[1] - r.c[1], o[2] - r.c[2]};
                quat_rotate_sums_host(rel, d.rot[r.index], rp);
                for (int i = 0; i < 3; i++) rp[i] = rp[i] + r.c[i];
            } else {
                for (int i = 0; i < 3; i++) rp[i] = (o[i] - r.c[i]) + r.c[i];
            }
            for (int i = 0; i < 3; i++) {
                q.c[i] = rp[i];
                q.lo[i] = r.lo[i] - rp[i];
                q.hi[i] = r.hi[i] - rp[i];
            }
        } else {
            const float l[3] = {r.c[0] - o[0], r.c[1] - o[1], r.c[2] - o[2]};
            for (int i = 0; i < 3; i++) q.c[i] = l[i];
            q.lo[0] = l[0] * l[0] + l[1] * l[1] + l[2] * l[2];
        }
    }
}

int compile_scene(const ptk_constants& k, int want_features, DevScene& d)
{
    if (k.res_h <= 0 || k.res_v <= 0 || (int64_t)k.res_h * k.res_v > (int64_t)1 << 28)
        return fail(PTK_ERR_ARG, "resolution %dx%d out of range", k.res_h, k.res_v);
    if (k.num_shapes < 0 || k.num_shapes > PTK_MAX_SHAPES)
        return fail(PTK_ERR_ARG, "num_shapes %d outside [0,%d]", k.num_shapes, PTK_MAX_SHAPES);
    if (k.max_depth < 1 || k.max_depth > PTK_MAX_DEPTH)
        return fail(PTK_ERR_ARG, "max_depth %d outside [1,%d]", k.max_depth, PTK_MAX_DEPTH);
    if (k.num_important_shapes > PTK_MAX_IMPORTANT)
        return fail(PTK_ERR_ARG, "num_important_shapes %u > %d", k.num_important_shapes,
                    PTK_MAX_IMPORTANT);

    std::memset(&d, 0, sizeof d);
    for (int i = 0; i < 3; i++) {
        d.cam_pos[i] = k.cam_pos[i];
        d.cam_fwd[i] = k.cam_forward[i];
        d.cam_right[i] = k.cam_right[i];
        d.cam_up[i] = k.cam_up[i];
    }
    d.rand_samp = k.rand_samp;
    d.max_angle_v = k.max_angle_v;
    d.max_angle_h = k.max_angle_h;
    d.focal = k.focal_length;
    d.num_shapes = k.num_shapes;
    d.max_depth = k.max_depth;
    d.res_v = k.res_v;
    d.res_h = k.res_h;
    d.num_imp = (int)k.num_important_shapes;
    d.want_features = want_features;
    d.inv_depth = 1.0f / (float)k.max_depth;
    d.imp_boost = k.num_important_shapes > 1 ? (float)k.num_important_shapes : 1.0f;
    for (uint32_t i = 0; i < k.num_important_shapes; i++) {
        if (k.important_shapes[i] >= (uint32_t)k.num_shapes)
            return fail(PTK_ERR_ARG, "important shape %u = %u is not a shape index", i,
                        k.important_shapes[i]);
        d.imp[i] = (int)k.important_shapes[i];
    }
    std::memcpy(d.mats, k.mat_list, sizeof d.mats);
    for (int m = 0; m < PTK_MAX_MATS; m++) {   // Renderers.cpp:330-346, per side of the surface
        const float eta_in = 1.0f / d.mats[m][4];
        const float eta_out = 1.0f / eta_in;
        float r_in = (1.0f - eta_in) / (1.0f + eta_in);
        float r_out = (1.0f - eta_out) / (1.0f + eta_out);
        d.mat_glass[m][0] = eta_in;
        d.mat_glass[m][1] = eta_out;
        d.mat_glass[m][2] = r_in * r_in;
        d.mat_glass[m][3] = r_out * r_out;
    }

    for (int s = 0; s < k.num_shapes; s++) {
        int type = k.shapes[s][0], mat = k.shapes[s][1], a = k.shapes[s][2];
        if (type != 0 && type != 1) return fail(PTK_ERR_ARG, "shape %d has type %d", s, type);
        if (mat < 0 || mat >= PTK_MAX_MATS)
            return fail(PTK_ERR_ARG, "shape %d has material %d", s, mat);
        int need = type == 0 ? 4 : 13;
        if (a < 0 || a + need > PTK_MAX_ATTRS)
            return fail(PTK_ERR_ARG, "shape %d attribute range [%d,%d) exceeds %d", s, a, a + need,
                        PTK_MAX_ATTRS);
        compile_shape(type, mat, (int)k.mat_list[mat][5], k.obj_attributes + a, d.shape[s], d.rot[s],
                      d.face_normal[s], d.scan[s]);
    }
    // the records again, grouped by kind (see DevScene::scan_kind)
    int at = 0;
    d.n_plain = d.n_rotated = d.n_spheres = d.any_dielectric = 0;
    for (int kind = 0; kind < 3; kind++)
        for (int s = 0; s < k.num_shapes; s++) {
            const ScanRec& r = d.scan[s];
            const int its = r.type != 1 ? 2 : (r.rotated ? 1 : 0);
            if (its != kind) continue;
            d.scan_kind[at] = r;
            d.scan_kind[at].index = s;
            at++;
            (kind == 0 ? d.n_plain : kind == 1 ? d.n_rotated : d.n_spheres)++;
            if (r.mat_type == 3) d.any_dielectric = 1;
        }
    // and by stretches between refracting shapes (see DevScene::scan_seg)
    at = 0;
    d.n_segments = 0;
    for (int s0 = 0; s0 < k.num_shapes;) {
        int s1 = s0;
        while (s1 < k.num_shapes && d.scan[s1].mat_type != 3) s1++;
        DevScene::Segment& g = d.seg[d.n_segments++];
        g.n_plain = g.n_rotated = g.n_spheres = 0;
        for (int kind = 0; kind < 3; kind++)
            for (int s = s0; s < s1; s++) {
                const ScanRec& r = d.scan[s];
                const int its = r.type != 1 ? 2 : (r.rotated ? 1 : 0);
                if (its != kind) continue;
                d.scan_seg[at] = r;
                d.scan_seg[at].index = s;
                at++;
                (kind == 0 ? g.n_plain : kind == 1 ? g.n_rotated : g.n_spheres)++;
            }
        g.ends_refracting = s1 < k.num_shapes ? 1 : 0;
        if (g.ends_refracting) {
            d.scan_seg[at] = d.scan[s1];
            d.scan_seg[at].index = s1;
            at++;
        }
        s0 = s1 + 1;
    }
    compile_camera_records(d);
    return PTK_OK;
}

// The reference's global seed step (Renderers.cpp:832-838).
inline void seed_step(uint64_t& g0, uint64_t& g1)
{
    uint64_t a = g0, b = g1;
    b ^= a;
    g0 = (a << 49) | ((a >> 15) ^ b ^ (b << 21));
    g1 = (b << 28) | (b >> 36);
}

double elapsed_ms(ptk_ctx* c)
{
    float ms = 0.0f;
    if (hipEventSynchronize(c->ev[1]) != hipSuccess) return 0.0;
    if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) != hipSuccess) return 0.0;
    return (double)ms;
}

}  // namespace

extern "C" {

const char* ptk_last_error(void) { return g_err; }

int ptk_create(int device, ptk_ctx** out)
{
    if (!out) return fail(PTK_ERR_ARG, "ptk_create: out is null");
    *out = nullptr;
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count)
        return fail(PTK_ERR_ARG, "ptk_create: device %d of %d", device, count);
    ptk_ctx* c = new (std::nothrow) ptk_ctx();
    if (!c) return fail(PTK_ERR_NOMEM, "ptk_create: out of host memory");
    c->device = device;
    std::memset(&c->times, 0, sizeof c->times);
    if (const char* v = std::getenv("PTK_SEGMENT_SCANS")) c->segment_scans_only = v[0] == '1';
    if (const char* v = std::getenv("PTK_RENDER_CHUNK_SPP")) {
        const long n = std::strtol(v, nullptr, 10);
        if (n >= 1 && n <= 0x7fffffff) c->render_chunk_spp = (int)n;
    }
    if (const char* v = std::getenv("PTK_SEED_CHUNK_BYTES")) {
        const unsigned long long b = std::strtoull(v, nullptr, 10);
        if (b >= 16) c->seed_chunk_bytes = (size_t)b;
    }
    int rc = bind(c);
    if (rc != PTK_OK) { delete c; return rc; }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev[0]);
    if (e == hipSuccess) e = hipEventCreate(&c->ev[1]);
    if (e == hipSuccess) e = hipEventCreate(&c->ev[2]);
    if (e == hipSuccess) e = hipMalloc((void**)&c->scene_d, sizeof(DevScene));
    if (e == hipSuccess) e = hipMalloc((void**)&c->rays_d, sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void**)&c->sum_d, PTK_EXPOSURE_SUM_DOUBLES * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&c->weights_d, PTK_DEV_WEIGHTS * sizeof(float));
    // this GPU's address of the array behind ptk_auto_exposure_static, while set-up may allocate
    if (e == hipSuccess) e = ptk_exposure_static_setup(nullptr);
    if (e != hipSuccess) {
        ptk_destroy(c);
        return fail(PTK_ERR_HIP, "ptk_create: %s", hipGetErrorString(e));
    }
    *out = c;
    return PTK_OK;
}

int ptk_destroy(ptk_ctx* c)
{
    if (!c) return PTK_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_image_buffers(c);
    free_big_scene(c);
    (void)hipFree(c->scene_d);
    (void)hipFree(c->rays_d);
    (void)hipFree(c->sum_d);
    (void)hipFree(c->weights_d);
    (void)hipFree(c->train_sums_d);
    if (c->ev[0]) (void)hipEventDestroy(c->ev[0]);
    if (c->ev[1]) (void)hipEventDestroy(c->ev[1]);
    if (c->ev[2]) (void)hipEventDestroy(c->ev[2]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PTK_OK;
}

int ptk_set_constants(ptk_ctx* c, const void* constants, size_t nbytes, int want_denoise_features)
{
    if (!c || !constants) return fail(PTK_ERR_ARG, "ptk_set_constants: null argument");
    if (nbytes != PTK_CONSTANTS_BYTES)
        return fail(PTK_ERR_ARG, "ptk_set_constants: expected %d bytes, got %zu",
                    PTK_CONSTANTS_BYTES, nbytes);
    ptk_constants k;
    std::memcpy(&k, constants, sizeof k);
    int feats = want_denoise_features >= 0 ? (want_denoise_features ? 1 : 0)
                                           : (k.get_denoiser_inf == 1 ? 1 : 0);
    DevScene d;
    int rc = compile_scene(k, feats, d);
    if (rc != PTK_OK) return rc;
    if ((rc = bind(c)) != PTK_OK) return rc;

    if ((rc = ensure_image(c, k.res_h, k.res_v)) != PTK_OK) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_big_scene(c);
    c->have_scene = false;   // until the new scene is complete on the device
    c->host_scene = d;
    c->max_depth = k.max_depth;
    c->features = feats != 0;
    c->has_blur = false;
    c->ordered_boxes = true;
    for (int i = 0; i < d.num_shapes; i++) {
        if (d.mats[d.shape[i].mat][3] > 0.0f) c->has_blur = true;
        if (d.shape[i].type == 1)
            for (int k3 = 0; k3 < 3; k3++)
                if (!(d.shape[i].lo[k3] <= d.shape[i].hi[k3])) c->ordered_boxes = false;
    }
    c->host_scene.ordered_boxes = c->ordered_boxes ? 1 : 0;
    HIP_TRY(hipMemcpyAsync(c->scene_d, &c->host_scene, sizeof(DevScene), hipMemcpyHostToDevice,
                           c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_scene = true;
    return ptk_reset_accumulators(c);
}

}  // extern "C"

namespace {

// Everything ptk_set_scene derives from a scene description on the host.
struct HostBigScene {
    DevScene d;
    std::vector<int32_t> imp;
    std::vector<DevShape> shapes;
    std::vector<float> rot, face_normal;
    std::vector<LeafRec> recs;
    std::vector<float> blo, bhi;      // padded bounds per shape
    BvhBuild bvh;
    std::vector<BvhNode4> nodes4;     // the tree the kernels walk (every other level folded away)
    int depth4 = 0;
    std::vector<FacePlane> face_planes;   // unrotated boxes' faces: x run, y run, z run, each sorted
    int face_plane_first[3] = {0, 0, 0}, face_plane_count[3] = {0, 0, 0};
    float face_plane_tol = 0.0f;
    // The camera may move (ptk_set_camera) within this box without a new hierarchy: the bounds
    // of the shapes and of the camera the scene was set with, grown by 2 % of their diagonal;
    // the spheres' bounds are padded for the longest ray that fits it.
    double cam_lo[3] = {0, 0, 0}, cam_hi[3] = {0, 0, 0};
    bool has_blur = false, any_dielectric = false, ordered_boxes = true;
    double build_ms = 0.0;
    double extent = 0.0;
};

int compile_big_scene(const ptk_scene_desc* k, int want_denoise_features, HostBigScene& H)
{
    if (k->res_h <= 0 || k->res_v <= 0 || (int64_t)k->res_h * k->res_v > (int64_t)1 << 28)
        return fail(PTK_ERR_ARG, "resolution %dx%d out of range", k->res_h, k->res_v);
    if (k->num_shapes < 0 || k->num_shapes > PTK_SCENE_MAX_SHAPES)
        return fail(PTK_ERR_ARG, "num_shapes %d outside [0,%d]", k->num_shapes, PTK_SCENE_MAX_SHAPES);
    if (k->num_mats < 1 || k->num_mats > PTK_SCENE_MAX_MATS)
        return fail(PTK_ERR_ARG, "num_mats %d outside [1,%d]", k->num_mats, PTK_SCENE_MAX_MATS);
    if (k->max_depth < 1 || k->max_depth > PTK_MAX_DEPTH)
        return fail(PTK_ERR_ARG, "max_depth %d outside [1,%d]", k->max_depth, PTK_MAX_DEPTH);
    if (k->num_important_shapes < 0 || k->num_important_shapes > k->num_shapes)
        return fail(PTK_ERR_ARG, "num_important_shapes %d outside [0,%d]", k->num_important_shapes,
                    k->num_shapes);
    if (!k->mat_list || (k->num_shapes > 0 && (!k->shape_type || !k->shape_mat || !k->shape_attrs)) ||
        (k->num_important_shapes > 0 && !k->important_shapes))
        return fail(PTK_ERR_ARG, "ptk_set_scene: null table");
    if (k->flags & ~PTK_SCENE_VISIT_ALL) return fail(PTK_ERR_ARG, "ptk_set_scene: flags %d", k->flags);

    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const int n = k->num_shapes;
    const int feats = want_denoise_features > 0 ? 1 : 0;

    DevScene& d = H.d;
    std::memset(&d, 0, sizeof d);
    for (int i = 0; i < 3; i++) {
        d.cam_pos[i] = k->cam_pos[i];
        d.cam_fwd[i] = k->cam_forward[i];
        d.cam_right[i] = k->cam_right[i];
        d.cam_up[i] = k->cam_up[i];
    }
    d.rand_samp = k->rand_samp;
    d.max_angle_v = k->max_angle_v;
    d.max_angle_h = k->max_angle_h;
    d.focal = k->focal_length;
    d.num_shapes = n;
    d.max_depth = k->max_depth;
    d.res_v = k->res_v;
    d.res_h = k->res_h;
    d.num_imp = k->num_important_shapes;
    d.want_features = feats;
    d.inv_depth = 1.0f / (float)k->max_depth;
    d.imp_boost = k->num_important_shapes > 1 ? (float)k->num_important_shapes : 1.0f;

    std::vector<int32_t>& imp = H.imp;
    imp.assign((size_t)k->num_important_shapes + 1, 0);
    for (int i = 0; i < k->num_important_shapes; i++) {
        if (k->important_shapes[i] >= (uint32_t)n)
            return fail(PTK_ERR_ARG, "important shape %d = %u is not a shape index", i,
                        k->important_shapes[i]);
        imp[(size_t)i] = (int32_t)k->important_shapes[i];
    }

    // ---- per-shape records, as for the POD
    std::vector<DevShape>& shapes = H.shapes;
    std::vector<float>& rot = H.rot;
    std::vector<float>& face_normal = H.face_normal;
    shapes.resize((size_t)n + 1);
    rot.assign(((size_t)n + 1) * 16, 0.0f);
    face_normal.assign(((size_t)n + 1) * 21, 0.0f);
    std::memset((void*)shapes.data(), 0, shapes.size() * sizeof(DevShape));
    bool& has_blur = H.has_blur;
    bool& any_dielectric = H.any_dielectric;
    has_blur = any_dielectric = false;
    double extent = 1.0;
    for (int i = 0; i < 3; i++) extent = std::max(extent, (double)std::fabs(k->cam_pos[i]));
    for (int s = 0; s < n; s++) {
        const int type = k->shape_type[s], mat = k->shape_mat[s];
        const float* at = k->shape_attrs + (size_t)s * PTK_SCENE_ATTRS;
        if (type != 0 && type != 1) return fail(PTK_ERR_ARG, "shape %d has type %d", s, type);
        if (mat < 0 || mat >= k->num_mats) return fail(PTK_ERR_ARG, "shape %d has material %d", s, mat);
        for (int i = 0; i < (type == 0 ? 4 : 13); i++)
            if (!std::isfinite(at[i])) return fail(PTK_ERR_ARG, "shape %d attribute %d is not finite", s, i);
        if (type == 1) {
            double q2 = 0.0;
            for (int i = 9; i < 13; i++) q2 += (double)at[i] * at[i];
            if (std::fabs(q2 - 1.0) > 1e-4)
                return fail(PTK_ERR_ARG, "shape %d: quaternion of squared length %g", s, q2);
        }
        const float* m = k->mat_list + (size_t)mat * 6;
        has_blur = has_blur || m[3] > 0.0f;
        any_dielectric = any_dielectric || (int)m[5] == 3;
        if (type == 1)
            for (int i = 0; i < 3; i++)
                if (!(at[3 + i] <= at[6 + i])) H.ordered_boxes = false;
        for (int i = 0; i < (type == 0 ? 3 : 9); i++) extent = std::max(extent, (double)std::fabs(at[i]));
        if (type == 0) extent = std::max(extent, (double)std::fabs(at[3]));
    }

    if (extent > 1e6)
        return fail(PTK_ERR_ARG, "ptk_set_scene: coordinates up to %g exceed 1e6", extent);
    H.extent = extent;
    ptk_parallel_for(n, [&](int s0, int s1) {
        for (int s = s0; s < s1; s++) {
            const int mat = k->shape_mat[s];
            ScanRec unused;     // the fixed-size kernels' record; these scenes use LeafRec
            std::memset((void*)&unused, 0, sizeof unused);
            compile_shape(k->shape_type[s], mat, (int)k->mat_list[(size_t)mat * 6 + 5],
                          k->shape_attrs + (size_t)s * PTK_SCENE_ATTRS, shapes[(size_t)s],
                          &rot[(size_t)s * 16],
                          reinterpret_cast<float(*)[3]>(&face_normal[(size_t)s * 21]), unused);
        }
    });

    // ---- bounds.  A shape test accepts a ray only if the ray passes the shape
    // within the test's rounding error, which grows with the distances involved.
    // For boxes it has two parts: a few 2^-24 of the distances between ray origin,
    // box and hit point (up to 1e-6 of them after a quaternion rotation), which
    // are at most the diagonal of the scene; and two or three units in the last
    // place of the coordinates themselves (the test's (o - c) + c, the bounds
    // rounded to float), which grow with the distance of the scene from the
    // origin.  The pad covers both more than ten times over: 1e-5 of the diagonal
    // plus 2e-6 (sixteen ulps) of the largest coordinate; the few ulps the slab
    // test against a node adds are of the first kind.  For spheres
    // the squared miss distance is a difference of two squares of the distance
    // |l| from the ray's origin to the centre (Renderers.cpp:131-152): in fp32,
    // with u = 2^-24 for one rounding, l.l is off by up to 3 u |l|^2, (l.d)^2 by
    // 7 u |l|^2 (the dot product's own three roundings, doubled by the square, and
    // the square's), the difference by another u, and a direction normalised in
    // fp32 is of unit length only to 3.5 u, which scales (l.d)^2 by up to 7 u more:
    // 18 u |l|^2 < 1.1e-6 |l|^2 in the worst case -- so rays passing within
    // sqrt(r^2 + 1.1e-6 |l|^2) of the centre may be accepted.  Ray origins are
    // the camera and points on shapes, all inside the box that holds the shapes and
    // the camera (grown a little, so that the camera may move: ptk_set_camera), so
    // |l| is at most the distance from the sphere's centre to the farthest corner
    // of that box -- whatever the distance of the scene from the origin.  The first
    // pass below measures the box on unpadded spheres, the second pads them.
    double pad = 2e-5 * extent;       // for the first pass: the diagonal is at most 2 sqrt(3) extent
    std::vector<float>& blo = H.blo;
    std::vector<float>& bhi = H.bhi;
    blo.assign((size_t)n * 3 + 3, 0.0f);
    bhi.assign((size_t)n * 3 + 3, 0.0f);
    double sphere_margin = 0.0;       // 1.1e-6 once the box of ray origins is known
    auto bounds_pass = [&] {
    ptk_parallel_for(n, [&](int s0, int s1) {
    for (int s = s0; s < s1; s++) {
        const DevShape& o = shapes[(size_t)s];
        double lo[3], hi[3];
        if (o.type == 0) {
            const double r = std::fabs((double)o.r);
            // |l| for this sphere: from its centre to the farthest corner of the box of ray origins
            double far2 = 0.0;
            for (int i = 0; i < 3; i++) {
                const double a = std::max((double)o.c[i] - H.cam_lo[i], H.cam_hi[i] - (double)o.c[i]);
                far2 += a * a;
            }
            const double reach = std::sqrt(r * r + sphere_margin * far2) + pad;
            for (int i = 0; i < 3; i++) { lo[i] = o.c[i] - reach; hi[i] = o.c[i] + reach; }
        } else if (!o.rotated) {
            for (int i = 0; i < 3; i++) {
                lo[i] = std::min(o.lo[i], o.hi[i]) - pad;
                hi[i] = std::max(o.lo[i], o.hi[i]) + pad;
            }
        } else {
            // box space b = R (w - c) + c with R v = conj(q) v q; so w = q (b - c) conj(q) + c
            const double w = o.q[0], x = o.q[1], y = o.q[2], z = o.q[3];
            const double R[3][3] = {
                {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)},
                {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)},
                {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
            for (int i = 0; i < 3; i++) { lo[i] = 1e300; hi[i] = -1e300; }
            for (int corner = 0; corner < 8; corner++) {
                double b[3];
                for (int i = 0; i < 3; i++) b[i] = ((corner >> i) & 1 ? o.hi[i] : o.lo[i]) - (double)o.c[i];
                for (int i = 0; i < 3; i++) {
                    const double wv = R[i][0] * b[0] + R[i][1] * b[1] + R[i][2] * b[2] + o.c[i];
                    lo[i] = std::min(lo[i], wv);
                    hi[i] = std::max(hi[i], wv);
                }
            }
            // the quaternion is of unit length only to 1e-4: scale slack on top of the pad
            double span = 0.0;
            for (int i = 0; i < 3; i++) span = std::max(span, hi[i] - lo[i]);
            for (int i = 0; i < 3; i++) { lo[i] -= pad + 4e-4 * span; hi[i] += pad + 4e-4 * span; }
        }
        for (int i = 0; i < 3; i++) {
            blo[(size_t)s * 3 + i] = std::nextafter((float)lo[i], -INFINITY);
            bhi[(size_t)s * 3 + i] = std::nextafter((float)hi[i], INFINITY);
        }
    }
    });
    };
    bounds_pass();
    {
        double lo[3], hi[3], diag2 = 0.0;
        for (int i = 0; i < 3; i++) lo[i] = hi[i] = k->cam_pos[i];
        for (int s = 0; s < n; s++)
            for (int i = 0; i < 3; i++) {
                lo[i] = std::min(lo[i], (double)blo[(size_t)s * 3 + i]);
                hi[i] = std::max(hi[i], (double)bhi[(size_t)s * 3 + i]);
            }
        for (int i = 0; i < 3; i++) diag2 += (hi[i] - lo[i]) * (hi[i] - lo[i]);
        pad = std::min(pad, 1e-5 * 1.05 * std::sqrt(diag2) + 2e-6 * extent);
        const double grow = 0.02 * std::sqrt(diag2) + pad;
        for (int i = 0; i < 3; i++) {
            H.cam_lo[i] = lo[i] - grow;
            H.cam_hi[i] = hi[i] + grow;
        }
        sphere_margin = 1.1e-6;
    }
    bounds_pass();
    std::vector<uint8_t> kind((size_t)n + 1, 0);
    std::vector<LeafRec>& recs = H.recs;
    recs.resize((size_t)n + 1);
    std::memset((void*)recs.data(), 0, recs.size() * sizeof(LeafRec));
    ptk_parallel_for(n, [&](int s0, int s1) {
        for (int s = s0; s < s1; s++) {
            const DevShape& o = shapes[(size_t)s];
            kind[(size_t)s] = o.type == 0 ? 0 : (o.rotated ? 2 : 1);
            LeafRec& r = recs[(size_t)s];
            for (int i = 0; i < 3; i++) { r.c[i] = o.c[i]; r.lo[i] = o.lo[i]; }
            r.rsq = o.rsq;
            r.hi_x = o.hi[0]; r.hi_y = o.hi[1]; r.hi_z = o.hi[2];
            r.same_mask = o.same_mask;
        }
    });
    BvhBuild& bvh = H.bvh;
    if (!ptk_build_bvh(n, blo.data(), bhi.data(), kind.data(), bvh))
        return fail(PTK_ERR_ARG, "ptk_set_scene: hierarchy of depth %d exceeds the traversal stack",
                    bvh.depth);
    ptk_collapse_bvh4(bvh.nodes, H.nodes4, H.depth4);
    {   // ---- face planes of the unrotated boxes, sorted per axis (what for: bvh_walk).  A box
        // test forms rp = (o - c) + c before it subtracts the planes: each of the two roundings
        // is below 2^-24 of a coordinate difference or a coordinate, so an origin within
        // 2^-21 * extent of a plane may be "in" it.
        size_t boxes = 0;
        for (int s = 0; s < n; s++) boxes += kind[(size_t)s] == 1;
        H.face_planes.resize(boxes * 6 + 1);
        for (int a = 0; a < 3; a++) {
            H.face_plane_first[a] = (int)(boxes * 2 * (size_t)a);
            H.face_plane_count[a] = (int)(boxes * 2);
        }
        size_t at = 0;
        for (int s = 0; s < n; s++) {
            if (kind[(size_t)s] != 1) continue;
            const DevShape& o = shapes[(size_t)s];
            for (int a = 0; a < 3; a++) {
                FacePlane* run = H.face_planes.data() + H.face_plane_first[a];
                run[2 * at] = {o.lo[a], s * 4 + 1};
                run[2 * at + 1] = {o.hi[a], s * 4 + 1};
            }
            at++;
        }
        auto sort_axis = [&](int a) {
            FacePlane* run = H.face_planes.data() + H.face_plane_first[a];
            std::sort(run, run + H.face_plane_count[a], [](const FacePlane& x, const FacePlane& y) {
                return x.coord < y.coord || (x.coord == y.coord && x.code < y.code);
            });
        };
        if (boxes >= 16384) {
            std::thread ty([&] { sort_axis(1); }), tz([&] { sort_axis(2); });
            sort_axis(0);
            ty.join();
            tz.join();
        } else {
            for (int a = 0; a < 3; a++) sort_axis(a);
        }
        H.face_plane_tol = (float)(extent * (1.0 / 2097152.0));
    }
    if (3 * H.depth4 > PTK_BVH4_STACK - 2)
        return fail(PTK_ERR_ARG, "ptk_set_scene: hierarchy of %d four-wide levels exceeds the traversal stack",
                    H.depth4);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    H.build_ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
    return PTK_OK;
}

void fill_scene_stats(const HostBigScene& H, uint64_t device_bytes, ptk_scene_stats* stats)
{
    if (!stats) return;
    stats->num_shapes = H.d.num_shapes;
    stats->nodes = (int32_t)H.bvh.nodes.size();
    stats->leaves = H.bvh.leaves;
    stats->depth = H.bvh.depth;
    stats->build_ms = H.build_ms;
    stats->device_bytes = device_bytes;
}

// Walks the tree below one child slot; returns false on any broken invariant:
// a shape in two leaves, a leaf code of the wrong kind, or grid bounds that do
// not contain the padded bounds of everything beneath them.
bool check_subtree(const HostBigScene& H, int child, const uint16_t* q, std::vector<uint8_t>& seen,
                   int depth, int& max_depth)
{
    // iterative, explicit stack of (child code, bounds it must fit in, depth)
    struct Item { int child; double lo[3], hi[3]; int depth; };
    std::vector<Item> todo;
    Item first;
    first.child = child;
    first.depth = depth;
    for (int k = 0; k < 3; k++) {
        first.lo[k] = (double)H.bvh.grid_base[k] + (double)q[k] * H.bvh.grid_scale[k];
        first.hi[k] = (double)H.bvh.grid_base[k] + (double)q[3 + k] * H.bvh.grid_scale[k];
    }
    todo.push_back(first);
    const int n = H.d.num_shapes;
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        max_depth = std::max(max_depth, it.depth);
        if (it.child < 0) {
            const int code = ~it.child;
            if ((code & 3) == 3) continue;
            const int s = code >> 2;
            if (s < 0 || s >= n || seen[(size_t)s]) return false;
            seen[(size_t)s] = 1;
            const DevShape& o = H.shapes[(size_t)s];
            const int kind = o.type == 0 ? 0 : (o.rotated ? 2 : 1);
            if ((code & 3) != kind) return false;
            for (int k = 0; k < 3; k++)
                if (H.blo[(size_t)s * 3 + k] < it.lo[k] || H.bhi[(size_t)s * 3 + k] > it.hi[k]) return false;
            continue;
        }
        if ((size_t)it.child >= H.bvh.nodes.size()) return false;
        const BvhNode& nd = H.bvh.nodes[(size_t)it.child];
        for (int c = 0; c < 2; c++) {
            Item next;
            next.child = nd.child[c];
            next.depth = it.depth + 1;
            const bool empty = nd.child[c] < 0 && ((~nd.child[c]) & 3) == 3;
            for (int k = 0; k < 3; k++) {
                next.lo[k] = (double)H.bvh.grid_base[k] + (double)nd.q[c * 6 + k] * H.bvh.grid_scale[k];
                next.hi[k] = (double)H.bvh.grid_base[k] + (double)nd.q[c * 6 + 3 + k] * H.bvh.grid_scale[k];
                // a child's box may be wider than its parent's (each is rounded outward on
                // its own) but must hold what lies beneath, which the leaves check
            }
            if (!empty) todo.push_back(next);
        }
    }
    return true;
}

// The same invariants for the four-wide tree the kernels walk: every shape in exactly one
// leaf, with the right kind, inside the grid box of its slot; depth as the collapse reported.
bool check_wide_tree(const HostBigScene& H, std::vector<uint8_t>& seen, int& max_depth)
{
    if (H.nodes4.empty()) return H.d.num_shapes == 0 || H.bvh.nodes.empty();
    struct Item { int32_t node, depth; };
    std::vector<Item> todo;
    todo.push_back({0, 1});
    const int n = H.d.num_shapes;
    size_t visited = 0;
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        if (it.node < 0 || (size_t)it.node >= H.nodes4.size() || ++visited > H.nodes4.size()) return false;
        max_depth = std::max(max_depth, (int)it.depth);
        const BvhNode4& nd = H.nodes4[(size_t)it.node];
        for (int s = 0; s < 4; s++) {
            const int32_t ch = nd.child[s];
            if (ch >= 0) {
                todo.push_back({ch, it.depth + 1});
                continue;
            }
            const int code = ~ch;
            if ((code & 3) == 3) continue;                // empty slot
            const int sh = code >> 2;
            if (sh < 0 || sh >= n || seen[(size_t)sh]) return false;
            seen[(size_t)sh] = 1;
            const DevShape& o = H.shapes[(size_t)sh];
            if ((code & 3) != (o.type == 0 ? 0 : (o.rotated ? 2 : 1))) return false;
            const uint32_t* b = nd.box[s];
            const uint32_t q[6] = {b[0] & 0xffffu, b[0] >> 16, b[1] & 0xffffu, b[1] >> 16, b[2] & 0xffffu, b[2] >> 16};
            for (int k = 0; k < 3; k++) {
                const double lo = (double)H.bvh.grid_base[k] + (double)q[k] * H.bvh.grid_scale[k];
                const double hi = (double)H.bvh.grid_base[k] + (double)q[3 + k] * H.bvh.grid_scale[k];
                if (H.blo[(size_t)sh * 3 + k] < lo || H.bhi[(size_t)sh * 3 + k] > hi) return false;
            }
        }
    }
    return visited == H.nodes4.size();
}

}  // namespace

extern "C" {

int ptk_check_scene(const ptk_scene_desc* k, ptk_scene_stats* stats)
{
    if (!k) return fail(PTK_ERR_ARG, "ptk_check_scene: null argument");
    HostBigScene H;
    int rc = compile_big_scene(k, 0, H);
    if (rc != PTK_OK) return rc;
    const int n = H.d.num_shapes;
    std::vector<uint8_t> seen((size_t)n + 1, 0);
    int max_depth = 0;
    const uint16_t whole[6] = {0, 0, 0, 65535, 65535, 65535};
    if (!check_subtree(H, 0, whole, seen, 0, max_depth))
        return fail(PTK_ERR_STATE, "ptk_check_scene: the hierarchy is inconsistent");
    for (int s = 0; s < n; s++)
        if (!seen[(size_t)s]) return fail(PTK_ERR_STATE, "ptk_check_scene: shape %d is in no leaf", s);
    if (max_depth > PTK_BVH_STACK - 2 || max_depth != H.bvh.depth)
        return fail(PTK_ERR_STATE, "ptk_check_scene: depth %d (builder says %d)", max_depth, H.bvh.depth);
    std::fill(seen.begin(), seen.end(), 0);
    int depth4 = 0;
    if (!check_wide_tree(H, seen, depth4))
        return fail(PTK_ERR_STATE, "ptk_check_scene: the four-wide hierarchy is inconsistent");
    for (int s = 0; s < n; s++)
        if (!seen[(size_t)s]) return fail(PTK_ERR_STATE, "ptk_check_scene: shape %d is in no leaf of the four-wide hierarchy", s);
    if (3 * depth4 > PTK_BVH4_STACK - 2 || depth4 != H.depth4)
        return fail(PTK_ERR_STATE, "ptk_check_scene: four-wide depth %d (collapse says %d)", depth4, H.depth4);
    // the face-plane runs: sorted, and both faces of every unrotated box on every axis
    for (int a = 0; a < 3; a++) {
        const FacePlane* run = H.face_planes.data() + H.face_plane_first[a];
        std::vector<uint8_t> faces((size_t)n + 1, 0);
        for (int i = 0; i < H.face_plane_count[a]; i++) {
            const int s = run[i].code >> 2;
            if (i > 0 && run[i].coord < run[i - 1].coord)
                return fail(PTK_ERR_STATE, "ptk_check_scene: face planes of axis %d are not sorted at %d", a, i);
            if ((run[i].code & 3) != 1 || s < 0 || s >= n || H.shapes[(size_t)s].type != 1 || H.shapes[(size_t)s].rotated ||
                (run[i].coord != H.shapes[(size_t)s].lo[a] && run[i].coord != H.shapes[(size_t)s].hi[a]))
                return fail(PTK_ERR_STATE, "ptk_check_scene: face plane %d of axis %d names no face of shape %d", i, a, s);
            faces[(size_t)s]++;
        }
        for (int s = 0; s < n; s++)
            if (faces[(size_t)s] != (H.shapes[(size_t)s].type == 1 && !H.shapes[(size_t)s].rotated ? 2 : 0))
                return fail(PTK_ERR_STATE, "ptk_check_scene: shape %d has %d face planes on axis %d", s, faces[(size_t)s], a);
    }
    fill_scene_stats(H, 0, stats);
    return PTK_OK;
}

int ptk_set_scene(ptk_ctx* c, const ptk_scene_desc* k, int want_denoise_features,
                  ptk_scene_stats* stats)
{
    if (!c || !k) return fail(PTK_ERR_ARG, "ptk_set_scene: null argument");
    HostBigScene H;
    int rc = compile_big_scene(k, want_denoise_features, H);
    if (rc != PTK_OK) return rc;
    const int feats = H.d.want_features;
    const bool has_blur = H.has_blur, any_dielectric = H.any_dielectric;
    const DevScene& d = H.d;
    BvhBuild& bvh = H.bvh;
    std::vector<LeafRec>& recs = H.recs;
    std::vector<float>& rot = H.rot;
    std::vector<float>& face_normal = H.face_normal;
    std::vector<DevShape>& shapes = H.shapes;
    std::vector<int32_t>& imp = H.imp;

    if ((rc = bind(c)) != PTK_OK) return rc;
    if ((rc = ensure_image(c, k->res_h, k->res_v)) != PTK_OK) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_big_scene(c);
    c->have_scene = false;   // until the new scene is complete on the device

    const void* src[9] = {bvh.nodes.data(), recs.data(), rot.data(), face_normal.data(),
                          shapes.data(), k->mat_list, imp.data(), H.nodes4.data(), H.face_planes.data()};
    const size_t bytes[9] = {bvh.nodes.size() * sizeof(BvhNode), recs.size() * sizeof(LeafRec),
                             rot.size() * sizeof(float), face_normal.size() * sizeof(float),
                             shapes.size() * sizeof(DevShape), (size_t)k->num_mats * 6 * sizeof(float),
                             imp.size() * sizeof(int32_t), H.nodes4.size() * sizeof(BvhNode4),
                             H.face_planes.size() * sizeof(FacePlane)};
    uint64_t total = 0;
    for (int i = 0; i < 9; i++) {
        hipError_t e = hipMalloc(&c->big_alloc[i], bytes[i]);
        const bool no_memory = e != hipSuccess;
        if (e == hipSuccess)
            e = hipMemcpyAsync(c->big_alloc[i], src[i], bytes[i], hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) {
            // copies queued so far read host vectors that die with this frame
            (void)hipStreamSynchronize(c->stream);
            free_big_scene(c);
            return fail(no_memory ? PTK_ERR_NOMEM : PTK_ERR_HIP, "ptk_set_scene: table %d (%zu bytes): %s",
                        i, bytes[i], hipGetErrorString(e));
        }
        total += bytes[i];
    }
    c->big.nodes = (const BvhNode*)c->big_alloc[0];
    c->big.recs = (const LeafRec*)c->big_alloc[1];
    c->big.rot = (const float*)c->big_alloc[2];
    c->big.face_normal = (const float*)c->big_alloc[3];
    c->big.shapes = (const float*)c->big_alloc[4];
    c->big.mats = (const float*)c->big_alloc[5];
    c->big.imp = (const int32_t*)c->big_alloc[6];
    c->big.nodes4 = (const BvhNode4*)c->big_alloc[7];
    c->big.face_planes = (const FacePlane*)c->big_alloc[8];
    for (int i = 0; i < 3; i++) {
        c->big.face_plane_first[i] = H.face_plane_first[i];
        c->big.face_plane_count[i] = H.face_plane_count[i];
    }
    c->big.face_plane_tol = H.face_plane_tol;
    for (int i = 0; i < 3; i++) {
        c->big_cam_lo[i] = H.cam_lo[i];
        c->big_cam_hi[i] = H.cam_hi[i];
    }
    for (int i = 0; i < 3; i++) {
        c->big.grid_base[i] = bvh.grid_base[i];
        c->big.grid_scale[i] = bvh.grid_scale[i];
    }
    c->big.any_dielectric = any_dielectric ? 1 : 0;
    c->big.visit_all = (k->flags & PTK_SCENE_VISIT_ALL) ? 1 : 0;
    c->big_scene = true;
    c->big_extent = H.extent;
    c->has_blur = has_blur;
    c->ordered_boxes = H.ordered_boxes;
    c->host_scene = d;
    c->host_scene.ordered_boxes = H.ordered_boxes ? 1 : 0;
    c->max_depth = k->max_depth;
    c->features = feats != 0;
    HIP_TRY(hipMemcpyAsync(c->scene_d, &c->host_scene, sizeof(DevScene), hipMemcpyHostToDevice,
                           c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_scene = true;
    fill_scene_stats(H, total, stats);
    return ptk_reset_accumulators(c);
}

int ptk_set_camera(ptk_ctx* c, const float* pos, const float* fwd, const float* right,
                   const float* up)
{
    if (!c || !pos || !fwd || !right || !up) return fail(PTK_ERR_ARG, "ptk_set_camera: null argument");
    if (!c->have_scene) return fail(PTK_ERR_STATE, "ptk_set_camera before a scene was set");
    for (int i = 0; i < 3; i++) {
        if (!std::isfinite(pos[i]) || !std::isfinite(fwd[i]) || !std::isfinite(right[i]) ||
            !std::isfinite(up[i]))
            return fail(PTK_ERR_ARG, "ptk_set_camera: component %d is not finite", i);
        if (c->big_scene && (std::fabs((double)pos[i]) > c->big_extent || (double)pos[i] < c->big_cam_lo[i] ||
                             (double)pos[i] > c->big_cam_hi[i]))
            return fail(PTK_ERR_ARG,
                        "ptk_set_camera: position %g lies outside [%g, %g] (the bounds of the scene and of the camera "
                        "it was set with, which its hierarchy was built for); call ptk_set_scene again",
                        (double)pos[i], std::max(c->big_cam_lo[i], -c->big_extent),
                        std::min(c->big_cam_hi[i], c->big_extent));
    }
    int rc = bind(c);
    if (rc != PTK_OK) return rc;
    for (int i = 0; i < 3; i++) {
        c->host_scene.cam_pos[i] = pos[i];
        c->host_scene.cam_fwd[i] = fwd[i];
        c->host_scene.cam_right[i] = right[i];
        c->host_scene.cam_up[i] = up[i];
    }
    if (!c->big_scene) compile_camera_records(c->host_scene);
    // the four vectors lead the structure; the records formed from the position end it
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpyAsync(c->scene_d, &c->host_scene, offsetof(DevScene, rand_samp),
                           hipMemcpyHostToDevice, c->stream));
    if (!c->big_scene)
        HIP_TRY(hipMemcpyAsync((char*)c->scene_d + offsetof(DevScene, cam_kind),
                               c->host_scene.cam_kind, sizeof c->host_scene.cam_kind,
                               hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ptk_reset_accumulators(c);
}

int ptk_reset_accumulators(ptk_ctx* c)
{
    if (!c) return fail(PTK_ERR_ARG, "ptk_reset_accumulators: null context");
    if (!c->planes_d) return fail(PTK_ERR_STATE, "ptk_reset_accumulators before ptk_set_constants");
    int rc = bind(c);
    if (rc != PTK_OK) return rc;
    HIP_TRY(hipMemsetAsync(c->planes_d, 0, PTK_N_PLANES * c->pixels * sizeof(float), c->stream));
    c->sample_count = 0;
    c->have_features = false;
    return PTK_OK;
}

int ptk_set_math_mode(ptk_ctx* c, int mode)
{
    if (!c) return fail(PTK_ERR_ARG, "ptk_set_math_mode: null context");
    if (mode != PTK_MATH_EXACT && mode != PTK_MATH_FAST)
        return fail(PTK_ERR_ARG, "ptk_set_math_mode: unknown mode %d", mode);
    c->math_mode = mode;
    return PTK_OK;
}

int ptk_set_stream_id(ptk_ctx* c, uint64_t stream_id)
{
    if (!c) return fail(PTK_ERR_ARG, "ptk_set_stream_id: null context");
    c->stream_id = stream_id;
    return PTK_OK;
}

int ptk_render(ptk_ctx* c, int spp, int seed_mode, uint64_t glo_seed[2], uint64_t* rays_out)
{
    if (!c) return fail(PTK_ERR_ARG, "ptk_render: null context");
    if (!c->have_scene || !c->planes_d)
        return fail(PTK_ERR_STATE, "ptk_render before ptk_set_constants / ptk_set_scene succeeded");
    if (spp < 0) return fail(PTK_ERR_ARG, "ptk_render: spp %d < 0", spp);
    if (seed_mode != PTK_SEED_REFERENCE && seed_mode != PTK_SEED_COUNTER)
        return fail(PTK_ERR_ARG, "ptk_render: unknown seed mode %d", seed_mode);
    if (seed_mode == PTK_SEED_REFERENCE && !glo_seed)
        return fail(PTK_ERR_ARG, "ptk_render: reference seed mode needs glo_seed");
    if ((int64_t)c->sample_count + spp > 0x7fffffff)
        return fail(PTK_ERR_ARG, "ptk_render: sample count would overflow");
    int rc = bind(c);
    if (rc != PTK_OK) return rc;

    const size_t P = c->pixels;
    HIP_TRY(hipMemsetAsync(c->rays_d, 0, sizeof(unsigned long long), c->stream));
    RenderArgs a;
    a.scene = c->scene_d;
    a.planes = c->planes_d;
    a.rays = c->rays_d;
    a.sched = c->sched_d;
    a.chunk_spp = c->render_chunk_spp;
    a.n_chunks = a.n_items = a.tiles_x = a.tiles_y = 0;   // the launcher's
    a.stream_id = c->stream_id;
    a.width = c->width;
    a.height = c->height;
    a.has_blur = c->has_blur ? 1 : 0;
    a.row_begin = 0;
    a.row_end = c->height;
    a.seed_pixel0 = 0;
    a.seed_stride = (int64_t)P;
    a.big = c->big;
    // A scene in which nothing refracts needs no bending code in its shadow scans (see
    // PTK_KIND_LOOPS in render_kernel.inc): kernels without it, 1 to 2 % faster.
    const bool by_kind = !c->big_scene && !c->host_scene.any_dielectric && !c->segment_scans_only;
    const bool fast = c->math_mode == PTK_MATH_FAST;
    auto launch = c->big_scene ? (fast ? ptk_launch_render_bvh_fast : ptk_launch_render_bvh_exact)
                  : by_kind    ? (fast ? ptk_launch_render_kind_fast : ptk_launch_render_kind_exact)
                               : (fast ? ptk_launch_render_fast : ptk_launch_render_exact);

    double render_ms = 0.0, seed_ms = 0.0;
    if (seed_mode == PTK_SEED_COUNTER) {
        a.seeds = nullptr;
        a.spp = spp;
        a.base_count = c->sample_count;
        HIP_TRY(hipEventRecord(c->ev[0], c->stream));
        if (spp > 0) HIP_TRY(launch(a, c->max_depth, c->features, c->stream));
        HIP_TRY(hipEventRecord(c->ev[1], c->stream));
        render_ms = elapsed_ms(c);
        c->sample_count += spp;
    } else {
        // The global seed recurrence is sequential (Renderers.cpp:1051-1060), so the host steps
        // it and stages 16 bytes per pixel per pass: whole passes while they fit in
        // the chunk size, otherwise one pass in bands of rows (a multiple of 16, the
        // workgroup's height), which the recurrence's row-major pixel order allows.
        const size_t per_pass = P * 2 * sizeof(int64_t);
        const size_t W = (size_t)c->width;
        size_t band_rows = (size_t)c->height;          // rows per launch
        size_t passes = c->seed_chunk_bytes / per_pass;   // passes per launch
        if (passes < 1) {
            passes = 1;
            band_rows = c->seed_chunk_bytes / (W * 2 * sizeof(int64_t)) / 16 * 16;
            if (band_rows < 16) band_rows = 16;
            if (band_rows > (size_t)c->height) band_rows = (size_t)c->height;
        }
        if (passes > (size_t)spp) passes = (size_t)spp;
        const size_t want = passes * band_rows * W;    // seeds (pairs) staged at a time
        if (want > c->seed_capacity) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            (void)hipFree(c->seeds_d); c->seeds_d = nullptr;
            if (c->seeds_h) (void)hipHostFree(c->seeds_h);
            c->seeds_h = nullptr;
            c->seed_capacity = 0;
            const size_t bytes = want * 2 * sizeof(int64_t);
            hipError_t ea = hipMalloc((void**)&c->seeds_d, bytes);
            if (ea == hipSuccess)
                ea = hipHostMalloc((void**)&c->seeds_h, bytes, hipHostMallocDefault);
            if (ea != hipSuccess) {
                (void)hipFree(c->seeds_d); c->seeds_d = nullptr;
                (void)hipGetLastError();
                return fail(ea == hipErrorOutOfMemory ? PTK_ERR_NOMEM : PTK_ERR_HIP,
                            "ptk_render: %zu bytes of seed staging (reference seeds take 16 bytes "
                            "per pixel per pass): %s", bytes, hipGetErrorString(ea));
            }
            c->seed_capacity = want;
        }
        uint64_t g0 = glo_seed[0], g1 = glo_seed[1];
        int done = 0;
        while (done < spp) {
            int chunk = spp - done;
            if ((size_t)chunk > passes) chunk = (int)passes;
            for (size_t row0 = 0; row0 < (size_t)c->height; row0 += band_rows) {
                const size_t row1 = std::min(row0 + band_rows, (size_t)c->height);
                const size_t band_pixels = (row1 - row0) * W;   // == P unless a pass is split
                timespec t0, t1;
                clock_gettime(CLOCK_MONOTONIC, &t0);
                int64_t* h = c->seeds_h;
                const size_t n = (size_t)chunk * band_pixels;
                for (size_t i = 0; i < n; i++) {
                    seed_step(g0, g1);
                    h[2 * i] = (int64_t)g0;
                    h[2 * i + 1] = (int64_t)g1;
                }
                HIP_TRY(hipMemcpyAsync(c->seeds_d, h, n * 2 * sizeof(int64_t),
                                       hipMemcpyHostToDevice, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                clock_gettime(CLOCK_MONOTONIC, &t1);
                seed_ms += (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
                a.seeds = c->seeds_d;
                a.spp = chunk;
                a.base_count = c->sample_count;
                a.row_begin = (int32_t)row0;
                a.row_end = (int32_t)row1;
                a.seed_pixel0 = (int64_t)(row0 * W);
                a.seed_stride = (int64_t)band_pixels;
                HIP_TRY(hipEventRecord(c->ev[0], c->stream));
                HIP_TRY(launch(a, c->max_depth, c->features, c->stream));
                HIP_TRY(hipEventRecord(c->ev[1], c->stream));
                render_ms += elapsed_ms(c);
            }
            c->sample_count += chunk;
            done += chunk;
        }
        glo_seed[0] = g0;
        glo_seed[1] = g1;
    }
    unsigned long long rays = 0;
    HIP_TRY(hipMemcpyAsync(&rays, c->rays_d, sizeof rays, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (rays_out) *rays_out = rays;
    if (spp > 0) c->have_features = c->features;   // a colour-only render leaves the feature planes behind
    c->times.render_ms = render_ms;
    c->times.seed_ms = seed_ms;
    c->times.render_samples = (uint64_t)P * (uint64_t)spp;
    c->times.render_rays = rays;
    return PTK_OK;
}

int ptk_read_planes(ptk_ctx* c, int first_plane, int n_planes, float* out)
{
    if (!c || !out) return fail(PTK_ERR_ARG, "ptk_read_planes: null argument");
    if (!c->planes_d) return fail(PTK_ERR_STATE, "ptk_read_planes before ptk_set_constants");
    if (first_plane < 0 || n_planes < 0 || first_plane + n_planes > PTK_N_PLANES)
        return fail(PTK_ERR_ARG, "ptk_read_planes: planes [%d,%d) outside [0,%d)", first_plane,
                    first_plane + n_planes, PTK_N_PLANES);
    int rc = bind(c);
    if (rc != PTK_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out, c->planes_d + (size_t)first_plane * c->pixels,
                           (size_t)n_planes * c->pixels * sizeof(float), hipMemcpyDeviceToHost,
                           c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PTK_OK;
}

int ptk_write_planes(ptk_ctx* c, int first_plane, int n_planes, const float* in, int sample_count)
{
    if (!c || !in) return fail(PTK_ERR_ARG, "ptk_write_planes: null argument");
    if (!c->planes_d) return fail(PTK_ERR_STATE, "ptk_write_planes before ptk_set_constants");
    if (first_plane < 0 || n_planes < 0 || first_plane + n_planes > PTK_N_PLANES)
        return fail(PTK_ERR_ARG, "ptk_write_planes: planes [%d,%d) outside [0,%d)", first_plane,
                    first_plane + n_planes, PTK_N_PLANES);
    if (sample_count < 0) return fail(PTK_ERR_ARG, "ptk_write_planes: sample_count %d", sample_count);
    int rc = bind(c);
    if (rc != PTK_OK) return rc;
    HIP_TRY(hipMemcpyAsync(c->planes_d + (size_t)first_plane * c->pixels, in,
                           (size_t)n_planes * c->pixels * sizeof(float), hipMemcpyHostToDevice,
                           c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->sample_count = sample_count;
    if (first_plane + n_planes > PTK_PL_NORMAL) c->have_features = true;
    return PTK_OK;
}

int ptk_read_accumulators(ptk_ctx* c, float* pre_screen, float* normal, float* albedo1,
                          float* albedo2, float* direct_light, float* world_pos, float* std_dev,
                          int* sample_count)
{
    if (!c) return fail(PTK_ERR_ARG, "ptk_read_accumulators: null context");
    if (!c->planes_d) return fail(PTK_ERR_STATE, "ptk_read_accumulators before ptk_set_constants");
    int rc = bind(c);
    if (rc != PTK_OK) return rc;
    const size_t P = c->pixels;
    if ((rc = ensure_staging(c, 6 * P)) != PTK_OK) return rc;
    struct { float* dst; int plane; int n; } jobs[] = {
        {pre_screen, PTK_PL_COL, 3}, {normal, PTK_PL_NORMAL, 3}, {albedo1, PTK_PL_ALB1, 3},
        {albedo2, PTK_PL_ALB2, 3},   {direct_light, PTK_PL_DL, 1}, {world_pos, PTK_PL_WPOS, 3},
        {std_dev, PTK_PL_SD, 6},
    };
    for (auto& j : jobs) {
        if (!j.dst) continue;
        HIP_TRY(hipMemcpyAsync(c->staging_h, c->planes_d + (size_t)j.plane * P,
                               (size_t)j.n * P * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (size_t i = 0; i < P; i++)
            for (int k = 0; k < j.n; k++) j.dst[i * j.n + k] = c->staging_h[(size_t)k * P + i];
    }
    if (sample_count) *sample_count = c->sample_count;
    return PTK_OK;
}

int ptk_planes_device(ptk_ctx* c, float** base_d, size_t* pixels_per_plane)
{
    if (!c || !base_d || !pixels_per_plane)
        return fail(PTK_ERR_ARG, "ptk_planes_device: null argument");
    if (!c->planes_d) return fail(PTK_ERR_STATE, "ptk_planes_device before ptk_set_constants");
    *base_d = c->planes_d;
    *pixels_per_plane = c->pixels;
    return PTK_OK;
}

int ptk_get_sample_count(ptk_ctx* c, int* sample_count)
{
    if (!c || !sample_count) return fail(PTK_ERR_ARG, "ptk_get_sample_count: null argument");
    *sample_count = c->sample_count;
    return PTK_OK;
}

int ptk_set_sample_count(ptk_ctx* c, int sample_count)
{
    if (!c || sample_count < 0) return fail(PTK_ERR_ARG, "ptk_set_sample_count: bad argument");
    c->sample_count = sample_count;
    // A caller that fills the planes through ptk_planes_device (a cross-GPU reduction) declares
    // them valid with this call: on a context set up for the denoiser's features that covers
    // the feature planes too.
    if (c->features && sample_count > 0) c->have_features = true;
    return PTK_OK;
}

int ptk_refresh_std_dev(ptk_ctx* c)
{
    if (!c) return fail(PTK_ERR_ARG, "ptk_refresh_std_dev: null context");
    if (!c->planes_d) return fail(PTK_ERR_STATE, "ptk_refresh_std_dev before ptk_set_constants");
    if (c->sample_count < 1) return fail(PTK_ERR_STATE, "ptk_refresh_std_dev: no samples");
    int rc = bind(c);
    if (rc != PTK_OK) return rc;
    HIP_TRY(ptk_launch_refresh_std_dev(c->planes_d, c->pixels, c->sample_count, c->stream));
    return PTK_OK;
}

int ptk_auto_exposure(ptk_ctx* c, int formula, float* exposure, double* raw_sum)
{
    if (!c) return fail(PTK_ERR_ARG, "ptk_auto_exposure: null context");
    if (!c->planes_d) return fail(PTK_ERR_STATE, "ptk_auto_exposure before ptk_set_constants");
    if (formula < 0 || formula > 2) return fail(PTK_ERR_ARG, "ptk_auto_exposure: formula %d", formula);
    if (exposure && c->sample_count < 1)
        return fail(PTK_ERR_STATE, "ptk_auto_exposure: no samples accumulated");
    int rc = bind(c);
    if (rc != PTK_OK) return rc;
    HIP_TRY(hipMemsetAsync(c->sum_d, 0, PTK_EXPOSURE_SUM_DOUBLES * sizeof(double), c->stream));
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    HIP_TRY(ptk_launch_exposure(c->planes_d, c->pixels, formula == 1 ? 1 : 0, c->sum_d, c->stream));
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    double shards[PTK_EXPOSURE_SUM_DOUBLES];
    HIP_TRY(hipMemcpyAsync(shards, c->sum_d, sizeof shards, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->times.exposure_ms = elapsed_ms(c);
    double sum = 0.0;
    for (int i = 0; i < PTK_EXPOSURE_SHARDS; i++) sum += shards[i * PTK_EXPOSURE_SHARD_STRIDE];
    if (raw_sum) *raw_sum = sum;
    if (exposure) {
        const float n = (float)c->sample_count;
        float e = (float)sum;
        if (formula == PTK_EXPOSURE_CPU)
            e *= 9.6f / (float)c->width * (float)c->height * n;  // as written, left to right
        else if (formula == PTK_EXPOSURE_SKEPU)
            e = 9.6f * e / (float)(int)c->pixels * n;
        else
            e = (float)(9.6 * sum / ((double)c->pixels * (double)c->sample_count));
        *exposure = e;
    }
    return PTK_OK;
}

int ptk_auto_exposure_static(const float* colour_planes_d, size_t pixels, int width, int height,
                             int sample_count, int formula, float* exposure_d, double* raw_sum_d,
                             void* hip_stream)
{
    if (!colour_planes_d) return fail(PTK_ERR_ARG, "ptk_auto_exposure_static: null planes");
    if (!exposure_d && !raw_sum_d)
        return fail(PTK_ERR_ARG, "ptk_auto_exposure_static: no output pointer");
    if (formula < 0 || formula > 2)
        return fail(PTK_ERR_ARG, "ptk_auto_exposure_static: formula %d", formula);
    if (pixels < 1 || pixels > PTK_STATIC_MAX_PIXELS)
        return fail(PTK_ERR_ARG, "ptk_auto_exposure_static: %zu pixels outside [1, %zu]", pixels,
                    (size_t)PTK_STATIC_MAX_PIXELS);
    if (width < 1 || height < 1 || (size_t)width * (size_t)height != pixels)
        return fail(PTK_ERR_ARG, "ptk_auto_exposure_static: %dx%d is not %zu pixels", width, height,
                    pixels);
    if (exposure_d && sample_count < 1)
        return fail(PTK_ERR_ARG, "ptk_auto_exposure_static: sample_count %d < 1", sample_count);
    HIP_TRY(ptk_launch_exposure_static(colour_planes_d, pixels, width, height, sample_count, formula,
                                       exposure_d, raw_sum_d, (hipStream_t)hip_stream));
    return PTK_OK;
}

namespace {
// Uploads the weights and runs the denoiser on rows [row_begin, row_end); the
// results stay in the context's device planes.
int denoise_device(ptk_ctx* c, const float* w12, const float* w23, const float* w34, int n,
                   int variant, int row_begin, int row_end, bool want_features,
                   bool want_layers = false)
{
    if (!c || !w12 || !w23 || !w34) return fail(PTK_ERR_ARG, "ptk_denoise: null argument");
    if (!c->planes_d) return fail(PTK_ERR_STATE, "ptk_denoise before ptk_set_constants");
    if (c->sample_count < 1) return fail(PTK_ERR_STATE, "ptk_denoise: no samples accumulated");
    if (!c->have_features)
        return fail(PTK_ERR_STATE, "ptk_denoise: the feature planes were not accumulated (render "
                    "with want_denoise_features, or write planes 3..37)");
    if (n < 0 || n > 8) return fail(PTK_ERR_ARG, "ptk_denoise: N %d outside [0,8]", n);
    if (variant < 0 || variant > 2)
        return fail(PTK_ERR_ARG, "ptk_denoise: variant %d (0: the denoiser; 1: the same with the "
                    "filter on LDS tiles at N = 1 as well; 2: the feature pass in its "
                    "tolerance-held form)", variant);
    if (row_begin < 0 || row_end > c->height || row_begin > row_end)
        return fail(PTK_ERR_ARG, "ptk_denoise: rows [%d,%d) outside [0,%d)", row_begin, row_end,
                    c->height);
    int rc = bind(c);
    if (rc != PTK_OK) return rc;
    const size_t P = c->pixels;
    // device layout: first layer regrouped [channel][feature group][node] so the
    // kernel's per-channel weights are 70 consecutive floats, then the 1/spp
    // column, then layers two and three transposed ([input][node], the third layer's rows
    // padded to eight), so the weights of two consecutive nodes are an aligned pair
    float w[PTK_DEV_WEIGHTS] = {};
    for (int k = 0; k < 5; k++)
        for (int g = 0; g < 7; g++)
            for (int nd = 0; nd < 10; nd++) w[70 * k + 10 * g + nd] = w12[36 * nd + 5 * g + k];
    for (int nd = 0; nd < 10; nd++) w[350 + nd] = w12[36 * nd + 35];
    for (int i = 0; i < 10; i++) {
        for (int nd = 0; nd < 10; nd++) w[360 + 10 * i + nd] = w23[10 * nd + i];
        for (int nd = 0; nd < 7; nd++) w[460 + 8 * i + nd] = w34[10 * nd + i];
    }
    // A frame loop passes the same weights every call: upload only what changed.  The copy
    // reads the context's own buffer, which outlives the call, so nothing waits for it.
    if (!c->weights_valid || std::memcmp(c->weights_up, w, sizeof w) != 0) {
        c->weights_valid = false;
        HIP_TRY(hipStreamSynchronize(c->stream));   // an earlier copy may still read weights_up
        std::memcpy(c->weights_up, w, sizeof w);
        HIP_TRY(hipMemcpyAsync(c->weights_d, c->weights_up, sizeof w, hipMemcpyHostToDevice, c->stream));
        c->weights_valid = true;
    }
    if (want_features && !c->features_d)
        HIP_TRY(hipMalloc((void**)&c->features_d, 36 * P * sizeof(float)));

    DenoiseArgs a;
    a.planes = c->planes_d;
    a.weights = c->weights_d;
    a.variances = c->variances_d;
    a.features = want_features ? c->features_d : nullptr;
    if (want_layers && !c->layers_d)
        HIP_TRY(hipMalloc((void**)&c->layers_d, 20 * P * sizeof(float)));
    a.layers = want_layers ? c->layers_d : nullptr;
    a.denoised = c->denoised_d;
    a.wc_sum = c->wcsum_d;
    a.width = c->width;
    a.height = c->height;
    a.sample_count = c->sample_count;
    a.n = n;
    a.row_begin = row_begin;
    a.row_end = row_end;
    if (row_begin == row_end) return PTK_OK;
    c->have_denoised = false;   // until the launches below are queued
    // both launches are queued back to back and the host waits once, after the second
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    HIP_TRY(ptk_launch_features_mlp(a, c->stream, variant == 2));
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    HIP_TRY(ptk_launch_filter(a, c->stream, variant == 1));
    HIP_TRY(hipEventRecord(c->ev[2], c->stream));
    HIP_TRY(hipEventSynchronize(c->ev[2]));
    float f_ms = 0.0f, g_ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&f_ms, c->ev[0], c->ev[1]));
    HIP_TRY(hipEventElapsedTime(&g_ms, c->ev[1], c->ev[2]));
    c->times.features_ms = f_ms;
    c->times.filter_ms = g_ms;
    c->times.denoise_ms = c->times.features_ms + c->times.filter_ms;
    c->have_denoised = true;
    return PTK_OK;
}
}  // namespace

int ptk_denoise(ptk_ctx* c, const float* w12, const float* w23, const float* w34, int n,
                int variant, float* denoised, float* wc_sum, float* variances, float* features)
{
    int rc = denoise_device(c, w12, w23, w34, n, variant, 0, c ? c->height : 0, features != nullptr);
    if (rc != PTK_OK) return rc;
    const size_t P = c->pixels;
    struct { float* dst; const float* src; int n; } jobs[] = {
        {denoised, c->denoised_d, 3}, {wc_sum, c->wcsum_d, 1},
        {variances, c->variances_d, 7}, {features, c->features_d, 36},
    };
    for (auto& j : jobs) {
        if (!j.dst) continue;
        if ((rc = ensure_staging(c, (size_t)j.n * P)) != PTK_OK) return rc;
        HIP_TRY(hipMemcpyAsync(c->staging_h, j.src, (size_t)j.n * P * sizeof(float),
                               hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (size_t i = 0; i < P; i++)
            for (int k = 0; k < j.n; k++) j.dst[i * j.n + k] = c->staging_h[(size_t)k * P + i];
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PTK_OK;
}

int ptk_denoise_rows(ptk_ctx* c, const float* w12, const float* w23, const float* w34, int n,
                     int variant, int row_begin, int row_end)
{
    int rc = denoise_device(c, w12, w23, w34, n, variant, row_begin, row_end, false);
    if (rc != PTK_OK) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PTK_OK;
}

int ptk_denoise_outputs_device(ptk_ctx* c, float** denoised_d, float** wc_sum_d, float** variances_d)
{
    if (!c) return fail(PTK_ERR_ARG, "ptk_denoise_outputs_device: null context");
    if (!c->planes_d) return fail(PTK_ERR_STATE, "ptk_denoise_outputs_device before ptk_set_constants");
    if (denoised_d) *denoised_d = c->denoised_d;
    if (wc_sum_d) *wc_sum_d = c->wcsum_d;
    if (variances_d) *variances_d = c->variances_d;
    return PTK_OK;
}

namespace {
// Rows [lo, hi) rank `rank` of `world` must hold to denoise its band: the band (contiguous,
// sizes differ by at most one row, the first height % world ranks take the extra one) plus
// `halo` rows either side, clipped to the image.
void band_span(int height, int world, int rank, int halo, int& lo, int& hi)
{
    const int base = height / world, extra = height % world;
    const int y0 = rank * base + std::min(rank, extra);
    const int y1 = y0 + base + (rank < extra ? 1 : 0);
    lo = std::max(0, y0 - halo);
    hi = std::min(height, y1 + halo);
}
int check_bands(ptk_ctx* c, int world, int halo, const char* who)
{
    if (!c) return fail(PTK_ERR_ARG, "%s: null context", who);
    if (!c->planes_d) return fail(PTK_ERR_STATE, "%s before ptk_set_constants", who);
    if (world < 1 || world > PTK_MAX_RANKS)
        return fail(PTK_ERR_ARG, "%s: %d ranks outside [1, %d]", who, world, PTK_MAX_RANKS);
    if (halo < 0 || halo > c->height) return fail(PTK_ERR_ARG, "%s: halo %d", who, halo);
    return PTK_OK;
}
}  // namespace

int ptk_band_layout(int height, int world, int rank, int halo, int* row_lo, int* row_hi, int* rows_max)
{
    if (height < 1 || world < 1 || rank < 0 || rank >= world || halo < 0)
        return fail(PTK_ERR_ARG, "ptk_band_layout: height %d, rank %d of %d, halo %d", height, rank,
                    world, halo);
    int lo = 0, hi = 0, most = 0;
    for (int d = 0; d < world; d++) {
        band_span(height, world, d, halo, lo, hi);
        most = std::max(most, hi - lo);
    }
    band_span(height, world, rank, halo, lo, hi);
    if (row_lo) *row_lo = lo;
    if (row_hi) *row_hi = hi;
    if (rows_max) *rows_max = most;
    return PTK_OK;
}

int ptk_pack_bands(ptk_ctx* c, int world, int halo, float* send_d, size_t send_floats)
{
    int rc = check_bands(c, world, halo, "ptk_pack_bands");
    if (rc != PTK_OK) return rc;
    if (!send_d) return fail(PTK_ERR_ARG, "ptk_pack_bands: null buffer");
    int rows_max = 0;
    ptk_band_layout(c->height, world, 0, halo, nullptr, nullptr, &rows_max);
    const size_t W = (size_t)c->width, block = 32 * (size_t)rows_max * W;
    if (send_floats < (size_t)world * block)
        return fail(PTK_ERR_ARG, "ptk_pack_bands: buffer of %zu floats, %zu needed", send_floats,
                    (size_t)world * block);
    if ((rc = bind(c)) != PTK_OK) return rc;
    for (int d = 0; d < world; d++) {
        int lo, hi;
        band_span(c->height, world, d, halo, lo, hi);
        // 32 planes, each a run of (hi - lo) rows: one strided copy per destination
        HIP_TRY(hipMemcpy2DAsync(send_d + (size_t)d * block, (size_t)rows_max * W * sizeof(float),
                                 c->planes_d + (size_t)lo * W, c->pixels * sizeof(float),
                                 (size_t)(hi - lo) * W * sizeof(float), 32, hipMemcpyDeviceToDevice,
                                 c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));   // the caller's exchange runs on a stream of its own
    return PTK_OK;
}

int ptk_merge_received(ptk_ctx* c, int world, int rank, int halo, const float* recv_d,
                       size_t recv_floats, const int* samples_per_rank)
{
    int rc = check_bands(c, world, halo, "ptk_merge_received");
    if (rc != PTK_OK) return rc;
    if (!recv_d || !samples_per_rank) return fail(PTK_ERR_ARG, "ptk_merge_received: null argument");
    if (rank < 0 || rank >= world) return fail(PTK_ERR_ARG, "ptk_merge_received: rank %d of %d", rank, world);
    BandMergeArgs a;
    a.recv = recv_d;
    a.planes = c->planes_d;
    a.width = c->width;
    a.height = c->height;
    a.world = world;
    ptk_band_layout(c->height, world, rank, halo, &a.row_lo, &a.row_hi, &a.rows_max);
    if (recv_floats < (size_t)world * 32 * (size_t)a.rows_max * (size_t)c->width)
        return fail(PTK_ERR_ARG, "ptk_merge_received: buffer of %zu floats, %zu needed", recv_floats,
                    (size_t)world * 32 * (size_t)a.rows_max * (size_t)c->width);
    int64_t total = 0;
    for (int r = 0; r < world; r++) {
        if (samples_per_rank[r] < 0)
            return fail(PTK_ERR_ARG, "ptk_merge_received: rank %d rendered %d samples", r, samples_per_rank[r]);
        a.n_rank[r] = (float)samples_per_rank[r];
        total += samples_per_rank[r];
    }
    if (total < 1 || total > 0x7fffffff)
        return fail(PTK_ERR_ARG, "ptk_merge_received: %lld samples in all", (long long)total);
    a.n_total = (float)total;
    if ((rc = bind(c)) != PTK_OK) return rc;
    HIP_TRY(ptk_launch_merge_received(a, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->sample_count = (int)total;
    c->have_features = true;
    c->have_denoised = false;
    return PTK_OK;
}

int ptk_save_target(ptk_ctx* c)
{
    if (!c) return fail(PTK_ERR_ARG, "ptk_save_target: null context");
    if (!c->planes_d) return fail(PTK_ERR_STATE, "ptk_save_target before ptk_set_constants");
    if (c->sample_count < 1) return fail(PTK_ERR_STATE, "ptk_save_target: no samples accumulated");
    int rc = bind(c);
    if (rc != PTK_OK) return rc;
    if (!c->target_d) HIP_TRY(hipMalloc((void**)&c->target_d, 3 * c->pixels * sizeof(float)));
    HIP_TRY(ptk_launch_save_target(c->planes_d, c->target_d, c->pixels, c->sample_count, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_target = true;
    return PTK_OK;
}

int ptk_set_target(ptk_ctx* c, const float* target)
{
    if (!c || !target) return fail(PTK_ERR_ARG, "ptk_set_target: null argument");
    if (!c->planes_d) return fail(PTK_ERR_STATE, "ptk_set_target before ptk_set_constants");
    int rc = bind(c);
    if (rc != PTK_OK) return rc;
    const size_t P = c->pixels;
    if (!c->target_d) HIP_TRY(hipMalloc((void**)&c->target_d, 3 * P * sizeof(float)));
    if ((rc = ensure_staging(c, 3 * P)) != PTK_OK) return rc;
    for (size_t i = 0; i < P; i++)
        for (int k = 0; k < 3; k++) c->staging_h[(size_t)k * P + i] = target[3 * i + k];
    HIP_TRY(hipMemcpyAsync(c->target_d, c->staging_h, 3 * P * sizeof(float), hipMemcpyHostToDevice,
                           c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_target = true;
    return PTK_OK;
}

int ptk_train_step(ptk_ctx* c, float* w12, float* w23, float* w34, int n, float learning_rate,
                   float* rel_mse)
{
    if (!c || !w12 || !w23 || !w34) return fail(PTK_ERR_ARG, "ptk_train_step: null argument");
    if (!c->have_target)
        return fail(PTK_ERR_STATE, "ptk_train_step before ptk_save_target / ptk_set_target");
    if (c->height > c->width)
        return fail(PTK_ERR_ARG, "ptk_train_step: %dx%d is taller than wide; the reference's filter "
                    "derivative (DenoiserNN.cpp:964-967) indexes outside the image there",
                    c->width, c->height);
    // forward pass keeping the features and both hidden layers, then the filter
    int rc = denoise_device(c, w12, w23, w34, n, 0, 0, c->height, true, true);
    if (rc != PTK_OK) return rc;
    if (!c->train_sums_d)
        HIP_TRY(hipMalloc((void**)&c->train_sums_d, PTK_TRAIN_SUM_DOUBLES * sizeof(double)));
    HIP_TRY(hipMemsetAsync(c->train_sums_d, 0, PTK_TRAIN_SUM_DOUBLES * sizeof(double), c->stream));
    TrainArgs a;
    a.planes = c->planes_d;
    a.denoised = c->denoised_d;
    a.wc_sum = c->wcsum_d;
    a.variances = c->variances_d;
    a.features = c->features_d;
    a.layers = c->layers_d;
    a.target = c->target_d;
    a.sums = c->train_sums_d;
    a.width = c->width;
    a.height = c->height;
    a.sample_count = c->sample_count;
    a.n = n;
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    HIP_TRY(ptk_launch_train(a, c->stream));
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    double shards[PTK_TRAIN_SUM_DOUBLES];
    HIP_TRY(hipMemcpyAsync(shards, c->train_sums_d, sizeof shards, hipMemcpyDeviceToHost,
                           c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->times.train_ms = elapsed_ms(c);
    double sum[PTK_TRAIN_TERMS];
    for (int k = 0; k < PTK_TRAIN_TERMS; k++) {
        sum[k] = 0.0;
        for (int sh = 0; sh < PTK_TRAIN_SHARDS; sh++) sum[k] += shards[sh * PTK_TRAIN_TERMS + k];
    }
    if (rel_mse) *rel_mse = (float)(sum[0] * ((double)c->sample_count / 2.0));
    // every weight of a layer whose index is congruent modulo the fan-in gets the same
    // increment (DenoiserNN.cpp:1083-1098)
    for (int w = 0; w < 360; w++) w12[w] += (float)((double)learning_rate * sum[1 + w % 36]);
    for (int w = 0; w < 100; w++) w23[w] += (float)((double)learning_rate * sum[37 + w % 10]);
    for (int w = 0; w < 70; w++) w34[w] += (float)((double)learning_rate * sum[47 + w % 10]);
    return PTK_OK;
}

int ptk_post_process(ptk_ctx* c, int metric, float exposure, float g, float* post_screen,
                     uint8_t* rgb8)
{
    if (!c) return fail(PTK_ERR_ARG, "ptk_post_process: null context");
    if (!c->planes_d) return fail(PTK_ERR_STATE, "ptk_post_process before ptk_set_constants");
    if (metric < 0 || metric > 7)
        return fail(PTK_ERR_ARG, "ptk_post_process: display metric %d outside [0,7]", metric);
    if (metric == 1 && !c->have_denoised)
        return fail(PTK_ERR_STATE, "ptk_post_process: metric 1 before ptk_denoise");
    if (metric == 7 && !c->have_target)
        return fail(PTK_ERR_STATE,
                    "ptk_post_process: metric 7 before ptk_save_target / ptk_set_target");
    if (metric != 1 && metric != 7 && c->sample_count < 1)
        return fail(PTK_ERR_STATE, "ptk_post_process: no samples accumulated");
    int rc = bind(c);
    if (rc != PTK_OK) return rc;
    const size_t P = c->pixels;
    if (!c->post_d) HIP_TRY(hipMalloc((void**)&c->post_d, 3 * P * sizeof(float)));
    if (rgb8 && !c->rgb8_d) HIP_TRY(hipMalloc((void**)&c->rgb8_d, 3 * P));

    PostArgs a;
    a.planes = c->planes_d;
    a.denoised = c->denoised_d;
    a.target = c->target_d;
    a.post = c->post_d;
    a.rgb8 = rgb8 ? c->rgb8_d : nullptr;
    a.pixels = P;
    a.width = c->width;
    a.height = c->height;
    a.sample_count = c->sample_count;
    a.metric = metric;
    a.exposure = exposure;
    a.inv_gamma = 1.0f / g;
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    HIP_TRY(ptk_launch_post_process(a, c->stream));
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    if (rgb8) HIP_TRY(hipMemcpyAsync(rgb8, c->rgb8_d, 3 * P, hipMemcpyDeviceToHost, c->stream));
    if (post_screen) {
        if ((rc = ensure_staging(c, 3 * P)) != PTK_OK) return rc;
        HIP_TRY(hipMemcpyAsync(c->staging_h, c->post_d, 3 * P * sizeof(float),
                               hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->times.post_ms = elapsed_ms(c);
    if (post_screen)
        for (size_t i = 0; i < P; i++)
            for (int k = 0; k < 3; k++) post_screen[3 * i + k] = c->staging_h[(size_t)k * P + i];
    return PTK_OK;
}

int ptk_timing(ptk_ctx* c, ptk_times* out)
{
    if (!c || !out) return fail(PTK_ERR_ARG, "ptk_timing: null argument");
    *out = c->times;
    return PTK_OK;
}

int ptk_synchronize(ptk_ctx* c)
{
    if (!c) return fail(PTK_ERR_ARG, "ptk_synchronize: null context");
    int rc = bind(c);
    if (rc != PTK_OK) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PTK_OK;
}

}  // extern "C"
