/*
 * ptk_oracle.c -- CPU restatement of the reference's per-frame hot path.
 * TEST INFRASTRUCTURE ONLY (see ptk_oracle.h for the pinning statement).
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fopenmp -fPIC -shared
 * (no -march flags: baseline x86-64 has no FMA, so every product and sum below
 * rounds once, exactly as the reference's own -O3 build does).
 *
 * Citations are to files under the reference checkout.
 */
#define _GNU_SOURCE
#include "ptk_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI 3.14159265358979323846 /* M_PI */

/* The reference calls sqrt/cos/sin/acos/pow unqualified.  Which overload that
 * binds to depends on the headers in scope of its build: the float overloads
 * (variant F, the default here and the canonical one) or the C double
 * functions (variant D, -DORC_LIBM_DOUBLE).  Only the render path honours the
 * switch; it exists to measure the gap between the two builds.
 *
 * Variant R (-DORC_LIBM_ROUNDED) is variant F with cos/sin/acos/pow replaced by
 * their correctly rounded float results: evaluated in double (glibc's double
 * functions are within one double ulp) and rounded to float once.  glibc's
 * cosf/sinf/acosf/powf are only faithfully rounded, so variant F can differ
 * from it by one float ulp at those four calls and nowhere else.  A second
 * implementation that rounds those calls correctly -- as the HIP path does --
 * must equal variant R bit for bit on every sample. */
#if defined(ORC_LIBM_ROUNDED)
typedef float lm_t;
#define LM_SQRT(x) sqrtf(x)
#define LM_COS(x) ((float)cos((double)(x)))
#define LM_SIN(x) ((float)sin((double)(x)))
#define LM_ACOS(x) ((float)acos((double)(x)))
#define LM_POW(x, y) ((float)pow((double)(x), (double)(y)))
#define LM_EXP(x) expf(x)
#elif defined(ORC_LIBM_DOUBLE)
typedef double lm_t;
#define LM_SQRT(x) sqrt((double)(x))
#define LM_COS(x) cos((double)(x))
#define LM_SIN(x) sin((double)(x))
#define LM_ACOS(x) acos((double)(x))
#define LM_POW(x, y) pow((double)(x), (double)(y))
#define LM_EXP(x) exp((double)(x))
#else
typedef float lm_t;
#define LM_SQRT(x) sqrtf(x)
#define LM_COS(x) cosf(x)
#define LM_SIN(x) sinf(x)
#define LM_ACOS(x) acosf(x)
#define LM_POW(x, y) powf(x, y)
#define LM_EXP(x) expf(x)
#endif
static const float EPS = 0.00001f;    /* Renderers.cpp:113 */

/* Op-counting build (-DORC_COUNT_OPS, single-threaded): tallies the algorithmic
 * work of the render path as written in the reference -- fp add/sub/mul/div/
 * sqrt/convert as flops, 64-bit integer RNG operations as two 32-bit integer
 * ops each, and libm calls (cos, sin, acos, pow) separately.  Comparisons,
 * selects and loads are not counted. */
#ifdef ORC_COUNT_OPS
static uint64_t g_flops, g_iops, g_libm, g_shape_tests, g_samples;
#define OPS(n) (g_flops += (n))
#define IOPS(n) (g_iops += (n))
#define LIBM(n) (g_libm += (n))
#define SHAPE_TEST() (g_shape_tests += 1)
void orc_opcount_reset(void) { g_flops = g_iops = g_libm = g_shape_tests = g_samples = 0; }
void orc_opcount_get(uint64_t out[5])
{
    out[0] = g_flops; out[1] = g_iops; out[2] = g_libm; out[3] = g_shape_tests; out[4] = g_samples;
}
#else
#define OPS(n) ((void)0)
#define IOPS(n) ((void)0)
#define LIBM(n) ((void)0)
#define SHAPE_TEST() ((void)0)
#endif

/* ------------------------------------------------------------------ RNG */

/* Global seed step, Renderers.cpp:832-838.  The OR binds the whole
 * right-hand side, so this is not a true xoroshiro step. */
void orc_seed_stream(uint64_t glo[2], size_t n, int64_t* out_pairs)
{
    uint64_t g0 = glo[0], g1 = glo[1];
    for (size_t i = 0; i < n; i++) {
        uint64_t a = g0, b = g1;
        b ^= a;
        g0 = (a << 49) | ((a >> 15) ^ b ^ (b << 21));
        g1 = (b << 28) | (b >> 36);
        if (out_pairs) {
            out_pairs[2 * i] = (int64_t)g0;
            out_pairs[2 * i + 1] = (int64_t)g1;
        }
    }
    glo[0] = g0;
    glo[1] = g1;
}

/* randBetween, Renderers.cpp:12-24: xoroshiro128++ output, u64 -> f64 -> f32. */
float orc_rand_between(int64_t state[2], float lo, float hi)
{
    IOPS(30);
    OPS(6);
    uint64_t a = (uint64_t)state[0];
    uint64_t b = (uint64_t)state[1];
    uint64_t sum = a + b;
    uint64_t word = ((sum << 17) | (sum >> 47)) + a;
    double two64 = (double)((uint64_t)1 << 63) * 2.0;
    float r = (float)((double)word / two64);
    b ^= a;
    state[0] = (int64_t)(((a << 49) | (a >> 15)) ^ b ^ (b << 21));
    state[1] = (int64_t)((b << 28) | (b >> 36));
    r *= hi - lo;
    r += lo;
    return r;
}

/* --------------------------------------------------------- vector helpers */

static inline float dot3(const float* a, const float* b)
{
    OPS(5);
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

static inline void normalize3(float* v)
{
    OPS(4);
    lm_t len = LM_SQRT(dot3(v, v));
    v[0] /= len;
    v[1] /= len;
    v[2] /= len;
}

/* Quaternion product lhs <- lhs * rhs in the reference's 9-multiply form
 * (Renderers.cpp:25-39, headers/Obj.h:26-40); components are w,x,y,z. */
static void quat_mul(float* lhs, const float* rhs)
{
    OPS(36);
    float a1 = (lhs[3] + lhs[1]) * (rhs[1] + rhs[2]);
    float a3 = (lhs[0] - lhs[2]) * (rhs[0] + rhs[3]);
    float a4 = (lhs[0] + lhs[2]) * (rhs[0] - rhs[3]);
    float a2 = a1 + a3 + a4;
    float a5 = (lhs[3] - lhs[1]) * (rhs[1] - rhs[2]);
    a5 = (a5 + a2) / 2.0f;
    float w = a5 - a1 + (lhs[3] - lhs[2]) * (rhs[2] - rhs[3]);
    float x = a5 - a2 + (lhs[1] + lhs[0]) * (rhs[1] + rhs[0]);
    float y = a5 - a3 + (lhs[0] - lhs[1]) * (rhs[2] + rhs[3]);
    float z = a5 - a4 + (lhs[3] + lhs[2]) * (rhs[0] - rhs[1]);
    lhs[0] = w; lhs[1] = x; lhs[2] = y; lhs[3] = z;
}

/* v <- conj(q) * (v * q), Renderers.cpp:40-47. */
static void quat_rotate(float* v, const float* q)
{
    OPS(3);
    float p[4] = {0.0f, v[0], v[1], v[2]};
    float qc[4] = {q[0], -q[1], -q[2], -q[3]};
    quat_mul(p, q);
    quat_mul(qc, p);
    v[0] = qc[1]; v[1] = qc[2]; v[2] = qc[3];
}

/* ---------------------------------------------------------- intersection */

/* Oriented-box slab test, Renderers.cpp:124-170 and 502-542.  Returns the hit
 * distance or INFINITY; rpos/rdir receive the ray in box space. */
static float box_distance(const float* a, const float* origin, const float* dir,
                          float* rpos, float* rdir)
{
    SHAPE_TEST();
    OPS(3 + 3 + 8); /* to box space and back, x and y slabs */
    rdir[0] = dir[0]; rdir[1] = dir[1]; rdir[2] = dir[2];
    float centre[3] = {a[0], a[1], a[2]};
    rpos[0] = origin[0] - centre[0];
    rpos[1] = origin[1] - centre[1];
    rpos[2] = origin[2] - centre[2];
    float q[4] = {a[9], a[10], a[11], a[12]};
    if (q[1] + q[2] + q[3] > EPS) {
        quat_rotate(rdir, q);
        normalize3(rdir);
        quat_rotate(rpos, q);
    }
    rpos[0] += centre[0]; rpos[1] += centre[1]; rpos[2] += centre[2];

    int neg[3] = {rdir[0] < 0, rdir[1] < 0, rdir[2] < 0};
    const float* lohi[2] = {a + 3, a + 6};
    float tmin = (lohi[neg[0]][0] - rpos[0]) / rdir[0];
    float tmax = (lohi[1 - neg[0]][0] - rpos[0]) / rdir[0];
    float tymin = (lohi[neg[1]][1] - rpos[1]) / rdir[1];
    float tymax = (lohi[1 - neg[1]][1] - rpos[1]) / rdir[1];
    if ((tmin > tymax) || (tymin > tmax))
        return INFINITY;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    OPS(4);
    float tzmin = (lohi[neg[2]][2] - rpos[2]) / rdir[2];
    float tzmax = (lohi[1 - neg[2]][2] - rpos[2]) / rdir[2];
    if ((tmin > tzmax) || (tzmin > tmax))
        return INFINITY;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    if (tmin > EPS) return tmin;
    if (tmax > EPS) return tmax;
    return INFINITY;
}

/* Geometric sphere test, Renderers.cpp:179-200 and 551-568. */
static float sphere_distance(const float* a, const float* origin, const float* dir)
{
    SHAPE_TEST();
    OPS(3);
    float l[3] = {a[0] - origin[0], a[1] - origin[1], a[2] - origin[2]};
    float tca = dot3(l, dir);
    if (tca < EPS) return INFINITY;
    OPS(4);
    float dsq = dot3(l, l) - tca * tca;
    float rsq = a[3] * a[3];
    if (rsq - dsq < EPS) return INFINITY;
    OPS(4);
    float thc = sqrtf(rsq - dsq);
    float t0 = tca - thc;
    float t1 = tca + thc;
    if (t0 > EPS) return t0;
    if (t1 > EPS) return t1;
    return INFINITY;
}

/* Box face normal in box space, then rotated back (Renderers.cpp:220-260).
 * shadow_variant reproduces the copy at 577-614: the flat test compares the
 * direction with 0 instead of EPS and the two z-face tests read the x bounds. */
static void box_normal(const float* a, const float* obb_hit, const float* dir,
                       int shadow_variant, float* n)
{
    const float* lo = a + 3;
    const float* hi = a + 6;
    float thr = shadow_variant ? 0.0f : EPS;
    /* one |a - b| < EPS test: a subtraction and a comparison, counted when it is made */
#define NEAR(x, y) (OPS(2), fabsf((x) - (y)) < EPS)
    n[0] = n[1] = n[2] = 0.0f;
    if (NEAR(lo[0], hi[0]))      n[0] = dir[0] > thr ? -1 : 1;
    else if (NEAR(lo[1], hi[1])) n[1] = dir[1] > thr ? -1 : 1;
    else if (NEAR(lo[2], hi[2])) n[2] = dir[2] > thr ? -1 : 1;
    else if (NEAR(obb_hit[0], lo[0])) n[0] = -1;
    else if (NEAR(obb_hit[0], hi[0])) n[0] = 1;
    else if (NEAR(obb_hit[1], lo[1])) n[1] = -1;
    else if (NEAR(obb_hit[1], hi[1])) n[1] = 1;
    else if (NEAR(obb_hit[2], shadow_variant ? lo[0] : lo[2])) n[2] = -1;
    else if (NEAR(obb_hit[2], shadow_variant ? hi[0] : hi[2])) n[2] = 1;
#undef NEAR
    float qc[4] = {a[9], -a[10], -a[11], -a[12]};
    quat_rotate(n, qc);
    normalize3(n);
}

/* Blur lobe shared by metal, dielectric and bent shadow rays
 * (Renderers.cpp:358-365, 400-407, 640-647): two draws. */
static void blur_direction(int64_t* rng, float* out)
{
    OPS(4 + 3);
    LIBM(7);
    float u = orc_rand_between(rng, 0, 1.0f);
    float a = (float)((double)LM_ACOS(2.0f * u - 1.0f) - ORC_PI / 2.0);
    float b = orc_rand_between(rng, 0, (float)(2.0 * ORC_PI));
    out[0] = LM_COS(a) * LM_COS(b);
    out[1] = LM_COS(a) * LM_SIN(b);
    out[2] = LM_SIN(a);
}

/* c < EPS ? EPS : c / pi, evaluated in double then stored to float
 * (Renderers.cpp:374, 387, 417, 782). */
static inline float cos_over_pi(float c)
{
    OPS(1);
    return c < 0.00001f ? 0.00001f : (float)((double)c / ORC_PI);
}

/* ------------------------------------------------------- one path sample */

/* The scene as the per-sample function reads it: the POD's fields, with the
 * tables behind pointers so the same function serves scenes of any size. */
typedef struct {
    float cam_pos[3], cam_fwd[3], cam_right[3], cam_up[3];
    float rand_samp;
    int32_t num_shapes, max_depth, res_v, res_h;
    float max_angle_v, max_angle_h, focal_length;
    const int32_t (*shapes)[3];
    const float* attrs;
    const float (*mats)[6];
    const uint32_t* imp;
    uint32_t num_imp, want_features;
} scene_view;

static void view_of_constants(const orc_constants* k, scene_view* v)
{
    memcpy(v->cam_pos, k->cam_pos, sizeof v->cam_pos);
    memcpy(v->cam_fwd, k->cam_fwd, sizeof v->cam_fwd);
    memcpy(v->cam_right, k->cam_right, sizeof v->cam_right);
    memcpy(v->cam_up, k->cam_up, sizeof v->cam_up);
    v->rand_samp = k->rand_samp;
    v->num_shapes = k->num_shapes; v->max_depth = k->max_depth;
    v->res_v = k->res_v; v->res_h = k->res_h;
    v->max_angle_v = k->max_angle_v; v->max_angle_h = k->max_angle_h;
    v->focal_length = k->focal_length;
    v->shapes = k->shapes; v->attrs = k->attrs; v->mats = k->mats; v->imp = k->imp;
    v->num_imp = k->num_imp; v->want_features = k->want_features;
}

static void render_sample_view(const scene_view* c, int row, int col, int64_t s1, int64_t s2,
                               orc_sample* out);

void orc_render_sample(const orc_constants* k, int row, int col,
                       int64_t s1, int64_t s2, orc_sample* out)
{
    scene_view v;
    view_of_constants(k, &v);
    render_sample_view(&v, row, col, s1, s2, out);
}

/* RenderFunc, Renderers.cpp:49-819. */
static void render_sample_view(const scene_view* c, int row, int col, int64_t s1, int64_t s2,
                               orc_sample* out)
{
    int64_t rng[2] = {s1, s2};
#ifdef ORC_COUNT_OPS
    g_samples++;
#endif
    OPS(2 + 14 + 27); /* jitter scale, pixel angles, pixel position */

    float jitter[2] = {0.0f, 0.0f};
    if (c->rand_samp > 0.001f) {
        jitter[0] = orc_rand_between(rng, -1, 1) * c->rand_samp;
        jitter[1] = orc_rand_between(rng, -1, 1) * c->rand_samp;
    }

    float py = -c->max_angle_v + 2.0f * c->max_angle_v * ((float)row / (float)c->res_v);
    float px = -c->max_angle_h + 2.0f * c->max_angle_h * ((float)col / (float)c->res_h);

    float dir[3];
    for (int k = 0; k < 3; k++) {
        float pix = c->cam_pos[k] + c->cam_fwd[k] * c->focal_length +
                    c->cam_right[k] * (px + jitter[0]) + c->cam_up[k] * (py + jitter[1]);
        dir[k] = pix - c->cam_pos[k];
    }
    normalize3(dir);

    float hit_pos[12][3];
    int hit_shape[12];
    float hit_n[12][3];
    float pdf[12];
    int lit[12];
    for (int v = 0; v < 12; v++) {
        hit_n[v][0] = hit_n[v][1] = hit_n[v][2] = 0.0f;
        pdf[v] = (float)(1.0 / ORC_PI);
        lit[v] = 0;
        hit_shape[v] = -1;
        hit_pos[v][0] = hit_pos[v][1] = hit_pos[v][2] = 0.0f;
    }

    float back[3] = {0.0f, 0.0f, 0.0f};
    int n_hits = 0;
    unsigned n_rays = 0;

    for (int depth = 0; depth < c->max_depth; depth++) {
        n_rays++;
        float origin[3];
        if (depth > 0) {
            origin[0] = hit_pos[depth - 1][0];
            origin[1] = hit_pos[depth - 1][1];
            origin[2] = hit_pos[depth - 1][2];
        } else {
            origin[0] = c->cam_pos[0]; origin[1] = c->cam_pos[1]; origin[2] = c->cam_pos[2];
        }

        /* closest hit, lowest index wins ties (Renderers.cpp:117-213) */
        float best = INFINITY;
        int shape = -1, shape_type = 0, attr = 0, mat = 0;
        float p[3] = {0.0f, 0.0f, 0.0f};
        float obb_hit[3] = {0.0f, 0.0f, 0.0f};
        for (int s = 0; s < c->num_shapes; s++) {
            int type = c->shapes[s][0];
            int a = c->shapes[s][2];
            float t = INFINITY;
            float rpos[3], rdir[3];
            if (type == 1) {
                t = box_distance(c->attrs + a, origin, dir, rpos, rdir);
                if (t < best) {
                    OPS(6);
                    obb_hit[0] = rpos[0] + rdir[0] * t;
                    obb_hit[1] = rpos[1] + rdir[1] * t;
                    obb_hit[2] = rpos[2] + rdir[2] * t;
                }
            } else if (type == 0) {
                t = sphere_distance(c->attrs + a, origin, dir);
            }
            if (t < best) {
                OPS(6);
                best = t;
                p[0] = origin[0] + dir[0] * t;
                p[1] = origin[1] + dir[1] * t;
                p[2] = origin[2] + dir[2] * t;
                shape = s; attr = a; mat = c->shapes[s][1]; shape_type = type;
            }
        }

        if (shape < 0) { /* miss, Renderers.cpp:739-744 */
            OPS(3);
            back[0] = 0.1f;
            back[1] = 0.1f;
            back[2] = (dir[1] + 1.0f) / 2.2f + 0.1f;
            break;
        }

        float* n = hit_n[depth];
        if (shape_type == 1) {
            box_normal(c->attrs + attr, obb_hit, dir, 0, n);
        } else {
            OPS(3);
            n[0] = p[0] - c->attrs[attr + 0];
            n[1] = p[1] - c->attrs[attr + 1];
            n[2] = p[2] - c->attrs[attr + 2];
            normalize3(n);
        }

        OPS(1 + 9 + 2 + 3 + 1 + 2 + 15); /* basis, phi, radii, local-to-world */
        LIBM(2);
        /* five draws and the cosine-weighted direction (Renderers.cpp:273-312) */
        float u[5];
        for (int k = 0; k < 5; k++) u[k] = orc_rand_between(rng, 0, 1);
        float ax1[3] = {0.0f, 0.0f, 0.0f};
        if ((double)fabsf(n[0]) > 0.9) {
            ax1[0] = -n[2];
            ax1[2] = n[0];
        } else {
            ax1[1] = n[2];
            ax1[2] = -n[1];
        }
        normalize3(ax1);
        float ax0[3] = {n[1] * ax1[2] - n[2] * ax1[1],
                        n[2] * ax1[0] - n[0] * ax1[2],
                        n[0] * ax1[1] - n[1] * ax1[0]};
        float phi = (float)(((double)2.0f * ORC_PI) * (double)u[0]);
        float lx = LM_COS(phi) * LM_SQRT(u[1]);
        float ly = LM_SIN(phi) * LM_SQRT(u[1]);
        float lz = LM_SQRT(1.0f - u[1]);
        float rand_dir[3] = {lx * ax0[0] + ly * ax1[0] + lz * n[0],
                             lx * ax0[1] + ly * ax1[1] + lz * n[1],
                             lx * ax0[2] + ly * ax1[2] + lz * n[2]};

        float mat_type = c->mats[mat][5];
        if (mat_type == 3) {
            /* dielectric, Renderers.cpp:328-389 */
            OPS(1 + 3 + 4 + 1 + 4 + 5 + 19);
            LIBM(1);
            lit[depth] = 1;
            float blur = c->mats[mat][3];
            float eta = 1.0f / c->mats[mat][4];
            float in[3] = {dir[0], dir[1], dir[2]};
            float rn[3] = {n[0], n[1], n[2]};
            float cosi = dot3(in, rn);
            if (cosi > 0) {
                rn[0] *= -1.0f; rn[1] *= -1.0f; rn[2] *= -1.0f;
                eta = 1.0f / eta;
            } else {
                cosi *= -1.0f;
            }
            float sin_sq = eta * eta * (1.0f - cosi * cosi);
            int can_refract = 1.0f - sin_sq > EPS;
            float r0 = (1.0f - eta) / (1.0f + eta);
            r0 = r0 * r0;
            float schlick = r0 + (1.0f - r0) * (float)LM_POW(1.0f - cosi, 5.0f);
            float pick = orc_rand_between(rng, 0, 1);
            float lobe[3] = {0.0f, 0.0f, 0.0f};
            if (blur > 0.0f) blur_direction(rng, lobe);
            if (!can_refract || schlick > pick) {
                for (int k = 0; k < 3; k++)
                    dir[k] = (in[k] - 2.0f * cosi * rn[k]) + blur * lobe[k];
            } else {
                float bend = eta * cosi - sqrtf(1.0f - sin_sq);
                for (int k = 0; k < 3; k++)
                    dir[k] = (eta * in[k] + bend * rn[k]) + blur * lobe[k];
            }
            normalize3(dir);
            pdf[depth] = cos_over_pi(dot3(n, dir));
        } else if (mat_type == 2) {
            /* metal, Renderers.cpp:390-418 */
            OPS(13);
            lit[depth] = 1;
            float in[3] = {dir[0], dir[1], dir[2]};
            float blur = c->mats[mat][3];
            float d_n = dot3(in, n);
            float lobe[3] = {0.0f, 0.0f, 0.0f};
            if (blur > 0.0f) blur_direction(rng, lobe);
            for (int k = 0; k < 3; k++)
                dir[k] = (in[k] - 2.0f * d_n * n[k]) + blur * lobe[k];
            normalize3(dir);
            pdf[depth] = cos_over_pi(dot3(dir, n));
        } else {
            /* lambert / light, Renderers.cpp:419-730 */
            dir[0] = rand_dir[0]; dir[1] = rand_dir[1]; dir[2] = rand_dir[2];
            int mix = 0;
            int target = 0;
            if (mat_type == 1) {
                lit[depth] = 1;
            } else {
                mix = c->num_imp > 0;
                if (mix) {
                    OPS(3);
                    int pick = (int)(u[3] * c->num_imp * 0.99999f);
                    target = (int)c->imp[pick];
                    if (target == shape) mix = 0;
                }
            }
            if (mix) {
                float p0 = (float)(1 / ORC_PI);
                int t_attr = c->shapes[target][2];
                int t_type = c->shapes[target][0];
                const float* ta = c->attrs + t_attr;
                if (u[4] > 0.65f) {
                    float aim[3] = {0.0f, 0.0f, 0.0f};
                    OPS(12 + 3 + 1 + 3 + 1 + 1); /* aim point, direction, length, cosine test, p0 */
                    if (t_type == 1) {
                        float w[3];
                        for (int k = 0; k < 3; k++) w[k] = orc_rand_between(rng, 0, 1);
                        for (int k = 0; k < 3; k++)
                            aim[k] = (1.0f - w[k]) * ta[3 + k] + w[k] * ta[6 + k];
                    } else if (t_type == 0) {
                        float w[3];
                        w[0] = orc_rand_between(rng, -1, 1);
                        w[1] = orc_rand_between(rng, -1, 1);
                        w[2] = orc_rand_between(rng, -1, 1);
                        normalize3(w);
                        for (int k = 0; k < 3; k++) aim[k] = ta[k] + w[k] * ta[3];
                    }
                    float to_light[3] = {aim[0] - p[0], aim[1] - p[1], aim[2] - p[2]};
                    float reach = sqrtf(dot3(to_light, to_light));
                    to_light[0] /= reach; to_light[1] /= reach; to_light[2] /= reach;

                    /* shadow scan, Renderers.cpp:490-668 */
                    int blocked = 0;
                    float sdir[3] = {to_light[0], to_light[1], to_light[2]};
                    for (int s = 0; s < c->num_shapes; s++) {
                        if (s == target) continue;
                        int type = c->shapes[s][0];
                        int smat = c->shapes[s][1];
                        int a = c->shapes[s][2];
                        float t = INFINITY;
                        /* the reference leaves this uninitialised for sphere
                         * occluders; zero is this restatement's definition */
                        float occ_obb[3] = {0.0f, 0.0f, 0.0f};
                        if (type == 1) {
                            float rpos[3], rdir[3];
                            t = box_distance(c->attrs + a, p, sdir, rpos, rdir);
                            if (t == INFINITY) {
                                /* slab miss or behind: the reference `continue`s on a
                                 * slab miss and leaves tempT infinite otherwise */
                                continue;
                            }
                            if ((int)c->mats[smat][5] == 3) {
                                occ_obb[0] = rpos[0] + rdir[0] * t;
                                occ_obb[1] = rpos[1] + rdir[1] * t;
                                occ_obb[2] = rpos[2] + rdir[2] * t;
                            }
                        } else if (type == 0) {
                            t = sphere_distance(c->attrs + a, p, sdir);
                        }
                        if (t < reach) {
                            if ((int)c->mats[smat][5] == 3) {
                                /* a dielectric occluder bends the shadow ray; the
                                 * normal is taken from the SHADED surface's shape */
                                OPS(1 + 6 + 3 + 5 + 19);
                                float blur = c->mats[smat][3];
                                float eta = 1.0f / c->mats[smat][4];
                                float rn[3] = {0.0f, 0.0f, 0.0f};
                                if (shape_type == 1) {
                                    box_normal(c->attrs + attr, occ_obb, sdir, 1, rn);
                                } else if (shape_type == 0) {
                                    float sp[3] = {p[0] + sdir[0] * t, p[1] + sdir[1] * t,
                                                   p[2] + sdir[2] * t};
                                    rn[0] = sp[0] - c->attrs[attr + 0];
                                    rn[1] = sp[1] - c->attrs[attr + 1];
                                    rn[2] = sp[2] - c->attrs[attr + 2];
                                    normalize3(rn);
                                }
                                float cosi = dot3(sdir, rn);
                                if (cosi > 0) {
                                    rn[0] *= -1.0f; rn[1] *= -1.0f; rn[2] *= -1.0f;
                                    eta = 1.0f / eta;
                                } else {
                                    cosi *= -1.0f;
                                }
                                float sin_sq = eta * eta * (1.0f - cosi * cosi);
                                int can_refract = 1.0f - sin_sq > EPS;
                                float lobe[3] = {0.0f, 0.0f, 0.0f};
                                if (blur > 0.0f) blur_direction(rng, lobe);
                                if (!can_refract) {
                                    for (int k = 0; k < 3; k++)
                                        sdir[k] = (sdir[k] - 2.0f * cosi * rn[k]) + blur * lobe[k];
                                } else {
                                    float bend = eta * cosi - sqrtf(1.0f - sin_sq);
                                    for (int k = 0; k < 3; k++)
                                        sdir[k] = (eta * sdir[k] + bend * rn[k]) + blur * lobe[k];
                                }
                                normalize3(sdir);
                                continue;
                            }
                            blocked = 1;
                            break;
                        }
                    }
                    if (!blocked) {
                        float cosine = fabsf(dot3(to_light, rand_dir));
                        if (cosine > 0.01f) {
                            lit[depth] = 1;
                            dir[0] = to_light[0]; dir[1] = to_light[1]; dir[2] = to_light[2];
                            p0 = (float)((double)fabsf(cosine) / ORC_PI);
                        }
                    }
                }
                float p1 = 0;
                OPS(t_type == 1 ? 12 + 3 + 1 + 2 + 3 : 3 + 2 + 1 + 1 + 1 + 2 + 1 + 2 + 1 + 3);
                if (t_type == 1) {
                    float area = 0;
                    float dx = ta[3] - ta[6];
                    float dy = ta[4] - ta[7];
                    float dz = ta[5] - ta[8];
                    area += dx * dy * 2.0f;
                    area += dz * dy * 2.0f;
                    area += dx * dz * 2.0f;
                    float cosine = dot3(dir, n);
                    cosine = cosine < 0.0001f ? 0.0001f : cosine;
                    float to_c[3] = {ta[0] - p[0], ta[1] - p[1], ta[2] - p[2]};
                    float dist = sqrtf(dot3(to_c, to_c));
                    p1 = dist / (cosine * area);
                } else if (t_type == 0) {
                    float to_c[3] = {ta[0] - p[0], ta[1] - p[1], ta[2] - p[2]};
                    float d2 = dot3(to_c, to_c);
                    lm_t cmax = LM_SQRT(1.0f - ta[3] * ta[3] / d2);
                    cmax = (cmax != cmax) ? 0.9999f : cmax;
                    double solid = ORC_PI * (double)(1.0f - cmax) * 2.0;
                    p1 = (float)((double)ta[3] / (solid * (double)LM_SQRT(d2) * 4.0));
                }
                pdf[depth] = 0.5f * p0 + 0.5f * p1;
            }
        }

        hit_pos[depth][0] = p[0]; hit_pos[depth][1] = p[1]; hit_pos[depth][2] = p[2];
        hit_shape[depth] = shape;
        n_hits++;
    }

    /* backward radiance accumulation, Renderers.cpp:748-793 */
    float colr[3] = {back[0], back[1], back[2]};
    for (int k = n_hits - 1; k >= 0; k--) {
        int m = c->shapes[hit_shape[k]][1];
        const float* alb = c->mats[m];
        int mtype = (int)c->mats[m][5];
        float out_dir[3];
        if (k == n_hits - 1) {
            out_dir[0] = dir[0]; out_dir[1] = dir[1]; out_dir[2] = dir[2];
        } else {
            out_dir[0] = hit_pos[k + 1][0] - hit_pos[k][0];
            out_dir[1] = hit_pos[k + 1][1] - hit_pos[k][1];
            out_dir[2] = hit_pos[k + 1][2] - hit_pos[k][2];
            normalize3(out_dir);
        }
        OPS((k == n_hits - 1 ? 0 : 3) + 1 + 12);
        float scatter = cos_over_pi(dot3(hit_n[k], out_dir));
        float boost = (lit[k] == 1 && c->num_imp > 1) ? (float)c->num_imp : 1.0f;
        float ratio = scatter / pdf[k];
        for (int ch = 0; ch < 3; ch++) {
            float emit = mtype == 1 ? alb[ch] : 0;
            float through = alb[ch] * colr[ch];
            colr[ch] = emit + through * ratio * boost;
        }
    }

    memset(out, 0, sizeof(*out));
    out->xyz[0] = colr[0]; out->xyz[1] = colr[1]; out->xyz[2] = colr[2];
    if (c->want_features == 1) {
        /* Renderers.cpp:798-816.  The reference reads hit records that were
         * never written when the path has fewer than one (two) hits; this
         * restatement defines those outputs as zero. */
        out->normal[0] = hit_n[0][0]; out->normal[1] = hit_n[0][1]; out->normal[2] = hit_n[0][2];
        if (n_hits >= 1) {
            const float* m1 = c->mats[c->shapes[hit_shape[0]][1]];
            out->albedo1[0] = m1[0]; out->albedo1[1] = m1[1]; out->albedo1[2] = m1[2];
            out->world_pos[0] = hit_pos[0][0];
            out->world_pos[1] = hit_pos[0][1];
            out->world_pos[2] = hit_pos[0][2];
        }
        if (n_hits >= 2) {
            const float* m2 = c->mats[c->shapes[hit_shape[1]][1]];
            out->albedo2[0] = m2[0]; out->albedo2[1] = m2[1]; out->albedo2[2] = m2[2];
        }
        OPS(3 * c->max_depth);
        out->direct_light = 0.0f;
        for (int k = 0; k < c->max_depth; k++)
            out->direct_light += (float)lit[k] / (float)c->max_depth;
    }
    out->rays_sent = n_rays;
}

/* ----------------------------------------------------- render + accumulate */

/* One pixel's accumulate step, Renderers.cpp:845-888.  sample_count is the
 * count AFTER this pass's increment. */
static void accumulate_pixel(const scene_view* c, float* acc, size_t P, size_t i,
                             const orc_sample* s, int sample_count)
{
    float n = (float)sample_count;
    OPS(c->want_features == 1 ? 3 + 13 + 64 + 18 : 3);
    acc[(ORC_PL_COL + 0) * P + i] += s->xyz[0];
    acc[(ORC_PL_COL + 1) * P + i] += s->xyz[1];
    acc[(ORC_PL_COL + 2) * P + i] += s->xyz[2];
    if (c->want_features != 1) return;

    const float* feat[5] = {s->xyz, s->normal, s->albedo1, s->albedo2, s->world_pos};
    const int plane[5] = {ORC_PL_COL, ORC_PL_NORMAL, ORC_PL_ALB1, ORC_PL_ALB2, ORC_PL_WPOS};
    for (int k = 1; k < 5; k++)
        for (int ch = 0; ch < 3; ch++)
            acc[(plane[k] + ch) * P + i] += feat[k][ch];
    acc[ORC_PL_DL * P + i] += s->direct_light;

    /* squared deviation of this sample from the running mean that already
     * includes it; pow(float,2) is an exact double square, stored as float */
    for (int k = 0; k < 5; k++) {
        for (int ch = 0; ch < 3; ch++) {
            float d = acc[(plane[k] + ch) * P + i] / n - feat[k][ch];
            acc[(ORC_PL_SDV + 3 * k + ch) * P + i] += (float)((double)d * (double)d);
        }
    }
    {
        float d = acc[ORC_PL_DL * P + i] / n - s->direct_light;
        acc[(ORC_PL_SDV + 15) * P + i] += (float)((double)d * (double)d);
    }
    for (int k = 0; k < 5; k++) {
        float v0 = acc[(ORC_PL_SDV + 3 * k + 0) * P + i];
        float v1 = acc[(ORC_PL_SDV + 3 * k + 1) * P + i];
        float v2 = acc[(ORC_PL_SDV + 3 * k + 2) * P + i];
        acc[(ORC_PL_SD + k) * P + i] = (v0 + v1 + v2) / n;
    }
    acc[(ORC_PL_SD + 5) * P + i] = acc[(ORC_PL_SDV + 15) * P + i] / n;
}

static void render_view(const scene_view* c, float* acc, int spp, int* sample_count,
                        uint64_t glo[2], int threads, uint64_t* rays_last_pass);

void orc_render(const orc_constants* k, float* acc, int spp, int* sample_count,
                uint64_t glo[2], int threads, uint64_t* rays_last_pass)
{
    scene_view v;
    view_of_constants(k, &v);
    render_view(&v, acc, spp, sample_count, glo, threads, rays_last_pass);
}

/* The reference's scans with the POD's size limits lifted: same per-sample
 * function over tables of any length (13 attribute floats per shape). */
int orc_render_scene(const orc_scene_desc* k, int want_features, float* acc, int spp,
                     int* sample_count, uint64_t glo[2], int threads, uint64_t* rays_last_pass)
{
    if (k->num_shapes < 0) return -1;
    int32_t (*shapes)[3] = malloc(sizeof(int32_t[3]) * (size_t)(k->num_shapes + 1));
    if (!shapes) return -1;
    for (int s = 0; s < k->num_shapes; s++) {
        shapes[s][0] = k->shape_type[s];
        shapes[s][1] = k->shape_mat[s];
        shapes[s][2] = 13 * s;
    }
    scene_view v;
    memcpy(v.cam_pos, k->cam_pos, sizeof v.cam_pos);
    memcpy(v.cam_fwd, k->cam_forward, sizeof v.cam_fwd);
    memcpy(v.cam_right, k->cam_right, sizeof v.cam_right);
    memcpy(v.cam_up, k->cam_up, sizeof v.cam_up);
    v.rand_samp = k->rand_samp;
    v.num_shapes = k->num_shapes; v.max_depth = k->max_depth;
    v.res_v = k->res_v; v.res_h = k->res_h;
    v.max_angle_v = k->max_angle_v; v.max_angle_h = k->max_angle_h;
    v.focal_length = k->focal_length;
    v.shapes = (const int32_t (*)[3])shapes;
    v.attrs = k->shape_attrs;
    v.mats = (const float (*)[6])k->mat_list;
    v.imp = k->important_shapes;
    v.num_imp = (uint32_t)k->num_important_shapes;
    v.want_features = want_features ? 1u : 0u;
    render_view(&v, acc, spp, sample_count, glo, threads, rays_last_pass);
    free(shapes);
    return 0;
}

/* One sample of each listed pixel of such a scene (for spot checks of frames too large to
 * recompute whole): out receives 17 floats per pixel -- colour, normal, albedo1, albedo2,
 * worldPos, directLight, rays. */
int orc_render_scene_pixels(const orc_scene_desc* k, int want_features, int n, const int32_t* rows,
                            const int32_t* cols, const int64_t* seeds, float* out, int threads)
{
    if (k->num_shapes < 0 || n < 0) return -1;
    int32_t (*shapes)[3] = malloc(sizeof(int32_t[3]) * (size_t)(k->num_shapes + 1));
    if (!shapes) return -1;
    for (int s = 0; s < k->num_shapes; s++) {
        shapes[s][0] = k->shape_type[s];
        shapes[s][1] = k->shape_mat[s];
        shapes[s][2] = 13 * s;
    }
    scene_view v;
    memcpy(v.cam_pos, k->cam_pos, sizeof v.cam_pos);
    memcpy(v.cam_fwd, k->cam_forward, sizeof v.cam_fwd);
    memcpy(v.cam_right, k->cam_right, sizeof v.cam_right);
    memcpy(v.cam_up, k->cam_up, sizeof v.cam_up);
    v.rand_samp = k->rand_samp;
    v.num_shapes = k->num_shapes; v.max_depth = k->max_depth;
    v.res_v = k->res_v; v.res_h = k->res_h;
    v.max_angle_v = k->max_angle_v; v.max_angle_h = k->max_angle_h;
    v.focal_length = k->focal_length;
    v.shapes = (const int32_t (*)[3])shapes;
    v.attrs = k->shape_attrs;
    v.mats = (const float (*)[6])k->mat_list;
    v.imp = k->important_shapes;
    v.num_imp = (uint32_t)k->num_important_shapes;
    v.want_features = want_features ? 1u : 0u;
    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads)
#endif
    for (int i = 0; i < n; i++) {
        orc_sample s;
        render_sample_view(&v, rows[i], cols[i], seeds[2 * i], seeds[2 * i + 1], &s);
        float* o = out + 17 * (size_t)i;
        memcpy(o, s.xyz, 3 * sizeof(float));
        memcpy(o + 3, s.normal, 3 * sizeof(float));
        memcpy(o + 6, s.albedo1, 3 * sizeof(float));
        memcpy(o + 9, s.albedo2, 3 * sizeof(float));
        memcpy(o + 12, s.world_pos, 3 * sizeof(float));
        o[15] = s.direct_light;
        o[16] = (float)s.rays_sent;
    }
    free(shapes);
    return 0;
}

static void render_view(const scene_view* c, float* acc, int spp, int* sample_count,
                        uint64_t glo[2], int threads, uint64_t* rays_last_pass)
{
    int W = c->res_h, H = c->res_v;
    size_t P = (size_t)W * (size_t)H;
    int64_t* seeds = (int64_t*)malloc(P * 2 * sizeof(int64_t));
    if (threads < 1) threads = 1;
    uint64_t rays = 0;
    for (int pass = 0; pass < spp; pass++) {
        (*sample_count)++;
        int sc = *sample_count;
        orc_seed_stream(glo, P, seeds);
        rays = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads) reduction(+ : rays)
#endif
        for (int j = 0; j < H; j++) {
            for (int i = 0; i < W; i++) {
                size_t idx = (size_t)j * W + i;
                orc_sample s;
                render_sample_view(c, j, i, seeds[2 * idx], seeds[2 * idx + 1], &s);
                accumulate_pixel(c, acc, P, idx, &s, sc);
                rays += s.rays_sent;
            }
        }
    }
    if (rays_last_pass) *rays_last_pass = rays;
    free(seeds);
}

/* --------------------------------------------------------------- exposure */

float orc_auto_exposure(const float* acc, int width, int height, int sample_count,
                        int formula, double* raw_sum)
{
    size_t P = (size_t)width * (size_t)height;
    const float* r = acc + (ORC_PL_COL + 0) * P;
    const float* g = acc + (ORC_PL_COL + 1) * P;
    const float* b = acc + (ORC_PL_COL + 2) * P;
    float e = 0.0f;
    double exact = 0.0;
    if (formula == 1) {
        /* SkePUExposureFuncCalc, Renderers.cpp:1174-1176: the middle and last
         * products are double, the sum is double, the return is float. */
        for (size_t i = 0; i < P; i++) {
            float term = (float)((double)(0.2125f * r[i]) + 0.7154 * (double)g[i] +
                                 0.0721 * (double)b[i]);
            e += term;
            exact += (double)term;
        }
        if (raw_sum) *raw_sum = exact;
        /* Renderers.cpp:1192 */
        return 9.6f * e / (float)(int)P * (float)sample_count;
    }
    const float lr = (float)0.2125, lg = (float)0.7154, lb = (float)0.0721;
    for (size_t i = 0; i < P; i++) {
        float term = lr * r[i] + lg * g[i] + lb * b[i];
        e += term;
        exact += (double)term;
    }
    if (raw_sum) *raw_sum = exact;
    if (formula == 0) {
        /* Renderers.cpp:1158: 9.6f / xRes * yRes * sampleCount, left to right */
        e *= 9.6f / (float)width * (float)height * (float)sample_count;
        return e;
    }
    return (float)(9.6 * exact / ((double)P * (double)sample_count));
}

/* ----------------------------------------------------- denoiser: features */

static inline int clampi(int v, int hi)
{
    return v < 0 ? 0 : (v >= hi ? hi - 1 : v);
}

/* The reference's "median getter" procedure, DenoiserNN.cpp:190-256.  It is
 * not a true median; it is restated literally. */
static float mad_procedure(float* v)
{
    float m[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = 0; i < 9; i++) {
        for (int k = 0; k < 5; k++) {
            if (v[i] > m[k]) {
                if (k == 4) {
                    m[3] = m[4]; m[2] = m[3]; m[1] = m[2]; m[0] = m[1];
                    m[4] = v[i];
                }
            } else if (k > 0) {
                m[k - 1] = v[i];
                break;
            }
        }
    }
    for (int i = 0; i < 9; i++) v[i] = fabsf(v[i] - m[0]);
    m[0] = m[1] = m[2] = m[3] = m[4] = 0.0f;
    for (int i = 0; i < 9; i++) {
        for (int k = 0; k < 5; k++) {
            if (v[i] > m[k]) {
                if (k == 4) {
                    m[3] = m[4]; m[2] = m[3]; m[1] = m[2]; m[0] = m[1];
                    m[4] = v[i];
                }
            } else if (k > 0) {
                for (int t = k - 1; t >= 0; t--) m[t] = m[t + 1];
                m[k] = v[i];
                break;
            }
        }
    }
    return m[0];
}

/* The procedure alone, for tests of other implementations of it: out[i] = its
 * result for the nine values v[9 i .. 9 i + 8]. */
void orc_mad_procedure(const float* v, size_t n, float* out)
{
    for (size_t i = 0; i < n; i++) {
        float w[9];
        memcpy(w, v + 9 * i, sizeof w);
        out[i] = mad_procedure(w);
    }
}

static inline float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }
static inline float softplusf(float x) { return logf(1.0f + expf(x)); }

/* CPUForwardProp, DenoiserNN.cpp:12-301. */
void orc_forward_prop_layers(const float* acc, int width, int height, int sample_count,
                             const float* w12, const float* w23, const float* w34,
                             float* features36, float* variances7, float* layer2, float* layer3,
                             int threads);

void orc_forward_prop(const float* acc, int width, int height, int sample_count,
                      const float* w12, const float* w23, const float* w34,
                      float* features36, float* variances7, int threads)
{
    orc_forward_prop_layers(acc, width, height, sample_count, w12, w23, w34, features36,
                            variances7, NULL, NULL, threads);
}

/* As orc_forward_prop, also keeping layerTwoValues / layerThreeValues (10 per
 * pixel each, DenoiserNN.cpp:271-284) when the pointers are not NULL. */
void orc_forward_prop_layers(const float* acc, int width, int height, int sample_count,
                             const float* w12, const float* w23, const float* w34,
                             float* features36, float* variances7, float* layer2, float* layer3,
                             int threads)
{
    int W = width, H = height;
    size_t P = (size_t)W * (size_t)H;
    float n = (float)sample_count;
    float* prim = (float*)malloc(P * 5 * sizeof(float));
    if (threads < 1) threads = 1;

    /* primary features, DenoiserNN.cpp:16-36 */
    const int plane[4] = {ORC_PL_NORMAL, ORC_PL_ALB1, ORC_PL_ALB2, ORC_PL_WPOS};
    for (size_t i = 0; i < P; i++) {
        for (int k = 0; k < 4; k++) {
            float v = 0.0f;
            for (int ch = 0; ch < 3; ch++) v += acc[(plane[k] + ch) * P + i] / (3.0f * n);
            prim[5 * i + k] = v;
        }
        prim[5 * i + 4] = acc[ORC_PL_DL * P + i] / n;
    }

    static const float gx[9] = {1, 2, 1, 0, 0, 0, -1, -2, -1};
    static const float gy[9] = {1, 0, -1, 2, 0, -2, 1, 0, -1};

#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
    for (int jm = 0; jm < H; jm++) {
        for (int im = 0; im < W; im++) {
            size_t idx = (size_t)jm * W + im;
            float* f = features36 + 36 * idx;
            float mean7[5] = {0}, sd7[5] = {0}, mean3[5] = {0};
            float gxs[5] = {0}, gys[5] = {0}, mdev[5] = {0};
            float win[5][9];

            for (int k = 0; k < 5; k++) {
                f[k] = prim[5 * idx + k];
                f[5 + k] = acc[(ORC_PL_SD + 1 + k) * P + idx];
            }
            for (int j = -3; j <= 3; j++) {
                int jf = clampi(jm + j, H);
                for (int i = -3; i <= 3; i++) {
                    int ifx = clampi(im + i, W);
                    const float* q = prim + 5 * ((size_t)jf * W + ifx);
                    for (int k = 0; k < 5; k++) mean7[k] += q[k] / 49.0f;
                    if (abs(j) <= 1 && abs(i) <= 1)
                        for (int k = 0; k < 5; k++) mean3[k] += q[k] / 9.0f;
                }
            }
            for (int j = -3; j <= 3; j++) {
                int jf = clampi(jm + j, H);
                for (int i = -3; i <= 3; i++) {
                    int ifx = clampi(im + i, W);
                    const float* q = prim + 5 * ((size_t)jf * W + ifx);
                    for (int k = 0; k < 5; k++) {
                        float d = q[k] - mean7[k];
                        /* float += pow(float,2)/49.0f : double sum, stored float */
                        sd7[k] = (float)((double)sd7[k] + ((double)d * (double)d) / 49.0);
                    }
                }
            }
            for (int k = 0; k < 5; k++) sd7[k] = sqrtf(sd7[k]);
            for (int j = -1; j <= 1; j++) {
                int jf = clampi(jm + j, H);
                for (int i = -1; i <= 1; i++) {
                    int ifx = clampi(im + i, W);
                    const float* q = prim + 5 * ((size_t)jf * W + ifx);
                    int l = (j + 1) * 3 + i + 1;
                    for (int k = 0; k < 5; k++) {
                        gxs[k] += gx[l] * q[k];
                        gys[k] += gy[l] * q[k];
                        mdev[k] += fabsf(q[k] - mean3[k]) / 9.0f;
                        win[k][l] = q[k];
                    }
                }
            }
            for (int k = 0; k < 5; k++) {
                f[10 + k] = mean7[k];
                f[15 + k] = sd7[k];
                f[20 + k] = sqrtf(gxs[k] * gxs[k] + gys[k] * gys[k]);
                f[25 + k] = mdev[k];
                f[30 + k] = mad_procedure(win[k]);
            }
            f[35] = 1.0f / n;

            /* MLP 36-10-10-7, no bias, DenoiserNN.cpp:266-299 */
            float l2[10], l3[10];
            for (int node = 0; node < 10; node++) {
                float a = 0.0f;
                for (int w = 0; w < 36; w++) a += f[w] * w12[36 * node + w];
                l2[node] = sigmoidf(a);
            }
            for (int node = 0; node < 10; node++) {
                float a = 0.0f;
                for (int w = 0; w < 10; w++) a += l2[w] * w23[10 * node + w];
                l3[node] = sigmoidf(a);
            }
            for (int node = 0; node < 7; node++) {
                float a = 0.0f;
                for (int w = 0; w < 10; w++) a += l3[w] * w34[10 * node + w];
                variances7[7 * idx + node] = softplusf(a);
            }
            if (layer2) memcpy(layer2 + 10 * idx, l2, sizeof l2);
            if (layer3) memcpy(layer3 + 10 * idx, l3, sizeof l3);
        }
    }
    free(prim);
}

/* ------------------------------------------------------ denoiser: training */

/* DenoiserNN::GenRelMSE (DenoiserNN.cpp:1461-1474): sequential fp32. */
float orc_rel_mse(const float* denoised, const float* target, int width, int height,
                  int sample_count)
{
    size_t P = (size_t)width * (size_t)height;
    float rel = 0.0f;
    for (size_t i = 0; i < P; i++)
        for (int ch = 0; ch < 3; ch++) {
            float d = denoised[3 * i + ch] - target[3 * i + ch];
            float t = target[3 * i + ch];
            rel += d * d / (t * t + 0.00001f);
        }
    rel *= (float)sample_count / 2.0f;
    return rel;
}

/* The seven vectors CPUFilterDerivative(j, i, var) returns for var = 0..6
 * (DenoiserNN.cpp:959-1054), as written: the flat index of "pixel (j, i)" is
 * j * yRes + i and its neighbours clamp rows at xRes - 1 and columns at
 * yRes - 1 (tW = yRes, tH = xRes); the spatial term uses the absolute j and i;
 * the colour difference is the raw sum minus the filtered colour. */
static void filter_derivative(const float* acc, const float* denoised, const float* wc_sum,
                              const float* variances7, int xRes, int yRes, int sample_count,
                              int nwin, int j, int i, float out[7][3])
{
    size_t P = (size_t)xRes * (size_t)yRes;
    int tW = yRes, tH = xRes;
    size_t iIndex = (size_t)j * tW + i;
    float n = (float)sample_count;
    const float* var = variances7 + 7 * iIndex;
    float sd_i[6];
    for (int k = 0; k < 6; k++) sd_i[k] = acc[(ORC_PL_SD + k) * P + iIndex];
    const int plane[5] = {ORC_PL_COL, ORC_PL_NORMAL, ORC_PL_ALB1, ORC_PL_ALB2, ORC_PL_WPOS};
    for (int v = 0; v < 7; v++) out[v][0] = out[v][1] = out[v][2] = 0.0f;

    for (int j1 = -nwin; j1 <= nwin; j1++) {
        for (int i1 = -nwin; i1 <= nwin; i1++) {
            size_t jIndex = (size_t)(j + j1 < 0 ? 0 : (j + j1 >= tH ? tH - 1 : j + j1));
            jIndex *= tW;
            jIndex += (size_t)(i + i1 < 0 ? 0 : (i + i1 >= tW ? tW - 1 : i + i1));

            float vecSum[3];
            for (int ch = 0; ch < 3; ch++)
                vecSum[ch] = (acc[(ORC_PL_COL + ch) * P + jIndex] - denoised[3 * iIndex + ch]) /
                             wc_sum[iIndex];

            float pd[7], dv[7];
            pd[0] = (float)(LM_POW((float)j, 2) + LM_POW((float)i, 2));
            dv[0] = pd[0] / (2.0f * var[0] + 0.000001f);
            for (int k = 0; k < 5; k++) {
                lm_t s2 = 0;
                for (int ch = 0; ch < 3; ch++) {
                    float a = acc[(plane[k] + ch) * P + iIndex] / n;
                    float b = acc[(plane[k] + ch) * P + jIndex] / n;
                    s2 += LM_POW(a - b, 2);
                }
                pd[1 + k] = (float)s2;
            }
            {
                float a = acc[ORC_PL_DL * P + iIndex] / n, b = acc[ORC_PL_DL * P + jIndex] / n;
                pd[6] = (float)LM_POW(a - b, 2);
            }
            float sd0_j = acc[ORC_PL_SD * P + jIndex];
            dv[1] = pd[1] / (2.0f * var[1] * (sd_i[0] + sd0_j) + 0.000001f);
            for (int k = 2; k < 7; k++) dv[k] = pd[k] / (2.0f * var[k] * sd_i[k - 1] + 0.000001f);

            float mult = 1.0f;
            for (int k = 0; k < 7; k++) mult *= (float)LM_EXP(-dv[k]) + 0.000001f;

            for (int v = 0; v < 7; v++) {
                float wop = (float)(mult * pd[v] / LM_POW(var[v], 3));
                for (int ch = 0; ch < 3; ch++) out[v][ch] += vecSum[ch] * wop;
            }
        }
    }
}

/* DenoiserNN::CPUBackProp (DenoiserNN.cpp:1056-1103): in-place update of the
 * three weight arrays, sequential over pixels, as written.  `target` and
 * `denoised` are interleaved rgb; features36 / layer2 / layer3 as produced by
 * orc_forward_prop_layers. */
int orc_back_prop(const float* acc, const float* denoised, const float* wc_sum,
                  const float* variances7, const float* features36, const float* layer2,
                  const float* layer3, const float* target, int width, int height,
                  int sample_count, int nwin, float learning_rate, float* w12, float* w23,
                  float* w34)
{
    int xRes = width, yRes = height;
    /* CPUFilterDerivative's flat index j * yRes + i leaves the image when the
     * image is taller than wide: undefined in the reference, refused here. */
    if (yRes > xRes) return -1;
    float n = (float)sample_count;
    for (int j = 0; j < yRes; j++) {
        for (int i = 0; i < xRes; i++) {
            size_t pixel = (size_t)j * xRes + i;
            float eoc[3];
            for (int ch = 0; ch < 3; ch++) {
                float t = target[3 * pixel + ch];
                eoc[ch] = n * (denoised[3 * pixel + ch] - t) / (t * t + 0.0001f);
            }
            float cop[7][3];
            filter_derivative(acc, denoised, wc_sum, variances7, xRes, yRes, sample_count, nwin, j,
                              i, cop);
            for (int v = 0; v < 7; v++) {
                float dot = eoc[0] * cop[v][0] + eoc[1] * cop[v][1] + eoc[2] * cop[v][2];
                for (int w = 0; w < 360; w++)
                    w12[w] += learning_rate * dot * features36[36 * pixel + w % 36];
                for (int w = 0; w < 100; w++)
                    w23[w] += learning_rate * dot * layer2[10 * pixel + w % 10];
                for (int w = 0; w < 70; w++)
                    w34[w] += learning_rate * dot * layer3[10 * pixel + w % 10];
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------- denoiser: filter */

/* CPUDenoise, Denoiser.cpp:222-355.  colour_variant 1 swaps in the OpenMP
 * copy's colour bandwidth (centre pixel only, Denoiser.cpp:436) and world
 * position denominator (Denoiser.cpp:452); it exists so the recorded output of
 * that copy can be reproduced as a second pin. */
void orc_filter_variant(const float* acc, const float* variances7, int width, int height,
                        int sample_count, int nwin, int colour_variant, float* denoised,
                        float* wc_sum, int threads);

void orc_filter(const float* acc, const float* variances7, int width, int height,
                int sample_count, int nwin, float* denoised, float* wc_sum, int threads)
{
    orc_filter_variant(acc, variances7, width, height, sample_count, nwin, 0, denoised, wc_sum,
                       threads);
}

void orc_filter_variant(const float* acc, const float* variances7, int width, int height,
                        int sample_count, int nwin, int colour_variant, float* denoised,
                        float* wc_sum, int threads)
{
    int W = width, H = height;
    size_t P = (size_t)W * (size_t)H;
    float n = (float)sample_count;
    if (threads < 1) threads = 1;
    const int plane[5] = {ORC_PL_COL, ORC_PL_NORMAL, ORC_PL_ALB1, ORC_PL_ALB2, ORC_PL_WPOS};

#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
    for (int jm = 0; jm < H; jm++) {
        for (int im = 0; im < W; im++) {
            size_t idx = (size_t)jm * W + im;
            float pv[5][3];
            for (int k = 0; k < 5; k++)
                for (int ch = 0; ch < 3; ch++) pv[k][ch] = acc[(plane[k] + ch) * P + idx] / n;
            float pdl = acc[ORC_PL_DL * P + idx] / n;
            float sd[6];
            for (int k = 0; k < 6; k++) sd[k] = acc[(ORC_PL_SD + k) * P + idx];
            const float* var = variances7 + 7 * idx;

            float wsum[3] = {0.0f, 0.0f, 0.0f};
            float wc = 0.0f;
            for (int j = -nwin; j <= nwin; j++) {
                int jf = clampi(jm + j, H);
                for (int i = -nwin; i <= nwin; i++) {
                    int ifx = clampi(im + i, W);
                    size_t q = (size_t)jf * W + ifx;
                    float qv[5][3];
                    for (int k = 0; k < 5; k++)
                        for (int ch = 0; ch < 3; ch++)
                            qv[k][ch] = acc[(plane[k] + ch) * P + q] / n;
                    float qdl = acc[ORC_PL_DL * P + q] / n;
                    float sd0q = acc[(ORC_PL_SD + 0) * P + q];

                    double dist2[5];
                    for (int k = 0; k < 5; k++) {
                        float dx = qv[k][0] - pv[k][0];
                        float dy = qv[k][1] - pv[k][1];
                        float dz = qv[k][2] - pv[k][2];
                        dist2[k] = (double)dx * (double)dx + (double)dy * (double)dy +
                                   (double)dz * (double)dz;
                    }
                    float e_idx = (float)(((double)(j * j) + (double)(i * i)) /
                                          (double)(2.0f * var[0]));
                    e_idx = expf(-e_idx);
                    float col_den = colour_variant ? (sd[0]) * 2.0f * var[1] + 0.000001f
                                                   : (sd[0] + sd0q) * 2.0f * var[1] + 0.000001f;
                    float e_col = (float)(dist2[0] / (double)col_den);
                    e_col = expf(-e_col);
                    float e_nrm = (float)(dist2[1] / (double)(sd[1] * 2.0f * var[2] + 0.000001f));
                    e_nrm = expf(-e_nrm);
                    float e_a1 = (float)(dist2[2] / (double)(sd[2] * 2.0f * var[3] + 0.000001f));
                    e_a1 = expf(-e_a1);
                    float e_a2 = (float)(dist2[3] / (double)(sd[3] * 2.0f * var[4] + 0.000001f));
                    e_a2 = expf(-e_a2);
                    float wp_den = colour_variant ? sd[4] * 2.0f * var[5] + 0.000001f
                                                  : 2.0f * sd[4] * var[5] + 0.000001f;
                    float e_wp = (float)(dist2[4] / (double)wp_den);
                    e_wp = expf(-e_wp);
                    float ddl = qdl - pdl;
                    float e_dl = (float)(((double)ddl * (double)ddl) /
                                         (double)(sd[5] * 2.0f * var[6] + 0.000001f));
                    e_dl = expf(-e_dl);

                    float w = e_idx * e_col * e_nrm * e_a1 * e_a2 * e_wp * e_dl;
                    wsum[0] += qv[0][0] * w;
                    wsum[1] += qv[0][1] * w;
                    wsum[2] += qv[0][2] * w;
                    wc += w;
                }
            }
            denoised[3 * idx + 0] = wsum[0] / wc;
            denoised[3 * idx + 1] = wsum[1] / wc;
            denoised[3 * idx + 2] = wsum[2] / wc;
            wc_sum[idx] = wc;
        }
    }
}

/* ------------------------------------------------------------------ loaders */

/* AABB::UpdateRot, headers/Obj.h:122-128: q = (qx * qy) * qz. */
static void euler_to_quat(const float* rot, float* q)
{
    float qx[4] = {cosf(rot[0] / 2.0f), sinf(rot[0] / 2.0f), 0, 0};
    float qz[4] = {cosf(rot[2] / 2.0f), 0, 0, sinf(rot[2] / 2.0f)};
    float qy[4] = {cosf(rot[1] / 2.0f), 0, sinf(rot[1] / 2.0f), 0};
    quat_mul(qx, qy);
    quat_mul(qx, qz);
    q[0] = qx[0]; q[1] = qx[1]; q[2] = qx[2]; q[3] = qx[3];
}

/* Scene::LoadScene, headers/Scene.h:127-224; camera defaults PT.cpp:62-72. */
int orc_load_scene(const char* path, orc_scene* s)
{
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    memset(s, 0, sizeof(*s));
    s->cam_pos[2] = -9; s->cam_fwd[2] = 1; s->cam_up[1] = 1; s->cam_right[0] = 1;
    s->focal_len = 1.0f; s->hfov = 120; s->vfov = 90; s->exposure = 2.0f; s->gamma = 2.1f;
    char line[1024];
    int cur = -1;
    while (fgets(line, sizeof line, f)) {
        if (!strncmp(line, "o ", 2)) {
            sscanf(line, "o %i", &cur);
        } else if (!strncmp(line, "i ", 2)) {
            int ind = 0;
            sscanf(line, "i %i", &ind);
            if (s->n_imp < 64) s->imp[s->n_imp++] = ind;
        } else if (!strncmp(line, "c ", 2)) {
            sscanf(line, "c %f/%f/%f %f/%f/%f %f/%f/%f %f/%f/%f %f %f",
                   &s->cam_pos[0], &s->cam_pos[1], &s->cam_pos[2],
                   &s->cam_fwd[0], &s->cam_fwd[1], &s->cam_fwd[2],
                   &s->cam_up[0], &s->cam_up[1], &s->cam_up[2],
                   &s->cam_right[0], &s->cam_right[1], &s->cam_right[2], &s->vfov, &s->hfov);
        } else if (!strncmp(line, "m ", 2)) {
            if (s->n_mats >= 64) { fclose(f); return -2; }
            float* m = s->mats[s->n_mats++];
            int type = 0;
            m[0] = m[1] = m[2] = 1; m[3] = 0; m[4] = 1;
            sscanf(line, "m %f/%f/%f %f %f %i", &m[0], &m[1], &m[2], &m[3], &m[4], &type);
            m[5] = (float)type;
        } else if (!strncmp(line, "d ", 2)) {
            if (s->n_shapes >= 64) { fclose(f); return -2; }
            int k = s->n_shapes;
            if (cur == 0) {
                s->shape_type[k] = 0;
                s->sphere_r[k] = 1;
                sscanf(line, "d %f/%f/%f %f %i", &s->shape_pos[k][0], &s->shape_pos[k][1],
                       &s->shape_pos[k][2], &s->sphere_r[k], &s->shape_mat[k]);
            } else if (cur == 1) {
                s->shape_type[k] = 1;
                sscanf(line, "d %f/%f/%f %f/%f/%f %f/%f/%f %f/%f/%f %i",
                       &s->shape_pos[k][0], &s->shape_pos[k][1], &s->shape_pos[k][2],
                       &s->box_min[k][0], &s->box_min[k][1], &s->box_min[k][2],
                       &s->box_max[k][0], &s->box_max[k][1], &s->box_max[k][2],
                       &s->box_rot[k][0], &s->box_rot[k][1], &s->box_rot[k][2],
                       &s->shape_mat[k]);
                euler_to_quat(s->box_rot[k], s->box_q[k]);
            } else {
                fclose(f);
                return -3;
            }
            s->n_shapes++;
        } else if (!strncmp(line, "pp ", 3)) {
            sscanf(line, "pp %f %f ", &s->exposure, &s->gamma);
        }
    }
    fclose(f);
    return 0;
}

/* Renderers::UpdateConstants + UpdateCam, Renderers.cpp:1051-1149. */
void orc_build_constants(const orc_scene* s, int width, int height, int max_depth,
                         float rand_samp, int denoising, orc_constants* c)
{
    memset(c, 0, sizeof(*c));
    for (int k = 0; k < 3; k++) {
        c->cam_pos[k] = s->cam_pos[k];
        c->cam_fwd[k] = s->cam_fwd[k];
        c->cam_right[k] = s->cam_right[k];
        c->cam_up[k] = s->cam_up[k];
    }
    c->res_h = width;
    c->res_v = height;
    c->max_angle_h = (float)tan(ORC_PI * (double)s->hfov / (double)360.0f);
    c->max_angle_v = (float)tan(ORC_PI * (double)s->vfov / (double)360.0f);
    c->focal_length = s->focal_len;
    c->max_depth = max_depth;
    c->rand_samp = rand_samp;
    int ni = s->n_imp < 10 ? s->n_imp : 10;
    for (int i = 0; i < ni; i++) c->imp[i] = (uint32_t)s->imp[i];
    c->num_imp = (uint32_t)ni;
    c->want_features = denoising ? 1 : 0;
    int nm = s->n_mats < 50 ? s->n_mats : 50;
    for (int m = 0; m < nm; m++)
        for (int k = 0; k < 6; k++) c->mats[m][k] = s->mats[m][k];
    int ns = s->n_shapes < 50 ? s->n_shapes : 50;
    int na = 0;
    for (int k = 0; k < ns; k++) {
        c->shapes[k][0] = s->shape_type[k];
        c->shapes[k][1] = s->shape_mat[k];
        c->shapes[k][2] = na;
        if (s->shape_type[k] == 0) {
            c->attrs[na++] = s->shape_pos[k][0];
            c->attrs[na++] = s->shape_pos[k][1];
            c->attrs[na++] = s->shape_pos[k][2];
            c->attrs[na++] = s->sphere_r[k];
        } else {
            for (int a = 0; a < 3; a++) c->attrs[na++] = s->shape_pos[k][a];
            for (int a = 0; a < 3; a++) c->attrs[na++] = s->shape_pos[k][a] + s->box_min[k][a];
            for (int a = 0; a < 3; a++) c->attrs[na++] = s->shape_pos[k][a] + s->box_max[k][a];
            for (int a = 0; a < 4; a++) c->attrs[na++] = s->box_q[k][a];
        }
    }
    c->num_shapes = ns;
}

/* DenoiserNN::LoadWeights, DenoiserNN.cpp:1625-1663: three lines of
 * whitespace-separated floats (360, 100, 70). */
int orc_load_weights(const char* path, float* w12, float* w23, float* w34)
{
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    float* dst[3] = {w12, w23, w34};
    const int cnt[3] = {360, 100, 70};
    size_t cap = 1 << 16;
    char* line = (char*)malloc(cap);
    for (int l = 0; l < 3; l++) {
        if (!fgets(line, (int)cap, f)) { free(line); fclose(f); return -2; }
        char* p = line;
        for (int i = 0; i < cnt[l]; i++) {
            char* end;
            float v = strtof(p, &end);
            if (end == p) v = 0.0f; /* stream extraction failure leaves zero */
            dst[l][i] = v;
            p = end;
        }
    }
    free(line);
    fclose(f);
    return 0;
}

/* CUDAPostProcess (CUDARender.cu:959-1022): the display source of each metric
 * (CUDARender::PostProcess, 1024-1066), then / exposure, then pow(., 1/g).
 * `denoised` is the interleaved screen metrics 1 and 7 display as it is: the
 * filter's output (denoisedCol) or the training target (targetCol). */
int orc_post_process(const float* acc, const float* denoised, int width, int height,
                     int sample_count, int metric, float exposure, float gamma, float* post)
{
    size_t P = (size_t)width * (size_t)height;
    float n = (float)sample_count;
    if (metric < 0 || metric > 7 || ((metric == 1 || metric == 7) && !denoised)) return -1;
    for (size_t i = 0; i < P; i++) {
        float c[3];
        for (int ch = 0; ch < 3; ch++) {
            switch (metric) {
            case 0: c[ch] = acc[(ORC_PL_COL + ch) * P + i] / n; break;
            case 1: case 7: c[ch] = denoised[3 * i + ch]; break;
            case 2: c[ch] = (acc[(ORC_PL_NORMAL + ch) * P + i] / n + 1.0f) / 2.0f; break;
            case 3: c[ch] = acc[(ORC_PL_ALB1 + ch) * P + i] / n; break;
            case 4: c[ch] = acc[(ORC_PL_ALB2 + ch) * P + i] / n; break;
            case 5: c[ch] = acc[ORC_PL_DL * P + i] / n; break;
            default: c[ch] = 1.0f / (acc[(ORC_PL_WPOS + ch) * P + i] / n); break;
            }
            c[ch] = c[ch] / exposure;
            post[3 * i + ch] = (float)LM_POW(c[ch], 1.0f / gamma);
        }
    }
    return 0;
}

/* PT::SaveImage's quantisation and row order (PT.cpp:484-501) of a post-processed
 * screen; NaN and out-of-range values (undefined in the reference's cast) clamp. */
void orc_quantise8(const float* post, int width, int height, uint8_t* out)
{
    for (int j = height - 1; j >= 0; --j)
        for (int i = 0; i < width; ++i)
            for (int ch = 0; ch < 3; ch++) {
                float v = post[3 * ((size_t)j * width + i) + ch];
                int q = (int)(255.999 * (double)v);
                if (!(v == v)) q = 0;
                if (q < 0) q = 0;
                if (q > 255) q = 255;
                *out++ = (uint8_t)q;
            }
}

/* CUDAPostProcess mode 0 (CUDARender.cu:971-975, 1013-1020) followed by
 * PT::SaveImage's int(255.999 * c) and row flip (PT.cpp:484-501). */
void orc_tonemap8(const float* acc, int width, int height, int sample_count,
                  float exposure, float gamma, uint8_t* out)
{
    size_t P = (size_t)width * (size_t)height;
    for (int j = height - 1; j >= 0; --j) {
        for (int i = 0; i < width; ++i) {
            size_t idx = (size_t)j * width + i;
            for (int ch = 0; ch < 3; ch++) {
                float v = acc[(ORC_PL_COL + ch) * P + idx] / (float)sample_count;
                v = v / exposure;
                v = powf(v, 1.0f / gamma);
                int q = (int)(255.999 * (double)v);
                if (!(v == v)) q = 0;
                if (q < 0) q = 0;
                if (q > 255) q = 255;
                *out++ = (uint8_t)q;
            }
        }
    }
}
