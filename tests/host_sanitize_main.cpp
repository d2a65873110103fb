// host_sanitize_main.cpp -- drives the host-only half of the library (scene and weight file
// parsers of the facade, scene validation, shape compilation, hierarchy builder and verifier,
// band partition, argument checks) over hostile inputs.  Built by tests/test_host_sanitizers.py
// with AddressSanitizer and UndefinedBehaviorSanitizer on the host code; needs no GPU.
//
//   host_sanitize FILE...      every *.scene / *.txt file is parsed, converted and re-saved
//   host_sanitize --tables N   N random table sets through ptk_check_scene
//
// Exit status 0 = every call returned (whatever its verdict); the sanitizers abort otherwise.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "ptk.h"

extern "C" {
int ptkhost_scene_constants(const char*, int, int, int, float, int, void*, float*, float*);
int ptkhost_scene_tables(const char*, int, int, int, float, int, ptk_scene_desc*, int32_t*, int32_t*,
                         float*, float*, uint32_t*, float*, float*);
int ptkhost_load_weights(const char*, float*, float*, float*);
int ptkhost_resave_scene(const char*, const char*);
int ptkhost_resave_weights(const char*, const char*);
}

static uint64_t rng_state = 0x2545F4914F6CDD1Dull;
static uint64_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}
static float uniform(float lo, float hi) { return lo + (hi - lo) * (float)((rnd() >> 40) * (1.0 / 16777216.0)); }

static bool ends_with(const std::string& s, const char* suffix)
{
    const size_t n = std::strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

static int run_scene_file(const std::string& path, const std::string& scratch, int* accepted)
{
    ptk_constants k;
    float exposure = 0, gamma = 0;
    if (ptkhost_scene_constants(path.c_str(), 96, 54, 4, 0.005f, 1, &k, &exposure, &gamma) != 0) return 0;
    ptk_scene_desc d;
    if (ptkhost_scene_tables(path.c_str(), 96, 54, 4, 0.005f, 1, &d, nullptr, nullptr, nullptr, nullptr,
                             nullptr, nullptr, nullptr) != 0)
        return 0;
    // exactly the sizes the query reported: an overrun of one element is a sanitizer report
    std::vector<int32_t> type((size_t)d.num_shapes), mat((size_t)d.num_shapes);
    std::vector<float> attrs((size_t)d.num_shapes * PTK_SCENE_ATTRS), mats((size_t)d.num_mats * 6);
    std::vector<uint32_t> imp((size_t)d.num_important_shapes);
    if (ptkhost_scene_tables(path.c_str(), 96, 54, 4, 0.005f, 1, &d, type.data(), mat.data(), attrs.data(),
                             mats.data(), imp.data(), &exposure, &gamma) != 0)
        return 1;
    d.shape_type = type.data();
    d.shape_mat = mat.data();
    d.shape_attrs = attrs.data();
    d.mat_list = mats.data();
    d.important_shapes = imp.data();
    ptk_scene_stats st;
    if (ptk_check_scene(&d, &st) == PTK_OK) ++*accepted;
    d.flags = PTK_SCENE_VISIT_ALL;
    ptk_check_scene(&d, nullptr);
    ptkhost_resave_scene(path.c_str(), scratch.c_str());
    return 0;
}

static int run_weight_file(const std::string& path, const std::string& scratch, int* accepted)
{
    std::vector<float> a(360), b(100), c(70);   // exact sizes
    if (ptkhost_load_weights(path.c_str(), a.data(), b.data(), c.data()) == 0) {
        ++*accepted;
        ptkhost_resave_weights(path.c_str(), scratch.c_str());
    }
    return 0;
}

static float hostile_float()
{
    switch (rnd() % 12) {
    case 0: return std::numeric_limits<float>::quiet_NaN();
    case 1: return std::numeric_limits<float>::infinity();
    case 2: return -std::numeric_limits<float>::infinity();
    case 3: return 3.0e38f;
    case 4: return -3.0e38f;
    case 5: return 1.0e-42f;
    case 6: return 0.0f;
    default: return uniform(-50.0f, 50.0f);
    }
}

// One random table set: mostly well-formed shapes, with a chosen amount of poison.
static void run_tables(int n, int poison_percent, int* accepted)
{
    std::vector<int32_t> type((size_t)n), mat((size_t)n);
    std::vector<float> attrs((size_t)n * PTK_SCENE_ATTRS, 0.0f);
    const int num_mats = 1 + (int)(rnd() % 6);
    std::vector<float> mats((size_t)num_mats * 6);
    for (int m = 0; m < num_mats; m++) {
        float* r = &mats[(size_t)m * 6];
        r[0] = uniform(0, 1); r[1] = uniform(0, 1); r[2] = uniform(0, 1);
        r[3] = rnd() % 4 ? 0.0f : uniform(0, 1);
        r[4] = uniform(1, 2);
        r[5] = (float)(rnd() % 4);
    }
    auto poisoned = [&] { return (int)(rnd() % 100) < poison_percent; };
    const bool coincident = rnd() % 5 == 0;
    for (int s = 0; s < n; s++) {
        float* a = &attrs[(size_t)s * PTK_SCENE_ATTRS];
        type[s] = (int32_t)(rnd() % 2);
        mat[s] = (int32_t)(rnd() % (uint64_t)num_mats);
        const float cx = coincident ? 1.0f : uniform(-20, 20), cy = coincident ? 2.0f : uniform(-20, 20),
                    cz = coincident ? 3.0f : uniform(-20, 20);
        a[0] = cx; a[1] = cy; a[2] = cz;
        if (type[s] == 0) {
            a[3] = uniform(0.05f, 1.5f);
        } else {
            const float hx = uniform(0.0f, 1.0f), hy = uniform(0.0f, 1.0f), hz = uniform(0.0f, 1.0f);
            a[3] = cx - hx; a[4] = cy - hy; a[5] = cz - hz;
            a[6] = cx + hx; a[7] = cy + hy; a[8] = cz + hz;
            if (rnd() % 3 == 0) {
                float q[4] = {uniform(-1, 1), uniform(-1, 1), uniform(-1, 1), uniform(-1, 1)};
                const float len = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]) + 1e-9f;
                for (int i = 0; i < 4; i++) a[9 + i] = q[i] / len;
            } else {
                a[9] = 1.0f;
            }
        }
        if (poisoned()) a[rnd() % PTK_SCENE_ATTRS] = hostile_float();
        if (poisoned()) type[s] = (int32_t)(rnd() % 7) - 2;
        if (poisoned()) mat[s] = (int32_t)(rnd() % 70000) - 100;
    }
    const int n_imp = n ? (int)(rnd() % (uint64_t)(n < 12 ? n + 1 : 12)) : 0;
    std::vector<uint32_t> imp((size_t)n_imp);
    for (auto& i : imp) i = poisoned() ? (uint32_t)rnd() : (uint32_t)(rnd() % (uint64_t)n);
    ptk_scene_desc d;
    std::memset(&d, 0, sizeof d);
    d.cam_pos[2] = -30.0f; d.cam_forward[2] = 1.0f; d.cam_right[0] = 1.0f; d.cam_up[1] = 1.0f;
    if (poisoned()) d.cam_pos[rnd() % 3] = hostile_float();
    d.rand_samp = 0.005f;
    d.max_depth = poisoned() ? (int32_t)(rnd() % 4000) - 2000 : 4;
    d.res_v = poisoned() ? (int32_t)(rnd() % 200000) - 1000 : 54;
    d.res_h = poisoned() ? (int32_t)(rnd() % 200000) - 1000 : 96;
    d.max_angle_v = 0.7f; d.max_angle_h = 1.2f; d.focal_length = 1.0f;
    d.num_shapes = n; d.num_mats = num_mats; d.num_important_shapes = n_imp;
    d.shape_type = type.data(); d.shape_mat = mat.data(); d.shape_attrs = attrs.data();
    d.mat_list = mats.data(); d.important_shapes = imp.data();
    ptk_scene_stats st;
    if (ptk_check_scene(&d, &st) == PTK_OK) ++*accepted;
}

static void run_argument_checks()
{
    // counts out of range and missing tables must be refused before anything is read
    ptk_scene_desc d;
    std::memset(&d, 0, sizeof d);
    ptk_check_scene(nullptr, nullptr);
    ptk_check_scene(&d, nullptr);
    for (int32_t n : {-1, 1, 50, PTK_SCENE_MAX_SHAPES, PTK_SCENE_MAX_SHAPES + 1, 0x7fffffff}) {
        d.num_shapes = n; d.num_mats = 1; d.res_v = 54; d.res_h = 96; d.max_depth = 4;
        if (ptk_check_scene(&d, nullptr) == PTK_OK && n != 0) {
            std::fprintf(stderr, "a scene of %d shapes without tables was accepted\n", n);
            std::exit(1);
        }
    }
    int lo, hi, mx;
    for (int h : {-5, 0, 1, 2, 7, 54, 2160, 1 << 30})
        for (int w : {-1, 0, 1, 2, 3, 8, 64, 65, 1000})
            for (int halo : {-1, 0, 3, 8, 100000})
                for (int r : {-1, 0, 1, w - 1, w}) {
                    ptk_band_layout(h, w, r, halo, &lo, &hi, &mx);
                    ptk_band_layout(h, w, r, halo, nullptr, nullptr, nullptr);
                }
    // entry points that need a context refuse a null one
    float w12[360] = {0}, w23[100] = {0}, w34[70] = {0};
    ptk_constants k;
    std::memset(&k, 0, sizeof k);
    if (ptk_set_constants(nullptr, &k, sizeof k, -1) == PTK_OK || ptk_render(nullptr, 1, PTK_SEED_COUNTER, nullptr, nullptr) == PTK_OK ||
        ptk_denoise(nullptr, w12, w23, w34, 1, 0, nullptr, nullptr, nullptr, nullptr) == PTK_OK) {
        std::fprintf(stderr, "a call on a null context succeeded\n");
        std::exit(1);
    }
}

int main(int argc, char** argv)
{
    const char* tmp = std::getenv("PTK_SANITIZE_SCRATCH");
    const std::string scratch = std::string(tmp ? tmp : "/tmp") + "/host_sanitize_resave.tmp";
    int accepted = 0, seen = 0;
    if (argc >= 3 && std::string(argv[1]) == "--tables") {
        const int rounds = std::atoi(argv[2]);
        run_argument_checks();
        const int sizes[] = {0, 1, 2, 3, 4, 5, 7, 50, 51, 333, 4096};
        for (int r = 0; r < rounds; r++) {
            const int n = r % 40 == 39 ? 40000 : sizes[rnd() % (sizeof sizes / sizeof sizes[0])];
            const int poison = r % 3 == 0 ? 0 : (r % 3 == 1 ? 1 : 20);
            run_tables(n, poison, &accepted);
            seen++;
        }
        std::printf("tables: %d sets, %d accepted\n", seen, accepted);
        return accepted > 0 && accepted < seen ? 0 : 1;   // both verdicts must have occurred
    }
    for (int i = 1; i < argc; i++) {
        const std::string path = argv[i];
        int rc = 0;
        if (ends_with(path, ".scene")) rc = run_scene_file(path, scratch, &accepted);
        else rc = run_weight_file(path, scratch, &accepted);
        if (rc) { std::fprintf(stderr, "%s: the fill call failed after the query succeeded\n", path.c_str()); return 1; }
        seen++;
    }
    std::printf("files: %d parsed, %d accepted\n", seen, accepted);
    return 0;
}
