// ptk_host.cpp -- the reference-shaped host façade over libptk (see ptk_host.h).
#include "ptk_host.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace ptkhost {

namespace {

// Splits a scene line into numeric fields: '/' and blanks both separate, the
// leading tag is dropped.  Each field is read with strtof/strtol, which is what
// the reference's sscanf("%f") / ("%i") conversions do.
std::vector<std::string> fields_of(const std::string& line)
{
    std::vector<std::string> out;
    std::string cur;
    size_t i = line.find(' ');
    if (i == std::string::npos) return out;
    for (; i < line.size(); i++) {
        char ch = line[i];
        if (ch == '/' || ch == ' ' || ch == '\t' || ch == '\r' || ch == '\n') {
            if (!cur.empty()) { out.push_back(cur); cur.clear(); }
        } else {
            cur.push_back(ch);
        }
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}

float f_at(const std::vector<std::string>& f, size_t i, float dflt)
{
    return i < f.size() ? std::strtof(f[i].c_str(), nullptr) : dflt;
}

int i_at(const std::vector<std::string>& f, size_t i, int dflt)
{
    return i < f.size() ? (int)std::strtol(f[i].c_str(), nullptr, 0) : dflt;
}

// Quaternion product in the reference's 9-multiply form (headers/Obj.h:26-40).
void quat_mul(const float* l, const float* r, float* o)
{
    float a1 = (l[3] + l[1]) * (r[1] + r[2]);
    float a3 = (l[0] - l[2]) * (r[0] + r[3]);
    float a4 = (l[0] + l[2]) * (r[0] - r[3]);
    float a2 = a1 + a3 + a4;
    float a5 = (l[3] - l[1]) * (r[1] - r[2]);
    a5 = (a5 + a2) / 2.0f;
    float w = a5 - a1 + (l[3] - l[2]) * (r[2] - r[3]);
    float x = a5 - a2 + (l[1] + l[0]) * (r[1] + r[0]);
    float y = a5 - a3 + (l[0] - l[1]) * (r[2] + r[3]);
    float z = a5 - a4 + (l[3] + l[2]) * (r[0] - r[1]);
    o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}

}  // namespace

void Obj::UpdateRot()
{
    float qx[4] = {cosf(rot.x / 2.0f), sinf(rot.x / 2.0f), 0, 0};
    float qz[4] = {cosf(rot.z / 2.0f), 0, 0, sinf(rot.z / 2.0f)};
    float qy[4] = {cosf(rot.y / 2.0f), 0, sinf(rot.y / 2.0f), 0};
    float t[4];
    quat_mul(qx, qy, t);
    quat_mul(t, qz, q);
}

// ------------------------------------------------------------------ Scene

void Scene::ResetScene()
{
    objList.clear();
    matList.clear();
    importantList.clear();
}

void Scene::AddToImpList(int index)
{
    importantList.push_back(index);
    if (index >= 0 && index < (int)objList.size()) objList[index].inImportantList = true;
}

void Scene::InitScene()
{
    ResetScene();
    const float wall[6][3] = {{0.65f, .05f, .05f}, {0.73f, 0.73f, 0.73f}, {0.73f, 0.73f, 0.73f},
                              {0.73f, 0.73f, 0.73f}, {0.73f, 0.73f, 0.73f}, {.12f, 0.45f, .15f}};
    for (auto& w : wall) { Mat m; m.alb = {w[0], w[1], w[2]}; matList.push_back(m); }
    Mat light; light.matType = 1; matList.push_back(light);
    auto box = [&](Vec3 p, int mat, Vec3 lo, Vec3 hi) {
        Obj o; o.type = 1; o.pos = p; o.mat_ind = mat; o.min = lo; o.max = hi;
        objList.push_back(o);
    };
    const float d = 10, r = 10;
    box({0, d - 1.0f, 0}, 6, {-2.5f, 0, -2.5f}, {2.5f, 1.0f, 2.5f});
    AddToImpList(0);
    box({0, 0, d}, 1, {-r, -r, 0}, {r, r, 0});
    box({d, 0, 0}, 5, {0, -r, -r}, {0, r, r});
    box({-d, 0, 0}, 0, {0, -r, -r}, {0, r, r});
    box({0, d, 0}, 2, {-r, 0, -r}, {r, 0, r});
    box({0, -d, 0}, 3, {-r, 0, -r}, {r, 0, r});
    box({0, 0, -d}, 4, {-r, -r, 0}, {r, r, 0});
}

bool Scene::LoadScene(const std::string& path)
{
    std::ifstream in(path);
    if (!in) {
        std::fprintf(stderr, "Cannot open scene file %s\n", path.c_str());
        return false;
    }
    ResetScene();
    std::string line;
    int currentType = -1;
    while (std::getline(in, line)) {
        auto f = fields_of(line);
        if (line.compare(0, 2, "o ") == 0) {
            currentType = i_at(f, 0, currentType);
        } else if (line.compare(0, 2, "i ") == 0) {
            AddToImpList(i_at(f, 0, 0));
        } else if (line.compare(0, 2, "c ") == 0) {
            Camera& c = st_->cam;
            c.pos = {f_at(f, 0, 0), f_at(f, 1, 0), f_at(f, 2, 0)};
            c.forward = {f_at(f, 3, 0), f_at(f, 4, 0), f_at(f, 5, 0)};
            c.up = {f_at(f, 6, 0), f_at(f, 7, 0), f_at(f, 8, 0)};
            c.right = {f_at(f, 9, 0), f_at(f, 10, 0), f_at(f, 11, 0)};
            c.vfov = f_at(f, 12, c.vfov);
            c.hfov = f_at(f, 13, c.hfov);
        } else if (line.compare(0, 2, "m ") == 0) {
            Mat m;
            m.alb = {f_at(f, 0, 1), f_at(f, 1, 1), f_at(f, 2, 1)};
            m.blur = f_at(f, 3, 0);
            m.RI = f_at(f, 4, 1);
            m.matType = i_at(f, 5, 0);
            matList.push_back(m);
        } else if (line.compare(0, 2, "d ") == 0) {
            Obj o;
            if (currentType == 0) {
                o.type = 0;
                o.pos = {f_at(f, 0, 0), f_at(f, 1, 0), f_at(f, 2, 0)};
                o.r = f_at(f, 3, 1);
                o.mat_ind = i_at(f, 4, 0);
            } else if (currentType == 1) {
                o.type = 1;
                o.pos = {f_at(f, 0, 0), f_at(f, 1, 0), f_at(f, 2, 0)};
                o.min = {f_at(f, 3, 0), f_at(f, 4, 0), f_at(f, 5, 0)};
                o.max = {f_at(f, 6, 0), f_at(f, 7, 0), f_at(f, 8, 0)};
                o.rot = {f_at(f, 9, 0), f_at(f, 10, 0), f_at(f, 11, 0)};
                o.UpdateRot();
                o.mat_ind = i_at(f, 12, 0);
            } else {
                std::fprintf(stderr, "Error parsing .scene file, unknown shape type!\n");
                return false;
            }
            objList.push_back(o);
        } else if (line.compare(0, 3, "pp ") == 0) {
            st_->exposure = f_at(f, 0, st_->exposure);
            st_->g = f_at(f, 1, st_->g);
        }
    }
    return true;
}

bool Scene::SaveScene(const std::string& path)
{
    std::ofstream out(path);
    if (!out) return false;
    for (auto& m : matList)
        out << "m " << m.alb.x << "/" << m.alb.y << "/" << m.alb.z << " " << m.blur << " " << m.RI
            << " " << m.matType << "\n";
    for (auto& o : objList) {
        out << "o " << o.type << "\n";
        if (o.type == 0)
            out << "d " << o.pos.x << "/" << o.pos.y << "/" << o.pos.z << " " << o.r << " "
                << o.mat_ind << "\n";
        else
            out << "d " << o.pos.x << "/" << o.pos.y << "/" << o.pos.z << " " << o.min.x << "/"
                << o.min.y << "/" << o.min.z << " " << o.max.x << "/" << o.max.y << "/" << o.max.z
                << " " << o.rot.x << "/" << o.rot.y << "/" << o.rot.z << " " << o.mat_ind << "\n";
    }
    for (int i : importantList) out << "i " << i << "\n";
    const Camera& c = st_->cam;
    out << "c " << c.pos.x << "/" << c.pos.y << "/" << c.pos.z << " " << c.forward.x << "/"
        << c.forward.y << "/" << c.forward.z << " " << c.up.x << "/" << c.up.y << "/" << c.up.z
        << " " << c.right.x << "/" << c.right.y << "/" << c.right.z << " " << c.vfov << " "
        << c.hfov << "\n";
    out << "pp " << st_->exposure << " " << st_->g << "\n";
    return (bool)out;
}

// -------------------------------------------------------------- Renderers

namespace {
void fill_constants(const HostState& st, const Scene& scene, ptk_constants& k)
{
    std::memset(&k, 0, sizeof k);
    const Camera& c = st.cam;
    const float* src[4] = {&c.pos.x, &c.forward.x, &c.right.x, &c.up.x};
    float* dst[4] = {k.cam_pos, k.cam_forward, k.cam_right, k.cam_up};
    for (int v = 0; v < 4; v++)
        for (int i = 0; i < 3; i++) dst[v][i] = src[v][i];
    k.res_h = st.xRes;
    k.res_v = st.yRes;
    const double pi = 3.14159265358979323846;
    k.max_angle_h = (float)std::tan(pi * (double)c.hfov / (double)360.0f);
    k.max_angle_v = (float)std::tan(pi * (double)c.vfov / (double)360.0f);
    k.focal_length = c.focalLen;
    k.max_depth = st.maxDepth;
    k.rand_samp = st.randSamp;
    size_t ni = scene.importantList.size() < 10 ? scene.importantList.size() : 10;
    for (size_t i = 0; i < ni; i++) k.important_shapes[i] = (uint32_t)scene.importantList[i];
    k.num_important_shapes = (uint32_t)ni;
    k.get_denoiser_inf = st.denoising ? 1 : 0;
    size_t nm = scene.matList.size() < 50 ? scene.matList.size() : 50;
    for (size_t m = 0; m < nm; m++) {
        const Mat& mm = scene.matList[m];
        float row[6] = {mm.alb.x, mm.alb.y, mm.alb.z, mm.blur, mm.RI, (float)mm.matType};
        std::memcpy(k.mat_list[m], row, sizeof row);
    }
    size_t ns = scene.objList.size() < 50 ? scene.objList.size() : 50;
    int na = 0;
    int added = 0;
    for (size_t s = 0; s < ns; s++) {
        const Obj& o = scene.objList[s];
        int need = o.type == 0 ? 4 : 13;
        if (na + need > PTK_MAX_ATTRS) break;  // the reference overruns here; stop instead
        k.shapes[added][0] = o.type;
        k.shapes[added][1] = o.mat_ind;
        k.shapes[added][2] = na;
        float* a = k.obj_attributes + na;
        a[0] = o.pos.x; a[1] = o.pos.y; a[2] = o.pos.z;
        if (o.type == 0) {
            a[3] = o.r;
        } else {
            a[3] = o.pos.x + o.min.x; a[4] = o.pos.y + o.min.y; a[5] = o.pos.z + o.min.z;
            a[6] = o.pos.x + o.max.x; a[7] = o.pos.y + o.max.y; a[8] = o.pos.z + o.max.z;
            for (int i = 0; i < 4; i++) a[9 + i] = o.q[i];
        }
        na += need;
        added++;
    }
    k.num_shapes = added;
}

// Whether the scene fits the reference's POD (50 shapes, 450 attribute floats, 50
// materials, 10 important shapes; headers/GLOBALS.h:80-94).
bool fits_pod(const Scene& scene)
{
    size_t attrs = 0;
    for (const Obj& o : scene.objList) attrs += o.type == 0 ? 4 : 13;
    return scene.objList.size() <= PTK_MAX_SHAPES && attrs <= PTK_MAX_ATTRS &&
           scene.matList.size() <= PTK_MAX_MATS && scene.importantList.size() <= PTK_MAX_IMPORTANT;
}

// The same tables without the limits, for ptk_set_scene.
struct SceneTables {
    ptk_scene_desc desc;
    std::vector<int32_t> type, mat;
    std::vector<float> attrs, mats;
    std::vector<uint32_t> imp;
};
void fill_tables(const HostState& st, const Scene& scene, const ptk_constants& k, SceneTables& t)
{
    std::memset(&t.desc, 0, sizeof t.desc);
    for (int i = 0; i < 3; i++) {
        t.desc.cam_pos[i] = k.cam_pos[i];
        t.desc.cam_forward[i] = k.cam_forward[i];
        t.desc.cam_right[i] = k.cam_right[i];
        t.desc.cam_up[i] = k.cam_up[i];
    }
    t.desc.rand_samp = k.rand_samp;
    t.desc.max_depth = k.max_depth;
    t.desc.res_v = k.res_v;
    t.desc.res_h = k.res_h;
    t.desc.max_angle_v = k.max_angle_v;
    t.desc.max_angle_h = k.max_angle_h;
    t.desc.focal_length = k.focal_length;
    const size_t n = scene.objList.size();
    t.type.assign(n, 0);
    t.mat.assign(n, 0);
    t.attrs.assign(n * PTK_SCENE_ATTRS, 0.0f);
    for (size_t s = 0; s < n; s++) {
        const Obj& o = scene.objList[s];
        t.type[s] = o.type;
        t.mat[s] = o.mat_ind;
        float* a = &t.attrs[s * PTK_SCENE_ATTRS];
        a[0] = o.pos.x; a[1] = o.pos.y; a[2] = o.pos.z;
        if (o.type == 0) {
            a[3] = o.r;
        } else {
            a[3] = o.pos.x + o.min.x; a[4] = o.pos.y + o.min.y; a[5] = o.pos.z + o.min.z;
            a[6] = o.pos.x + o.max.x; a[7] = o.pos.y + o.max.y; a[8] = o.pos.z + o.max.z;
            for (int i = 0; i < 4; i++) a[9 + i] = o.q[i];
        }
    }
    t.mats.assign(std::max<size_t>(1, scene.matList.size()) * 6, 0.0f);
    for (size_t m = 0; m < scene.matList.size(); m++) {
        const Mat& mm = scene.matList[m];
        const float row[6] = {mm.alb.x, mm.alb.y, mm.alb.z, mm.blur, mm.RI, (float)mm.matType};
        std::memcpy(&t.mats[m * 6], row, sizeof row);
    }
    t.imp.clear();
    for (int i : scene.importantList) t.imp.push_back((uint32_t)i);
    t.desc.num_shapes = (int32_t)n;
    t.desc.num_mats = (int32_t)(t.mats.size() / 6);
    t.desc.num_important_shapes = (int32_t)t.imp.size();
    t.desc.shape_type = t.type.data();
    t.desc.shape_mat = t.mat.data();
    t.desc.shape_attrs = t.attrs.data();
    t.desc.mat_list = t.mats.data();
    t.desc.important_shapes = t.imp.data();
    (void)st;
}

// Hands the scene to the device: the POD when it fits (the reference's path,
// CUDARender::UpdateConstants), the tables behind a hierarchy otherwise.
int push_scene(HostState& st, const ptk_constants& k)
{
    if (fits_pod(st.scene)) return ptk_set_constants(st.ctx, &k, sizeof k, -1);
    SceneTables t;
    fill_tables(st, st.scene, k, t);
    return ptk_set_scene(st.ctx, &t.desc, k.get_denoiser_inf == 1 ? 1 : 0, nullptr);
}
}  // namespace

void Renderers::UpdateCam(bool push)
{
    const Camera& c = st_->cam;
    const float* src[4] = {&c.pos.x, &c.forward.x, &c.right.x, &c.up.x};
    float* dst[4] = {constants.cam_pos, constants.cam_forward, constants.cam_right, constants.cam_up};
    for (int v = 0; v < 4; v++)
        for (int i = 0; i < 3; i++) dst[v][i] = src[v][i];
    constants.res_h = st_->xRes;
    constants.res_v = st_->yRes;
    if (push && st_->ctx) {
        // only the camera moved since the scene was handed over: keep the compiled scene
        // (and, for a scene beyond the POD, its hierarchy)
        int rc = -1;
        if (scenePushed_ && pushedW_ == st_->xRes && pushedH_ == st_->yRes)
            rc = ptk_set_camera(st_->ctx, constants.cam_pos, constants.cam_forward,
                                constants.cam_right, constants.cam_up);
        if (rc != PTK_OK) {
            st_->check(push_scene(*st_, constants), "ptk_set_constants / ptk_set_scene");
            scenePushed_ = st_->ok();
            pushedW_ = st_->xRes;
            pushedH_ = st_->yRes;
        }
        st_->sampleCount = 0;
    }
}

void Renderers::UpdateConstants()
{
    fill_constants(*st_, st_->scene, constants);
    // the reference reseeds from the wall clock here (Renderers.cpp:1062-1063);
    // a caller that wants a fixed stream overwrites GloRandS afterwards
    using namespace std::chrono;
    uint64_t ms = (uint64_t)duration_cast<milliseconds>(system_clock::now().time_since_epoch()).count();
    GloRandS[0] = ms;
    GloRandS[1] = ms;
    if (!st_->ctx) st_->check(ptk_create(st_->device, &st_->ctx), "ptk_create");
    if (!st_->ctx) return;
    st_->check(push_scene(*st_, constants), "ptk_set_constants / ptk_set_scene");
    scenePushed_ = st_->ok();
    pushedW_ = st_->xRes;
    pushedH_ = st_->yRes;
    if (!scenePushed_) return;   // a refused scene or resolution: no host screens of that size either
    st_->check(ptk_set_math_mode(st_->ctx, st_->mathMode), "ptk_set_math_mode");
    st_->sampleCount = 0;
    st_->InitScreens();
}

void Renderers::Render() { Render(1); }

void Renderers::Render(int spp)
{
    if (st_->currentRenderer != 2) {
        st_->lastError = "Renderers::Render: backend slot " + std::to_string(st_->currentRenderer) +
                         " is not built; only slot 2 (GPU) exists";
        return;
    }
    if (!st_->ctx) { st_->lastError = "Renderers::Render before UpdateConstants"; return; }
    uint64_t rays = 0;
    st_->check(ptk_render(st_->ctx, spp, st_->seedMode, GloRandS, &rays), "ptk_render");
    if (!st_->ok()) return;
    st_->sampleCount += spp;
    st_->rayCount = rays;
    ptk_times t;
    if (ptk_timing(st_->ctx, &t) == PTK_OK) st_->renderTime = t.render_ms;
}

void Renderers::AutoExposure()
{
    if (st_->currentRenderer != 2) {
        st_->lastError = "Renderers::AutoExposure: only backend slot 2 (GPU) exists";
        return;
    }
    if (!st_->ctx) { st_->lastError = "Renderers::AutoExposure before UpdateConstants"; return; }
    float e = 0;
    st_->check(ptk_auto_exposure(st_->ctx, PTK_EXPOSURE_CPU, &e, nullptr), "ptk_auto_exposure");
    if (!st_->ok()) return;
    st_->exposure = e;
    ptk_times t;
    if (ptk_timing(st_->ctx, &t) == PTK_OK) st_->exposureTime = t.exposure_ms;
}

// ------------------------------------------------------------- DenoiserNN

bool DenoiserNN::LoadWeights(const std::string& path)
{
    std::ifstream in(path);
    if (!in.is_open()) {
        std::fprintf(stderr, "Cannot open %s!\n", path.c_str());
        return false;
    }
    float* dst[3] = {onetwo, twothree, threefour};
    const int cnt[3] = {360, 100, 70};
    std::string line;
    for (int l = 0; l < 3; l++) {
        if (!std::getline(in, line)) {
            std::fprintf(stderr, "Invalid file %s!\n", path.c_str());
            return false;
        }
        std::istringstream ls(line);
        for (int i = 0; i < cnt[l]; i++) ls >> dst[l][i];
    }
    st_->weightsLoaded = true;
    return true;
}

void DenoiserNN::OutputWeights(const std::string& path)
{
    std::ofstream out(path);
    if (!out.is_open()) {
        std::fprintf(stderr, "Could not write weights file !\n");
        return;
    }
    for (int i = 0; i < 360; i++) out << onetwo[i] << " ";
    out << "\n";
    for (int i = 0; i < 100; i++) out << twothree[i] << " ";
    out << "\n";
    for (int i = 0; i < 70; i++) out << threefour[i] << " ";
    out << "\n";
}

void DenoiserNN::ForwardProp()
{
    // Inference runs as one device call together with the filter (the
    // bandwidths are consumed where they are produced); Denoiser::denoise()
    // drives it.  Called on its own, this runs the same call with N = 0 and
    // keeps only the bandwidths (and the features when asked).
    if (st_->denoisingBackend != 2) {
        st_->lastError = "DenoiserNN::ForwardProp: only backend slot 2 (GPU) exists";
        return;
    }
    if (!st_->ctx) { st_->lastError = "DenoiserNN::ForwardProp before UpdateConstants"; return; }
    size_t P = (size_t)st_->xRes * st_->yRes;
    st_->variances.resize(7 * P);
    if (st_->wantFeaturesOut) st_->sFeatures.resize(36 * P);
    st_->check(ptk_denoise(st_->ctx, onetwo, twothree, threefour, 0, 0, nullptr, nullptr,
                           st_->variances.data(),
                           st_->wantFeaturesOut ? st_->sFeatures.data() : nullptr),
               "ptk_denoise");
}

void DenoiserNN::RandomizeWeights()
{
    // xoroshiro128++ on the renderer's global state, (double)u64 / 2^63 / 2 scaled to
    // [-0.5, 0.5] (DenoiserNN.cpp:1567-1601)
    auto rotl = [](uint64_t x, int k) { return (x << k) | (x >> (64 - k)); };
    uint64_t* S = st_->renderer.GloRandS;
    auto next = [&]() {
        const uint64_t s0 = S[0];
        uint64_t s1 = S[1];
        const uint64_t r = rotl(s0 + s1, 17) + s0;
        s1 ^= s0;
        S[0] = rotl(s0, 49) ^ s1 ^ (s1 << 21);
        S[1] = rotl(s1, 28);
        return r;
    };
    auto between = [&](float lo, float hi) {
        double f = (double)next() / (double)(1ULL << 63) / 2.0;
        f *= hi - lo;
        f += lo;
        return (float)f;
    };
    for (float& w : onetwo) w = between(-0.5f, 0.5f);
    for (float& w : twothree) w = between(-0.5f, 0.5f);
    for (int i = 0; i < 70; i++) threefour[i] = between(-0.5f, 0.5f);
}

void DenoiserNN::InitTraining()
{
    if (!st_->ctx) { st_->lastError = "DenoiserNN::InitTraining before UpdateConstants"; return; }
    training = true;
    trainingTime = 0.0;
    st_->check(ptk_save_target(st_->ctx), "ptk_save_target");
    if (trainedWeightsPath.empty() || !LoadWeights(trainedWeightsPath)) RandomizeWeights();
    st_->weightsLoaded = true;
    trainingEpoch = 0;
    if (errorLogPath.empty()) return;
    std::ofstream log(errorLogPath);
    if (log.is_open())
        log << "Res: (" << st_->xRes << "x" << st_->yRes << ") Samples: " << samplesWhenTraining
            << " L Rate: " << learningRate << "," << std::endl;
}

void DenoiserNN::AppendTrainingFile()
{
    if (errorLogPath.empty()) return;
    std::ofstream log(errorLogPath, std::ios_base::app);
    if (log.is_open()) log << relMSE << "," << std::endl;
}

void DenoiserNN::EndTraining()
{
    training = false;
    if (!errorLogPath.empty()) {
        std::ofstream log(errorLogPath, std::ios_base::app);
        if (log.is_open()) log << trainingTime << std::endl;
    }
    if (!trainedWeightsPath.empty()) OutputWeights(trainedWeightsPath);
}

void DenoiserNN::TrainNN()
{
    if (!st_->ctx) { st_->lastError = "DenoiserNN::TrainNN before UpdateConstants"; return; }
    const auto t0 = std::chrono::high_resolution_clock::now();
    float inv = lRateInt;
    if (trainingLimitBool)
        inv = lRateInt + (lRateIntMax - lRateInt) * ((float)trainingEpoch / (float)trainingCount);
    learningRate = (float)(1.0 / std::pow(10.0, (double)inv));

    st_->check(ptk_reset_accumulators(st_->ctx), "ptk_reset_accumulators");
    st_->sampleCount = 0;
    st_->renderer.Render(samplesWhenTraining);
    if (!st_->ok()) return;
    // denoise, relMSE (of the weights before the update) and back-prop in one device call
    st_->check(ptk_train_step(st_->ctx, onetwo, twothree, threefour, st_->denoisingN, learningRate,
                              &relMSE),
               "ptk_train_step");
    if (!st_->ok()) return;
    AppendTrainingFile();
    trainingEpoch++;
    epochTime = std::chrono::duration<double, std::milli>(
                    std::chrono::high_resolution_clock::now() - t0).count();
    trainingTime += epochTime;
    if (trainingLimitBool && trainingEpoch == trainingCount) training = false;
}

// --------------------------------------------------------------- Denoiser

void Denoiser::denoise()
{
    if (st_->denoisingBackend != 2) {
        st_->lastError = "Denoiser::denoise: only backend slot 2 (GPU) exists";
        return;
    }
    if (!st_->ctx) { st_->lastError = "Denoiser::denoise before UpdateConstants"; return; }
    size_t P = (size_t)st_->xRes * st_->yRes;
    st_->denoisedCol.resize(3 * P);
    st_->wcSum.resize(P);
    st_->variances.resize(7 * P);
    DenoiserNN& nn = st_->denoiserNN;
    st_->check(ptk_denoise(st_->ctx, nn.onetwo, nn.twothree, nn.threefour, st_->denoisingN,
                           st_->denoiseVariant,
                           st_->denoisedCol.data(), st_->wcSum.data(), st_->variances.data(),
                           nullptr),
               "ptk_denoise");
    ptk_times t;
    if (st_->ok() && ptk_timing(st_->ctx, &t) == PTK_OK) st_->denoiseTime = t.denoise_ms;
}

// -------------------------------------------------------------- HostState

HostState::HostState() : scene(this), renderer(this), denoiserNN(this), denoiser(this)
{
    scene.InitScene();
}

HostState::~HostState()
{
    if (ctx) ptk_destroy(ctx);
}

void HostState::check(int rc, const char* what)
{
    if (rc != PTK_OK) lastError = std::string(what) + ": " + ptk_last_error();
}

void HostState::InitScreens()
{
    // sizes the library refuses (ptk_set_constants: 1 .. 2^28 pixels) get no host screens
    const bool valid = xRes > 0 && yRes > 0 && (int64_t)xRes * yRes <= ((int64_t)1 << 28);
    size_t P = valid ? (size_t)xRes * yRes : 0;
    preScreen.assign(3 * P, 0.0f);
    postScreen.assign(3 * P, 0.0f);
    normal.assign(3 * P, 0.0f);
    albedo1.assign(3 * P, 0.0f);
    albedo2.assign(3 * P, 0.0f);
    worldPos.assign(3 * P, 0.0f);
    directLight.assign(P, 0.0f);
    denoisedCol.assign(3 * P, 0.0f);
    stdDev.assign(6 * P, 0.0f);
    variances.assign(7 * P, 0.0f);
    wcSum.assign(P, 0.0f);
}

void HostState::SyncScreens()
{
    if (!ctx) { lastError = "SyncScreens before UpdateConstants"; return; }
    if (xRes < 1 || yRes < 1) { lastError = "SyncScreens: no image"; return; }
    if (preScreen.size() != (size_t)3 * xRes * yRes) InitScreens();
    int sc = 0;
    check(ptk_read_accumulators(ctx, preScreen.data(), denoising ? normal.data() : nullptr,
                                denoising ? albedo1.data() : nullptr,
                                denoising ? albedo2.data() : nullptr,
                                denoising ? directLight.data() : nullptr,
                                denoising ? worldPos.data() : nullptr,
                                denoising ? stdDev.data() : nullptr, &sc),
          "ptk_read_accumulators");
    if (ok()) sampleCount = sc;
}

void HostState::PostProcess(bool denoised)
{
    size_t P = (size_t)xRes * yRes;
    postScreen.resize(3 * P);
    check(ptk_post_process(ctx, denoised ? 1 : displayMetric, exposure, g, postScreen.data(),
                           nullptr),
          "ptk_post_process");
}

bool HostState::SaveImage(const std::string& path) const
{
    std::ofstream out(path);
    if (!out) return false;
    out << "P3\n" << xRes << ' ' << yRes << "\n255\n";
    for (int j = yRes - 1; j >= 0; --j)
        for (int i = 0; i < xRes; ++i) {
            size_t k = 3 * ((size_t)i + (size_t)xRes * j);
            out << static_cast<int>(255.999 * postScreen[k]) << ' '
                << static_cast<int>(255.999 * postScreen[k + 1]) << ' '
                << static_cast<int>(255.999 * postScreen[k + 2]) << '\n';
        }
    return (bool)out;
}

}  // namespace ptkhost

// ---------------------------------------------------------------- C helpers

extern "C" int ptkhost_scene_constants(const char* scene_path, int width, int height,
                                       int max_depth, float rand_samp, int denoising,
                                       void* constants_out, float* exposure_out, float* gamma_out)
{
    if (!scene_path || !constants_out) return -1;
    ptkhost::HostState st;
    st.xRes = width;
    st.yRes = height;
    st.maxDepth = max_depth;
    st.randSamp = rand_samp;
    st.denoising = denoising != 0;
    if (!st.scene.LoadScene(scene_path)) return -1;
    ptk_constants k;
    ptkhost::fill_constants(st, st.scene, k);
    std::memcpy(constants_out, &k, sizeof k);
    if (exposure_out) *exposure_out = st.exposure;
    if (gamma_out) *gamma_out = st.g;
    return 0;
}

// The tables the facade hands to ptk_set_scene for a scene file of any size.
// Arrays may be null to query the counts (returned in desc_out's num_* fields).
extern "C" int ptkhost_scene_tables(const char* scene_path, int width, int height, int max_depth,
                                    float rand_samp, int denoising, ptk_scene_desc* desc_out,
                                    int32_t* shape_type, int32_t* shape_mat, float* shape_attrs,
                                    float* mat_list, uint32_t* important, float* exposure_out,
                                    float* gamma_out)
{
    if (!scene_path || !desc_out) return -1;
    ptkhost::HostState st;
    st.xRes = width;
    st.yRes = height;
    st.maxDepth = max_depth;
    st.randSamp = rand_samp;
    st.denoising = denoising != 0;
    if (!st.scene.LoadScene(scene_path)) return -1;
    ptk_constants k;
    ptkhost::fill_constants(st, st.scene, k);
    ptkhost::SceneTables t;
    ptkhost::fill_tables(st, st.scene, k, t);
    *desc_out = t.desc;
    desc_out->shape_type = nullptr;
    desc_out->shape_mat = nullptr;
    desc_out->shape_attrs = nullptr;
    desc_out->mat_list = nullptr;
    desc_out->important_shapes = nullptr;
    if (shape_type) std::memcpy(shape_type, t.type.data(), t.type.size() * sizeof(int32_t));
    if (shape_mat) std::memcpy(shape_mat, t.mat.data(), t.mat.size() * sizeof(int32_t));
    if (shape_attrs) std::memcpy(shape_attrs, t.attrs.data(), t.attrs.size() * sizeof(float));
    if (mat_list) std::memcpy(mat_list, t.mats.data(), t.mats.size() * sizeof(float));
    if (important) std::memcpy(important, t.imp.data(), t.imp.size() * sizeof(uint32_t));
    if (exposure_out) *exposure_out = st.exposure;
    if (gamma_out) *gamma_out = st.g;
    return 0;
}

extern "C" int ptkhost_load_weights(const char* path, float* w12, float* w23, float* w34)
{
    if (!path || !w12 || !w23 || !w34) return -1;
    ptkhost::HostState st;
    if (!st.denoiserNN.LoadWeights(path)) return -1;
    std::memcpy(w12, st.denoiserNN.onetwo, 360 * sizeof(float));
    std::memcpy(w23, st.denoiserNN.twothree, 100 * sizeof(float));
    std::memcpy(w34, st.denoiserNN.threefour, 70 * sizeof(float));
    return 0;
}

extern "C" int ptkhost_resave_scene(const char* in_path, const char* out_path)
{
    if (!in_path || !out_path) return -1;
    ptkhost::HostState st;
    if (!st.scene.LoadScene(in_path)) return -1;
    return st.scene.SaveScene(out_path) ? 0 : -1;
}

extern "C" int ptkhost_resave_weights(const char* in_path, const char* out_path)
{
    if (!in_path || !out_path) return -1;
    ptkhost::HostState st;
    if (!st.denoiserNN.LoadWeights(in_path)) return -1;
    st.denoiserNN.OutputWeights(out_path);
    return 0;
}

// Writes a training log with the façade's own formatting (header, one line per epoch,
// total time) without touching a GPU: the on-disk format build/TrainingVisualizer.py reads.
extern "C" int ptkhost_write_training_log(const char* path, int width, int height, int samples,
                                          float learning_rate, const float* rel_mse, int epochs,
                                          double training_time_ms)
{
    if (!path || (epochs > 0 && !rel_mse)) return -1;
    ptkhost::HostState st;
    st.xRes = width;
    st.yRes = height;
    ptkhost::DenoiserNN& nn = st.denoiserNN;
    nn.errorLogPath = path;
    nn.trainedWeightsPath.clear();
    nn.samplesWhenTraining = samples;
    nn.learningRate = learning_rate;
    {   // InitTraining's header line
        std::ofstream log(nn.errorLogPath);
        if (!log.is_open()) return -1;
        log << "Res: (" << st.xRes << "x" << st.yRes << ") Samples: " << nn.samplesWhenTraining
            << " L Rate: " << nn.learningRate << "," << std::endl;
    }
    for (int e = 0; e < epochs; e++) {
        nn.relMSE = rel_mse[e];
        nn.AppendTrainingFile();
    }
    nn.trainingTime = training_time_ms;
    nn.EndTraining();
    return 0;
}

// ---- the denoiser's "median getter" network (csrc/mad_network.h), instantiated for the host
// so that it can be checked against the literal loop without a GPU: out[i] = the network's
// result for the nine values v[9 * i .. 9 * i + 8].
#define PTK_MAD_FN inline
#define PTK_MAD_UNROLL
#include "../csrc/mad_network.h"
namespace {
struct HostOps {
    using Mask = bool;
    static bool gt(float a, float b) { return a > b; }
    static bool land(bool a, bool b) { return a && b; }
    static bool lor(bool a, bool b) { return a || b; }
    static bool andn(bool a, bool b) { return a && !b; }
    static bool lnot(bool a) { return !a; }
    static float sel(bool c, float x, float y) { return c ? x : y; }
    static float fabs(float x) { return std::fabs(x); }
};
}  // namespace

extern "C" void ptkhost_mad_network(const float* v, size_t n, float* out)
{
    for (size_t i = 0; i < n; i++) {
        float w[9];
        for (int k = 0; k < 9; k++) w[k] = v[9 * i + k];
        out[i] = ptk_mad::Network<HostOps>::run(w);
    }
}
