// ptk_host.h -- C++ host side above the C ABI: the reference's entry points
// (Renderers / Denoiser / DenoiserNN / Scene, same names and call order) with
// backend slot 2 served by libptk.  State that the reference keeps in globals
// (headers/GLOBALS.h:15-76) lives in one HostState object here.
//
// Only slot 2 (GPU) exists: selecting any other backend is an error, there is
// no CPU path in the product.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ptk.h"

namespace ptkhost {

struct Vec3 {
    float x = 0, y = 0, z = 0;
};

// headers/Mat.h:8-48.  Types: 0 lambert, 1 light, 2 metal, 3 dielectric.
struct Mat {
    int matType = 0;
    Vec3 alb{1, 1, 1};
    float blur = 0, RI = 1;
};

// headers/Obj.h:60-164.  type 0 sphere, 1 oriented box.
struct Obj {
    int type = 0, mat_ind = 0;
    bool inImportantList = false;
    Vec3 pos;
    float r = 1;                 // sphere
    Vec3 min, max, rot;          // box: relative bounds, Euler radians
    float q[4] = {1, 0, 0, 0};   // box: quaternion w,x,y,z = (qx*qy)*qz
    void UpdateRot();            // headers/Obj.h:122-128
};

// headers/Camera.h:56-58 with the defaults of PT.cpp:62-69.
struct Camera {
    Vec3 pos{0, 0, -9}, forward{0, 0, 1}, up{0, 1, 0}, right{1, 0, 0};
    float focalLen = 1.0f, vfov = 90, hfov = 120;
};

struct HostState;

// headers/Scene.h.  File grammar: `m r/g/b blur RI type`, `o type`,
// `d ...` (sphere: pos r mat; box: pos min max rot mat), `i index`,
// `c pos fwd up right vfov hfov`, `pp exposure gamma`.
class Scene {
public:
    explicit Scene(HostState* st) : st_(st) {}
    void ResetScene();
    void InitScene();                        // headers/Scene.h:10-36 (default Cornell box)
    bool LoadScene(const std::string& path); // headers/Scene.h:127-224 (takes a path, not a name)
    bool SaveScene(const std::string& path); // headers/Scene.h:226-274
    void AddToImpList(int index);

    std::vector<Obj> objList;
    std::vector<Mat> matList;
    std::vector<int> importantList;

private:
    HostState* st_;
};

// headers/Renderers.h:26-110.
class Renderers {
public:
    explicit Renderers(HostState* st) : st_(st) {}
    void UpdateConstants();          // Renderers.cpp:1051-1131 (+ CUDARender::UpdateConstants)
    void UpdateCam(bool push = true);// Renderers.cpp:1132-1149
    void Render();                   // one sample per pixel (headers/Renderers.h:31-60)
    void Render(int spp);            // spp passes in one device launch
    void AutoExposure();             // headers/Renderers.h:72-101
    uint64_t GloRandS[2] = {0, 0};
    ptk_constants constants{};

private:
    HostState* st_;
    bool scenePushed_ = false;       // the device holds this scene at pushedW_ x pushedH_
    int pushedW_ = 0, pushedH_ = 0;
};

// headers/DenoiserNN.h:95-219 (inference part).
class DenoiserNN {
public:
    explicit DenoiserNN(HostState* st) : st_(st) {}
    bool LoadWeights(const std::string& path);    // DenoiserNN.cpp:1625-1663
    void OutputWeights(const std::string& path);  // DenoiserNN.cpp:1603-1624
    void ForwardProp();  // headers/DenoiserNN.h:103-131; fills variances (and sFeatures)
    float onetwo[360] = {0}, twothree[100] = {0}, threefour[80] = {0};

    // Training (DenoiserNN.cpp:1476-1563).  InitTraining takes the current frame
    // as the target (Denoiser::saveTargetCol) and loads the weights to continue
    // from, or randomises them when that file cannot be read; TrainNN is one epoch: a fresh frame
    // of samplesWhenTraining samples, denoise + relMSE + back-prop in one
    // device call (ptk_train_step), epoch counter and learning-rate schedule.
    void InitTraining();                          // DenoiserNN.cpp:1476-1499
    void TrainNN();                               // DenoiserNN.cpp:1520-1563
    void AppendTrainingFile();                    // DenoiserNN.cpp:1500-1506
    void EndTraining();                           // DenoiserNN.cpp:1507-1518
    void RandomizeWeights();                      // DenoiserNN.cpp:1591-1601
    // The reference reads and writes "../Weights/NNWeights.txt" and "ErrorLog.txt"
    // relative to its working directory; the same names are the defaults here.
    // InitTraining loads trainedWeightsPath (randomising on failure) and writes the
    // log's header line; every epoch appends its relMSE; EndTraining appends the
    // total time in ms and saves the weights.  An empty path skips that file.
    std::string trainedWeightsPath = "../Weights/NNWeights.txt";
    std::string errorLogPath = "ErrorLog.txt";
    double trainingTime = 0.0;                    // ms, sum of epochTime
    bool training = false, trainingLimitBool = false;
    int trainingEpoch = 0, trainingCount = 0, samplesWhenTraining = 4;
    float lRateInt = 5.0f, lRateIntMax = 5.0f;    // learningRate = 10^-lRateInt
    float learningRate = 0.0001f, relMSE = 0.0f;
    double epochTime = 0.0;                       // ms, wall clock of the epoch (render included)

private:
    HostState* st_;
};

// headers/Denoiser.h:16-31.
class Denoiser {
public:
    explicit Denoiser(HostState* st) : st_(st) {}
    void denoise();  // Denoiser.cpp:3-38: ForwardProp then the filter
private:
    HostState* st_;
};

// What headers/GLOBALS.h:15-76 declares as globals.
struct HostState {
    HostState();
    ~HostState();
    HostState(const HostState&) = delete;
    HostState& operator=(const HostState&) = delete;

    // settings (PT.cpp:50-82 defaults)
    int xRes = 1536, yRes = 864, maxDepth = 4;
    int currentRenderer = 2, denoisingBackend = 2;
    bool denoising = false;
    int denoisingN = 1;
    // feature pass of Denoiser::denoise(): 0 = the reference's arithmetic (bit-exact against the
    // CPU restatement), 2 = the tolerance-held pass (7x7 statistics from shared column sums)
    int denoiseVariant = 0;
    float randSamp = 0.005f, exposure = 2.0f, g = 2.1f;
    int sampleCount = 0;
    uint64_t rayCount = 0;
    bool weightsLoaded = false;
    bool wantFeaturesOut = false;   // also read back the 36 features in ForwardProp
    int seedMode = PTK_SEED_REFERENCE;
    int mathMode = PTK_MATH_EXACT;
    int device = 0;
    double renderTime = 0, exposureTime = 0, denoiseTime = 0;  // ms, device time

    // screens (interleaved xyz per pixel, as the reference's vec3 arrays)
    std::vector<float> preScreen, normal, albedo1, albedo2, worldPos, denoisedCol;
    std::vector<float> directLight;  // x component only
    std::vector<float> stdDev;       // 6 per pixel
    std::vector<float> variances;    // 7 per pixel
    std::vector<float> wcSum;        // 1 per pixel
    std::vector<float> sFeatures;    // 36 per pixel (when wantFeaturesOut)
    std::vector<float> postScreen;   // tone-mapped rgb

    Camera cam;
    Scene scene;
    Renderers renderer;
    DenoiserNN denoiserNN;
    Denoiser denoiser;

    // GLOBALS::InitScreens (headers/GLOBALS.h:243-259): sizes the host screens.
    void InitScreens();
    // Copies the device accumulators into the host screens (what the reference's
    // CUDA backend does in its host loop, CUDARender.cu:886-929).
    void SyncScreens();
    // CUDARender::PostProcess (CUDARender.cu:959-1066) on the device: the source
    // named by displayMetric (or the denoised colour), / exposure, pow(1/g) ->
    // postScreen.
    int displayMetric = 0;
    void PostProcess(bool denoised);
    // PT::SaveImage (PT.cpp:484-501): ASCII P3, rows flipped, int(255.999*c).
    bool SaveImage(const std::string& path) const;

    ptk_ctx* ctx = nullptr;
    std::string lastError;
    bool ok() const { return lastError.empty(); }
    void check(int rc, const char* what);
};

}  // namespace ptkhost

// C helpers over the façade, for bindings that cannot consume C++ classes.
extern "C" {
// Parses a .scene file and fills the 3732-byte POD exactly as
// Scene::LoadScene + Renderers::UpdateConstants would.  Returns 0 or -1.
int ptkhost_scene_constants(const char* scene_path, int width, int height, int max_depth,
                            float rand_samp, int denoising, void* constants_out,
                            float* exposure_out, float* gamma_out);
// Reads a three-line weights file.  Returns 0 or -1.
int ptkhost_load_weights(const char* path, float* w12_360, float* w23_100, float* w34_70);
// Scene::LoadScene then Scene::SaveScene / LoadWeights then OutputWeights: the
// on-disk formats either side of the hot path, for round-trip checks.
int ptkhost_resave_scene(const char* in_path, const char* out_path);
int ptkhost_resave_weights(const char* in_path, const char* out_path);
}
