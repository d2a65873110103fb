// ptk_headless.cpp -- headless driver: the reference's frame loop
// (PT.cpp:105-131) without the window.  Loads a scene, renders, optionally
// auto-exposes and denoises, and writes a PPM and/or raw float dumps.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

#include "ptk_host.h"

using namespace ptkhost;

static void usage()
{
    std::fprintf(stderr,
                 "usage: ptk_headless --scene FILE [--res WxH] [--spp N] [--depth D]\n"
                 "       [--rand-samp F] [--seed HEX0,HEX1 | --counter-seeds] [--fast]\n"
                 "       [--auto-exposure] [--denoise N --weights FILE [--denoise-variant 0|2]] [--device K]\n"
                 "       [--ppm OUT.ppm] [--dump PREFIX]\n"
                 "       [--train EPOCHS --train-spp N --lr-exp X --save-weights FILE --error-log FILE]\n"
                 "         (the rendered frame becomes the target; needs --denoise N)\n");
}

static bool dump(const std::string& path, const std::vector<float>& v)
{
    std::ofstream f(path, std::ios::binary);
    f.write((const char*)v.data(), (std::streamsize)(v.size() * sizeof(float)));
    return (bool)f;
}

int main(int argc, char** argv)
{
    std::string scene, weights, ppm, prefix, save_weights;
    int epochs = 0, train_spp = 4;
    std::string error_log;
    float lr_exp = 5.0f;
    int W = 1536, H = 864, spp = 16, depth = 4, nwin = -1, device = 0, variant = 0;
    float rand_samp = 0.005f;
    bool counter = false, fast = false, auto_exp = false, have_seed = false;
    uint64_t seed[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--scene") scene = next();
        else if (a == "--res") { if (std::sscanf(next(), "%dx%d", &W, &H) != 2) { usage(); return 2; } }
        else if (a == "--spp") spp = std::atoi(next());
        else if (a == "--depth") depth = std::atoi(next());
        else if (a == "--rand-samp") rand_samp = (float)std::atof(next());
        else if (a == "--seed") {
            unsigned long long s0, s1;
            if (std::sscanf(next(), "%llx,%llx", &s0, &s1) != 2) { usage(); return 2; }
            seed[0] = s0; seed[1] = s1; have_seed = true;
        }
        else if (a == "--counter-seeds") counter = true;
        else if (a == "--fast") fast = true;
        else if (a == "--auto-exposure") auto_exp = true;
        else if (a == "--denoise") nwin = std::atoi(next());
        else if (a == "--weights") weights = next();
        else if (a == "--denoise-variant") variant = std::atoi(next());
        else if (a == "--device") device = std::atoi(next());
        else if (a == "--ppm") ppm = next();
        else if (a == "--dump") prefix = next();
        else if (a == "--train") epochs = std::atoi(next());
        else if (a == "--train-spp") train_spp = std::atoi(next());
        else if (a == "--error-log") error_log = next();
        else if (a == "--lr-exp") lr_exp = (float)std::atof(next());
        else if (a == "--save-weights") save_weights = next();
        else { usage(); return 2; }
    }
    if (scene.empty()) { usage(); return 2; }

    HostState st;
    st.device = device;
    st.xRes = W; st.yRes = H; st.maxDepth = depth; st.randSamp = rand_samp;
    st.denoising = nwin >= 0;
    st.denoisingN = nwin >= 0 ? nwin : 1;
    st.denoiseVariant = variant;
    st.seedMode = counter ? PTK_SEED_COUNTER : PTK_SEED_REFERENCE;
    st.mathMode = fast ? PTK_MATH_FAST : PTK_MATH_EXACT;
    if (!st.scene.LoadScene(scene)) return 1;
    st.renderer.UpdateConstants();
    if (!st.ok()) { std::fprintf(stderr, "%s\n", st.lastError.c_str()); return 1; }
    if (have_seed || !counter) { st.renderer.GloRandS[0] = seed[0]; st.renderer.GloRandS[1] = seed[1]; }

    st.renderer.Render(spp);
    if (!st.ok()) { std::fprintf(stderr, "%s\n", st.lastError.c_str()); return 1; }
    std::printf("render: %dx%d %d spp  %.3f ms  %.1f Msamples/s  %llu rays\n", W, H, spp,
                st.renderTime, st.renderTime > 0 ? (double)W * H * spp / st.renderTime / 1e3 : 0.0,
                (unsigned long long)st.rayCount);
    if (auto_exp) {
        st.renderer.AutoExposure();
        if (!st.ok()) { std::fprintf(stderr, "%s\n", st.lastError.c_str()); return 1; }
        std::printf("auto exposure: %.9g (%.4f ms)\n", st.exposure, st.exposureTime);
    }
    bool denoised = false;
    if (epochs > 0) {
        // the frame just rendered is the target; each epoch renders a fresh noisy frame
        if (nwin < 0) { std::fprintf(stderr, "--train needs --denoise N\n"); return 1; }
        if (!weights.empty() && !st.denoiserNN.LoadWeights(weights)) {
            std::fprintf(stderr, "cannot read %s\n", weights.c_str());
            return 1;
        }
        st.denoiserNN.samplesWhenTraining = train_spp;
        st.denoiserNN.lRateInt = st.denoiserNN.lRateIntMax = lr_exp;
        // continue from --weights (random weights without it); the log goes next to the other outputs
        st.denoiserNN.trainedWeightsPath = weights;
        st.denoiserNN.errorLogPath = error_log;
        st.denoiserNN.InitTraining();
        for (int e = 0; e < epochs && st.ok(); e++) {
            st.denoiserNN.TrainNN();
            std::printf("epoch %d: relMSE %.6g  lr %.3g  %.3f ms\n", e, st.denoiserNN.relMSE,
                        st.denoiserNN.learningRate, st.denoiserNN.epochTime);
        }
        if (!st.ok()) { std::fprintf(stderr, "%s\n", st.lastError.c_str()); return 1; }
        st.denoiserNN.trainedWeightsPath = save_weights;   // empty: keep them in memory only
        st.denoiserNN.EndTraining();
        std::printf("training: %d epochs, %.3f ms\n", epochs, st.denoiserNN.trainingTime);
        weights.clear();   // the trained weights are in place for the denoise below
    }
    if (nwin >= 0) {
        if (epochs == 0 && (weights.empty() || !st.denoiserNN.LoadWeights(weights))) {
            std::fprintf(stderr, "--denoise needs --weights FILE\n");
            return 1;
        }
        st.denoiser.denoise();
        if (!st.ok()) { std::fprintf(stderr, "%s\n", st.lastError.c_str()); return 1; }
        std::printf("denoise: N=%d variant %d  %.4f ms\n", st.denoisingN, st.denoiseVariant,
                    st.denoiseTime);
        denoised = true;
    }
    st.SyncScreens();
    if (!st.ok()) { std::fprintf(stderr, "%s\n", st.lastError.c_str()); return 1; }
    if (!prefix.empty()) {
        dump(prefix + ".preScreen.f32", st.preScreen);
        if (st.denoising) {
            dump(prefix + ".normal.f32", st.normal);
            dump(prefix + ".albedo1.f32", st.albedo1);
            dump(prefix + ".albedo2.f32", st.albedo2);
            dump(prefix + ".worldPos.f32", st.worldPos);
            dump(prefix + ".directLight.f32", st.directLight);
            dump(prefix + ".stdDev.f32", st.stdDev);
        }
        if (denoised) {
            dump(prefix + ".denoisedCol.f32", st.denoisedCol);
            dump(prefix + ".variances.f32", st.variances);
            dump(prefix + ".wcSum.f32", st.wcSum);
        }
    }
    if (!ppm.empty()) {
        st.PostProcess(denoised);
        if (!st.SaveImage(ppm)) { std::fprintf(stderr, "cannot write %s\n", ppm.c_str()); return 1; }
    }
    return 0;
}
