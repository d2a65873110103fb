// ptk_multigpu.cpp -- one frame on several GPUs, driven from C++ through the C ABI:
// one process per GPU, the frame's samples split across them, one accumulation
// exchange over RCCL, and (with --denoise) the row-band denoiser.
//
// The reference renders and denoises on one GPU (PT.cpp:105-131); this is the
// same frame loop with the samples of every pixel shared out:
//   render only   each rank renders its share on its own seed stream; all-reduce
//                 of the three colour planes in place in HBM; rank 0 writes the image.
//   with denoise  each rank needs the merged accumulators of its own row band only:
//                 ptk_pack_bands -> one all-to-all (ncclSend / ncclRecv pairs, every
//                 GPU to every other at once) -> ptk_merge_received (one kernel) ->
//                 ptk_denoise_rows on the band -> the filtered bands gathered in place.
//
// Start one process per GPU, e.g. for two:
//   ptk_multigpu --ranks 2 --rank 0 --rendezvous /tmp/ptk.id --scene S --spp 2000 --ppm out.ppm &
//   ptk_multigpu --ranks 2 --rank 1 --rendezvous /tmp/ptk.id --scene S --spp 2000
// Rank 0 creates the communicator's id and leaves it in the rendezvous file (written
// under a temporary name and renamed, so a reader never sees half of it); the others
// wait for the file.  --device defaults to the rank.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "ptk_host.h"

using namespace ptkhost;

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define NCCL_OK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { std::fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 1; } } while (0)
#define PTK_OK_(x) do { int rc_ = (x); if (rc_ != PTK_OK) { std::fprintf(stderr, "%s: %s\n", #x, ptk_last_error()); return 1; } } while (0)

static void usage()
{
    std::fprintf(stderr,
                 "usage: ptk_multigpu --ranks N --rank R --rendezvous FILE --scene FILE [--res WxH]\n"
                 "       [--spp TOTAL] [--depth D] [--device K] [--denoise N --weights FILE] [--ppm OUT.ppm]\n");
}

// Samples rendered by `rank`: the first (total % world) ranks take one extra.
static int shard_spp(int total, int world, int rank)
{
    return total / world + (rank < total % world ? 1 : 0);
}

static bool exchange_id(const std::string& path, int rank, ncclUniqueId& id)
{
    if (rank == 0) {
        if (ncclGetUniqueId(&id) != ncclSuccess) return false;
        const std::string tmp = path + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f) return false;
        const bool ok = std::fwrite(&id, sizeof id, 1, f) == 1;
        std::fclose(f);
        return ok && std::rename(tmp.c_str(), path.c_str()) == 0;
    }
    for (int tries = 0; tries < 600; tries++) {       // up to a minute
        if (FILE* f = std::fopen(path.c_str(), "rb")) {
            const bool ok = std::fread(&id, sizeof id, 1, f) == 1;
            std::fclose(f);
            if (ok) return true;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    return false;
}

int main(int argc, char** argv)
{
    std::string scene, weights, ppm, rendezvous;
    int W = 1920, H = 1080, total_spp = 64, depth = 4, nwin = -1, device = -1, world = 1, rank = 0;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--scene") scene = next();
        else if (a == "--res") { if (std::sscanf(next(), "%dx%d", &W, &H) != 2) { usage(); return 2; } }
        else if (a == "--spp") total_spp = std::atoi(next());
        else if (a == "--depth") depth = std::atoi(next());
        else if (a == "--denoise") nwin = std::atoi(next());
        else if (a == "--weights") weights = next();
        else if (a == "--device") device = std::atoi(next());
        else if (a == "--ppm") ppm = next();
        else if (a == "--ranks") world = std::atoi(next());
        else if (a == "--rank") rank = std::atoi(next());
        else if (a == "--rendezvous") rendezvous = next();
        else { usage(); return 2; }
    }
    if (scene.empty() || world < 1 || rank < 0 || rank >= world || (world > 1 && rendezvous.empty()) ||
        total_spp < world) {
        usage();
        return 2;
    }
    if (device < 0) device = rank;
    HIP_OK(hipSetDevice(device));

    ncclComm_t comm = nullptr;
    ncclUniqueId id;
    if (world > 1) {
        if (!exchange_id(rendezvous, rank, id)) { std::fprintf(stderr, "rank %d: no communicator id through %s\n", rank, rendezvous.c_str()); return 1; }
    } else {
        NCCL_OK(ncclGetUniqueId(&id));
    }
    NCCL_OK(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    HostState st;
    st.device = device;
    st.xRes = W; st.yRes = H; st.maxDepth = depth;
    st.denoising = nwin >= 0;
    st.denoisingN = nwin >= 0 ? nwin : 1;
    st.seedMode = PTK_SEED_COUNTER;
    if (!st.scene.LoadScene(scene)) return 1;
    if (nwin >= 0 && (weights.empty() || !st.denoiserNN.LoadWeights(weights))) {
        std::fprintf(stderr, "--denoise needs --weights FILE\n");
        return 1;
    }
    st.renderer.UpdateConstants();
    if (!st.ok()) { std::fprintf(stderr, "%s\n", st.lastError.c_str()); return 1; }
    PTK_OK_(ptk_set_stream_id(st.ctx, (uint64_t)rank));

    const int n_local = shard_spp(total_spp, world, rank);
    const auto t0 = std::chrono::steady_clock::now();
    st.renderer.Render(n_local);
    if (!st.ok()) { std::fprintf(stderr, "%s\n", st.lastError.c_str()); return 1; }
    const auto t1 = std::chrono::steady_clock::now();

    float* planes = nullptr;
    size_t P = 0;
    PTK_OK_(ptk_planes_device(st.ctx, &planes, &P));
    std::vector<int> per_rank((size_t)world);
    for (int r = 0; r < world; r++) per_rank[(size_t)r] = shard_spp(total_spp, world, r);

    if (nwin < 0) {
        // ---- render only: the colour sums of all ranks, in place
        NCCL_OK(ncclAllReduce(planes, planes, 3 * P, ncclFloat, ncclSum, comm, stream));
        HIP_OK(hipStreamSynchronize(stream));
        PTK_OK_(ptk_set_sample_count(st.ctx, total_spp));
        st.sampleCount = total_spp;
    } else {
        // ---- band exchange, merge, row-band denoise, gather
        const int halo = std::max(3, nwin);
        int lo = 0, hi = 0, rows_max = 0;
        PTK_OK_(ptk_band_layout(H, world, rank, halo, &lo, &hi, &rows_max));
        const size_t block = 32 * (size_t)rows_max * (size_t)W, n = (size_t)world * block;
        float *send = nullptr, *recv = nullptr;
        HIP_OK(hipMalloc((void**)&send, n * sizeof(float)));
        HIP_OK(hipMalloc((void**)&recv, n * sizeof(float)));
        PTK_OK_(ptk_pack_bands(st.ctx, world, halo, send, n));
        NCCL_OK(ncclGroupStart());
        for (int r = 0; r < world; r++) {
            NCCL_OK(ncclSend(send + (size_t)r * block, block, ncclFloat, r, comm, stream));
            NCCL_OK(ncclRecv(recv + (size_t)r * block, block, ncclFloat, r, comm, stream));
        }
        NCCL_OK(ncclGroupEnd());
        HIP_OK(hipStreamSynchronize(stream));
        PTK_OK_(ptk_merge_received(st.ctx, world, rank, halo, recv, n, per_rank.data()));
        HIP_OK(hipFree(send));
        HIP_OK(hipFree(recv));
        st.sampleCount = total_spp;

        const int base = H / world, extra = H % world;
        const int y0 = rank * base + std::min(rank, extra), y1 = y0 + base + (rank < extra ? 1 : 0);
        PTK_OK_(ptk_denoise_rows(st.ctx, st.denoiserNN.onetwo, st.denoiserNN.twothree,
                                 st.denoiserNN.threefour, nwin, 0, y0, y1));
        float *den = nullptr, *wc = nullptr;
        PTK_OK_(ptk_denoise_outputs_device(st.ctx, &den, &wc, nullptr));
        // A band's rows of one plane are contiguous and the bands tile the plane in rank order:
        // each rank broadcasts its rows of each plane, in place (any band sizes).
        NCCL_OK(ncclGroupStart());
        for (int r = 0; r < world; r++) {
            const int a = r * base + std::min(r, extra), b = a + base + (r < extra ? 1 : 0);
            const size_t off = (size_t)a * W, cnt = (size_t)(b - a) * W;
            for (int k = 0; k < 3; k++)
                NCCL_OK(ncclBroadcast(den + k * P + off, den + k * P + off, cnt, ncclFloat, r, comm, stream));
            NCCL_OK(ncclBroadcast(wc + off, wc + off, cnt, ncclFloat, r, comm, stream));
        }
        NCCL_OK(ncclGroupEnd());
        HIP_OK(hipStreamSynchronize(stream));
    }
    const auto t2 = std::chrono::steady_clock::now();
    const double render_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    const double rest_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
    std::printf("rank %d of %d (device %d): %d of %d spp in %.3f ms (%.1f Msamples/s); %s %.3f ms\n", rank,
                world, device, n_local, total_spp, render_ms, (double)W * H * n_local / render_ms / 1e3,
                nwin < 0 ? "all-reduce" : "band exchange + merge + band denoise + gather", rest_ms);

    if (rank == 0 && !ppm.empty()) {
        st.renderer.AutoExposure();
        st.PostProcess(nwin >= 0);
        if (!st.ok() || !st.SaveImage(ppm)) { std::fprintf(stderr, "cannot write %s: %s\n", ppm.c_str(), st.lastError.c_str()); return 1; }
        std::printf("rank 0: exposure %.6g, wrote %s\n", st.exposure, ppm.c_str());
    }
    NCCL_OK(ncclCommDestroy(comm));
    HIP_OK(hipStreamDestroy(stream));
    if (rank == 0 && world > 1) std::remove(rendezvous.c_str());
    return 0;
}
