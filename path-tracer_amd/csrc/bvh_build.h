// bvh_build.h -- host-side construction of the bounding-volume hierarchy that
// the ptk_set_scene kernels walk (layout: BvhNode in ptk_dev.h).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

#include "ptk_dev.h"

// Runs fn(begin, end) over [0, n) in contiguous chunks on a few host threads
// (one when n is small); fn must only write what belongs to its own indices.
template <class Fn>
void ptk_parallel_for(int n, Fn fn)
{
    int threads = 1;
    if (n >= 32768) {
        threads = (int)std::thread::hardware_concurrency();
        if (const char* e = std::getenv("OMP_NUM_THREADS")) {
            const int v = std::atoi(e);
            if (v > 0) threads = std::min(threads, v);
        }
        threads = std::max(1, std::min(threads, 16));
    }
    if (threads == 1) {
        fn(0, n);
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) {
        const int b = (int)((int64_t)n * t / threads), e = (int)((int64_t)n * (t + 1) / threads);
        pool.emplace_back([=] { fn(b, e); });
    }
    for (std::thread& th : pool) th.join();
}

struct BvhBuild {
    std::vector<BvhNode> nodes;   // nodes[0] is the root
    float grid_base[3], grid_scale[3];
    int depth = 0;                // internal nodes on the longest root-to-leaf path
    int leaves = 0;
};

// lo / hi: per-shape bounds, 3 floats each, already padded so that no ray a shape
// test accepts can miss them.  kind: per shape, the two low bits of its leaf code
// (0 sphere, 1 box, 2 rotated box).  Returns false when n is out of range or a
// tree within the traversal stack's depth cannot be built (never for n <= 2^24).
bool ptk_build_bvh(int n, const float* lo, const float* hi, const uint8_t* kind, BvhBuild& out);

// Folds every other level of the binary tree away (layout: BvhNode4 in ptk_dev.h).  nodes4[0]
// is the root; depth4 = nodes on the longest root-to-leaf path.  The boxes are the binary
// nodes' own grid boxes, copied, so whatever held for them holds here.
void ptk_collapse_bvh4(const std::vector<BvhNode>& nodes, std::vector<BvhNode4>& nodes4, int& depth4);
