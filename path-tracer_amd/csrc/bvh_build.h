// bvh_build.h -- host-side construction of the bounding-volume hierarchy that
// the ptk_set_scene kernels walk (layout: BvhNode in ptk_dev.h).
#pragma once

#include <cstdint>
#include <vector>

#include "ptk_dev.h"

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
