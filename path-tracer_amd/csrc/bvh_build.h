// bvh_build.h -- host-side construction of the bounding-volume hierarchy that
// the ptk_set_scene kernels walk (layout: BvhNode in ptk_dev.h).
#pragma once

#include <cstdint>
#include <vector>

#include "ptk_dev.h"

struct BvhBuild {
    std::vector<BvhNode> nodes;   // nodes[0] is the root
    std::vector<int32_t> order;   // shape index of each leaf record, leaf by leaf
    int depth = 0;                // internal nodes on the longest root-to-leaf path
    int leaves = 0;
};

// lo / hi: per-shape bounds, 3 floats each, already padded so that no ray a shape
// test accepts can miss them.  visit_all gives every node infinite bounds, which
// turns each walk into a test of every shape (a checking aid: same kernels, no culling).
// Returns false when n is out of range or a tree within the traversal stack's
// depth cannot be built (never for n <= 2^24).
bool ptk_build_bvh(int n, const float* lo, const float* hi, bool visit_all, BvhBuild& out);
