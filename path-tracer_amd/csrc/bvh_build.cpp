// bvh_build.cpp -- binned surface-area-heuristic build, top down, with a switch
// to median splits once a branch is deep enough that an unbalanced tree could
// outgrow the kernels' traversal stack.
#include "bvh_build.h"

#include <algorithm>
#include <cmath>
#include <limits>

namespace {

constexpr int BINS = 16;
constexpr int SAH_DEPTH_LIMIT = 36;   // beyond this only median splits: depth <= 36 + log2(n)

struct Box {
    float lo[3], hi[3];
    void clear()
    {
        for (int k = 0; k < 3; k++) {
            lo[k] = std::numeric_limits<float>::infinity();
            hi[k] = -std::numeric_limits<float>::infinity();
        }
    }
    void grow(const float* l, const float* h)
    {
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(lo[k], l[k]);
            hi[k] = std::max(hi[k], h[k]);
        }
    }
    void grow(const Box& b) { grow(b.lo, b.hi); }
    double half_area() const
    {
        double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
        if (!(x >= 0) || !(y >= 0) || !(z >= 0)) return 0.0;
        return x * y + y * z + z * x;
    }
};

struct Task { int node, slot, begin, end, depth; };

// A node before its bounds are put on the grid.
struct WideNode {
    float lo[2][3], hi[2][3];
    int32_t child[2];
};

}  // namespace

bool ptk_build_bvh(int n, const float* lo, const float* hi, const uint8_t* kind, BvhBuild& out)
{
    if (n < 0 || n > (1 << 24)) return false;
    // One shape per leaf: a shape test costs two to four node visits, and a lane
    // testing a shape its neighbours do not idles them (grid of 97 k shapes: 384 /
    // 373 / 359 / 342 Msamples/s at 1 / 2 / 3 / 4 shapes per leaf).
    const int leaf_max = 1;
    out.nodes.clear();
    out.depth = 0;
    out.leaves = 0;
    std::vector<int32_t> order_v((size_t)n + 1);
    for (int i = 0; i < n; i++) order_v[(size_t)i] = i;
    std::vector<WideNode> wide;
    std::vector<float> centre((size_t)n * 3);
    for (size_t i = 0; i < (size_t)n * 3; i++) centre[i] = 0.5f * lo[i] + 0.5f * hi[i];

    WideNode blank;
    for (int k = 0; k < 3; k++) blank.lo[0][k] = blank.hi[0][k] = blank.lo[1][k] = blank.hi[1][k] = 0.0f;
    blank.child[0] = blank.child[1] = ~3;           // leaves of nothing
    wide.push_back(blank);

    int32_t* order = order_v.data();
    std::vector<Task> todo;
    if (n <= 1) {
        if (n > 0) todo.push_back({0, 0, 0, n, 1});
    } else {
        // the root's two children are made by splitting the whole range like any other
        todo.push_back({-1, 0, 0, n, 0});
    }

    while (!todo.empty()) {
        const Task t = todo.back();
        todo.pop_back();
        const int count = t.end - t.begin;
        Box bounds, cbounds;
        bounds.clear();
        cbounds.clear();
        for (int i = t.begin; i < t.end; i++) {
            const size_t s = (size_t)order[i];
            bounds.grow(lo + 3 * s, hi + 3 * s);
            cbounds.grow(&centre[3 * s], &centre[3 * s]);
        }

        int self;   // the node whose children this range becomes, when it is split
        if (t.node < 0) {
            self = 0;
        } else {
            WideNode& parent = wide[(size_t)t.node];
            for (int k = 0; k < 3; k++) {
                parent.lo[t.slot][k] = bounds.lo[k];
                parent.hi[t.slot][k] = bounds.hi[k];
            }
            if (count == 1) {
                const int32_t s = order[t.begin];
                parent.child[t.slot] = ~(s * 4 + (int32_t)kind[s]);
                out.leaves++;
                out.depth = std::max(out.depth, t.depth);
                continue;
            }
            self = (int)wide.size();
            wide[(size_t)t.node].child[t.slot] = self;
            wide.push_back(blank);
        }

        // ---- choose the split
        int mid = -1;
        if (t.depth < SAH_DEPTH_LIMIT) {
            double best_cost = std::numeric_limits<double>::infinity();
            int best_axis = -1, best_bin = -1;
            for (int axis = 0; axis < 3; axis++) {
                const float c0 = cbounds.lo[axis], c1 = cbounds.hi[axis];
                if (!(c1 > c0)) continue;
                const double scale = BINS / ((double)c1 - c0);
                Box bin_box[BINS];
                int bin_n[BINS];
                for (int b = 0; b < BINS; b++) { bin_box[b].clear(); bin_n[b] = 0; }
                for (int i = t.begin; i < t.end; i++) {
                    const size_t s = (size_t)order[i];
                    int b = (int)(((double)centre[3 * s + axis] - c0) * scale);
                    b = std::min(std::max(b, 0), BINS - 1);
                    bin_box[b].grow(lo + 3 * s, hi + 3 * s);
                    bin_n[b]++;
                }
                double right_area[BINS];
                Box acc;
                acc.clear();
                for (int b = BINS - 1; b > 0; b--) {
                    acc.grow(bin_box[b]);
                    right_area[b] = acc.half_area();
                }
                acc.clear();
                int left_n = 0;
                for (int b = 0; b < BINS - 1; b++) {
                    acc.grow(bin_box[b]);
                    left_n += bin_n[b];
                    if (left_n == 0 || left_n == count) continue;
                    const double cost = acc.half_area() * left_n + right_area[b + 1] * (count - left_n);
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                const float c0 = cbounds.lo[best_axis], c1 = cbounds.hi[best_axis];
                const double scale = BINS / ((double)c1 - c0);
                int32_t* m = std::partition(order + t.begin, order + t.end, [&](int32_t s) {
                    int b = (int)(((double)centre[3 * (size_t)s + best_axis] - c0) * scale);
                    b = std::min(std::max(b, 0), BINS - 1);
                    return b <= best_bin;
                });
                mid = (int)(m - order);
            }
        }
        if (mid <= t.begin || mid >= t.end) {
            // median along the widest axis of the centres (or of the index order
            // when all centres coincide)
            int axis = 0;
            float widest = -1.0f;
            for (int k = 0; k < 3; k++) {
                const float w = cbounds.hi[k] - cbounds.lo[k];
                if (w > widest) { widest = w; axis = k; }
            }
            mid = t.begin + count / 2;
            std::nth_element(order + t.begin, order + mid, order + t.end, [&](int32_t a, int32_t b) {
                const float ca = centre[3 * (size_t)a + axis], cb = centre[3 * (size_t)b + axis];
                return ca < cb || (ca == cb && a < b);
            });
        }
        todo.push_back({self, 0, t.begin, mid, t.depth + 1});
        todo.push_back({self, 1, mid, t.end, t.depth + 1});
    }
    if (out.depth > PTK_BVH_STACK - 2) return false;

    // ---- the grid: 16 bits per axis over the bounds of everything
    Box all;
    all.clear();
    for (int i = 0; i < n; i++) all.grow(lo + 3 * (size_t)i, hi + 3 * (size_t)i);
    double base[3], scale[3];
    for (int k = 0; k < 3; k++) {
        if (n == 0) { all.lo[k] = 0.0f; all.hi[k] = 1.0f; }
        base[k] = all.lo[k];
        scale[k] = std::max(((double)all.hi[k] - all.lo[k]) / 65535.0 * (1.0 + 1e-6), 1e-30);
        out.grid_base[k] = (float)base[k];
        if ((double)out.grid_base[k] > base[k]) out.grid_base[k] = std::nextafter(out.grid_base[k], -INFINITY);
        out.grid_scale[k] = (float)scale[k];
        if ((double)out.grid_scale[k] < scale[k]) out.grid_scale[k] = std::nextafter(out.grid_scale[k], INFINITY);
        base[k] = out.grid_base[k];
        scale[k] = out.grid_scale[k];
    }
    out.nodes.resize(wide.size());
    for (size_t i = 0; i < wide.size(); i++) {
        const WideNode& w = wide[i];
        BvhNode& o = out.nodes[i];
        for (int c = 0; c < 2; c++)
            for (int k = 0; k < 3; k++) {
                double ql = std::floor(((double)w.lo[c][k] - base[k]) / scale[k]);
                double qh = std::ceil(((double)w.hi[c][k] - base[k]) / scale[k]);
                ql = std::min(std::max(ql, 0.0), 65535.0);
                qh = std::min(std::max(qh, 0.0), 65535.0);
                o.q[c * 6 + k] = (uint16_t)ql;
                o.q[c * 6 + 3 + k] = (uint16_t)qh;
            }
        o.child[0] = w.child[0];
        o.child[1] = w.child[1];
    }
    return true;
}
