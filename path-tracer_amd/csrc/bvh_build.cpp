// bvh_build.cpp -- binned surface-area-heuristic build, top down, with a switch
// to median splits once a branch is deep enough that an unbalanced tree could
// outgrow the kernels' traversal stack.
//
// One shape per leaf, so a range of m shapes becomes exactly m - 1 nodes: every
// range knows the block of node indices it will fill before it is split (the
// node itself, then the left subtree's block, then the right one's -- depth-first
// order, whatever the schedule).  Ranges are therefore independent tasks, and a
// few host threads share them for large scenes; the tree does not depend on how
// many threads built it.
#include "bvh_build.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <limits>
#include <mutex>
#include <thread>

namespace {

constexpr int BINS = 16;
constexpr int SAH_DEPTH_LIMIT = 36;   // beyond this only median splits: depth <= 36 + log2(n)
constexpr int SHARE_ABOVE = 4096;     // ranges larger than this may go to another thread
constexpr int THREADS_MAX = 16;

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

// The range [begin, end) of `order` becomes child `slot` of node `parent`
// (parent < 0: the range is the whole scene and becomes node 0 itself); if it is
// split, its own node is `node` and its subtree fills [node, node + count - 1).
struct Task { int parent, slot, begin, end, node, depth; };

// A node before its bounds are put on the grid.
struct WideNode {
    float lo[2][3], hi[2][3];
    int32_t child[2];
};

struct Builder {
    const float* lo;
    const float* hi;
    const uint8_t* kind;
    const float* centre;
    int32_t* order;
    WideNode* wide;
    std::atomic<int> depth{0}, leaves{0};

    // work shared between threads
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Task> shared;
    int busy = 0;          // threads holding work
    bool sharing = false;

    void note_depth(int d)
    {
        int cur = depth.load(std::memory_order_relaxed);
        while (d > cur && !depth.compare_exchange_weak(cur, d, std::memory_order_relaxed)) {}
    }

    // Bounds and leaf / node decision of one range; pushes the two halves on `local`.
    void process(const Task& t, std::vector<Task>& local)
    {
        const int count = t.end - t.begin;
        Box bounds, cbounds;
        bounds.clear();
        cbounds.clear();
        for (int i = t.begin; i < t.end; i++) {
            const size_t s = (size_t)order[i];
            bounds.grow(lo + 3 * s, hi + 3 * s);
            cbounds.grow(centre + 3 * s, centre + 3 * s);
        }
        if (t.parent >= 0) {
            WideNode& parent = wide[(size_t)t.parent];
            for (int k = 0; k < 3; k++) {
                parent.lo[t.slot][k] = bounds.lo[k];
                parent.hi[t.slot][k] = bounds.hi[k];
            }
            if (count == 1) {
                const int32_t s = order[t.begin];
                parent.child[t.slot] = ~(s * 4 + (int32_t)kind[s]);
                leaves.fetch_add(1, std::memory_order_relaxed);
                note_depth(t.depth);
                return;
            }
            parent.child[t.slot] = t.node;
        }

        // ---- choose the split
        int mid = -1;
        if (t.depth < SAH_DEPTH_LIMIT) {
            double best_cost = std::numeric_limits<double>::infinity();
            int best_axis = -1, best_bin = -1;
            // one pass over the range fills the bins of all three axes
            Box bin_box[3][BINS];
            int bin_n[3][BINS];
            double scale[3];
            bool usable[3];
            for (int axis = 0; axis < 3; axis++) {
                const float c0 = cbounds.lo[axis], c1 = cbounds.hi[axis];
                usable[axis] = c1 > c0;
                scale[axis] = usable[axis] ? BINS / ((double)c1 - c0) : 0.0;
                for (int b = 0; b < BINS; b++) { bin_box[axis][b].clear(); bin_n[axis][b] = 0; }
            }
            for (int i = t.begin; i < t.end; i++) {
                const size_t s = (size_t)order[i];
                for (int axis = 0; axis < 3; axis++) {
                    if (!usable[axis]) continue;
                    int b = (int)(((double)centre[3 * s + axis] - cbounds.lo[axis]) * scale[axis]);
                    b = std::min(std::max(b, 0), BINS - 1);
                    bin_box[axis][b].grow(lo + 3 * s, hi + 3 * s);
                    bin_n[axis][b]++;
                }
            }
            for (int axis = 0; axis < 3; axis++) {
                if (!usable[axis]) continue;
                double right_area[BINS];
                Box acc;
                acc.clear();
                for (int b = BINS - 1; b > 0; b--) {
                    acc.grow(bin_box[axis][b]);
                    right_area[b] = acc.half_area();
                }
                acc.clear();
                int left_n = 0;
                for (int b = 0; b < BINS - 1; b++) {
                    acc.grow(bin_box[axis][b]);
                    left_n += bin_n[axis][b];
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
        const int left_count = mid - t.begin;
        // the node, then the left subtree's left_count - 1 nodes, then the right subtree
        local.push_back({t.node, 1, mid, t.end, t.node + left_count, t.depth + 1});
        local.push_back({t.node, 0, t.begin, mid, t.node + 1, t.depth + 1});
    }

    void worker()
    {
        std::vector<Task> local;
        for (;;) {
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return !shared.empty() || busy == 0; });
                if (shared.empty()) return;       // nothing queued and nobody could queue more
                local.push_back(shared.back());
                shared.pop_back();
                busy++;
            }
            while (!local.empty()) {
                const Task t = local.back();
                local.pop_back();
                process(t, local);
                // hand the larger pending ranges to idle threads
                if (sharing && local.size() > 1 && local.front().end - local.front().begin > SHARE_ABOVE) {
                    std::lock_guard<std::mutex> lock(mu);
                    shared.push_back(local.front());
                    local.erase(local.begin());
                    cv.notify_one();
                }
            }
            {
                std::lock_guard<std::mutex> lock(mu);
                busy--;
                if (busy == 0) cv.notify_all();
            }
        }
    }
};

}  // namespace

bool ptk_build_bvh(int n, const float* lo, const float* hi, const uint8_t* kind, BvhBuild& out)
{
    if (n < 0 || n > (1 << 24)) return false;
    out.nodes.clear();
    out.depth = 0;
    out.leaves = 0;
    std::vector<int32_t> order((size_t)n + 1);
    for (int i = 0; i < n; i++) order[(size_t)i] = i;
    std::vector<float> centre((size_t)n * 3 + 3);
    for (size_t i = 0; i < (size_t)n * 3; i++) centre[i] = 0.5f * lo[i] + 0.5f * hi[i];

    WideNode blank;
    for (int k = 0; k < 3; k++) blank.lo[0][k] = blank.hi[0][k] = blank.lo[1][k] = blank.hi[1][k] = 0.0f;
    blank.child[0] = blank.child[1] = ~3;           // leaves of nothing
    std::vector<WideNode> wide((size_t)std::max(1, n - 1), blank);

    Builder B;
    B.lo = lo;
    B.hi = hi;
    B.kind = kind;
    B.centre = centre.data();
    B.order = order.data();
    B.wide = wide.data();
    if (n == 1) B.shared.push_back({0, 0, 0, 1, 0, 1});       // the root's only child is a leaf
    else if (n > 1) B.shared.push_back({-1, 0, 0, n, 0, 0});  // the whole scene becomes node 0

    int threads = 1;
    if (n >= 4 * SHARE_ABOVE) {
        threads = (int)std::thread::hardware_concurrency();
        if (const char* e = std::getenv("OMP_NUM_THREADS")) {
            const int v = std::atoi(e);
            if (v > 0) threads = std::min(threads, v);
        }
        threads = std::max(1, std::min(threads, THREADS_MAX));
    }
    B.sharing = threads > 1;
    if (threads == 1) {
        B.worker();
    } else {
        std::vector<std::thread> pool;
        for (int i = 0; i < threads; i++) pool.emplace_back([&B] { B.worker(); });
        for (std::thread& th : pool) th.join();
    }
    out.depth = B.depth.load();
    out.leaves = B.leaves.load();
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
    ptk_parallel_for((int)wide.size(), [&](int i0, int i1) {
    for (size_t i = (size_t)i0; i < (size_t)i1; i++) {
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
    });
    return true;
}

void ptk_collapse_bvh4(const std::vector<BvhNode>& nodes, std::vector<BvhNode4>& nodes4, int& depth4)
{
    nodes4.clear();
    depth4 = 0;
    if (nodes.empty()) return;
    constexpr int32_t EMPTY = ~3;
    auto pack = [](const uint16_t* q, uint32_t* box) {
        box[0] = (uint32_t)q[0] | ((uint32_t)q[1] << 16);
        box[1] = (uint32_t)q[2] | ((uint32_t)q[3] << 16);
        box[2] = (uint32_t)q[4] | ((uint32_t)q[5] << 16);
    };
    nodes4.reserve(nodes.size() / 2 + 2);
    // depth-first, a node's index fixed when it is first met; (binary node, wide index, depth)
    struct Item { int32_t bin, wide, depth; };
    std::vector<Item> todo;
    nodes4.emplace_back();
    todo.push_back({0, 0, 1});
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        depth4 = std::max(depth4, it.depth);
        const BvhNode& b = nodes[(size_t)it.bin];
        BvhNode4 w;
        for (int s = 0; s < 4; s++) {
            w.box[s][0] = w.box[s][1] = w.box[s][2] = 0;
            w.child[s] = EMPTY;
        }
        int32_t below[4] = {-1, -1, -1, -1};   // binary node each slot continues with, if any
        for (int c = 0; c < 2; c++) {
            const int32_t ch = b.child[c];
            if (ch < 0) {                       // a leaf (or nothing) keeps its own box
                pack(b.q + 6 * c, w.box[2 * c]);
                w.child[2 * c] = ch;
            } else {                            // an inner node is replaced by its two children
                const BvhNode& g = nodes[(size_t)ch];
                for (int k = 0; k < 2; k++) {
                    pack(g.q + 6 * k, w.box[2 * c + k]);
                    w.child[2 * c + k] = g.child[k];
                    if (g.child[k] >= 0) below[2 * c + k] = g.child[k];
                }
            }
        }
        for (int s = 0; s < 4; s++)
            if (below[s] >= 0) {
                w.child[s] = (int32_t)nodes4.size();
                nodes4.emplace_back();
                todo.push_back({below[s], w.child[s], it.depth + 1});
            }
        nodes4[(size_t)it.wide] = w;
    }
}
