// render_bvh_fast.hip -- PTK_MATH_FAST path-trace kernels for scenes set with
// ptk_set_scene: the same per-sample function as render_fast.hip, with the two
// scans over the shapes going through a bounding-volume hierarchy in HBM.
#include <hip/hip_runtime.h>

#include "ptk_dev.h"

#define PTK_FAST 1
#pragma clang fp contract(off)
#include "render_kernel.inc"

hipError_t ptk_launch_render_bvh_fast(const RenderArgs& a, int max_depth, bool features,
                                       hipStream_t stream)
{
    if (max_depth <= 4) return launch_depth<4, true>(a, features, stream);
    if (max_depth <= 8) return launch_depth<8, true>(a, features, stream);
    return launch_depth<12, true>(a, features, stream);
}
