// render_kind_exact.hip -- scenes in which no shape refracts (scans by kind of shape):
// PTK_MATH_EXACT instantiation of the path-trace kernel:
// IEEE divide/sqrt, no FMA contraction, reference operation order.
#include <hip/hip_runtime.h>

#include "ptk_dev.h"

#define PTK_FAST 0
#define PTK_KIND_LOOPS 1
#pragma clang fp contract(off)
#include "render_kernel.inc"

hipError_t ptk_launch_render_kind_exact(const RenderArgs& a, int max_depth, bool features,
                                   hipStream_t stream)
{
    if (max_depth <= 4) return launch_depth<4, false>(a, features, stream);
    if (max_depth <= 8) return launch_depth<8, false>(a, features, stream);
    return launch_depth<12, false>(a, features, stream);
}
