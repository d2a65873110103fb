// render_kind_fast.hip -- scenes in which no shape refracts (scans by kind of shape):
// PTK_MATH_FAST instantiation of the path-trace kernel:
// reciprocal-based slab tests and normalisation, hardware sin/cos, fp32 pi
// sites.  Same sampling decisions and estimator; values differ from EXACT mode
// at the 1e-7 relative level per operation.
#include <hip/hip_runtime.h>

#include "ptk_dev.h"

#define PTK_FAST 1
#define PTK_KIND_LOOPS 1
#ifndef PTK_F_CONTRACT
#define PTK_F_CONTRACT 0   // see render_kernel.inc
#endif
#if PTK_F_CONTRACT
#pragma clang fp contract(fast)
#else
#pragma clang fp contract(off)
#endif
#include "render_kernel.inc"

hipError_t ptk_launch_render_kind_fast(const RenderArgs& a, int max_depth, bool features,
                                  hipStream_t stream)
{
    if (max_depth <= 4) return launch_depth<4, false>(a, features, stream);
    if (max_depth <= 8) return launch_depth<8, false>(a, features, stream);
    return launch_depth<12, false>(a, features, stream);
}
