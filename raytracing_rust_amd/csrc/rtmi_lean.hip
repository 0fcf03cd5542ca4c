// rtmi_lean.hip — second translation unit of librtmi.so: the LEAN instantiations of the cooperative render kernel
// (scenes without BVH items: cornell_box, cornell_smoke, ...), compiled with the backend's DEFAULT machine scheduler.
// The rest of the library (rtmi_device.hip) is compiled with -amdgpu-sched-strategy=iterative-maxocc, which is worth
// +2.4 % on final_scene and +2.2 % on random_spheres and costs these two instantiations 2 % (measured);
// the strategy is a per-translation-unit option, so they live here.  Device code only: the host stubs defined here are
// what rtmi_device.hip launches through its `extern template` declarations.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "rtmi.h"
#include "rtmi_math.h"

#define RTMI_LEAN_TU 1
#include "rtmi_kernels.hpp"

template __global__ void rtmi_render_coop<false, false, 4, false, 0>(DevScene, DevCamera, DevParams);
template __global__ void rtmi_render_coop<false, false, 4, false, 1>(DevScene, DevCamera, DevParams);
template __global__ void rtmi_render_coop<false, false, 4, false, 2>(DevScene, DevCamera, DevParams);
