// rtmi_alt.hip — third translation unit of librtmi.so: the two render kernels that are not the default path
// (csrc/rtmi_kernels_alt.hpp: workgroup-cooperative traversal, per-lane state machine), both measured slower than
// rtmi_render_coop and kept as independent implementations of the same per-lane program for the parity tests.  Device
// code only: the host stubs defined here are what rtmi_device.hip launches through its `extern template` declarations.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "rtmi.h"
#include "rtmi_math.h"

#define RTMI_LEAN_TU 1 /* the plain kernels and the lean instantiations are defined elsewhere */
#define RTMI_ALT_TU 1
#include "rtmi_kernels_alt.hpp"

template __global__ void rtmi_render_bcoop<true, false>(DevScene, DevCamera, DevParams);
template __global__ void rtmi_render_bcoop<false, false>(DevScene, DevCamera, DevParams);
template __global__ void rtmi_render_async<true, false, true>(DevScene, DevCamera, DevParams);
template __global__ void rtmi_render_async<false, false, true>(DevScene, DevCamera, DevParams);
template __global__ void rtmi_render_async<true, true, false>(DevScene, DevCamera, DevParams);
template __global__ void rtmi_render_async<true, false, false>(DevScene, DevCamera, DevParams);
template __global__ void rtmi_render_async<false, true, false>(DevScene, DevCamera, DevParams);
template __global__ void rtmi_render_async<false, false, false>(DevScene, DevCamera, DevParams);
