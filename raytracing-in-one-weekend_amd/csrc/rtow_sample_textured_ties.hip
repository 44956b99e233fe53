// every instantiation of sample_batch_kernel for SCENE_KIND_TEXTURED scenes that hold duplicate primitives: nearest-hit ties are settled by
// the reference's whole procedure (resolve_nearest_tie, rtow_sample_kernel.hip.h)
#include "rtow_sample_kernel.hip.h"

namespace rtow {
RTOW_DEFINE_KIND_LAUNCHER(launchSampleTexturedTies, SCENE_KIND_TEXTURED | kExactTiesBit)
}
