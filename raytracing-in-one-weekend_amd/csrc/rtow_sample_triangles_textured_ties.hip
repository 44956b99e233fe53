// every instantiation of sample_batch_kernel for SCENE_KIND_TRIANGLES_TEXTURED scenes that need the exact-tie resolver (more than 16 triangles, or duplicates)
#include "rtow_sample_kernel.hip.h"

namespace rtow {
RTOW_DEFINE_KIND_LAUNCHER(launchSampleTrianglesTexturedTies, SCENE_KIND_TRIANGLES_TEXTURED | kExactTiesBit)
}
