// every instantiation of sample_batch_kernel for SCENE_KIND_TRIANGLES scenes - all entities triangles, the reference's live scenes (see rtow_sample_kernel.hip.h)
#include "rtow_sample_kernel.hip.h"

namespace rtow {
RTOW_DEFINE_KIND_LAUNCHER(launchSampleTriangles, SCENE_KIND_TRIANGLES)
}
