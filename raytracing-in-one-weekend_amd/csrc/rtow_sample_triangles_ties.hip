// every instantiation of sample_batch_kernel for SCENE_KIND_TRIANGLES scenes of more than 16 triangles (or with duplicates): nearest-hit ties are settled by
// the reference's whole procedure (resolve_nearest_tie, rtow_sample_kernel.hip.h)
#include "rtow_sample_kernel.hip.h"

namespace rtow {
RTOW_DEFINE_KIND_LAUNCHER(launchSampleTrianglesTies, SCENE_KIND_TRIANGLES | kExactTiesBit)
}
