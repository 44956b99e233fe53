// every instantiation of sample_batch_kernel for SCENE_KIND_TRIANGLES_TEXTURED scenes - all entities triangles, some material with an Image texture
#include "rtow_sample_kernel.hip.h"

namespace rtow {
RTOW_DEFINE_KIND_LAUNCHER(launchSampleTrianglesTextured, SCENE_KIND_TRIANGLES_TEXTURED)
}
