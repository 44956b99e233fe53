// every instantiation of sample_batch_kernel for SCENE_KIND_VOLUMES scenes (see rtow_sample_kernel.hip.h)
#include "rtow_sample_kernel.hip.h"

namespace rtow {
RTOW_DEFINE_KIND_LAUNCHER(launchSampleVolumes, SCENE_KIND_VOLUMES)
}
