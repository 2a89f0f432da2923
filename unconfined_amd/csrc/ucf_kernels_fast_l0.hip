// "fast" flavour, translation unit of lane layout 0 (see UCF_TU in ucf_device.h; ucf_kernels_fast.hip = all of them,
// used by the inspection tools)
#define UCF_FAST 1
#define UCF_NS ucf_fast
#define UCF_TU 0
#include "ucf_device.h"
