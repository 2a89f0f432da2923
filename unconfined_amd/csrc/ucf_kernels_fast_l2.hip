// "fast" flavour, translation unit of lane layout 2 (see UCF_TU in ucf_device.h; ucf_kernels_fast.hip = all of them,
// used by the inspection tools)
#define UCF_FAST 1
#define UCF_NS ucf_fast
#define UCF_TU 2
#include "ucf_device.h"
