// "fast" flavour, translation unit of lane layout 3 (see UCF_TU in ucf_device.h)
#define UCF_FAST 1
#define UCF_NS ucf_fast
#define UCF_TU 3
#include "ucf_device.h"
