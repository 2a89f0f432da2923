// "fast" flavour: FMA contraction on, shared exponentials, reciprocal-based complex division.
#define UCF_FAST 1
#define UCF_NS ucf_fast
#include "ucf_device.h"
