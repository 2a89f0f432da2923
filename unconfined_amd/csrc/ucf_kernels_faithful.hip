// "faithful" flavour of the kernels: compiled with -ffp-contract=off so that every
// floating point operation happens in the reference's order with no fused multiply-add.
#define UCF_FAST 0
#define UCF_NS ucf_faithful
#include "ucf_device.h"
