#define WALDO_LP 32
#include "warp_composite_inst.hip.h"
