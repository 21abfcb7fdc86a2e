#define WALDO_LP 17
#include "warp_composite_inst.hip.h"
