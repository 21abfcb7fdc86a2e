#define WALDO_LP 8
#include "warp_composite_inst.hip.h"
