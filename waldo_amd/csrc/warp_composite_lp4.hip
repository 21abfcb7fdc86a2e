#define WALDO_LP 4
#include "warp_composite_inst.hip.h"
