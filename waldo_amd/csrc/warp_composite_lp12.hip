#define WALDO_LP 12
#include "warp_composite_inst.hip.h"
