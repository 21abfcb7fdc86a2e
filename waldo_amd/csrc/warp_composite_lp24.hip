#define WALDO_LP 24
#include "warp_composite_inst.hip.h"
