#define WALDO_LP 8
#include "warp_composite_inst.hip.h"

#ifdef WALDO_K1_STAMPS
namespace waldo {
int k1_stamps_read(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(waldo_k1_stamps), sizeof(unsigned long long) * (size_t)n);
}
}  // namespace waldo
#endif
#ifdef WALDO_FWD_STAMPS
namespace waldo {
int fwd_stamps_read(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(waldo_fwd_stamps), sizeof(unsigned long long) * (size_t)n);
}
}  // namespace waldo
#endif
