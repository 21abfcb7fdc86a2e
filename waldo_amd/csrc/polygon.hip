// f3: the point-in-polygon test of WIF.inpaint (models/nets/wif.py:228-235: matplotlib.path.Path(corners)
// .contains_points(pts), radius 0, no transform) for the region an object enters the frame from (wif.py:140-160).
// The reference runs it on the host: a device -> host copy of every pixel coordinate (4 MB at 512 x 1024), 3.3 ms of
// matplotlib with the GPU idle, a host -> device copy of the mask.  matplotlib's test (src/_path.h:point_in_path_impl)
// is the crossings-multiply test of Haines ("Point in Polygon Strategies", Graphics Gems IV) in DOUBLE precision over
// the path's vertices, closed back to its first vertex:
//
//     inside = false;  yflag0 = (v0.y >= ty)
//     for every edge (a, b) of (v0, v0), (v0, v1), ..., (v[K-2], v[K-1]), (v[K-1], v0):
//         yflag1 = (b.y >= ty)
//         if (yflag0 != yflag1  and  ((b.y - ty) * (a.x - b.x) >= (b.x - tx) * (a.y - b.y)) == yflag1)  inside = !inside
//         yflag0 = yflag1
//
// restated here operation by operation -- the float32 coordinates widened to double as numpy's conversion does, the
// products and differences in the same order, no fused multiply-add (the library is built with -ffp-contract=off; an
// x86-64 wheel has none either) -- so that a pixel ON an edge falls on the side matplotlib puts it
// (tests/test_inpaint.py::test_points_in_polygon_is_matplotlibs: random, concave and degenerate polygons, points on
// vertices and edges, against matplotlib itself).  The corners are read on the HOST (they come from host scalars,
// wif.py:146-157) and travel as kernel arguments: no copy is queued.
#include "waldo_common.hip.h"

namespace waldo {

constexpr int kMaxCorners = 16;

struct PolygonArg {
  double x[kMaxCorners];
  double y[kMaxCorners];
};

__global__ __launch_bounds__(kBlock) void points_in_polygon_kernel(const float* __restrict__ pts, PolygonArg poly, int K,
                                                                   float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const double tx = (double)pts[2 * i], ty = (double)pts[2 * i + 1];
  if (!(fabs(tx) <= 1.7976931348623157e308) || !(fabs(ty) <= 1.7976931348623157e308)) {  // not finite: outside
    out[i] = 0.0f;
    return;
  }
  bool inside = false;
  double ax = poly.x[0], ay = poly.y[0];        // the edge's first vertex (vtx0, vty0)
  double bx = ax, by = ay;                      // its second (vtx1, vty1): the walk starts with the edge (v0, v0)
  bool yflag0 = ay >= ty;
  for (int e = 0; e <= K; ++e) {                // e < K: the next vertex read is v[e + 1] (v0 again behind the last)
    const bool yflag1 = by >= ty;
    if (yflag0 != yflag1) {
      const double lhs = (by - ty) * (ax - bx);
      const double rhs = (bx - tx) * (ay - by);
      if ((lhs >= rhs) == yflag1) inside = !inside;
    }
    yflag0 = yflag1;
    ax = bx;
    ay = by;
    const int nxt = (e + 1 < K) ? e + 1 : 0;
    bx = poly.x[nxt];
    by = poly.y[nxt];
  }
  out[i] = inside ? 1.0f : 0.0f;
}

}  // namespace waldo

using namespace waldo;

extern "C" int waldo_points_in_polygon_fwd(const float* pts, const double* corners_host, int K, float* out, int64_t N,
                                           waldo_stream_t stream) {
  if (N < 0 || K < 0 || K > kMaxCorners) {
    set_error("waldo_points_in_polygon_fwd: bad arguments N=%lld K=%d (at most %d corners)", (long long)N, K, kMaxCorners);
    return WALDO_EINVAL;
  }
  if (N == 0) return WALDO_OK;
  if (!pts || !out || (K > 0 && !corners_host)) {
    set_error("waldo_points_in_polygon_fwd: null pointer");
    return WALDO_EINVAL;
  }
  if (K < 3) {  // matplotlib: a path of fewer than three vertices contains nothing
    fill_words(out, 0u, (size_t)N * sizeof(float), (hipStream_t)stream);
    return launch_status("waldo_points_in_polygon_fwd");
  }
  PolygonArg poly;
  for (int k = 0; k < kMaxCorners; ++k) {
    poly.x[k] = k < K ? corners_host[2 * k] : 0.0;
    poly.y[k] = k < K ? corners_host[2 * k + 1] : 0.0;
  }
  const int64_t blocks = (N + kBlock - 1) / kBlock;
  if (blocks > 2147483647) {
    set_error("waldo_points_in_polygon_fwd: too many points for one launch");
    return WALDO_EINVAL;
  }
  points_in_polygon_kernel<<<dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream>>>(pts, poly, K, out, N);
  return launch_status("waldo_points_in_polygon_fwd");
}
