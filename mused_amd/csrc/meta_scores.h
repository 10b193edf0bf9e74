// Scores of the two-column metadata records (/root/reference/matrix_operations.py:22-54, 250-263), shared by
// meta.hip (score matrices) and knn.hip (selection straight from the records).
#pragma once
#include <hip/hip_runtime.h>

namespace mused {

// Every operation is rounded on its own (no fused multiply-add), in the order the reference's Python expression
// evaluates it, so that the only difference to the host arithmetic is the last-bit accuracy of sin / cos / asin.
__device__ __forceinline__ double haversine_km(double lat1d, double lon1d, double lat2d, double lon2d) {
#pragma clang fp contract(off)
  const double d2r = 3.14159265358979323846 / 180.0;  // math.radians: x * (pi / 180)
  const double lat1 = __dmul_rn(lat1d, d2r), lon1 = __dmul_rn(lon1d, d2r);
  const double lat2 = __dmul_rn(lat2d, d2r), lon2 = __dmul_rn(lon2d, d2r);
  const double sdlat = sin(__dmul_rn(__dsub_rn(lat2, lat1), 0.5));
  const double sdlon = sin(__dmul_rn(__dsub_rn(lon2, lon1), 0.5));
  const double cc = __dmul_rn(cos(lat1), cos(lat2));
  const double a = __dadd_rn(__dmul_rn(sdlat, sdlat), __dmul_rn(cc, __dmul_rn(sdlon, sdlon)));
  return __dmul_rn(__dmul_rn(2.0, asin(sqrt(a))), 6371.0);
}

// |datetaken_j - datetaken_i| + |dateupload_j - dateupload_i| (:40-50)
__device__ __forceinline__ double time_l1(double a0, double a1, double b0, double b1) {
#pragma clang fp contract(off)
  return __dadd_rn(fabs(__dsub_rn(b0, a0)), fabs(__dsub_rn(b1, a1)));
}

}  // namespace mused
