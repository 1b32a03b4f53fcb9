// Corridor wire format: msgs/corridor of the reference in ROS 1 serialisation, the
// format its recorder publishes (writeCorridorMsg / readCorridorMsg, teach_repeat_planner.cpp:354-410;
// msgs/msg/corridor.msg, polyhedron.msg, facet3.msg).  ROS 1 serialises little-endian, arrays with a
// uint32 length prefix, geometry_msgs/Vector3 as three float64:
//
//   int32  path_id
//   uint32 n_polyhedrons
//   n x {  float64 center[3];  float64 seed_coord[3];  uint32 n_facets;  n_facets x float64 (a, b, c, d)  }
//
// and the replay protocol of corridorRecCallBack / fastTrajPlanning (TRP:308-352, 796-812): for every
// requested count n the problem is the first n polytopes, start = center of polytope 0, goal = center
// of polytope n-1, at rest.  Host code only.
#pragma once
#include <cstdint>
#include <cstring>

namespace direct {

struct ByteReader {
  const uint8_t* p;
  size_t left;
  bool ok = true;
  template <typename T>
  T get() {
    T v{};
    if (left < sizeof(T)) { ok = false; left = 0; return v; }
    std::memcpy(&v, p, sizeof(T));  // the host is little-endian (x86-64), as the wire format
    p += sizeof(T); left -= sizeof(T);
    return v;
  }
};

inline size_t corridor_wire_size(int n_seg, const int32_t* n_planes) {
  size_t s = 4 + 4;
  for (int k = 0; k < n_seg; k++) s += 24 + 24 + 4 + (size_t)n_planes[k] * 32;
  return s;
}

inline size_t corridor_pack(int32_t path_id, int n_seg, const int32_t* n_planes, const double* planes, int p_max,
                            const double* seeds, const double* centers, uint8_t* buf) {
  uint8_t* q = buf;
  auto put = [&](const void* v, size_t n) { std::memcpy(q, v, n); q += n; };
  const uint32_t n = (uint32_t)n_seg;
  put(&path_id, 4);
  put(&n, 4);
  for (int k = 0; k < n_seg; k++) {
    put(centers + (size_t)k * 3, 24);
    put(seeds + (size_t)k * 3, 24);
    const uint32_t m = (uint32_t)n_planes[k];
    put(&m, 4);
    for (uint32_t j = 0; j < m; j++) put(planes + ((size_t)k * p_max + j) * 4, 32);
  }
  return (size_t)(q - buf);
}

// returns 0 on success, 1 malformed / truncated, 2 more polytopes or facets than the caller's arrays hold
inline int corridor_unpack(const uint8_t* buf, size_t len, int n_seg_max, int p_max, int32_t* path_id, int32_t* n_seg,
                           int32_t* n_planes, double* planes, double* seeds, double* centers, size_t* used) {
  ByteReader r{buf, len};
  *path_id = r.get<int32_t>();
  const uint32_t n = r.get<uint32_t>();
  if (!r.ok) return 1;
  if (n > (uint32_t)n_seg_max) return 2;
  for (uint32_t k = 0; k < n; k++) {
    for (int d = 0; d < 3; d++) centers[(size_t)k * 3 + d] = r.get<double>();
    for (int d = 0; d < 3; d++) seeds[(size_t)k * 3 + d] = r.get<double>();
    const uint32_t m = r.get<uint32_t>();
    if (!r.ok) return 1;
    if (m > (uint32_t)p_max) return 2;
    n_planes[k] = (int32_t)m;
    for (uint32_t j = 0; j < m; j++)
      for (int c = 0; c < 4; c++) planes[((size_t)k * p_max + j) * 4 + c] = r.get<double>();
    if (!r.ok) return 1;
  }
  *n_seg = (int32_t)n;
  if (used) *used = len - r.left;
  return 0;
}

}  // namespace direct
