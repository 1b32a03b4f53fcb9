// Host-side C++ mirror of the reference's corridor generator on top of the C-ABI (include/direct_cluster.h).
//
// The reference class is `polyhedronGenerator` (global_planner/include/global_planner/utils/poly_utils.h:27-118,
// global_planner/src/utils/poly_utils.cpp): paramSet -> map -> corridorGeneration(gridPath) walks a grid path and
// asks getConvexPoly(seed) for a new polytope whenever the path leaves the latest one (:508-557).  This header keeps
// those names and that walk (coord2Index / index2Coord :3-40, isOutsidePolytope with its 0.01 margin :42-52, the pop of
// the last polytope when the path re-enters the last but one :526-530) and replaces what getConvexPoly, Polyhedron::hrep
// and polyHrep2Utils do per seed (:127-206, 282-389: voxel clustering, quickhull, cdd) by ONE device call for all the
// seeds that are due:
//   * corridorGeneration(gridPath, corridor)       the reference's single-path walk (one seed per device call)
//   * corridorInsertGeneration(coordSet, corridor) the walk the LIVE caller uses (teach_repeat_planner.cpp:172, 228;
//     poly_utils.cpp:391-449): continues the corridor it is given (isOutsideLatestPolytope against the last existing
//     polytope), never pops, works on a copy and returns 0 / 1 - on 0 (no map, cdd error) the corridor is untouched
//   * corridorGenerationBatch(gridPaths, corridors) many paths in lock step: every round collects the seed each
//     unfinished walk is waiting for and runs them as one batch - the form the device wants.  A walk's result does not
//     depend on the other walks (getConvexPoly is a function of the seed voxel and the map), so both give the same
//     corridors.
// Corridor / Polytope types are template parameters read through the member names of
// global_planner/include/global_planner/utils/data_type.h:124-245 (polyhedrons, planes, center, seed_coord,
// appendPlane, setSeed / setCenter where they exist); direct::PlainCorridor of ddp_optimizer.hpp fits.
#pragma once
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/direct_cluster.h"
#include "ddp_optimizer.hpp"

namespace direct {

class polyhedronGenerator {
 public:
  // paramSet (poly_utils.h:60-85): resolution, map origin and size in voxels, clustering limits; max_batch seeds per device call
  polyhedronGenerator(double resolution, const std::array<double, 3>& map_lower, int max_x_id, int max_y_id, int max_z_id,
                      int itr_inflate_max = 1000, int itr_cluster_max = 50, int max_batch = 64, int cluster_capacity = 50000,
                      int candidate_capacity = 10000, int device = 0)
      : res_(resolution), inv_res_(1.0 / resolution), lower_(map_lower), mx_(max_x_id), my_(max_y_id), mz_(max_z_id),
        itr_inflate_(itr_inflate_max), itr_cluster_(itr_cluster_max), max_batch_(max_batch) {
    direct_cluster_config_t cfg{device, max_x_id, max_y_id, max_z_id, max_batch, cluster_capacity, candidate_capacity, 0};
    if (direct_cluster_create(&cfg, &h_) != DIRECT_OK) throw std::runtime_error(direct_cluster_last_error());
  }
  ~polyhedronGenerator() { direct_cluster_destroy(h_); }
  polyhedronGenerator(const polyhedronGenerator&) = delete;
  polyhedronGenerator& operator=(const polyhedronGenerator&) = delete;

  // the occupancy grid, [x][y][z], 0 free / 1 obstacle (what setObs / mapUpload build, cluster_server_cpu.cpp:83-120)
  void setMap(const uint8_t* map_data) {
    if (direct_cluster_set_map(h_, DIRECT_MEM_HOST, map_data) != DIRECT_OK) throw std::runtime_error(direct_cluster_last_error());
    has_map_ = true;
  }

  std::array<int, 3> coord2Index(const std::array<double, 3>& c) const {  // :10-18
    auto f = [&](double v, double lo, int mx) {
      int i = (int)((v - lo) * inv_res_);
      return i < 0 ? 0 : (i > mx - 1 ? mx - 1 : i);
    };
    return {{f(c[0], lower_[0], mx_), f(c[1], lower_[1], my_), f(c[2], lower_[2], mz_)}};
  }
  std::array<double, 3> index2Coord(const std::array<int, 3>& i) const {  // :20-29
    return {{i[0] * res_ + 0.5 * res_ + lower_[0], i[1] * res_ + 0.5 * res_ + lower_[1], i[2] * res_ + 0.5 * res_ + lower_[2]}};
  }
  template <class Polytope>
  static bool isOutsidePolytope(const std::array<double, 3>& c, const Polytope& p) {  // :42-52
    for (const auto& pl : p.planes)
      if (c[0] * elem(pl, 0) + c[1] * elem(pl, 1) + c[2] * elem(pl, 2) + elem(pl, 3) > 0.01) return true;
    return false;
  }

  template <class Corridor>
  static bool isOutsideLatestPolytope(const std::array<double, 3>& c, const Corridor& cor) {  // :59-62
    return isOutsidePolytope(c, cor.polyhedrons.back());
  }
  template <class Corridor>
  static bool isInsideLastSecondPolytope(const std::array<double, 3>& c, const Corridor& cor) {  // :64-70
    const size_t n = cor.polyhedrons.size();
    return n > 1 ? !isOutsidePolytope(c, cor.polyhedrons[n - 2]) : false;
  }

  // corridorInsertGeneration (:391-449), the walk of the live caller (teach_repeat_planner.cpp:172, 228): new polytopes
  // are appended behind the ones `corridor` already holds; a point inside the latest polytope adds nothing; there is no
  // pop.  Returns 1 and replaces `corridor` when the whole path went through, 0 (corridor untouched) without a map or
  // where the reference's cdd call fails (:437-439).  The reference's second argument, the visualisation message, has
  // no counterpart here (RViz is out of scope).
  template <class Corridor>
  int corridorInsertGeneration(const std::vector<std::array<double, 3>>& coordSet, Corridor& corridor) {
    if (!has_map_) return 0;  // :394
    std::vector<Corridor*> cs{&corridor};
    return corridorInsertGenerationBatch(std::vector<std::vector<std::array<double, 3>>>{coordSet}, cs)[0];
  }
  // the same for many (path, corridor) pairs in lock step (one device batch per round of due seeds)
  template <class Corridor>
  std::vector<int> corridorInsertGenerationBatch(const std::vector<std::vector<std::array<double, 3>>>& coordSets,
                                                 const std::vector<Corridor*>& corridors) {
    std::vector<int> rc(coordSets.size(), 0);
    if (!has_map_) return rc;
    std::vector<Corridor> beg;  // beg_corridor (:399-403): the copy the walk extends
    for (Corridor* c : corridors) beg.push_back(*c);
    std::vector<Corridor*> ptr;
    for (Corridor& c : beg) ptr.push_back(&c);
    const std::vector<bool> ok = walk(coordSets, ptr, false);
    for (size_t p = 0; p < coordSets.size(); p++)
      if (ok[p]) {
        *corridors[p] = beg[p];  // :447
        rc[p] = 1;
      }
    return rc;
  }

  // corridorGeneration (:508-557) for one path.  Returns false where the reference prints "corridor generation broke".
  template <class Corridor>
  bool corridorGeneration(const std::vector<std::array<double, 3>>& gridPath, Corridor& corridor) {
    std::vector<Corridor*> cs{&corridor};
    return corridorGenerationBatch(std::vector<std::vector<std::array<double, 3>>>{gridPath}, cs)[0];
  }

  // The same walk for many paths in lock step; corridors[p] is extended like the reference's member `corridor`.
  template <class Corridor>
  std::vector<bool> corridorGenerationBatch(const std::vector<std::vector<std::array<double, 3>>>& gridPaths,
                                            const std::vector<Corridor*>& corridors) {
    if (!has_map_) throw std::runtime_error("polyhedronGenerator: no map");
    return walk(gridPaths, corridors, true);
  }

 private:
  // The walk all entry points share.  pop_back: corridorGeneration's return into the last but one polytope (:524-528);
  // corridorInsertGeneration has none.
  template <class Corridor>
  std::vector<bool> walk(const std::vector<std::vector<std::array<double, 3>>>& gridPaths, const std::vector<Corridor*>& corridors,
                         bool pop_back) {
    const size_t np = gridPaths.size();
    struct Walk { size_t next = 0; std::array<double, 3> lst{{-INFINITY, -INFINITY, -INFINITY}}; bool done = false, ok = true; };
    std::vector<Walk> w(np);
    std::vector<size_t> due;          // walks waiting for a polytope
    std::vector<int32_t> seeds;
    std::vector<std::array<double, 3>> due_coord;
    rounds_ = 0;
    polytopes_ = 0;
    for (;;) {
      due.clear(); seeds.clear(); due_coord.clear();
      for (size_t p = 0; p < np; p++) {
        Walk& s = w[p];
        Corridor& cor = *corridors[p];
        while (!s.done && s.next < gridPaths[p].size()) {
          const std::array<int, 3> idx = coord2Index(gridPaths[p][s.next]);
          const std::array<double, 3> cur = index2Coord(idx);
          if (cur == s.lst) { s.next++; continue; }
          if (pop_back && isInsideLastSecondPolytope(cur, cor)) cor.polyhedrons.pop_back();  // :524-528
          if (cor.polyhedrons.empty() || isOutsideLatestPolytope(cur, cor)) {  // :413, :530
            due.push_back(p);
            seeds.insert(seeds.end(), idx.begin(), idx.end());
            due_coord.push_back(cur);
            break;  // resumes behind this point once the polytope is there
          }
          s.lst = cur;
          s.next++;
        }
        if (s.next >= gridPaths[p].size()) s.done = true;
      }
      if (due.empty()) break;
      rounds_++;
      for (size_t b0 = 0; b0 < due.size(); b0 += (size_t)max_batch_) {
        const int nb = (int)std::min((size_t)max_batch_, due.size() - b0);
        std::vector<PlainPolytope> poly;
        std::vector<int32_t> rtn;
        getConvexPolyBatch(nb, &seeds[3 * b0], poly, rtn);
        for (int b = 0; b < nb; b++) {
          Walk& s = w[due[b0 + b]];
          if (rtn[b] != DIRECT_HULL_OK) {  // the reference's cdd error branch (:548-552): the walk stops
            s.ok = false;
            s.done = true;
            continue;
          }
          Corridor& cor = *corridors[due[b0 + b]];
          typename std::decay<decltype(cor.polyhedrons[0])>::type pt{};
          for (const auto& pl : poly[b].planes) pt.appendPlane({pl[0], pl[1], pl[2], pl[3]});
          for (int a = 0; a < 3; a++) {
            set_elem(pt.center, a, poly[b].center[a]);
            set_elem(pt.seed_coord, a, due_coord[b0 + b][a]);
          }
          cor.polyhedrons.push_back(pt);
          polytopes_++;
          s.lst = due_coord[b0 + b];
          s.next++;
        }
      }
    }
    std::vector<bool> ok(np);
    for (size_t p = 0; p < np; p++) ok[p] = w[p].ok;
    return ok;
  }

 public:

  // getConvexPoly + hrep + polyHrep2Utils for `batch` seed voxels: clusters and planes never leave the device in between
  void getConvexPolyBatch(int batch, const int32_t* seed_idx, std::vector<PlainPolytope>& out, std::vector<int32_t>& rtn) {
    const int pcap = 128;  // clusters of a few thousand voxels have 20 - 50 facets
    std::vector<double> planes((size_t)batch * pcap * 4), center((size_t)batch * 3);
    std::vector<int32_t> npl(batch), crtn(batch);
    rtn.assign(batch, 0);
    if (direct_cluster_polygon_generation_batch(h_, batch, seed_idx, itr_inflate_, itr_cluster_, DIRECT_MEM_HOST, nullptr, nullptr,
                                                nullptr, nullptr, crtn.data()) != DIRECT_OK)
      throw std::runtime_error(direct_cluster_last_error());
    if (direct_cluster_hull_planes_batch(h_, batch, DIRECT_MEM_HOST, nullptr, nullptr, res_, lower_.data(), pcap, 1, DIRECT_MEM_HOST,
                                         planes.data(), nullptr, npl.data(), nullptr, nullptr, center.data(), nullptr,
                                         rtn.data()) != DIRECT_OK)
      throw std::runtime_error(direct_cluster_last_error());
    out.assign(batch, PlainPolytope());
    for (int b = 0; b < batch; b++) {
      if (crtn[b] != DIRECT_CLUSTER_OK) rtn[b] = DIRECT_HULL_OVERFLOW;
      if (rtn[b] != DIRECT_HULL_OK) continue;
      for (int k = 0; k < npl[b]; k++) {
        const double* p = &planes[((size_t)b * pcap + k) * 4];
        out[b].appendPlane({{p[0], p[1], p[2], p[3]}});
      }
      for (int a = 0; a < 3; a++) out[b].center[a] = center[(size_t)b * 3 + a];
    }
  }

  int lastRounds() const { return rounds_; }        // device batches of the last corridorGenerationBatch
  int lastPolytopes() const { return polytopes_; }  // polytopes it generated (popped ones included)
  direct_cluster_handle_t handle() const { return h_; }

 private:
  double res_, inv_res_;
  std::array<double, 3> lower_;
  int mx_, my_, mz_, itr_inflate_, itr_cluster_, max_batch_;
  direct_cluster_handle_t h_ = nullptr;
  bool has_map_ = false;
  int rounds_ = 0, polytopes_ = 0;
};

}  // namespace direct
