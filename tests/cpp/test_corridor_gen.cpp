// C++ host-side check (GPU needed): drives direct::polyhedronGenerator (direct_amd/host/poly_utils.hpp) like the
// reference drives its corridor generator - a map, grid paths, corridorGeneration - once path by path and once for all
// paths in lock step, and writes the corridors for tests/test_gpu_hull.py to compare with the CPU restatement.
//   usage: test_corridor_gen <in.bin> <out.bin> [seeds per device call, default 16]
//   in : int32 X Y Z, double res, double lower[3], int32 n_paths, {int32 len, double pts[len][3]}*, uint8 map[X*Y*Z]
//   out: per mode (0 = corridorGeneration one by one, 1 = all paths in lock step, 2 = corridorInsertGeneration: the first half of
//        every path into an empty corridor, then the second half into that corridor - the live caller's pattern,
//        teach_repeat_planner.cpp:172 / 228 -, 3 = the same two calls for all paths in lock step): per path: int32 ok, int32 n_poly, {int32 n_planes, double planes[n][4], center[3], seed[3]}*
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../direct_amd/host/poly_utils.hpp"

template <class T>
static void rd(FILE* f, T* p, size_t n) {
  if (fread(p, sizeof(T), n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); }
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t dims[3], np;
  double res;
  std::array<double, 3> lower;
  rd(f, dims, 3); rd(f, &res, 1); rd(f, lower.data(), 3); rd(f, &np, 1);
  std::vector<std::vector<std::array<double, 3>>> paths(np);
  for (auto& p : paths) {
    int32_t len;
    rd(f, &len, 1);
    p.resize(len);
    rd(f, &p[0][0], (size_t)len * 3);
  }
  std::vector<uint8_t> map((size_t)dims[0] * dims[1] * dims[2]);
  rd(f, map.data(), map.size());
  std::fclose(f);
  const int max_batch = argc > 3 ? std::atoi(argv[3]) : 16;
  direct::polyhedronGenerator gen(res, lower, dims[0], dims[1], dims[2], 1000, 50, max_batch);
  gen.setMap(map.data());
  FILE* o = std::fopen(argv[2], "wb");
  for (int mode = 0; mode < 4; mode++) {
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<direct::PlainCorridor> cors(np);
    std::vector<bool> ok(np);
    if (mode == 0) {
      for (int p = 0; p < np; p++) ok[p] = gen.corridorGeneration(paths[p], cors[p]);
    } else if (mode == 1) {
      std::vector<direct::PlainCorridor*> ptr;
      for (auto& c : cors) ptr.push_back(&c);
      ok = gen.corridorGenerationBatch(paths, ptr);
      std::printf("batch: %d device rounds for %d polytopes on %d paths\n", gen.lastRounds(), gen.lastPolytopes(), np);
    } else {
      std::vector<std::vector<std::array<double, 3>>> first(np), second(np);
      for (int p = 0; p < np; p++) {
        const size_t h = paths[p].size() / 2;
        first[p].assign(paths[p].begin(), paths[p].begin() + h);
        second[p].assign(paths[p].begin() + h, paths[p].end());
      }
      if (mode == 2) {
        for (int p = 0; p < np; p++)
          ok[p] = gen.corridorInsertGeneration(first[p], cors[p]) == 1 && gen.corridorInsertGeneration(second[p], cors[p]) == 1;
      } else {
        std::vector<direct::PlainCorridor*> ptr;
        for (auto& c : cors) ptr.push_back(&c);
        const std::vector<int> r1 = gen.corridorInsertGenerationBatch(first, ptr), r2 = gen.corridorInsertGenerationBatch(second, ptr);
        for (int p = 0; p < np; p++) ok[p] = r1[p] == 1 && r2[p] == 1;
      }
    }
    std::printf("mode %d: %.3f ms\n", mode, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    for (int p = 0; p < np; p++) {
      const int32_t okv = ok[p] ? 1 : 0, n = (int32_t)cors[p].polyhedrons.size();
      std::fwrite(&okv, 4, 1, o); std::fwrite(&n, 4, 1, o);
      for (const auto& pl : cors[p].polyhedrons) {
        const int32_t k = (int32_t)pl.planes.size();
        std::fwrite(&k, 4, 1, o);
        for (const auto& q : pl.planes) std::fwrite(q.data(), 8, 4, o);
        std::fwrite(pl.center.data(), 8, 3, o);
        std::fwrite(pl.seed_coord.data(), 8, 3, o);
      }
    }
  }
  std::fclose(o);
  std::printf("PASS\n");
  return 0;
}
