/*
 * direct_cluster.h -- C-ABI of the MI355X-native corridor-cluster generator: the step that
 * produces the polytopes the DDP path consumes.  Replaces, for a BATCH of seed voxels on one voxel map,
 *
 *   void cudaPolytopeGeneration::polygonGeneration(vector<int>& x, vector<int>& y, vector<int>& z)
 *       polyhedron_generator/include/polyhedron_generator/cluster_server_cpu.h:61   (shipped CPU build)
 *       polyhedron_generator/src/cluster_server_cpu.cpp:394-528
 *   with paramSet / setObs / mapClear (cluster_server_cpu.cpp:8-46, 83-120) for the map,
 *
 * whose inner loops are cubeInflation_cpu (:257-293), polytopeCluster_cpu (:295-392) and serialConvexTest
 * (cluster_engine_cpu.cpp:31-136); the reference's optional CUDA twins are paraCubeInflation / paraConvexTest /
 * paraResultCheck (cluster_engine.cu:37-350).  RESULTS ARE THOSE OF THE SHIPPED CPU BUILD, bit for bit: the same
 * cluster voxels in the same order (the CUDA twins use a different DDA - double precision, different termination
 * order - and are not what poly_utils.h:13-14 links).  The caller is polyhedronGenerator::getConvexPoly
 * (global_planner/src/utils/poly_utils.cpp:285-299), which passes ONE seed voxel per call; a batch here is many
 * such calls (every seed along an A* path, or many paths) on the same map.
 *
 * Voxel (x, y, z) lives at index x * max_y * max_z + y * max_z + z (cluster_server_cpu.cpp:30).  Map bytes are
 * 0 (free) or 1 (obstacle), as setObs / setFr write them.  Plain pointers and sizes only.
 */
#ifndef DIRECT_CLUSTER_H_
#define DIRECT_CLUSTER_H_

#include <stddef.h>
#include <stdint.h>

#include "direct_ddp.h" /* direct_status_t, direct_mem_t */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int32_t device;             /* HIP device ordinal */
  int32_t max_x, max_y, max_z; /* paramSet's max_x_id, max_y_id, max_z_id (voxels per axis) */
  int32_t max_batch;          /* seeds per call */
  int32_t cluster_capacity;   /* voxels per cluster (_cluster_buffer_size: 50000 in the CPU build) */
  int32_t candidate_capacity; /* candidates per round (_candidate_buffer_size: 10000) */
  int32_t reserved;
} direct_cluster_config_t;

typedef struct direct_cluster_handle_s* direct_cluster_handle_t;

direct_status_t direct_cluster_create(const direct_cluster_config_t* cfg, direct_cluster_handle_t* out);
direct_status_t direct_cluster_destroy(direct_cluster_handle_t h);
const char* direct_cluster_last_error(void);

/* mapClear + setObs for every obstacle voxel + mapUpload: the whole occupancy grid at once.
 * map_data[max_x*max_y*max_z] in memory kind `mem`. */
direct_status_t direct_cluster_set_map(direct_cluster_handle_t h, int32_t mem, const uint8_t* map_data);

/* Return codes per seed (the reference has none: it writes past its buffers instead). */
#define DIRECT_CLUSTER_OK 0
#define DIRECT_CLUSTER_OVERFLOW 1  /* cluster_capacity / candidate_capacity exceeded: result truncated, not usable */
#define DIRECT_CLUSTER_BAD_SEED 2  /* seed voxel outside the map */

/* polygonGeneration for `batch` seeds (host array seeds[batch][3]).  itr_inflate_max / itr_cluster_max as given to
 * paramSet with is_cluster_on (is_cluster_on == false is (1000, 0)).  Outputs in memory kind `mem`, any may be NULL:
 *   vertex_idx[batch][24]       the inflated cube (cluster_server_cpu.cpp:48-60 layout: x 0..7, y 8..15, z 16..23)
 *   cluster_xyz[batch][cluster_capacity][3]   cluster voxels in the reference's order
 *   cluster_num[batch], cluster_iters[batch] (completed rounds of polytopeCluster_cpu), rtn[batch] (codes above) */
direct_status_t direct_cluster_polygon_generation_batch(direct_cluster_handle_t h, int32_t batch, const int32_t* seeds,
                                                        int32_t itr_inflate_max, int32_t itr_cluster_max, int32_t mem,
                                                        int32_t* vertex_idx, int32_t* cluster_xyz, int32_t* cluster_num,
                                                        int32_t* cluster_iters, int32_t* rtn);

/* ---- kernel-level entry point (parity tests against the reference's own serialConvexTest) ----------------------
 * One convex test round on caller-provided state: for every candidate i
 *   can_clu[i]  = serialConvexTest(candidate i, cluster[0..n_cluster), inside_data, map)      (uint8 0/1)
 *   can_can[i*(i-1)/2 + j], j < i  = the same ray test from candidate i towards candidate j alone
 *                 (the packed lower triangle of paraResultCheck, cluster_engine.cu:37-67; may be NULL)
 *   accept[i]   = can_clu[i] && for all j < i with accept[j]: can_can[i][j]     -- what polytopeCluster_cpu's
 *                 sequential loop decides (cluster_server_cpu.cpp:360-384), since serialConvexTest is an AND over
 *                 targets and accepted candidates join the cluster at once (may be NULL)
 * All arrays are HOST memory; inside_data is the reference's per-voxel flag array.
 * The call uses the handle's cluster storage: it INVALIDATES the clusters a preceding
 * direct_cluster_polygon_generation_batch left resident (a following direct_cluster_hull_planes_batch with
 * cluster_xyz == NULL fails with DIRECT_ERR_INVALID until the next generation). */
direct_status_t direct_cluster_convex_test(direct_cluster_handle_t h, const uint8_t* inside_data, int32_t n_candidate,
                                           const int32_t* candidate_xyz, int32_t n_cluster, const int32_t* cluster_xyz,
                                           uint8_t* can_clu, uint8_t* can_can, uint8_t* accept);

/* ---- hull -> planes (the tail of SURVEY.md 8f-4): cluster voxels -> the polytope the DDP path consumes ---------
 * Replaces, for a batch of clusters, what polyhedronGenerator does with the result of polygonGeneration
 * (global_planner/src/utils/poly_utils.cpp):
 *   getConvexPoly   :301-389  point set (voxel centres; the eight corners of every voxel when checkDegeneratePoly
 *                             :236-273 finds the cluster flat along an axis), third_party/quickhull, vertex buffer
 *   Polyhedron::hrep (eigen-cdd / cddlib), call sites :404-449, :470-480   V-rep -> A x <= b
 *   polyHrep2Utils  :127-206  unit normals pointing outwards, plane (a, b, c, d) with a x + b y + c z + d <= 0 inside,
 *                             axis-aligned faces of a solid cluster moved out by half a voxel, centre
 * All of the reference's points lie on the half-voxel lattice q = 2 index + 1 (+/- 1 for corners), world coordinate
 * x = q * resolution / 2 + map_lower, so the hull is computed in exact integer arithmetic: the planes are THE facet
 * planes the two floating-point libraries approximate.  Not reproducible and therefore defined here: the ORDER of the
 * rows (cdd's; here ascending (nx, ny, nz, K) of the primitive integer normal) and the per-plane vertex behind the
 * centre (an argmin over residuals that are all ~1e-16; here the first corner, in cluster order, on the plane).
 * Vertices are the corners of the hull in cluster order (quickhull's vertex buffer may also hold boundary points that
 * are not corners; cdd's H-rep does not depend on them).
 *
 * cluster_xyz == NULL: the clusters of the last polygon_generation_batch, still resident on the device (no copy of
 * the voxels in either direction; `batch` may not exceed that call's, and a seed whose generation did not end with
 * DIRECT_CLUSTER_OK has no usable cluster: DIRECT_HULL_OVERFLOW / DIRECT_HULL_FLAT).  Otherwise
 * cluster_xyz[batch][cluster_capacity][3] / cluster_num[batch] in memory kind mem_in REPLACE them in the handle's storage
 * (a later call with cluster_xyz == NULL needs a new generation first); every voxel must lie inside the map.  Outputs in memory kind `mem`, any may be NULL:
 *   planes[batch][plane_capacity][4] (double), plane_int[batch][plane_capacity][4] (int64: primitive normal and
 *   offset on the lattice, n . q + K <= 0), n_planes[batch], vertices[batch][vertex_capacity][3], n_vertices[batch],
 *   center[batch][3], degenerate[batch] (checkDegeneratePoly), rtn[batch] (codes below). */
#define DIRECT_HULL_OK 0
#define DIRECT_HULL_OVERFLOW 1 /* more planes / vertices than the capacity of an output that was asked for, more than 2048 line-extreme points or 8192 plane reports of hull edges, more than 2048 planes with `center` asked for, or a resident cluster whose generation overflowed */
#define DIRECT_HULL_BAD_VOXEL 2 /* a caller-provided voxel lies outside the map [0, max_x) x [0, max_y) x [0, max_z): nothing is computed for the cluster */
#define DIRECT_HULL_FLAT 3     /* empty cluster, or the points do not span three dimensions (the reference's cdd call fails) */
direct_status_t direct_cluster_hull_planes_batch(direct_cluster_handle_t h, int32_t batch, int32_t mem_in,
                                                 const int32_t* cluster_xyz, const int32_t* cluster_num, double resolution,
                                                 const double* map_lower, int32_t plane_capacity, int32_t vertex_capacity,
                                                 int32_t mem, double* planes, int64_t* plane_int, int32_t* n_planes,
                                                 double* vertices, int32_t* n_vertices, double* center, int32_t* degenerate,
                                                 int32_t* rtn);

/* The HIP stream (hipStream_t) the handle enqueues its copies, kernels and timing events on; NULL (the default) is
 * the legacy default stream.  Mirrors direct_ddp_set_stream. */
direct_status_t direct_cluster_set_stream(direct_cluster_handle_t h, void* hip_stream);
/* HIP-event time [ms] of the kernels of the last polygon_generation_batch / convex_test / hull_planes_batch call */
direct_status_t direct_cluster_last_ms(direct_cluster_handle_t h, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* DIRECT_CLUSTER_H_ */
