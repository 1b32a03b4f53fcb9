// TEST INFRASTRUCTURE.  C entry point for the REFERENCE's serialConvexTest, compiled together with the reference's
// own polyhedron_generator/src/cluster_engine_cpu.cpp (from where it lies under /root/reference, never copied) into
// oracle/_ref/libcluster_engine_ref.so by `make -C oracle _ref`.  The declaration below is the one in
// polyhedron_generator/include/polyhedron_generator/cluster_engine_cpu.h:9-11.
#include <stdint.h>

bool serialConvexTest(const int& can_x_index, const int& can_y_index, const int& can_z_index, const int& cluster_grid_num,
                      const int& max_yz_id, const int& max_z_id, const int* cluster_xyz_id, const uint8_t* inside_data,
                      const uint8_t* map_data);

extern "C" int ref_serial_convex_test(int can_x, int can_y, int can_z, int cluster_grid_num, int max_yz_id, int max_z_id,
                                      const int* cluster_xyz_id, const uint8_t* inside_data, const uint8_t* map_data) {
  return serialConvexTest(can_x, can_y, can_z, cluster_grid_num, max_yz_id, max_z_id, cluster_xyz_id, inside_data, map_data) ? 1 : 0;
}
