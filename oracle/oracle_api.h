/* oracle/oracle_api.h -- TEST INFRASTRUCTURE ONLY. Never linked into, imported by or called from the
 * product path (fiesta_amd/, include/, libfiesta_hip.so). Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load the libraries that export this API.
 *
 * One C API, two implementations that export the very same symbols:
 *   oracle/_ref/libfiesta_ref_{array,hash}.so   the reference's own src/ESDFMap.cpp + src/raycast.cpp
 *                                               compiled VERBATIM from /root/reference against the header
 *                                               shims in oracle/shim (built by oracle/Makefile; binaries
 *                                               only, never sources, never committed);
 *   oracle/libfiesta_port.so                    oracle/esdf_port.cpp, a from-scratch CPU restatement
 *                                               of the same algorithm (travels to the GPU box).
 * The restatement is pinned against the verbatim build in tests/test_oracle_port_vs_ref.py and against
 * fixtures under tests/golden/ that were generated from the verbatim build.
 */
#ifndef FIESTA_ORACLE_API_H
#define FIESTA_ORACLE_API_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_map oracle_map;

typedef struct oracle_esdf_stats {
  int64_t inserted;   /* insert_queue_ size at entry   (reference prints it, src/ESDFMap.cpp:277) */
  int64_t deleted;    /* delete_queue_ size at entry */
  int64_t expanded;   /* "Expanding N nodes"           (src/ESDFMap.cpp:394) */
  int64_t change_num; /* "with change_num = C" */
  double seconds;     /* steady_clock around UpdateESDF only, stdout muted */
} oracle_esdf_stats;

/* mode 0 = dense array (ESDFMap(origin,res,map_size), src/ESDFMap.cpp:171), 1 = hash blocks
 * (ESDFMap(origin,res,reserve), :130). A library built for one flavour returns NULL for the other. */
oracle_map *oracle_create(int mode, const double origin[3], double resolution, const double map_size[3],
                          int reserve_size);
void oracle_destroy(oracle_map *m);
const char *oracle_kind(void); /* "reference-array", "reference-hash" or "port" */

void oracle_set_parameters(oracle_map *m, double p_hit, double p_miss, double p_min, double p_max,
                           double p_occ);
int64_t oracle_grid_total_size(oracle_map *m); /* array: grid_total_size_; hash: count */
void oracle_grid_size(oracle_map *m, int32_t out[3]);
void oracle_set_original_range(oracle_map *m);
void oracle_set_update_range(oracle_map *m, const double min_pos[3], const double max_pos[3], int new_vec);

/* SetOccupancy(Vector3i,int) / SetOccupancy(Vector3d,int) applied in array order; ret (nullable)
 * receives each call's return value. */
void oracle_set_occupancy_vox(oracle_map *m, const int32_t *vox, const int32_t *occ, int64_t n, int32_t *ret);
void oracle_set_occupancy_pos(oracle_map *m, const double *pos, const int32_t *occ, int64_t n, int32_t *ret);
int oracle_check_update(oracle_map *m);
int oracle_update_occupancy(oracle_map *m, int global_map, int64_t *n_insert, int64_t *n_delete);
void oracle_update_esdf(oracle_map *m, oracle_esdf_stats *stats);

void oracle_get_distance_vox(oracle_map *m, const int32_t *vox, int64_t n, double *out);
void oracle_get_distance_pos(oracle_map *m, const double *pos, int64_t n, double *out);
void oracle_get_dist_grad(oracle_map *m, const double *pos, int64_t n, double *dist, double *grad);
void oracle_get_occupancy_vox(oracle_map *m, const int32_t *vox, int64_t n, int32_t *out);
void oracle_get_occupancy_pos(oracle_map *m, const double *pos, int64_t n, int32_t *out);

/* Dense dump (array mode): every output is optional (NULL to skip) and has grid_total_size entries in the
 * reference's linear order x*Ny*Nz + y*Nz + z (src/ESDFMap.cpp:91).
 *   dist   distance_buffer_ (metres; -10000 never observed, +10000 observed/no obstacle)
 *   coc    closest_obstacle_ (3 x int32; -10000 undefined)
 *   occ    Exist(idx) as 0/1
 *   logodds occupancy_buffer_ */
void oracle_dump_dense(oracle_map *m, double *dist, int32_t *coc, uint8_t *occ, double *logodds);
/* num_hit_ / num_miss_ (dense order, array mode). */
void oracle_dump_counts(oracle_map *m, int32_t *num_hit, int32_t *num_miss);
/* Hash dump: entries 1..count-1 of the block store in allocation order; returns count-1. With all
 * outputs NULL it only returns the count. */
int64_t oracle_dump_hash(oracle_map *m, int32_t *vox, double *dist, int32_t *coc, uint8_t *occ);

int oracle_check_consistency(oracle_map *m); /* CheckConsistency(), src/ESDFMap.cpp:856-902 */

/* GetPointCloud(m, vis_lower_bound, vis_upper_bound), src/ESDFMap.cpp:544-582: the points of the message (float xyz,
 * message order), at most cap written; returns the number the reference produced. */
int64_t oracle_get_point_cloud(oracle_map *m, int vis_lower_bound, int vis_upper_bound, float *xyz, int64_t cap);
/* GetSliceMarker(m, slice, id, color, max_dist), src/ESDFMap.cpp:639-699: points (double xyz) and colours (float
 * rgba) of the marker, message order; returns the number the reference produced. */
int64_t oracle_get_slice_marker(oracle_map *m, int slice, double max_dist, double *xyz, float *rgba, int64_t cap);

/* Raycast(start,end,min,max,&out), src/raycast.cpp:56-158; all in voxel units. Writes at most cap voxels
 * (3 doubles each), returns the number the reference produced, or -1 if it threw (>1500 voxels). */
int oracle_raycast(const double start[3], const double end[3], const double minv[3], const double maxv[3],
                   double *out, int cap);

/* One frame of Fiesta::RaycastProcess(0, n, tt) (include/Fiesta.h:194-278), single thread, restated
 * (the header is inseparable from ROS/PCL). points are float xyz in the sensor frame, transform is the
 * row-major 4x4 transform_, origin the raycast_origin_. The per-map stamp arrays set_occ_/set_free_ and
 * the frame counter tot_ live inside the oracle map. */
typedef struct oracle_raycast_params {
  double min_ray_length, max_ray_length;
  double l_cornor[3], r_cornor[3];
} oracle_raycast_params;
void oracle_raycast_frame(oracle_map *m, const float *points, int64_t n, const double transform[16],
                          const double origin[3], const oracle_raycast_params *p);

/* The same frame with -DSIGNED_NEEDED (include/Fiesta.h:216-218,249-251): every SetOccupancy on the map is followed by
 * one on the INVERSE map -- end points as free (0), traversed voxels as occupied (1) -- whose return value then feeds
 * the per-frame de-duplication.  inv must have the same geometry as m. */
void oracle_raycast_frame_signed(oracle_map *m, oracle_map *inv, const float *points, int64_t n,
                                 const double transform[16], const double origin[3], const oracle_raycast_params *p);

/* Fiesta::DepthConversion (include/Fiesta.h:319-382), restated: see depth_filter.inc. */
int64_t oracle_depth_conversion(const uint16_t *cur, const uint16_t *last, int rows, int cols, double fx, double fy,
                                double cx, double cy, int use_filter, const double rel[16], double tolerance,
                                double max_dist, double min_dist, int margin, float *out);

#ifdef __cplusplus
}
#endif
#endif
