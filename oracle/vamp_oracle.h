/*
 * vamp_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * A plain-C restatement of the reference's motion-validation hot path
 * (chingchennn/vamp_mvt @ /root/reference; citations are file:line under
 * src/impl/vamp/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load this library, and only as the checker — never
 * as the thing measured or shipped.  The product path (vamp_mvt_amd/) never
 * links, imports or calls anything in this directory.
 *
 * Parity pins: the FK programs are bit-exact against
 * the reference's generated robots/<robot>.hh fkcc text evaluated by
 * tools/ref_fk_eval.py; sin/cos/l2_norm/Halton are checked against the
 * reference's own vector.hh / halton.hh compiled in place (oracle/_ref); whole
 * pipeline known answers come from SURVEY.md §8c (sphere cage, Halton counts).
 * Not pinned by any reference artefact: cuboid/capsule tests and the CAPT
 * build/query beyond the survey's recorded statistics ("parity unpinned").
 *
 * Arithmetic contract: IEEE fp32, one rounding per written operation
 * (compile with -ffp-contract=off, no fast-math).  The only deliberate
 * deviation from the reference: `max_extent` in the sorted early-break uses a
 * correctly rounded sqrt instead of AVX `v * rsqrt_ps(v)` (vector/avx.hh:411-415),
 * whose low bits are vendor-defined and therefore not reproducible.
 */
#ifndef VAMP_ORACLE_H
#define VAMP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VO_RAKE 8 /* vamp::FloatVectorWidth on AVX2 (vector.hh:24) */

typedef struct vo_env vo_env;

/* test hook for the one documented deviation (max_extent's sqrt, collision/validity.hh:59): fn(in, out, n) replaces
 * sqrtf for the sorted early-break; NULL restores the default. */
typedef void (*vo_sqrt_fn)(const float *in, float *out, size_t n);
void vo_set_max_extent_sqrt(vo_sqrt_fn fn);

/* collision/environment.hh:16-88 + bindings/environment.cc:111-163 */
vo_env *vo_env_create(void);
void vo_env_destroy(vo_env *e);
void vo_env_add_sphere(vo_env *e, float x, float y, float z, float r);
/* 15 floats: x y z | axis_1 xyz | axis_2 xyz | axis_3 xyz | half extents 1..3 (collision/shapes.hh:32-49) */
void vo_env_add_cuboid(vo_env *e, const float *p15);
/* 8 floats: x1 y1 z1 | xv yv zv | r | rdv (collision/shapes.hh:128-143) */
void vo_env_add_capsule(vo_env *e, const float *p8);
/* make_heightfield + add_heightfield (collision/factory.hh:363-423, bindings/environment.cc:100,149-151): row-major
 * data[yd][xd]; `scale` as given to make_heightfield (the shape stores the reciprocals).  Returns 0 on success. */
int vo_env_add_heightfield(vo_env *e, const float center[3], const float scale[3], size_t xd, size_t yd, const float *data);
/* Environment.attach(Attachment(tf) + spheres) / detach (bindings/environment.cc:178-181, :241-259;
 * collision/attachments.hh).  tf: 4x4 row-major, relative to the end-effector frame; spheres: [n][4] = x y z r in
 * that frame (at most 256).  With an attachment the rake check is Robot::fkcc_attach (planning/validate.hh:43,58). */
void vo_env_attach(vo_env *e, const float tf_rowmajor_4x4[16], const float *spheres_xyzr, size_t n);
void vo_env_detach(vo_env *e);
/* collision/capt.hh:296-369; returns 0 on success */
int vo_env_add_capt(vo_env *e, const float *points_xyz, size_t n, float r_min, float r_max, float r_point);

/* collision/mvt.hh:147-170 (Multi-level Voxel Table).  Returns 0 on success; a positive code names the condition under
 * which the reference throws inside its noexcept constructor (std::terminate): 1 voxel capacity exceeded,
 * 2 point coordinate pool exhausted, 3 voxel index (z-table) pool exhausted, 4 degenerate grid. */
int vo_env_add_mvt(vo_env *e, const float *points_xyz, size_t n, float r_min, float r_max, const float *ws_min3,
                   const float *ws_max3, float r_point);
/* MVT::collides (scalar, mvt.hh:204-279) and collides_simd (mvt.hh:282-403) */
int vo_mvt_collides(const vo_env *e, size_t index, const float c[3], float r);
int vo_mvt_collides_simd(const vo_env *e, size_t index, const float *cx, const float *cy, const float *cz,
                         const float *r, int lanes);
typedef struct vo_mvt_view
{
    uint32_t grid_width, capacity, n_voxels, n_y_tables, n_z_tables;
    float inverse_scale_factor;
    float global_min[3], global_max[3];
    const uint32_t *x_table, *y_tables, *z_tables, *voxel_count;
    const float *voxel_bbox, *px, *py, *pz;
} vo_mvt_view;
int vo_env_mvt_view(const vo_env *e, size_t index, vo_mvt_view *out);

/* counts after sorting/splitting: spheres, capsules, z_capsules, cuboids, z_cuboids, capts */
void vo_env_counts(const vo_env *e, size_t counts[6]);
/* copy the sorted primitive tables (for cross-checking the device upload) */
size_t vo_env_get_spheres(const vo_env *e, float *out5 /* x y z r min_distance */);
size_t vo_env_get_cuboids(const vo_env *e, int z_aligned, float *out16 /* 15 params + min_distance */);
size_t vo_env_get_capsules(const vo_env *e, int z_aligned, float *out9 /* 8 params + min_distance */);

/* CAPT array access (collision/capt.hh:588-623) */
typedef struct vo_capt_view
{
    uint32_t nlog2;
    uint32_t n_tests;       /* 2^nlog2 - 1 */
    uint32_t n_leaves;      /* 2^nlog2 */
    uint32_t n_aff_vectors; /* affordance vectors of VO_RAKE points each */
    const float *tests;
    const uint32_t *aff_starts; /* n_leaves + 1 */
    const float *aabbs;         /* n_leaves * 6: lower xyz, upper xyz */
    const float *aff_x, *aff_y, *aff_z; /* n_aff_vectors * VO_RAKE */
    float aabb_top[6];
    float r_min, r_max, r_point;
} vo_capt_view;
int vo_env_capt_view(const vo_env *e, size_t index, vo_capt_view *out);
/* CAPT::collides (scalar, capt.hh:374-415) and one lane of collides_simd (capt.hh:428-512) */
int vo_capt_collides(const vo_env *e, size_t index, const float c[3], float r);
int vo_capt_collides_simd(const vo_env *e, size_t index, const float *cx, const float *cy, const float *cz,
                          const float *r, int lanes);

/* sphere_environment_in_collision (collision/validity.hh:47-158) for one sphere (a rake of one lane); 1 = collides */
int vo_sphere_environment_in_collision(const vo_env *e, const float c[3], float r);

/* filter_pointcloud(pc, min_dist, max_range, origin, workspace_min, workspace_max, cull) — collision/filter.hh:175-275
 * ("scdf", the space-filling-curve filter); out_xyz has room for n points; returns how many were kept. */
size_t vo_filter_scdf(const float *pc, size_t n, float min_dist, float max_range, const float origin[3],
                      const float ws_min[3], const float ws_max[3], int cull, float *out_xyz);
/* filter_pointcloud_centervox — collision/filter_centervox.hh:16-313; (size_t) -1 where the reference throws */
size_t vo_filter_centervox(const float *pc, size_t n, float voxel_size, float max_range, const float origin[3],
                           const float ws_min[3], const float ws_max[3], float *out_xyz);

/* robots */
int vo_robot_id(const char *name); /* -1 if unknown */
size_t vo_robot_dimension(int robot);
size_t vo_robot_n_spheres(int robot);
size_t vo_robot_n_total_spheres(int robot);
size_t vo_robot_resolution(int robot);
void vo_robot_bounds(int robot, float *lower, float *span);

/* arithmetic contract probes (vector/avx.hh:455-548, vector/interface.hh:447-458, :397-420) */
float vo_sin(float x);
float vo_cos(float x);
float vo_l2_norm(const float *v, size_t dim);

/* Robot::sphere_fk<1> (robots/panda.hh:116-462): out[n_spheres][4] = x y z r */
void vo_fk(int robot, const float *q, float *out);
/* Robot::eefk (bindings/robot_helper.hh:279-282): end-effector frame, 4x4 row-major */
void vo_eefk(int robot, const float *q, float *out16);
/* FK of fine + bounding spheres as used by fkcc: out[n_total][4] */
void vo_fk_all(int robot, const float *q, float *out);

/* Robot::fkcc<8> (robots/panda.hh:5226-10262) on a rake given as block[dim][VO_RAKE]; 1 = valid */
int vo_fkcc_rake(int robot, const vo_env *e, const float *block);
/* bindings/robot_helper.hh:255-267 (check_bounds optional) */
int vo_validate(int robot, const vo_env *e, const float *q, int check_bounds);
/* planning/validate.hh:70-77 with the robot's resolution */
int vo_validate_motion(int robot, const vo_env *e, const float *start, const float *goal);
/* batch helpers: out[i] = 1 valid / 0 invalid */
void vo_validate_batch(int robot, const vo_env *e, const float *q, size_t n, uint8_t *out);
void vo_validate_motion_batch(int robot, const vo_env *e, const float *a, const float *b, size_t n, uint8_t *out);
/* same, spread over `threads` pthreads (cpu_baseline leg of bench.py) */
void vo_validate_batch_mt(int robot, const vo_env *e, const float *q, size_t n, uint8_t *out, int threads);
void vo_validate_motion_batch_mt(int robot, const vo_env *e, const float *a, const float *b, size_t n, uint8_t *out,
                                 int threads);
/* <robot>.filter_self_from_pointcloud (bindings/robot_helper.hh:284-322): out[m][3], returns m */
size_t vo_filter_self_from_pointcloud(int robot, const vo_env *e, const float *q, const float *points_xyz, size_t n,
                                      float point_radius, float *out_xyz);
/* AVX2 build of vo_validate_batch: one vector lane per configuration, 8 distinct configurations per rake (the shape of
 * the reference's vector/avx.hh), bit-identical answers.  Primitive environments only; -1 otherwise or without AVX2. */
int vo_has_avx2(void);
int vo_validate_batch_avx2(int robot, const vo_env *e, const float *q, size_t n, uint8_t *out);
int vo_validate_batch_avx2_mt(int robot, const vo_env *e, const float *q, size_t n, uint8_t *out, int threads);

#ifdef __cplusplus
}
#endif
#endif
