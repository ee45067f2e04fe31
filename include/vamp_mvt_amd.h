/*
 * vamp_mvt_amd.h — C ABI of the MI355X-native motion-validation hot path.
 *
 * Drop-in boundary for one path of chingchennn/vamp_mvt: configuration-rake forward kinematics + sphere
 * collision check (+ CAPT point-cloud query).  The reference exposes this path only through its nanobind
 * module `vamp._core` (no C ABI exists there); each entry point below names the reference interface it
 * replaces (file:line under /root/reference/src/impl/vamp/).  INTEGRATION.md shows the binding a maintainer
 * would add on the reference side.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, opaque handles, `int` status (0 = VMV_OK); nothing throws/aborts.
 *   - `d_*` pointers are DEVICE pointers (HBM) on the current HIP device; `*_host` variants take host
 *     buffers and do the H2D/D2H copies themselves.  `stream` is a hipStream_t passed as void* (NULL = default).
 *   - configurations are fp32, row-major [n][dimension], joint values in radians / metres (not normalised).
 *   - validity results are packed bitmasks, little-endian within 64-bit words: bit (i % 64) of word (i / 64)
 *     is 1 iff configuration/edge i is VALID (collision free).  d_bits must hold ceil(n / 64) words.
 *   - an environment is immutable after vmv_env_finalize() and may then be used from any thread/stream.
 *   - there is no CPU fallback: every compute entry point fails with VMV_ERR_NO_DEVICE without a GPU.
 *   - non-finite input is DEFINED: a configuration with a NaN or +-inf joint is INVALID (bit 0), an edge with such an
 *     endpoint is INVALID; nothing is evaluated for it.  (The reference has no rule: its sign-bit predicates read the
 *     sign of a propagated NaN, vector/interface.hh:257-277 — an artefact of the instruction set.)  vmv_fk_batch /
 *     vmv_eefk_batch return NaN spheres / frames for such a configuration.
 *   - bits of the last word at or beyond n are written as 0; the entry points that AND into existing words
 *     (vmv_validate_batch_self) ignore and clear them.
 */
#ifndef VAMP_MVT_AMD_H
#define VAMP_MVT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum
{
    VMV_OK = 0,
    VMV_ERR_INVALID_ARGUMENT = 1,
    VMV_ERR_NO_DEVICE = 2,
    VMV_ERR_HIP = 3,
    VMV_ERR_CAPACITY = 4,      /* environment does not fit the on-chip staging budget */
    VMV_ERR_NOT_FINALIZED = 5,
    VMV_ERR_UNKNOWN_ROBOT = 6,
    VMV_ERR_FINALIZED = 7      /* mutation after finalize */
};

const char *vmv_status_string(int status);
int vmv_abi_version(void);
/* last HIP error text of the calling thread (empty if none) */
const char *vmv_last_error(void);

/* ---- devices ----------------------------------------------------------------------------------------- */
int vmv_device_count(int *count);
int vmv_set_device(int device);
int vmv_get_device(int *device);

/* ---- robots (replaces the per-robot submodule constants, bindings/robot_helper.hh:326-360) ------------ */
int vmv_num_robots(void);
const char *vmv_robot_name(int robot);
int vmv_robot_id(const char *name);           /* -1 if unknown ("panda", "ur5", "fetch", "baxter") */
int vmv_robot_dimension(int robot);           /* Robot::dimension */
int vmv_robot_n_spheres(int robot);           /* Robot::n_spheres */
int vmv_robot_resolution(int robot);          /* Robot::resolution */
int vmv_robot_min_max_radii(int robot, float *min_radius, float *max_radius);
/* lower[d], span[d], descale[d]: Robot::s_a, s_m, d_m (robots/panda.hh:50-75) */
int vmv_robot_bounds(int robot, float *lower, float *span, float *descale);
const char *vmv_robot_joint_name(int robot, int joint);
const char *vmv_robot_end_effector(int robot);

/* ---- point-cloud filters (replaces vamp.filter_pointcloud, bindings/environment.cc:183-239) ------------ */
/* filter_type 0 = "scdf" (collision/filter.hh:175-275; uses min_dist, max_range, cull), 1 = "centervox"
 * (collision/filter_centervox.hh:288-313; uses voxel_size, max_range).  points / out: host pointers, [n][3] fp32; the
 * kept points come back in the reference's order.  out may be NULL (count only); VMV_ERR_CAPACITY if out is too small
 * or where the reference throws "Voxel pool exhausted".  nanoseconds: wall time including transfers (what the
 * reference's binding reports), device_nanoseconds: HIP-event time of the device work alone; both may be NULL. */
int vmv_filter_pointcloud(const float *points_xyz, size_t n, float min_dist, float max_range, float voxel_size,
                          const float *origin3, const float *workspace_min3, const float *workspace_max3, int cull,
                          int filter_type, float *out_xyz, size_t capacity, size_t *n_out, uint64_t *nanoseconds,
                          uint64_t *device_nanoseconds);

/* ---- environment (replaces vamp.Environment, bindings/environment.cc:111-163) -------------------------- */
typedef struct vmv_env vmv_env;

int vmv_env_create(vmv_env **out);
int vmv_env_destroy(vmv_env *env);
/* Environment.add_sphere(Sphere(center, r)) — environment.cc:113-119, collision/shapes.hh:226-239 */
int vmv_env_add_sphere(vmv_env *env, float x, float y, float z, float r);
/* Environment.add_cuboid — environment.cc:120-133.  15 floats: centre xyz | axis_1 xyz | axis_2 xyz |
 * axis_3 xyz | half extents 1..3 (collision/shapes.hh:32-49).  Filed as z-aligned iff axis_3_z == 1. */
int vmv_env_add_cuboid(vmv_env *env, const float *params15);
/* Environment.add_capsule — environment.cc:134-147.  8 floats: x1 y1 z1 | xv yv zv | r | rdv
 * (collision/shapes.hh:128-143).  Filed as z-aligned iff xv == 0 and yv == 0. */
int vmv_env_add_capsule(vmv_env *env, const float *params8);
/* make_heightfield(center, scaling, dimensions, data) + Environment.add_heightfield — collision/factory.hh:363-423,
 * bindings/environment.cc:100,149-151, collision/shapes.hh:250-312.  data: host pointer, row-major [yd][xd] fp32;
 * scale3 as given to make_heightfield (the shape stores the reciprocals).  At most 4 per environment. */
int vmv_env_add_heightfield(vmv_env *env, const float *center3, const float *scale3, size_t xd, size_t yd,
                            const float *data);
int vmv_env_heightfield_count(const vmv_env *env, size_t *count);
/* Environment.attach(Attachment) / detach — bindings/environment.cc:178-181; Attachment(tf) + add_sphere(s)
 * (:241-259), collision/attachments.hh.  tf: 4 x 4 row-major, the attachment's frame relative to the end-effector
 * frame; spheres: [n][4] = x y z r in that frame, n <= 256.  With an attachment (and n > 0) every validate call is
 * Robot::fkcc_attach (planning/validate.hh:43,58): plain fkcc, then the posed spheres against the environment and
 * against the links of the reference's "Attachment vs. <link>" blocks. */
int vmv_env_attach(vmv_env *env, const float *tf_rowmajor_4x4, const float *spheres_xyzr, size_t n);
int vmv_env_detach(vmv_env *env);
/* Environment.add_capt_pointcloud(points, r_min, r_max, r_point) -> build ns — environment.cc:152-163,
 * collision/capt.hh:296-369.  points: host pointer, [n][3] fp32. */
int vmv_env_add_capt_pointcloud(vmv_env *env, const float *points_xyz, size_t n, float r_min, float r_max,
                                float r_point, uint64_t *build_nanoseconds);
/* The same point cloud structure built on the GPU: identical arrays, level-synchronous build
 * (csrc/vmv_capt_gpu.hip).  build_nanoseconds: wall time including the upload of the points and the download of the
 * arrays; device_nanoseconds: HIP-event time of the build on the device alone.  Either may be NULL. */
int vmv_env_add_capt_pointcloud_gpu(vmv_env *env, const float *points_xyz, size_t n, float r_min, float r_max,
                                    float r_point, uint64_t *build_nanoseconds, uint64_t *device_nanoseconds);
/* Environment.add_mvt_pointcloud(points, r_min, r_max, workspace_aabb_min, workspace_aabb_max, r_point) -> build ns
 * — environment.cc:164-177, collision/mvt.hh:147-170 (the fork's Multi-level Voxel Table).  Where the reference
 * throws inside its noexcept constructor (a pool it sized up front runs out: mvt.hh:66-70, 634-648), this returns
 * VMV_ERR_CAPACITY and `*reason` (may be NULL) = 1 voxel capacity, 2 point pool (> 10 % of the voxels occupied),
 * 3 z-table pool (> 50 % of the (x, y) columns occupied), 4 degenerate grid, 5 grid too large for the dense table. */
int vmv_env_add_mvt_pointcloud(vmv_env *env, const float *points_xyz, size_t n, float r_min, float r_max,
                               const float *workspace_min3, const float *workspace_max3, float r_point,
                               uint64_t *build_nanoseconds, int *reason);
/* Sorts every primitive list by min_distance (collision/environment.hh:46-72) and uploads the environment to
 * the current device.  The reference re-sorts on every add and converts per call (robot_helper.hh:266). */
int vmv_env_finalize(vmv_env *env);
/* counts[6]: spheres, capsules, z_capsules, cuboids, z_cuboids, capt point clouds; vmv_env_mvt_count: MVT clouds */
int vmv_env_counts(const vmv_env *env, size_t *counts6);
int vmv_env_mvt_count(const vmv_env *env, size_t *count);
/* MVT cloud `index`: grid_width, per-voxel capacity, occupied voxels, inverse scale factor, global box (6 floats) */
int vmv_env_mvt_info(const vmv_env *env, size_t index, uint32_t *grid_width, uint32_t *capacity, uint32_t *n_voxels,
                     float *inverse_scale_factor, float *global_box6);
/* sorted host copies for inspection: spheres [n][5] (x y z r min_distance), cuboids [n][16], capsules [n][9] */
int vmv_env_get_spheres(const vmv_env *env, float *out, size_t capacity, size_t *n);
int vmv_env_get_cuboids(const vmv_env *env, int z_aligned, float *out, size_t capacity, size_t *n);
int vmv_env_get_capsules(const vmv_env *env, int z_aligned, float *out, size_t capacity, size_t *n);
/* CAPT arrays of point cloud `index` (collision/capt.hh:588-623) — sizes first, then copies (NULL = skip) */
int vmv_env_capt_sizes(const vmv_env *env, size_t index, uint32_t *nlog2, uint32_t *n_aff_vectors);
int vmv_env_capt_arrays(const vmv_env *env, size_t index, float *tests, uint32_t *aff_starts, float *aabbs,
                        float *aff_x, float *aff_y, float *aff_z, float *aabb_top6);

/* ---- batched hot path ---------------------------------------------------------------------------------- */
/* <robot>.fk(q) -> list[Sphere] — robot_helper.hh:234-247, Robot::sphere_fk (robots/panda.hh:116-462).
 * d_out: [n][n_spheres][4] = x y z r. */
int vmv_fk_batch(int robot, const float *d_q, size_t n, float *d_out, void *stream);
/* <robot>.eefk(q) -> 4 x 4 — robot_helper.hh:279-282, Robot::eefk.  d_out: [n][16] row-major frames. */
int vmv_eefk_batch(int robot, const float *d_q, size_t n, float *d_out, void *stream);
/* <robot>.validate(q, env) — robot_helper.hh:255-267 -> validate_motion<Robot, 8, 1>(q, q, env)
 * (planning/validate.hh:70-77) -> Robot::fkcc (robots/panda.hh:5226-10262).  One bit per configuration. */
int vmv_validate_batch(int robot, const vmv_env *env, const float *d_q, size_t n, uint64_t *d_bits, void *stream);
/* The two halves vmv_validate_batch launches back to back, exposed for per-kernel measurement and for callers that
 * pipeline them: `_env` WRITES the validity words (environment half of fkcc), `_self` ANDs the self-collision half
 * into words already written.  vmv_validate_batch == _env then _self on the same stream. */
int vmv_validate_batch_env(int robot, const vmv_env *env, const float *d_q, size_t n, uint64_t *d_bits, void *stream);
int vmv_validate_batch_self(int robot, const float *d_q, size_t n, uint64_t *d_bits, void *stream);
/* validate_motion<Robot, 8, Robot::resolution>(start, goal, env) — planning/validate.hh:24-77, the call every
 * planner makes per edge (rrtc.hh:136-140, prm.hh:59, fcit.hh:238 ...).  One bit per edge.
 * A sequence of kernels on `stream` (rake 0 of every edge, a scan, the remaining rakes of the surviving edges), with 8
 * bytes of internal device scratch per edge kept per (device, stream); batches beyond 2^20 edges run slice by slice.
 * Rake counts are 32-bit: the rakes of one slice must number below 2^32 (edges averaging 4,096 rakes = 512 rad at
 * resolution 64 — far beyond any joint range; the reference's walk of such an edge would not end either). */
int vmv_validate_motion_batch(int robot, const vmv_env *env, const float *d_start, const float *d_goal, size_t n,
                              uint64_t *d_bits, void *stream);

/* <robot>.debug(q, env) — robot_helper.hh:249-253 -> Robot::fkcc_debug: per fine sphere the environment objects it
 * collides with (sphere_environment_get_collisions, collision/validity.hh:161-256: the five sorted primitive lists with
 * their early break, then the heightfields; no point clouds), and the fine sphere pairs of the self-collision groups
 * that overlap (all pairs, no bounding gates).  Host buffers.  env_words: [n][n_spheres][9]; words 0..7 hold 32
 * sorted-list positions each, the lists back to back (vmv_env_report_layout gives each list's first word), word 8 the
 * heightfields.  pair_words: [n][ceil(n_pairs / 32)], bit p = pair p of vmv_robot_self_pairs.  Environments whose
 * lists need more than 8 words: VMV_ERR_CAPACITY. */
int vmv_contacts_batch_host(int robot, const vmv_env *env, const float *q, size_t n, uint32_t *env_words,
                            uint32_t *pair_words);
/* first word of each sorted list in the report: spheres, capsules, z_capsules, cuboids, z_cuboids */
int vmv_env_report_layout(const vmv_env *env, uint32_t *first_word5);
/* the fine pairs the report covers: n_pairs, and (a, b) sphere indices into pairs2 (may be NULL) */
int vmv_robot_self_pairs(int robot, size_t *n_pairs, uint16_t *pairs2);

/* sphere_environment_in_collision(environment, x, y, z, r) — collision/validity.hh:47-158, the predicate every fkcc
 * check is made of (and CAPT::collides / MVT::collides behind it, collision/capt.hh:374-415), for a batch of free
 * spheres: spheres [n][4] = x y z r, hits[i] = 1 iff sphere i collides with the environment.  Each sphere is its own
 * replicated rake (lanes independent). */
int vmv_spheres_in_collision_batch(const vmv_env *env, const float *d_spheres, size_t n, uint8_t *d_hits, void *stream);
int vmv_spheres_in_collision_batch_host(const vmv_env *env, const float *spheres, size_t n, uint8_t *hits);

/* <robot>.filter_self_from_pointcloud(pc, point_radius, configuration, environment) — bindings/robot_helper.hh:284-322:
 * keeps the points whose sphere (x, y, z, point_radius) neither overlaps a collision sphere of the robot at `q`
 * (sphere_sphere_sql2 < 0) nor collides with the environment; order preserved.  Host buffers; out may be NULL (count only);
 * VMV_ERR_CAPACITY if out is too small. */
int vmv_filter_self_from_pointcloud(int robot, const vmv_env *env, const float *q, const float *points_xyz, size_t n,
                                    float point_radius, float *out_xyz, size_t capacity, size_t *n_out);

/* host-buffer variants (copies included; the PCIe-inclusive path) */
int vmv_fk_batch_host(int robot, const float *q, size_t n, float *out);
int vmv_eefk_batch_host(int robot, const float *q, size_t n, float *out);
int vmv_validate_batch_host(int robot, const vmv_env *env, const float *q, size_t n, uint64_t *bits);
int vmv_validate_motion_batch_host(int robot, const vmv_env *env, const float *start, const float *goal, size_t n,
                                   uint64_t *bits);
/* The host-buffer variants stage through a per-thread device arena that is reused between calls (requests above 64 MiB
 * are not kept), and vmv_validate_motion_batch keeps 8 bytes of device scratch per edge per (device, stream) for its task
 * lists.  Frees the calling thread's arena and every stream's scratch (waits for edge batches in flight); optional — none
 * of this memory is touched at thread or process exit. */
int vmv_release_staging(void);

/* ---- multi-GPU: one process per GPU, every unit independent given the read-only environment ------- */
/* Contiguous shard [*lo, *hi) of an n-unit batch (configurations or edges) for `rank` of `world`: every boundary except
 * the last is a multiple of 64, so a shard owns whole validity words and - an edge being a whole sequence of 8-lane
 * rakes - no rake straddles two GPUs.  Each rank finalizes its own copy of the environment on its device, calls
 * vmv_validate_batch / vmv_validate_motion_batch on its shard (d_q + lo * dimension, hi - lo units) and the ranks
 * all-gather the packed words (ncclAllGather of vmv_shard_words(n, world) uint64 per rank; ranks whose shard is
 * shorter pad with zero words).  That all-gather is the path's only exchange step.  Python: vamp_mvt_amd.sharding. */
int vmv_shard_range(size_t n, int rank, int world, size_t *lo, size_t *hi);
/* uint64 words each rank contributes to the all-gather: ceil(ceil(n / 64) / world) */
size_t vmv_shard_words(size_t n, int world);

/* ---- sampler --------------------------------------------------------------------------------------------- */
/* <robot>.halton() / RNG.next() — random/halton.hh:75-108 with the default prime bases, generated on the device:
 * fills d_q[n][dimension] with samples skip+1 .. skip+n of the reference's sequence (bit-exact: the sequence's
 * n/d are exact small integers, so element j of sample i is float(radical_inverse_numerator) / float(b_j^k),
 * scaled by Robot::scale_configuration).  Valid while skip + n <= 1,000,000 (the reference re-seeds after that). */
int vmv_halton_configs(int robot, uint64_t skip, float *d_q, size_t n, void *stream);

/* ---- measurement support (bench.py) ---------------------------------------------------------------------- */
/* Runs vmv_validate_batch `iters` times on `stream` between two HIP events recorded on that same stream and
 * returns the average kernel time in milliseconds. */
int vmv_time_validate_batch(int robot, const vmv_env *env, const float *d_q, size_t n, uint64_t *d_bits, int iters,
                            void *stream, float *avg_ms);
/* fills d_q[n][dimension] with uniform configurations inside the joint bounds (splitmix64 counter RNG) */
int vmv_fill_uniform_configs(int robot, float *d_q, size_t n, uint64_t seed, void *stream);
/* name of the dominant kernel symbol for a robot (to match rocprofv3 --kernel-trace rows) */
const char *vmv_kernel_name(int robot, const char *entry_point);

#ifdef __cplusplus
}
#endif
#endif
