/*
 * vamp_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see vamp_oracle.h).
 *
 * Restates, function by function, the reference's hot path.  `file:line`
 * citations are relative to /root/reference/src/impl/vamp/.
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 */
#include "vamp_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* L0: SIMD value-type semantics, one lane at a time                          */
/* ------------------------------------------------------------------------- */

static inline uint32_t f2u(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
static inline float u2f(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}
/* test_zero() is "no lane has its sign bit set" (vector/interface.hh:257-277, avx.hh:385-389);
 * every collision predicate is `not expr.test_zero()`, i.e. some lane's sign bit is set. */
static inline int signbit_set(float f) { return (int) (f2u(f) >> 31); }
/* _mm256_max_ps(a, b): a > b ? a : b (returns b when either is NaN) (avx.hh:435-439) */
static inline float x86_max(float a, float b) { return a > b ? a : b; }
/* _mm256_min_ps(a, b): a < b ? a : b */
static inline float x86_min(float a, float b) { return a < b ? a : b; }
/* clamp = min(max(v, lower), upper) (avx.hh:429-433) */
static inline float vclamp(float v, float lo, float hi) { return x86_min(x86_max(v, lo), hi); }

/* vector/avx.hh:455-548 — cephes/sse_mathfun sine, all steps as separate fp32 operations */
static float vo_sinf(float x)
{
    uint32_t sign_bit = f2u(x) & 0x80000000u;
    x = u2f(f2u(x) & 0x7fffffffu);
    float y = x * 1.27323954473516f; /* 4/pi */
    /* _mm256_cvtps_epi32: round to nearest even; out of range -> 0x80000000 */
    int32_t j;
    if (!(y < 2147483648.0f))
        j = INT32_MIN;
    else
        j = (int32_t) lrintf(y);
    j = (int32_t) (((uint32_t) j + 1u) & ~1u);
    y = (float) j;
    uint32_t swap_sign = ((uint32_t) j & 4u) << 29;
    int poly_sin = (((uint32_t) j & 2u) == 0u);
    sign_bit ^= swap_sign;
    /* extended precision modular arithmetic: x = ((x - y*DP1) - y*DP2) - y*DP3 */
    float xmm1 = y * -0.78515625f;
    float xmm2 = y * -2.4187564849853515625e-4f;
    float xmm3 = y * -3.77489497744594108e-8f;
    x = x + xmm1;
    x = x + xmm2;
    x = x + xmm3;
    /* first polynomial (0 <= x <= pi/4) */
    float z = x * x;
    float yc = 2.443315711809948E-005f;
    yc = yc * z;
    yc = yc + -1.388731625493765E-003f;
    yc = yc * z;
    yc = yc + 4.166664568298827E-002f;
    yc = yc * z;
    yc = yc * z;
    float tmp = z * 0.5f;
    yc = yc - tmp;
    yc = yc + 1.0f;
    /* second polynomial */
    float y2 = -1.9515295891E-4f;
    y2 = y2 * z;
    y2 = y2 + 8.3321608736E-3f;
    y2 = y2 * z;
    y2 = y2 + -1.6666654611E-1f;
    y2 = y2 * z;
    y2 = y2 * x;
    y2 = y2 + x;
    /* select: and/andnot with the mask, then add (adds +0.0f to the selected value) */
    float sel = poly_sin ? (0.0f + y2) : (yc + 0.0f);
    return u2f(f2u(sel) ^ sign_bit);
}

/* vector/interface.hh:447-458 — cos through the shifted sine */
static float vo_cosf(float x)
{
    const float PI = 3.14159265359f;
    const float v_sq = x + (float) (PI / 2.);
    const float sub = (v_sq >= PI) ? (float) (2 * PI) : 0.0f;
    const float vsq_sq = v_sq - sub;
    return vo_sinf(vsq_sq);
}

float vo_sin(float x) { return vo_sinf(x); }
float vo_cos(float x) { return vo_cosf(x); }

/* hsum (avx.hh:441-452) over the 8 lanes of one row: ((v4+v0)+(v6+v2)) + ((v5+v1)+(v7+v3)) */
static float hsum8(const float *v)
{
    const float s0 = v[4] + v[0], s1 = v[5] + v[1], s2 = v[6] + v[2], s3 = v[7] + v[3];
    const float a = s0 + s2, b = s1 + s3;
    return a + b;
}

/* l2_norm (interface.hh:397-410): rows are added first (unpack::sum_), then hsum, then exact sqrt */
float vo_l2_norm(const float *v, size_t dim)
{
    float sq[16] = {0};
    for (size_t i = 0; i < dim && i < 16; ++i) sq[i] = v[i] * v[i];
    float row[8];
    if (dim > 8)
        for (int k = 0; k < 8; ++k) row[k] = sq[k] + sq[8 + k];
    else
        memcpy(row, sq, sizeof(row));
    return sqrtf(hsum8(row));
}

/* ------------------------------------------------------------------------- */
/* L1a: environment                                                           */
/* ------------------------------------------------------------------------- */

typedef struct
{
    float x, y, z, r, min_distance;
} vo_sphere;
typedef struct
{
    float p[15];
    float min_distance;
} vo_cuboid;
typedef struct
{
    float x1, y1, z1, xv, yv, zv, r, rdv, min_distance;
} vo_capsule;

typedef struct
{
    uint32_t nlog2, n_tests, n_leaves, n_aff;
    float *tests;
    uint32_t *aff_starts;
    float *aabbs;
    float *aff[3];
    float aabb_top[6];
    float r_min, r_max, r_point;
} vo_capt;

/* collision/mvt.hh: the three-level table (x -> y tables -> z tables -> voxel index) flattened to index arrays */
#define VO_MVT_INVALID 0xffffffffu
typedef struct
{
    float r_min, r_max, r_point;
    float ws_min[3], ws_max[3], gmin[3], gmax[3];
    float inv_scale;
    uint32_t grid_width, cap;
    uint32_t *x_table;  /* [grid_width] */
    uint32_t *y_tables; /* [n_y][grid_width] */
    uint32_t *z_tables; /* [n_z][grid_width] */
    uint32_t n_y, n_z, n_vox;
    float *bbox; /* [n_vox][6] */
    uint32_t *count;
    float *px, *py, *pz; /* [n_vox][cap], +inf padded (mvt.hh:650) */
} vo_mvt;

/* collision/shapes.hh:250-312 (xs, ys, zs are the reciprocal scales, factory.hh:365-386) */
typedef struct
{
    float x, y, z, xs, ys, zs;
    size_t xd, yd, xd2, yd2;
    float *data;
} vo_heightfield;

struct vo_env
{
    vo_sphere *spheres;
    size_t n_spheres;
    vo_capsule *capsules;
    size_t n_capsules;
    vo_capsule *z_capsules;
    size_t n_z_capsules;
    vo_cuboid *cuboids;
    size_t n_cuboids;
    vo_cuboid *z_cuboids;
    size_t n_z_cuboids;
    vo_capt *capts;
    size_t n_capts;
    vo_mvt *mvts;
    size_t n_mvts;
    vo_heightfield *heightfields;
    size_t n_heightfields;
    /* collision/attachments.hh: relative frame (row-major 3x4: R | t) + spheres (x y z r) in that frame */
    int attached;
    float attach_tf[12];
    float *attach_spheres;
    size_t n_attach_spheres;
};

vo_env *vo_env_create(void) { return (vo_env *) calloc(1, sizeof(vo_env)); }

static void capt_free(vo_capt *c)
{
    free(c->tests);
    free(c->aff_starts);
    free(c->aabbs);
    free(c->aff[0]);
    free(c->aff[1]);
    free(c->aff[2]);
}

static void mvt_free(vo_mvt *m)
{
    free(m->x_table);
    free(m->y_tables);
    free(m->z_tables);
    free(m->bbox);
    free(m->count);
    free(m->px);
    free(m->py);
    free(m->pz);
}

void vo_env_destroy(vo_env *e)
{
    if (!e) return;
    free(e->spheres);
    free(e->capsules);
    free(e->z_capsules);
    free(e->cuboids);
    free(e->z_cuboids);
    for (size_t i = 0; i < e->n_capts; ++i) capt_free(&e->capts[i]);
    free(e->capts);
    for (size_t i = 0; i < e->n_mvts; ++i) mvt_free(&e->mvts[i]);
    free(e->mvts);
    for (size_t i = 0; i < e->n_heightfields; ++i) free(e->heightfields[i].data);
    free(e->heightfields);
    free(e->attach_spheres);
    free(e);
}

static float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return (ax * bx) + (ay * by) + (az * bz); /* collision/math.hh:16-26 */
}
/* scalar clamp<float> (collision/math.hh:47-51): max(min(v, upper), lower) */
static float sclamp(float v, float lo, float hi) { return fmaxf(fminf(v, hi), lo); }

#define SORT_BY_MIN_DISTANCE(T, name)                                     \
    static int name(const void *a, const void *b)                         \
    {                                                                     \
        const float da = ((const T *) a)->min_distance;                   \
        const float db = ((const T *) b)->min_distance;                   \
        return (da < db) ? -1 : (da > db) ? 1 : 0;                        \
    }
SORT_BY_MIN_DISTANCE(vo_sphere, cmp_sphere)
SORT_BY_MIN_DISTANCE(vo_cuboid, cmp_cuboid)
SORT_BY_MIN_DISTANCE(vo_capsule, cmp_capsule)

/* collision/shapes.hh:236-239 + environment.cc:113-119 (push_back; sort()) */
void vo_env_add_sphere(vo_env *e, float x, float y, float z, float r)
{
    e->spheres = (vo_sphere *) realloc(e->spheres, (e->n_spheres + 1) * sizeof(vo_sphere));
    vo_sphere s = {x, y, z, r, sqrtf(x * x + y * y + z * z) - r};
    e->spheres[e->n_spheres++] = s;
    qsort(e->spheres, e->n_spheres, sizeof(vo_sphere), cmp_sphere);
}

/* collision/shapes.hh:52-67 */
static float cuboid_min_distance(const float *p)
{
    const float x = p[0], y = p[1], z = p[2];
    const float d1 = dot3(-x, -y, -z, p[3], p[4], p[5]);
    const float d2 = dot3(-x, -y, -z, p[6], p[7], p[8]);
    const float d3 = dot3(-x, -y, -z, p[9], p[10], p[11]);
    const float v1 = sclamp(d1, -p[12], p[12]);
    const float v2 = sclamp(d2, -p[13], p[13]);
    const float v3 = sclamp(d3, -p[14], p[14]);
    const float xn = x + p[3] * v1 + p[6] * v2 + p[9] * v3;
    const float yn = y + p[4] * v1 + p[7] * v2 + p[10] * v3;
    const float zn = z + p[5] * v1 + p[8] * v2 + p[11] * v3;
    return sqrtf(xn * xn + yn * yn + zn * zn);
}

/* environment.cc:120-133: z-aligned iff axis_3_z == 1 */
void vo_env_add_cuboid(vo_env *e, const float *p)
{
    vo_cuboid c;
    memcpy(c.p, p, sizeof(c.p));
    c.min_distance = cuboid_min_distance(p);
    if (p[11] == 1.0f)
    {
        e->z_cuboids = (vo_cuboid *) realloc(e->z_cuboids, (e->n_z_cuboids + 1) * sizeof(vo_cuboid));
        e->z_cuboids[e->n_z_cuboids++] = c;
        qsort(e->z_cuboids, e->n_z_cuboids, sizeof(vo_cuboid), cmp_cuboid);
    }
    else
    {
        e->cuboids = (vo_cuboid *) realloc(e->cuboids, (e->n_cuboids + 1) * sizeof(vo_cuboid));
        e->cuboids[e->n_cuboids++] = c;
        qsort(e->cuboids, e->n_cuboids, sizeof(vo_cuboid), cmp_cuboid);
    }
}

/* collision/shapes.hh:165-189 */
static float capsule_min_distance(const vo_capsule *c)
{
    const float dot = sclamp(dot3(-c->x1, -c->y1, -c->z1, c->xv, c->yv, c->zv) * c->rdv, 0.F, 1.F);
    const float xp = c->x1 + c->xv * dot, yp = c->y1 + c->yv * dot, zp = c->z1 + c->zv * dot;
    float xo = -xp, yo = -yp, zo = -zp;
    const float ol = sqrtf(dot3(xo, yo, zo, xo, yo, zo));
    xo = xo / ol;
    yo = yo / ol;
    zo = zo / ol;
    const float ro = sclamp(ol, 0.F, c->r);
    const float xn = xp + ro * xo, yn = yp + ro * yo, zn = zp + ro * zo;
    return sqrtf(xn * xn + yn * yn + zn * zn);
}

/* environment.cc:134-147: z-aligned iff xv == 0 and yv == 0 */
void vo_env_add_capsule(vo_env *e, const float *p)
{
    vo_capsule c = {p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], 0.F};
    c.min_distance = capsule_min_distance(&c);
    if (c.xv == 0.F && c.yv == 0.F)
    {
        e->z_capsules = (vo_capsule *) realloc(e->z_capsules, (e->n_z_capsules + 1) * sizeof(vo_capsule));
        e->z_capsules[e->n_z_capsules++] = c;
        qsort(e->z_capsules, e->n_z_capsules, sizeof(vo_capsule), cmp_capsule);
    }
    else
    {
        e->capsules = (vo_capsule *) realloc(e->capsules, (e->n_capsules + 1) * sizeof(vo_capsule));
        e->capsules[e->n_capsules++] = c;
        qsort(e->capsules, e->n_capsules, sizeof(vo_capsule), cmp_capsule);
    }
}

/* Environment.attach / detach (bindings/environment.cc:178-181); Attachment(tf) + add_spheres (:241-259) */
void vo_env_attach(vo_env *e, const float tf_rowmajor_4x4[16], const float *spheres_xyzr, size_t n)
{
    e->attached = 1;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) e->attach_tf[4 * i + j] = tf_rowmajor_4x4[4 * i + j];
    free(e->attach_spheres);
    e->attach_spheres = (float *) malloc((n ? n : 1) * 4 * sizeof(float));
    memcpy(e->attach_spheres, spheres_xyzr, n * 4 * sizeof(float));
    e->n_attach_spheres = n;
}
void vo_env_detach(vo_env *e) { e->attached = 0; }

/* bindings/environment.cc:149-151 + factory.hh:365-386: scale -> reciprocal; no sorting, min_distance unused */
int vo_env_add_heightfield(vo_env *e, const float center[3], const float scale[3], size_t xd, size_t yd, const float *data)
{
    if (!xd || !yd) return -1;
    vo_heightfield h = {center[0], center[1], center[2], 1.F / scale[0], 1.F / scale[1], 1.F / scale[2],
                        xd,        yd,        xd / 2,    yd / 2,         NULL};
    h.data = (float *) malloc(xd * yd * sizeof(float));
    memcpy(h.data, data, xd * yd * sizeof(float));
    e->heightfields = (vo_heightfield *) realloc(e->heightfields, (e->n_heightfields + 1) * sizeof(vo_heightfield));
    e->heightfields[e->n_heightfields++] = h;
    return 0;
}

void vo_env_counts(const vo_env *e, size_t counts[6])
{
    counts[0] = e->n_spheres;
    counts[1] = e->n_capsules;
    counts[2] = e->n_z_capsules;
    counts[3] = e->n_cuboids;
    counts[4] = e->n_z_cuboids;
    counts[5] = e->n_capts;
}

size_t vo_env_get_spheres(const vo_env *e, float *out)
{
    for (size_t i = 0; i < e->n_spheres; ++i) memcpy(out + 5 * i, &e->spheres[i], 5 * sizeof(float));
    return e->n_spheres;
}
size_t vo_env_get_cuboids(const vo_env *e, int z, float *out)
{
    const vo_cuboid *c = z ? e->z_cuboids : e->cuboids;
    const size_t n = z ? e->n_z_cuboids : e->n_cuboids;
    for (size_t i = 0; i < n; ++i) memcpy(out + 16 * i, &c[i], 16 * sizeof(float));
    return n;
}
size_t vo_env_get_capsules(const vo_env *e, int z, float *out)
{
    const vo_capsule *c = z ? e->z_capsules : e->capsules;
    const size_t n = z ? e->n_z_capsules : e->n_capsules;
    for (size_t i = 0; i < n; ++i) memcpy(out + 9 * i, &c[i], 9 * sizeof(float));
    return n;
}

/* ------------------------------------------------------------------------- */
/* CAPT build (collision/capt.hh:106-369), bug-for-bug                        */
/* ------------------------------------------------------------------------- */

typedef struct
{
    float lower[3], upper[3];
} vo_volume;

/* capt.hh:30-37 */
static void vol_extend(vo_volume *v, const float *p)
{
    for (int k = 0; k < 3; ++k)
    {
        v->lower[k] = fminf(v->lower[k], p[k]);
        v->upper[k] = fmaxf(v->upper[k], p[k]);
    }
}
/* capt.hh:39-46 */
static int vol_contained_by_internal_ball(const vo_volume *v, const float *p, float r)
{
    const float d0 = fmaxf(p[0] - v->lower[0], v->upper[0] - p[0]);
    const float d1 = fmaxf(p[1] - v->lower[1], v->upper[1] - p[1]);
    const float d2 = fmaxf(p[2] - v->lower[2], v->upper[2] - p[2]);
    return (d0 * d0 + d1 * d1 + d2 * d2) <= r;
}
/* std::clamp(v, lo, hi) */
static float stdclamp(float v, float lo, float hi) { return (v < lo) ? lo : (hi < v) ? hi : v; }
/* capt.hh:48-55 */
static float vol_distsq_to(const vo_volume *v, const float *p)
{
    const float d0 = p[0] - stdclamp(p[0], v->lower[0], v->upper[0]);
    const float d1 = p[1] - stdclamp(p[1], v->lower[1], v->upper[1]);
    const float d2 = p[2] - stdclamp(p[2], v->lower[2], v->upper[2]);
    return d0 * d0 + d1 * d1 + d2 * d2;
}

typedef struct
{
    const float *points; /* [n][3] padded */
    uint32_t *argsort;
    vo_capt *c;
    float max_aff_l2, min_aff_l2;
    /* growing outputs */
    size_t aff_cap, n_leaves_done;
} capt_builder;

static const float *g_sort_points;
static int g_sort_k;
static int cmp_argsort(const void *a, const void *b)
{
    const float fa = g_sort_points[3 * (size_t) (*(const uint32_t *) a) + g_sort_k];
    const float fb = g_sort_points[3 * (size_t) (*(const uint32_t *) b) + g_sort_k];
    if (fa < fb) return -1;
    if (fb < fa) return 1;
    /* the reference's pdqsort_branchless leaves the order of equal keys unspecified;
     * index order is used here (fixtures avoid duplicate coordinates) */
    const uint32_t ia = *(const uint32_t *) a, ib = *(const uint32_t *) b;
    return (ia < ib) ? -1 : (ia > ib);
}

/* capt.hh:106-123 */
static float median_partition(capt_builder *b, uint32_t begin, uint32_t end, int k)
{
    g_sort_points = b->points;
    g_sort_k = k;
    qsort(b->argsort + begin, end - begin, sizeof(uint32_t), cmp_argsort);
    const uint32_t middle = begin + (end - begin) / 2;
    const float lo = b->points[3 * (size_t) b->argsort[middle - 1] + k];
    const float hi = b->points[3 * (size_t) b->argsort[middle] + k];
    return (float) ((double) (lo + hi) / 2.0);
}

static void aff_push(capt_builder *b, const float *xs, const float *ys, const float *zs)
{
    vo_capt *c = b->c;
    if (c->n_aff == b->aff_cap)
    {
        b->aff_cap = b->aff_cap ? b->aff_cap * 2 : 1024;
        for (int k = 0; k < 3; ++k) c->aff[k] = (float *) realloc(c->aff[k], b->aff_cap * VO_RAKE * sizeof(float));
    }
    memcpy(c->aff[0] + (size_t) c->n_aff * VO_RAKE, xs, VO_RAKE * sizeof(float));
    memcpy(c->aff[1] + (size_t) c->n_aff * VO_RAKE, ys, VO_RAKE * sizeof(float));
    memcpy(c->aff[2] + (size_t) c->n_aff * VO_RAKE, zs, VO_RAKE * sizeof(float));
    c->n_aff++;
}

/* capt.hh:125-290; `afford` is owned by the callee (freed here) */
static void subdivide(capt_builder *b, uint32_t points_begin, uint32_t how_many, uint32_t i, uint32_t *afford,
                      uint32_t n_afford, vo_volume volume, int d)
{
    vo_capt *c = b->c;
    const float *points = b->points;
    uint32_t *argsort = b->argsort;
    if (how_many == 1)
    {
        const float *rep = points + 3 * (size_t) argsort[points_begin];
        vo_volume aabb;
        memcpy(aabb.lower, rep, 12);
        memcpy(aabb.upper, rep, 12);
        if (isfinite(rep[0]))
        {
            vo_volume top;
            memcpy(top.lower, c->aabb_top, 12);
            memcpy(top.upper, c->aabb_top + 3, 12);
            vol_extend(&top, rep);
            memcpy(c->aabb_top, top.lower, 12);
            memcpy(c->aabb_top + 3, top.upper, 12);

            float xs[VO_RAKE] = {rep[0]}, ys[VO_RAKE] = {rep[1]}, zs[VO_RAKE] = {rep[2]};
            int j = 1;
            if (!vol_contained_by_internal_ball(&volume, rep, b->min_aff_l2))
            {
                for (uint32_t t = 0; t < n_afford; ++t)
                {
                    const float *p = points + 3 * (size_t) afford[t];
                    if (vol_distsq_to(&volume, p) <= b->max_aff_l2)
                    {
                        vol_extend(&aabb, p);
                        xs[j] = p[0];
                        ys[j] = p[1];
                        zs[j] = p[2];
                        j++;
                        if (j == VO_RAKE)
                        {
                            aff_push(b, xs, ys, zs);
                            j = 0;
                        }
                    }
                }
            }
            if (j > 0)
            {
                for (int jj = j; jj < VO_RAKE; ++jj) xs[jj] = ys[jj] = zs[jj] = INFINITY;
                aff_push(b, xs, ys, zs);
            }
        }
        memcpy(c->aabbs + 6 * b->n_leaves_done, aabb.lower, 12);
        memcpy(c->aabbs + 6 * b->n_leaves_done + 3, aabb.upper, 12);
        b->n_leaves_done++;
        c->aff_starts[b->n_leaves_done] = c->n_aff;
        free(afford);
        return;
    }

    const float test = median_partition(b, points_begin, points_begin + how_many, d);
    c->tests[i] = test;
    const uint32_t next_width = how_many / 2;
    vo_volume lo_vol = volume, hi_vol = volume;
    lo_vol.upper[d] = test;
    hi_vol.lower[d] = test;

    const float r_max = c->r_max;
    uint32_t *hi_afford = afford; /* moved */
    uint32_t *lo_afford = (uint32_t *) malloc((n_afford + how_many + 1) * sizeof(uint32_t));
    uint32_t hi_len = 0, lo_len = 0;
    for (uint32_t t = 0; t < n_afford; ++t)
    {
        const uint32_t idx = hi_afford[t];
        const float v = points[3 * (size_t) idx + d];
        if (v <= test + r_max) lo_afford[lo_len++] = idx;
        if (v >= test - r_max) hi_afford[hi_len++] = idx;
    }
    uint32_t new_hi = points_begin;
    uint32_t new_lo = points_begin + next_width;
    while (new_hi < points_begin + next_width && points[3 * (size_t) argsort[new_hi] + d] >= test - r_max &&
           isfinite(points[3 * (size_t) argsort[new_hi] + d]))
        ++new_hi;
    while (new_lo < points_begin + how_many && points[3 * (size_t) argsort[new_lo] + d] <= test + r_max &&
           isfinite(points[3 * (size_t) argsort[new_lo] + d]))
        ++new_lo;
    const uint32_t num_new_hi = new_hi - points_begin;
    const uint32_t num_new_lo = new_lo - (points_begin + next_width);
    hi_afford = (uint32_t *) realloc(hi_afford, (hi_len + num_new_hi + 1) * sizeof(uint32_t));
    memcpy(hi_afford + hi_len, argsort + points_begin, num_new_hi * sizeof(uint32_t));
    memcpy(lo_afford + lo_len, argsort + points_begin + next_width, num_new_lo * sizeof(uint32_t));

    const int next_d = (d + 1) % 3;
    subdivide(b, points_begin, next_width, 2 * i + 1, lo_afford, lo_len + num_new_lo, lo_vol, next_d);
    subdivide(b, points_begin + next_width, next_width, 2 * i + 2, hi_afford, hi_len + num_new_hi, hi_vol, next_d);
}

/* capt.hh:296-369 */
int vo_env_add_capt(vo_env *e, const float *pts, size_t n, float r_min, float r_max, float r_point)
{
    if (n < 2) return -1;
    vo_capt c;
    memset(&c, 0, sizeof(c));
    c.r_min = r_min;
    c.r_max = r_max;
    c.r_point = r_point;
    const float max_aff_l1 = r_max + r_point;
    capt_builder b;
    memset(&b, 0, sizeof(b));
    b.max_aff_l2 = max_aff_l1 * max_aff_l1;
    b.min_aff_l2 = (r_min + r_point) * (r_min + r_point);
    while (((size_t) 1u << c.nlog2) < n) c.nlog2++;
    const size_t pow2 = (size_t) 1u << c.nlog2;
    float *points2 = (float *) malloc(pow2 * 3 * sizeof(float));
    memcpy(points2, pts, n * 3 * sizeof(float));
    for (size_t i = n * 3; i < pow2 * 3; ++i) points2[i] = INFINITY;
    for (int k = 0; k < 3; ++k)
    {
        c.aabb_top[k] = INFINITY;
        c.aabb_top[3 + k] = -INFINITY;
    }
    c.n_tests = (uint32_t) pow2 - 1;
    c.n_leaves = (uint32_t) pow2;
    c.tests = (float *) malloc((pow2) * sizeof(float));
    for (size_t i = 0; i < pow2; ++i) c.tests[i] = NAN;
    c.aff_starts = (uint32_t *) calloc(pow2 + 1, sizeof(uint32_t));
    c.aabbs = (float *) malloc(pow2 * 6 * sizeof(float));
    uint32_t *argsort = (uint32_t *) malloc(pow2 * sizeof(uint32_t));
    for (size_t i = 0; i < pow2; ++i) argsort[i] = (uint32_t) i;
    b.points = points2;
    b.argsort = argsort;
    b.c = &c;
    vo_volume all;
    for (int k = 0; k < 3; ++k)
    {
        all.lower[k] = -INFINITY;
        all.upper[k] = INFINITY;
    }
    subdivide(&b, 0u, (uint32_t) pow2, 0u, (uint32_t *) malloc(sizeof(uint32_t)), 0u, all, 0);
    free(argsort);
    free(points2);
    e->capts = (vo_capt *) realloc(e->capts, (e->n_capts + 1) * sizeof(vo_capt));
    e->capts[e->n_capts++] = c;
    return 0;
}

int vo_env_capt_view(const vo_env *e, size_t index, vo_capt_view *out)
{
    if (index >= e->n_capts) return -1;
    const vo_capt *c = &e->capts[index];
    out->nlog2 = c->nlog2;
    out->n_tests = c->n_tests;
    out->n_leaves = c->n_leaves;
    out->n_aff_vectors = c->n_aff;
    out->tests = c->tests;
    out->aff_starts = c->aff_starts;
    out->aabbs = c->aabbs;
    out->aff_x = c->aff[0];
    out->aff_y = c->aff[1];
    out->aff_z = c->aff[2];
    memcpy(out->aabb_top, c->aabb_top, sizeof(out->aabb_top));
    out->r_min = c->r_min;
    out->r_max = c->r_max;
    out->r_point = c->r_point;
    return 0;
}

/* CAPT::collides (capt.hh:374-415) */
static int capt_collides_scalar(const vo_capt *c, const float ctr[3], float r)
{
    vo_volume top;
    memcpy(top.lower, c->aabb_top, 12);
    memcpy(top.upper, c->aabb_top + 3, 12);
    if (vol_distsq_to(&top, ctr) > r * r) return 0;
    size_t test_idx = 0;
    for (uint32_t i = 0, k = 0; i < c->nlog2; ++i)
    {
        test_idx = 2 * test_idx + 1 + (ctr[k] >= c->tests[test_idx]);
        k = (k + 1) % 3;
    }
    const size_t z = test_idx - c->n_tests;
    r += c->r_point;
    const float radius_sq = r * r;
    vo_volume bb;
    memcpy(bb.lower, c->aabbs + 6 * z, 12);
    memcpy(bb.upper, c->aabbs + 6 * z + 3, 12);
    if (vol_distsq_to(&bb, ctr) > radius_sq) return 0;
    for (uint32_t i = c->aff_starts[z]; i < c->aff_starts[z + 1]; ++i)
        for (int l = 0; l < VO_RAKE; ++l)
        {
            const float xs = c->aff[0][(size_t) i * VO_RAKE + l] - ctr[0];
            const float ys = c->aff[1][(size_t) i * VO_RAKE + l] - ctr[1];
            const float zs = c->aff[2][(size_t) i * VO_RAKE + l] - ctr[2];
            if (dot3(xs, ys, zs, xs, ys, zs) <= radius_sq) return 1;
        }
    return 0;
}

/* CAPT::collides_simd (capt.hh:428-512) over `lanes` lanes; result = any lane collides */
static int capt_collides_simd(const vo_capt *c, const float *cx, const float *cy, const float *cz, const float *radii,
                              int lanes)
{
    const float *ctr[3] = {cx, cy, cz};
    int inbounds[VO_RAKE];
    int any = 0;
    for (int l = 0; l < lanes; ++l)
    {
        int in = 1;
        for (int k = 0; k < 3; ++k)
            in = in && (ctr[k][l] + radii[l] >= c->aabb_top[k]) && (ctr[k][l] - radii[l] <= c->aabb_top[3 + k]);
        inbounds[l] = in;
        any |= in;
    }
    if (!any) return 0;

    uint32_t zs[VO_RAKE];
    float rc_sq[VO_RAKE];
    any = 0;
    for (int l = 0; l < lanes; ++l)
    {
        uint32_t idx = (uint32_t) (ctr[0][l] >= c->tests[0]) + 1u;
        for (uint32_t i = 1, k = 1; i < c->nlog2; ++i)
        {
            idx = (idx << 1) + (uint32_t) (ctr[k][l] >= c->tests[idx]) + 1u;
            k = (k + 1) % 3;
        }
        zs[l] = idx - c->n_tests;
        const float rr = radii[l] + c->r_point;
        rc_sq[l] = rr * rr;
        const float *bb = c->aabbs + 6 * (size_t) zs[l];
        const float d0 = ctr[0][l] - vclamp(ctr[0][l], bb[0], bb[3]);
        const float d1 = ctr[1][l] - vclamp(ctr[1][l], bb[1], bb[4]);
        const float d2 = ctr[2][l] - vclamp(ctr[2][l], bb[2], bb[5]);
        const float distsq = d0 * d0 + d1 * d1 + d2 * d2;
        inbounds[l] = inbounds[l] && (distsq <= rc_sq[l]);
        any |= inbounds[l];
    }
    if (!any) return 0;

    for (int l = 0; l < lanes; ++l)
    {
        const uint32_t start = c->aff_starts[zs[l]];
        const uint32_t end = inbounds[l] ? c->aff_starts[zs[l] + 1] : 0u;
        for (uint32_t i = start; i < end; ++i)
            for (int p = 0; p < VO_RAKE; ++p)
            {
                const float xs = c->aff[0][(size_t) i * VO_RAKE + p] - ctr[0][l];
                const float ys = c->aff[1][(size_t) i * VO_RAKE + p] - ctr[1][l];
                const float z_ = c->aff[2][(size_t) i * VO_RAKE + p] - ctr[2][l];
                if (dot3(xs, ys, z_, xs, ys, z_) <= rc_sq[l]) return 1;
            }
    }
    return 0;
}

int vo_capt_collides(const vo_env *e, size_t index, const float c[3], float r)
{
    return capt_collides_scalar(&e->capts[index], c, r);
}
int vo_capt_collides_simd(const vo_env *e, size_t index, const float *cx, const float *cy, const float *cz,
                          const float *r, int lanes)
{
    return capt_collides_simd(&e->capts[index], cx, cy, cz, r, lanes);
}

/* ------------------------------------------------------------------------- */
/* MVT build + query (collision/mvt.hh)                                       */
/* ------------------------------------------------------------------------- */

/* mvt.hh:754-764 */
static unsigned next_power_of_two(unsigned n)
{
    if (n == 0) return 1;
    n--;
    n |= n >> 1;
    n |= n >> 2;
    n |= n >> 4;
    n |= n >> 8;
    n |= n >> 16;
    n++;
    return n;
}

int vo_env_add_mvt(vo_env *e, const float *pts, size_t n, float r_min, float r_max, const float *ws_min,
                   const float *ws_max, float r_point)
{
    vo_mvt m;
    memset(&m, 0, sizeof(m));
    m.r_min = r_min;
    m.r_max = r_max;
    m.r_point = r_point;
    memcpy(m.ws_min, ws_min, 12);
    memcpy(m.ws_max, ws_max, 12);
    /* configure_grid (mvt.hh:438-447) */
    const float workspace_width = ws_max[0] - ws_min[0];
    const float cells = floorf(workspace_width / r_max);
    if (!(cells >= 1.0f) || n == 0) return 4;
    const uint32_t gw32 = (cells >= 4294967296.0f) ? 0xffffffffu : (uint32_t) cells;
    m.grid_width = gw32 < 65535u ? gw32 : 65535u;
    m.inv_scale = (float) m.grid_width / workspace_width;
    const uint32_t gw = m.grid_width;
    /* initialize_point_coord_pool (mvt.hh:455-477) */
    size_t est = (size_t) pow((double) r_max / 0.02, (double) 3.0f);
    unsigned bytes = next_power_of_two((unsigned) est * (unsigned) sizeof(float));
    if (bytes < VO_RAKE * sizeof(float)) bytes = VO_RAKE * sizeof(float);
    const size_t pool_bytes = (size_t) ((double) ((size_t) gw * gw * gw) * 0.1 * (double) (size_t) bytes * 3);
    const size_t pool_floats = pool_bytes / sizeof(float);
    m.cap = bytes / (unsigned) sizeof(float);
    const size_t max_voxels = pool_floats / (3 * (size_t) m.cap);
    /* initialize_voxel_index_pool (mvt.hh:494-508): room for grid_width^2 / 2 z-tables */
    const size_t max_z_tables = (size_t) ((double) ((size_t) gw * gw) * 0.5);

    m.x_table = (uint32_t *) malloc(gw * sizeof(uint32_t));
    for (uint32_t i = 0; i < gw; ++i) m.x_table[i] = VO_MVT_INVALID;
    size_t cap_y = 0, cap_z = 0, cap_v = 0;
    int rc = 0;
    /* build_spatial_grid (mvt.hh:530-593) */
    for (size_t i = 0; i < n && rc == 0; ++i)
    {
        const float *p = pts + 3 * i;
        uint32_t v[3];
        for (int k = 0; k < 3; ++k)
        {
            const float f = (p[k] - ws_min[k]) * m.inv_scale;
            const float c = stdclamp(f, 0.0f, (float) (gw - 1));
            v[k] = (uint32_t) (uint16_t) c;
        }
        uint32_t yt = m.x_table[v[0]];
        if (yt == VO_MVT_INVALID)
        {
            if (m.n_y == cap_y)
            {
                cap_y = cap_y ? cap_y * 2 : 16;
                m.y_tables = (uint32_t *) realloc(m.y_tables, cap_y * gw * sizeof(uint32_t));
            }
            yt = m.n_y++;
            for (uint32_t j = 0; j < gw; ++j) m.y_tables[(size_t) yt * gw + j] = VO_MVT_INVALID;
            m.x_table[v[0]] = yt;
        }
        uint32_t zt = m.y_tables[(size_t) yt * gw + v[1]];
        if (zt == VO_MVT_INVALID)
        {
            if (m.n_z + 1 > max_z_tables)
            {
                rc = 3;
                break;
            }
            if (m.n_z == cap_z)
            {
                cap_z = cap_z ? cap_z * 2 : 64;
                m.z_tables = (uint32_t *) realloc(m.z_tables, cap_z * gw * sizeof(uint32_t));
            }
            zt = m.n_z++;
            for (uint32_t j = 0; j < gw; ++j) m.z_tables[(size_t) zt * gw + j] = VO_MVT_INVALID;
            m.y_tables[(size_t) yt * gw + v[1]] = zt;
        }
        uint32_t vi = m.z_tables[(size_t) zt * gw + v[2]];
        if (vi == VO_MVT_INVALID)
        {
            if ((size_t) m.n_vox + 1 > max_voxels)
            {
                rc = 2;
                break;
            }
            if (m.n_vox == cap_v)
            {
                cap_v = cap_v ? cap_v * 2 : 256;
                m.bbox = (float *) realloc(m.bbox, cap_v * 6 * sizeof(float));
                m.count = (uint32_t *) realloc(m.count, cap_v * sizeof(uint32_t));
                m.px = (float *) realloc(m.px, cap_v * m.cap * sizeof(float));
                m.py = (float *) realloc(m.py, cap_v * m.cap * sizeof(float));
                m.pz = (float *) realloc(m.pz, cap_v * m.cap * sizeof(float));
            }
            vi = m.n_vox++;
            m.count[vi] = 0;
            for (uint32_t j = 0; j < m.cap; ++j)
                m.px[(size_t) vi * m.cap + j] = m.py[(size_t) vi * m.cap + j] = m.pz[(size_t) vi * m.cap + j] = INFINITY;
            m.z_tables[(size_t) zt * gw + v[2]] = vi;
        }
        if (m.count[vi] >= m.cap)
        {
            rc = 1;
            break;
        }
        const uint32_t c = m.count[vi];
        m.px[(size_t) vi * m.cap + c] = p[0];
        m.py[(size_t) vi * m.cap + c] = p[1];
        m.pz[(size_t) vi * m.cap + c] = p[2];
        float *bb = m.bbox + 6 * (size_t) vi;
        if (c == 0)
        {
            memcpy(bb, p, 12);
            memcpy(bb + 3, p, 12);
        }
        else
            for (int k = 0; k < 3; ++k)
            {
                bb[k] = fminf(bb[k], p[k]);
                bb[3 + k] = fmaxf(bb[3 + k], p[k]);
            }
        m.count[vi] = c + 1;
    }
    if (rc != 0)
    {
        mvt_free(&m);
        return rc;
    }
    /* compute_global_bounds (mvt.hh:595-609) */
    for (int k = 0; k < 3; ++k)
    {
        m.gmin[k] = 3.402823466e+38f;
        m.gmax[k] = -3.402823466e+38f;
    }
    for (uint32_t vi = 0; vi < m.n_vox; ++vi)
        for (int k = 0; k < 3; ++k)
        {
            m.gmin[k] = fminf(m.gmin[k], m.bbox[6 * (size_t) vi + k]);
            m.gmax[k] = fmaxf(m.gmax[k], m.bbox[6 * (size_t) vi + 3 + k]);
        }
    e->mvts = (vo_mvt *) realloc(e->mvts, (e->n_mvts + 1) * sizeof(vo_mvt));
    e->mvts[e->n_mvts++] = m;
    return 0;
}

int vo_env_mvt_view(const vo_env *e, size_t index, vo_mvt_view *out)
{
    if (index >= e->n_mvts) return -1;
    const vo_mvt *m = &e->mvts[index];
    out->grid_width = m->grid_width;
    out->capacity = m->cap;
    out->n_voxels = m->n_vox;
    out->n_y_tables = m->n_y;
    out->n_z_tables = m->n_z;
    out->inverse_scale_factor = m->inv_scale;
    memcpy(out->global_min, m->gmin, 12);
    memcpy(out->global_max, m->gmax, 12);
    out->x_table = m->x_table;
    out->y_tables = m->y_tables;
    out->z_tables = m->z_tables;
    out->voxel_count = m->count;
    out->voxel_bbox = m->bbox;
    out->px = m->px;
    out->py = m->py;
    out->pz = m->pz;
    return 0;
}

/* MVT::collides (mvt.hh:204-279) == one lane of collides_simd (mvt.hh:282-403; the 8-point SIMD chunks read the
 * voxel's +inf padding, which never collides).  Float -> uint16 casts are applied to in-range values exactly as
 * written; where the reference's cast would be undefined (negative upper bound) the range is empty here. */
static int mvt_collides_lane(const vo_mvt *m, float cx, float cy, float cz, float radius)
{
    const float c[3] = {cx, cy, cz};
    const float qr = radius + m->r_point;
    const float qr2 = qr * qr;
    for (int k = 0; k < 3; ++k)
        if (c[k] + qr < m->gmin[k] || c[k] - qr > m->gmax[k]) return 0;
    const float gqr = fminf(1.0f, qr * m->inv_scale);
    int lo[3], hi[3];
    for (int k = 0; k < 3; ++k)
    {
        const float g = (c[k] - m->ws_min[k]) * m->inv_scale;
        const float a = fmaxf(0.0f, g - gqr);
        const float b = fminf((float) (m->grid_width - 1), g + gqr);
        if (b < 0.0f) return 0;
        lo[k] = (int) (uint16_t) a;
        hi[k] = (int) (uint16_t) b;
    }
    const uint32_t gw = m->grid_width;
    for (int vx = lo[0]; vx <= hi[0]; ++vx)
    {
        const uint32_t yt = m->x_table[vx];
        if (yt == VO_MVT_INVALID) continue;
        for (int vy = lo[1]; vy <= hi[1]; ++vy)
        {
            const uint32_t zt = m->y_tables[(size_t) yt * gw + vy];
            if (zt == VO_MVT_INVALID) continue;
            for (int vz = lo[2]; vz <= hi[2]; ++vz)
            {
                const uint32_t vi = m->z_tables[(size_t) zt * gw + vz];
                if (vi == VO_MVT_INVALID) continue;
                const float *bb = m->bbox + 6 * (size_t) vi;
                if (c[0] + qr < bb[0] || c[0] - qr > bb[3] || c[1] + qr < bb[1] || c[1] - qr > bb[4] ||
                    c[2] + qr < bb[2] || c[2] - qr > bb[5])
                    continue;
                const float *x = m->px + (size_t) vi * m->cap, *y = m->py + (size_t) vi * m->cap,
                            *z = m->pz + (size_t) vi * m->cap;
                for (uint32_t i = 0; i < m->count[vi]; ++i)
                {
                    const float dx = c[0] - x[i], dy = c[1] - y[i], dz = c[2] - z[i];
                    if (dx * dx + dy * dy + dz * dz <= qr2) return 1;
                }
            }
        }
    }
    return 0;
}

int vo_mvt_collides(const vo_env *e, size_t index, const float c[3], float r)
{
    return mvt_collides_lane(&e->mvts[index], c[0], c[1], c[2], r);
}
int vo_mvt_collides_simd(const vo_env *e, size_t index, const float *cx, const float *cy, const float *cz,
                         const float *r, int lanes)
{
    for (int l = 0; l < lanes; ++l)
        if (mvt_collides_lane(&e->mvts[index], cx[l], cy[l], cz[l], r[l])) return 1;
    return 0;
}

/* ------------------------------------------------------------------------- */
/* collision primitives (collision/sphere_*.hh), one lane                     */
/* ------------------------------------------------------------------------- */

/* collision/math.hh:28-42 */
static inline float sql2_3(float ax, float ay, float az, float bx, float by, float bz)
{
    const float xs = ax - bx, ys = ay - by, zs = az - bz;
    return dot3(xs, ys, zs, xs, ys, zs);
}
/* collision/sphere_sphere.hh:9-23 */
static inline float sphere_sphere_sql2(float ax, float ay, float az, float ar, float bx, float by, float bz, float br)
{
    const float sum = sql2_3(ax, ay, az, bx, by, bz);
    const float rs = ar + br;
    return sum - rs * rs;
}
/* collision/sphere_cuboid.hh:8-27 */
static inline float sphere_cuboid(const float *c, float x, float y, float z, float rsq)
{
    const float xs = x - c[0], ys = y - c[1], zs = z - c[2];
    const float a1 = x86_max(fabsf(dot3(c[3], c[4], c[5], xs, ys, zs)) - c[12], 0.f);
    const float a2 = x86_max(fabsf(dot3(c[6], c[7], c[8], xs, ys, zs)) - c[13], 0.f);
    const float a3 = x86_max(fabsf(dot3(c[9], c[10], c[11], xs, ys, zs)) - c[14], 0.f);
    return dot3(a1, a2, a3, a1, a2, a3) - rsq;
}
/* collision/sphere_cuboid.hh:35-52 */
static inline float sphere_z_aligned_cuboid(const float *c, float x, float y, float z, float rsq)
{
    const float xs = x - c[0], ys = y - c[1], zs = z - c[2];
    const float a1 = x86_max(fabsf((c[3] * xs) + (c[4] * ys)) - c[12], 0.f);
    const float a2 = x86_max(fabsf((c[6] * xs) + (c[7] * ys)) - c[13], 0.f);
    const float a3 = x86_max(fabsf(zs) - c[14], 0.f);
    return dot3(a1, a2, a3, a1, a2, a3) - rsq;
}
/* collision/sphere_capsule.hh:8-23 */
static inline float sphere_capsule(const vo_capsule *c, float x, float y, float z, float r)
{
    const float dot = dot3(x - c->x1, y - c->y1, z - c->z1, c->xv, c->yv, c->zv);
    const float cdf = vclamp(dot * c->rdv, 0.F, 1.F);
    const float sum = sql2_3(x, y, z, c->x1 + c->xv * cdf, c->y1 + c->yv * cdf, c->z1 + c->zv * cdf);
    const float rs = r + c->r;
    return sum - rs * rs;
}
/* collision/sphere_capsule.hh:31-45 */
static inline float sphere_z_aligned_capsule(const vo_capsule *c, float x, float y, float z, float r)
{
    const float dot = (z - c->z1) * c->zv;
    const float cdf = vclamp(dot * c->rdv, 0.F, 1.F);
    const float sum = sql2_3(x, y, z, c->x1, c->y1, c->z1 + c->zv * cdf);
    const float rs = r + c->r;
    return sum - rs * rs;
}

/* collision/sphere_heightfield.hh:8-31, one lane.  The cell index is clamped to [0, xd] x [0, yd] by the reference, one
 * past the image on both axes, so it can address up to xd + 1 floats beyond the data (an out-of-bounds gather, UB).
 * Here such an index reads the last pixel; every in-bounds index is the reference's. */
static inline float sphere_heightfield(const vo_heightfield *a, float x, float y, float z, float r)
{
    const float xo = a->x - x, yo = a->y - y;
    const float xs = floorf(vclamp(a->xs * xo + (float) a->xd2, 0.F, (float) a->xd));
    const float ys = floorf(vclamp(a->ys * yo + (float) a->yd2, 0.F, (float) a->yd));
    const float index = ys * (float) a->xd + xs;
    long idx = lrintf(index); /* _mm256_cvtps_epi32: round to nearest even (vector/avx.hh:629) */
    const long last = (long) (a->xd * a->yd) - 1;
    if (idx > last) idx = last;
    if (idx < 0) idx = 0;
    const float zh = a->data[idx];
    const float zhs = a->zs * zh + a->z;
    return z - r - zhs;
}

/* ------------------------------------------------------------------------- */
/* sphere_environment_in_collision (collision/validity.hh:47-158) on a rake   */
/* ------------------------------------------------------------------------- */

/* test hook: replaces the exact sqrt of `max_extent` (validity.hh:59) by a caller-supplied one (the reference's
 * approximate sqrt exported by oracle/_ref).  NULL (default) = correctly rounded sqrtf.  Set before any worker runs. */
static vo_sqrt_fn vo_sqrt_hook = NULL;
void vo_set_max_extent_sqrt(vo_sqrt_fn fn) { vo_sqrt_hook = fn; }

static int sphere_environment_in_collision(const vo_env *e, const float *sx, const float *sy, const float *sz,
                                           float sr, int lanes)
{
    float max_extent[VO_RAKE];
    /* validity.hh:59 — the reference's approximate sqrt is replaced by the correctly rounded one (header note);
     * a test may install the reference's own `v * rsqrt_ps(v)` (oracle/_ref) to show the booleans do not depend on it */
    if (vo_sqrt_hook)
    {
        float sq[VO_RAKE] = {0}, rt[VO_RAKE];
        for (int l = 0; l < lanes; ++l) sq[l] = dot3(sx[l], sy[l], sz[l], sx[l], sy[l], sz[l]);
        vo_sqrt_hook(sq, rt, VO_RAKE);
        for (int l = 0; l < lanes; ++l) max_extent[l] = rt[l] + sr;
    }
    else
        for (int l = 0; l < lanes; ++l) max_extent[l] = sqrtf(dot3(sx[l], sy[l], sz[l], sx[l], sy[l], sz[l])) + sr;

#define LIST_LOOP(COUNT, MIN_DISTANCE, TEST)                                  \
    for (size_t i = 0; i < (COUNT); ++i)                                     \
    {                                                                        \
        int any_neg = 0;                                                     \
        for (int l = 0; l < lanes; ++l) any_neg |= signbit_set((MIN_DISTANCE) - max_extent[l]); \
        if (!any_neg) break; /* diff.test_zero() */                          \
        for (int l = 0; l < lanes; ++l)                                      \
            if (signbit_set(TEST)) return 1;                                 \
    }

    LIST_LOOP(e->n_spheres, e->spheres[i].min_distance,
              sphere_sphere_sql2(e->spheres[i].x, e->spheres[i].y, e->spheres[i].z, e->spheres[i].r, sx[l], sy[l],
                                 sz[l], sr))
    LIST_LOOP(e->n_capsules, e->capsules[i].min_distance, sphere_capsule(&e->capsules[i], sx[l], sy[l], sz[l], sr))
    LIST_LOOP(e->n_z_capsules, e->z_capsules[i].min_distance,
              sphere_z_aligned_capsule(&e->z_capsules[i], sx[l], sy[l], sz[l], sr))
    const float rsq = sr * sr;
    LIST_LOOP(e->n_cuboids, e->cuboids[i].min_distance, sphere_cuboid(e->cuboids[i].p, sx[l], sy[l], sz[l], rsq))
    LIST_LOOP(e->n_z_cuboids, e->z_cuboids[i].min_distance,
              sphere_z_aligned_cuboid(e->z_cuboids[i].p, sx[l], sy[l], sz[l], rsq))
#undef LIST_LOOP
    /* validity.hh:131-137 */
    for (size_t i = 0; i < e->n_heightfields; ++i)
        for (int l = 0; l < lanes; ++l)
            if (signbit_set(sphere_heightfield(&e->heightfields[i], sx[l], sy[l], sz[l], sr))) return 1;
    float radii[VO_RAKE];
    for (int l = 0; l < lanes; ++l) radii[l] = sr;
    for (size_t i = 0; i < e->n_capts; ++i)
        if (capt_collides_simd(&e->capts[i], sx, sy, sz, radii, lanes)) return 1;
    /* validity.hh:149-155 */
    for (size_t i = 0; i < e->n_mvts; ++i)
        for (int l = 0; l < lanes; ++l)
            if (mvt_collides_lane(&e->mvts[i], sx[l], sy[l], sz[l], sr)) return 1;
    return 0;
}

int vo_sphere_environment_in_collision(const vo_env *e, const float c[3], float r)
{
    return sphere_environment_in_collision(e, &c[0], &c[1], &c[2], r, 1);
}

/* ------------------------------------------------------------------------- */
/* L1b: robots (generated FK + tables)                                        */
/* ------------------------------------------------------------------------- */

typedef struct
{
    uint16_t bound, n_fine, fine_offset;
} vo_env_group;
typedef struct
{
    uint16_t bound_a, bound_b, n_pairs;
    uint32_t pair_offset;
} vo_self_group;
typedef struct
{
    const char *name;
    size_t dimension, n_spheres, n_total, resolution;
    float min_radius, max_radius;
    void (*fk_all)(const float *q, float *c);
    void (*fk_fine)(const float *q, float *c);
    const float *radii;
    const vo_env_group *env_groups;
    size_t n_env_groups;
    const uint16_t *env_fine;
    const vo_self_group *self_groups;
    size_t n_self_groups;
    const uint16_t (*self_pairs)[2];
    const float *lower, *span, *descale;
    void (*ee_frame)(const float *q, float *out12); /* fkcc_attach's last 12 outputs: translation, rotation col-major */
    const uint16_t *attach_groups;                  /* env_groups indices of the "Attachment vs. <link>" blocks */
    size_t n_attach_groups;
} vo_robot;

#include "gen/robots_gen.inc"

int vo_robot_id(const char *name)
{
    for (int i = 0; i < VO_N_ROBOTS; ++i)
        if (strcmp(vo_robots[i].name, name) == 0) return i;
    return -1;
}
size_t vo_robot_dimension(int r) { return vo_robots[r].dimension; }
size_t vo_robot_n_spheres(int r) { return vo_robots[r].n_spheres; }
size_t vo_robot_n_total_spheres(int r) { return vo_robots[r].n_total; }
size_t vo_robot_resolution(int r) { return vo_robots[r].resolution; }
void vo_robot_bounds(int r, float *lower, float *span)
{
    memcpy(lower, vo_robots[r].lower, vo_robots[r].dimension * sizeof(float));
    memcpy(span, vo_robots[r].span, vo_robots[r].dimension * sizeof(float));
}

void vo_fk(int robot, const float *q, float *out)
{
    const vo_robot *R = &vo_robots[robot];
    float c[3 * 256];
    R->fk_fine(q, c);
    for (size_t s = 0; s < R->n_spheres; ++s)
    {
        out[4 * s + 0] = c[3 * s + 0];
        out[4 * s + 1] = c[3 * s + 1];
        out[4 * s + 2] = c[3 * s + 2];
        out[4 * s + 3] = R->radii[s];
    }
}
void vo_fk_all(int robot, const float *q, float *out)
{
    const vo_robot *R = &vo_robots[robot];
    float c[3 * 256];
    R->fk_all(q, c);
    for (size_t s = 0; s < R->n_total; ++s)
    {
        out[4 * s + 0] = c[3 * s + 0];
        out[4 * s + 1] = c[3 * s + 1];
        out[4 * s + 2] = c[3 * s + 2];
        out[4 * s + 3] = R->radii[s];
    }
}

/* Robot::fkcc<rake> (robots/panda.hh:5226-10262): block[dim][VO_RAKE], first `lanes` lanes used.
 * lanes = 1 is the same computation for a rake whose 8 lanes hold one replicated configuration. */
static int fkcc_lanes(int robot, const vo_env *e, const float *block, const int lanes)
{
    const vo_robot *R = &vo_robots[robot];
    /* FK of every lane; centres stored [sphere][xyz][lane] */
    static _Thread_local float C[256][3][VO_RAKE];
    for (int l = 0; l < lanes; ++l)
    {
        float q[16], c[3 * 256];
        for (size_t j = 0; j < R->dimension; ++j) q[j] = block[j * VO_RAKE + l];
        R->fk_all(q, c);
        for (size_t s = 0; s < R->n_total; ++s)
            for (int k = 0; k < 3; ++k) C[s][k][l] = c[3 * s + k];
    }
#define ENV_HIT(S) sphere_environment_in_collision(e, C[S][0], C[S][1], C[S][2], R->radii[S], lanes)
    for (size_t g = 0; g < R->n_env_groups; ++g)
    {
        const vo_env_group *G = &R->env_groups[g];
        if (ENV_HIT(G->bound))
            for (size_t f = 0; f < G->n_fine; ++f)
            {
                const unsigned s = R->env_fine[G->fine_offset + f];
                if (ENV_HIT(s)) return 0;
            }
    }
#undef ENV_HIT
    /* sphere_sphere_self_collision (collision/validity.hh:23-44) */
#define SELF_HIT(A, B, RES)                                                                               \
    do                                                                                                    \
    {                                                                                                     \
        (RES) = 0;                                                                                        \
        for (int l = 0; l < lanes; ++l)                                                                   \
            (RES) |= signbit_set(sphere_sphere_sql2(C[A][0][l], C[A][1][l], C[A][2][l], R->radii[A],     \
                                                    C[B][0][l], C[B][1][l], C[B][2][l], R->radii[B]));   \
    } while (0)
    for (size_t g = 0; g < R->n_self_groups; ++g)
    {
        const vo_self_group *G = &R->self_groups[g];
        int hit;
        SELF_HIT(G->bound_a, G->bound_b, hit);
        if (!hit) continue;
        for (size_t p = 0; p < G->n_pairs; ++p)
        {
            const unsigned a = R->self_pairs[G->pair_offset + p][0], b = R->self_pairs[G->pair_offset + p][1];
            SELF_HIT(a, b, hit);
            if (hit) return 0;
        }
    }
#undef SELF_HIT
    if (!e->attached) return 1; /* planning/validate.hh:43,58: fkcc_attach only when the environment has attachments */

    /* Robot::fkcc_attach (robots/panda.hh:15308-15445): pose the attachment at the end-effector frame
     * (collision/attachments.hh:43-56: n_tf = p_tf * tf; sphere = n_tf * s), then attachment vs. environment
     * (validity.hh:258-275) and attachment vs. the links of the "Attachment vs." blocks (validity.hh:277-303).
     * PARITY UNPINNED: the two products are Eigen expressions on the vector type; restated here as the coefficient sums
     * ((a0 b0 + a1 b1) + a2 b2) [+ t], the order Eigen's fixed-size 3-term redux uses.  Eigen is not available offline. */
    static _Thread_local float A[256][3][VO_RAKE];
    const size_t na = e->n_attach_spheres < 256 ? e->n_attach_spheres : 256;
    for (int l = 0; l < lanes; ++l)
    {
        float q[16], f[12];
        for (size_t j = 0; j < R->dimension; ++j) q[j] = block[j * VO_RAKE + l];
        R->ee_frame(q, f);
        const float *T = e->attach_tf; /* row-major 3x4 */
        float Rn[3][3], tn[3];
        for (int i = 0; i < 3; ++i)
        {
            /* p_tf rotation is column-major behind the translation: R(i, k) = f[3 + 3 k + i] (vector/math.hh:40-51) */
            const float r0 = f[3 + i], r1 = f[6 + i], r2 = f[9 + i];
            for (int j = 0; j < 3; ++j) Rn[i][j] = ((r0 * T[j]) + (r1 * T[4 + j])) + (r2 * T[8 + j]);
            tn[i] = (((r0 * T[3]) + (r1 * T[7])) + (r2 * T[11])) + f[i];
        }
        for (size_t s = 0; s < na; ++s)
        {
            const float *sp = e->attach_spheres + 4 * s;
            for (int i = 0; i < 3; ++i) A[s][i][l] = (((Rn[i][0] * sp[0]) + (Rn[i][1] * sp[1])) + (Rn[i][2] * sp[2])) + tn[i];
        }
    }
    for (size_t s = 0; s < na; ++s)
        if (sphere_environment_in_collision(e, A[s][0], A[s][1], A[s][2], e->attach_spheres[4 * s + 3], lanes)) return 0;
#define ATT_HIT(S, RES)                                                                                          \
    do                                                                                                           \
    {                                                                                                            \
        (RES) = 0;                                                                                               \
        for (size_t s_ = 0; s_ < na && !(RES); ++s_)                                                             \
            for (int l = 0; l < lanes; ++l)                                                                      \
                (RES) |= signbit_set(sphere_sphere_sql2(C[S][0][l], C[S][1][l], C[S][2][l], R->radii[S],         \
                                                        A[s_][0][l], A[s_][1][l], A[s_][2][l],                   \
                                                        e->attach_spheres[4 * s_ + 3]));                         \
    } while (0)
    for (size_t g = 0; g < R->n_attach_groups; ++g)
    {
        const vo_env_group *G = &R->env_groups[R->attach_groups[g]];
        int hit;
        ATT_HIT(G->bound, hit);
        if (!hit) continue;
        for (size_t f = 0; f < G->n_fine; ++f)
        {
            const unsigned s = R->env_fine[G->fine_offset + f];
            ATT_HIT(s, hit);
            if (hit) return 0;
        }
    }
#undef ATT_HIT
    return 1;
}

/* Robot::eefk (robots/panda.hh, bindings/robot_helper.hh:279-282): 4x4 row-major */
void vo_eefk(int robot, const float *q, float *out16)
{
    float f[12];
    vo_robots[robot].ee_frame(q, f);
    for (int i = 0; i < 3; ++i)
    {
        for (int k = 0; k < 3; ++k) out16[4 * i + k] = f[3 + 3 * k + i];
        out16[4 * i + 3] = f[i];
    }
    out16[12] = out16[13] = out16[14] = 0.F;
    out16[15] = 1.F;
}

int vo_fkcc_rake(int robot, const vo_env *e, const float *block) { return fkcc_lanes(robot, e, block, VO_RAKE); }

/* ------------------------------------------------------------------------- */
/* L2: the rake (planning/validate.hh:24-77)                                  */
/* ------------------------------------------------------------------------- */

static int validate_vector(int robot, const vo_env *e, const float *start, const float *vector, float distance,
                           size_t resolution, int lanes)
{
    const vo_robot *R = &vo_robots[robot];
    const size_t dim = R->dimension;
    float block[16 * VO_RAKE] = {0};
    /* validate.hh:31-39: block[i] = start[i] + vector[i] * percents, percents[k] = (k+1)/8 */
    for (size_t i = 0; i < dim; ++i)
        for (int k = 0; k < VO_RAKE; ++k)
        {
            const float percent = (float) (k + 1) / (float) VO_RAKE;
            block[i * VO_RAKE + k] = start[i] + (vector[i] * percent);
        }
    /* validate.hh:41 */
    const size_t n = (size_t) fmaxf(ceilf(distance / (float) VO_RAKE * (float) resolution), 1.F);
    int valid = fkcc_lanes(robot, e, block, lanes);
    if (!valid || n == 1) return valid;
    /* validate.hh:50-64 */
    float backstep[16];
    for (size_t j = 0; j < dim; ++j) backstep[j] = vector[j] / (float) (VO_RAKE * n);
    for (size_t i = 1; i < n; ++i)
    {
        for (size_t j = 0; j < dim; ++j)
            for (int k = 0; k < VO_RAKE; ++k) block[j * VO_RAKE + k] = block[j * VO_RAKE + k] - backstep[j];
        if (!fkcc_lanes(robot, e, block, lanes)) return 0;
    }
    return 1;
}

int vo_validate_motion(int robot, const vo_env *e, const float *start, const float *goal)
{
    const vo_robot *R = &vo_robots[robot];
    float vector[16];
    for (size_t j = 0; j < R->dimension; ++j) vector[j] = goal[j] - start[j];
    return validate_vector(robot, e, start, vector, vo_l2_norm(vector, R->dimension), R->resolution, VO_RAKE);
}

int vo_validate(int robot, const vo_env *e, const float *q, int check_bounds)
{
    const vo_robot *R = &vo_robots[robot];
    if (check_bounds)
    {
        /* robot_helper.hh:258-262 with descale_configuration (robots/panda.hh:82-85): (q - s_a) * d_m in [0, 1] */
        for (size_t j = 0; j < R->dimension; ++j)
        {
            const float t = (q[j] - R->lower[j]) * R->descale[j];
            if (!(t <= 1.F) || !(t >= 0.F)) return 0;
        }
    }
    /* validate_motion<Robot, rake, 1>(q, q): vector = 0, distance = 0 -> n = 1, all 8 lanes hold q + 0 * percent */
    float vector[16] = {0};
    return validate_vector(robot, e, q, vector, 0.F, 1, 1);
}

/* The batch entry points mirror the C ABI's (include/vamp_mvt_amd.h), including its one rule the reference does not
 * have: a unit with a NaN or +-inf joint is INVALID.  vo_validate / vo_validate_motion above stay the plain restatement:
 * for such input the reference reads the sign bit of whichever NaN each x86 instruction propagates, an artefact no
 * caller of the path relies on (no sampler or planner produces non-finite joints). */
static int all_finite(const float *q, size_t dim)
{
    for (size_t j = 0; j < dim; ++j)
        if (!isfinite(q[j])) return 0;
    return 1;
}

void vo_validate_batch(int robot, const vo_env *e, const float *q, size_t n, uint8_t *out)
{
    const size_t dim = vo_robots[robot].dimension;
    for (size_t i = 0; i < n; ++i)
        out[i] = (uint8_t) (all_finite(q + i * dim, dim) && vo_validate(robot, e, q + i * dim, 0));
}

void vo_validate_motion_batch(int robot, const vo_env *e, const float *a, const float *b, size_t n, uint8_t *out)
{
    const size_t dim = vo_robots[robot].dimension;
    for (size_t i = 0; i < n; ++i)
        out[i] = (uint8_t) (all_finite(a + i * dim, dim) && all_finite(b + i * dim, dim) &&
                            vo_validate_motion(robot, e, a + i * dim, b + i * dim));
}

typedef struct
{
    int robot;
    const vo_env *e;
    const float *q, *goal; /* goal != NULL: edges */
    size_t n;
    uint8_t *out;
} mt_job;
static void *mt_run(void *p)
{
    mt_job *j = (mt_job *) p;
    if (j->goal)
        vo_validate_motion_batch(j->robot, j->e, j->q, j->goal, j->n, j->out);
    else
        vo_validate_batch(j->robot, j->e, j->q, j->n, j->out);
    return NULL;
}
static void mt_dispatch(int robot, const vo_env *e, const float *q, const float *goal, size_t n, uint8_t *out, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    const size_t dim = vo_robots[robot].dimension;
    pthread_t tid[256];
    mt_job jobs[256];
    /* small interleaved-free blocks: contiguous shards of ceil(n / threads) units */
    const size_t per = (n + (size_t) threads - 1) / (size_t) threads;
    int started = 0;
    for (int t = 0; t < threads; ++t)
    {
        const size_t b = per * (size_t) t;
        if (b >= n) break;
        const size_t cnt = (b + per <= n) ? per : n - b;
        jobs[t] = (mt_job){robot, e, q + b * dim, goal ? goal + b * dim : NULL, cnt, out + b};
        pthread_create(&tid[t], NULL, mt_run, &jobs[t]);
        started++;
    }
    for (int t = 0; t < started; ++t) pthread_join(tid[t], NULL);
}
void vo_validate_batch_mt(int robot, const vo_env *e, const float *q, size_t n, uint8_t *out, int threads)
{
    mt_dispatch(robot, e, q, NULL, n, out, threads);
}
void vo_validate_motion_batch_mt(int robot, const vo_env *e, const float *a, const float *b, size_t n, uint8_t *out,
                                 int threads)
{
    mt_dispatch(robot, e, a, b, n, out, threads);
}

/* ------------------------------------------------------------------------- */
/* point-cloud filters (collision/filter.hh, collision/filter_centervox.hh)   */
/* ------------------------------------------------------------------------- */

/* float -> uint32_t as x86-64 compilers emit it for `return <float expr>;` in a function returning uint32_t
 * (filter.hh:129-132): cvttss2si to 64 bits, low 32 bits kept; NaN / out of range give the "integer indefinite"
 * 0x8000000000000000, i.e. 0.  (The C++ conversion is undefined for negative values; this is what the reference's
 * build does with them, and culled copies of point 0 do reach it - see vo_filter_scdf.) */
static uint32_t cvt_f32_u32_x86(float v)
{
    if (!(v > -9223372036854775808.0f && v < 9223372036854775808.0f)) return 0u;
    return (uint32_t) (uint64_t) (int64_t) v;
}
/* filter.hh:129-132 */
static uint32_t remap_point(float x, float mn, float mx) { return cvt_f32_u32_x86(((x - mn) / (mx - mn)) * 1000.0f); }
/* _pdep_u32 */
static uint32_t pdep32(uint32_t src, uint32_t mask)
{
    uint32_t out = 0;
    for (uint32_t bit = 1; mask; bit <<= 1)
    {
        const uint32_t low = mask & (0u - mask);
        if (src & bit) out |= low;
        mask &= mask - 1;
    }
    return out;
}
/* filter.hh:149-152 */
static uint32_t morton_pdep(uint32_t x, uint32_t y, uint32_t z)
{
    return pdep32(x, 0x49249249u) | pdep32(y, 0x92492492u) | pdep32(z, 0x24924924u);
}

typedef struct
{
    uint32_t first, second, pos;
} vo_morton;
static int cmp_morton(const void *a, const void *b)
{
    const vo_morton *x = (const vo_morton *) a, *y = (const vo_morton *) b;
    if (x->second != y->second) return x->second < y->second ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos); /* ties: current order (the reference's pdqsort leaves it open) */
}

/* filter_pointcloud (collision/filter.hh:175-275), "scdf".  out_xyz: capacity n points; returns the number kept.
 * Restated quirks: `max` is the MIN of origin + range (:193); the index vector is resized to n before culling, so the
 * n - hi entries behind the survivors all refer to point 0 (:195-216) and take part in every later step; min/max of
 * the next curve are averaged with the extremes over all three coordinates (:232-233, :261-262). */
size_t vo_filter_scdf(const float *pc, size_t n, float min_dist, float max_range, const float origin[3],
                      const float ws_min[3], const float ws_max[3], int cull, float *out_xyz)
{
    if (n == 0) return 0;
    const float sqdist = min_dist * min_dist, sqrange = max_range * max_range;
    float mn = fminf(fminf(origin[0] - max_range, origin[1] - max_range), origin[2] - max_range);
    float mx = fminf(fminf(origin[0] + max_range, origin[1] + max_range), origin[2] + max_range);
    vo_morton *m = (vo_morton *) calloc(n, sizeof(vo_morton));
    vo_morton *f = (vo_morton *) calloc(n, sizeof(vo_morton));
    size_t size = n, hi = 0;
    for (size_t i = 0; i < n; ++i)
    {
        const float *p = pc + 3 * i;
        if (!cull || (sql2_3(p[0], p[1], p[2], origin[0], origin[1], origin[2]) < sqrange && ws_min[0] <= p[0] &&
                      p[0] <= ws_max[0] && ws_min[1] <= p[1] && p[1] <= ws_max[1] && ws_min[2] <= p[2] && p[2] <= ws_max[2]))
            m[hi++].first = (uint32_t) i;
    }
    static const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    for (int it = 0; it < 6; ++it)
    {
        const int *c = perms[it];
        float new_min = mx, new_max = mn;
        for (size_t i = 0; i < size; ++i)
        {
            const float *p = pc + 3 * (size_t) m[i].first;
            const uint32_t c0 = remap_point(p[c[0]], mn, mx), c1 = remap_point(p[c[1]], mn, mx),
                           c2 = remap_point(p[c[2]], mn, mx);
            new_min = fminf(fminf(fminf(new_min, p[0]), p[1]), p[2]);
            new_max = fmaxf(fmaxf(fmaxf(new_max, p[0]), p[1]), p[2]);
            m[i].second = morton_pdep(c0, c1, c2);
            m[i].pos = (uint32_t) i;
        }
        qsort(m, size, sizeof(vo_morton), cmp_morton);
        size_t k = 0;
        f[k++] = m[0];
        for (size_t i = 1; i < size; ++i)
        {
            const float *p1 = pc + 3 * (size_t) m[i].first, *p2 = pc + 3 * (size_t) f[k - 1].first;
            if (sql2_3(p1[0], p1[1], p1[2], p2[0], p2[1], p2[2]) > sqdist) f[k++] = m[i];
        }
        vo_morton *t = m;
        m = f;
        f = t;
        size = k;
        mx = (float) ((double) (new_max + mx) / 2.);
        mn = (float) ((double) (new_min + mn) / 2.);
    }
    for (size_t i = 0; i < size; ++i) memcpy(out_xyz + 3 * i, pc + 3 * (size_t) m[i].first, 3 * sizeof(float));
    free(m);
    free(f);
    return size;
}

/* filter_pointcloud_centervox (collision/filter_centervox.hh:16-313).  Returns the number kept, or (size_t) -1 where
 * the reference throws "Voxel pool exhausted" (:132-134).  Output order = the reference's extract_points(): x tables,
 * y tables and voxels each in order of first appearance (:150-166). */
size_t vo_filter_centervox(const float *pc, size_t n, float voxel_size, float max_range, const float origin[3],
                           const float ws_min[3], const float ws_max[3], float *out_xyz)
{
    if (n == 0) return 0;
    const float max_range_sq = max_range * max_range;
    const float width = fmaxf(fmaxf(ws_max[0] - ws_min[0], ws_max[1] - ws_min[1]), ws_max[2] - ws_min[2]);
    int grid_width = (int) ceilf(width / voxel_size);
    if (grid_width > 255) grid_width = 255;
    const float isf = (float) grid_width / width;
    const float per_dim = width / voxel_size;
    size_t pool = (size_t) (powf(per_dim, 3.0f) * 0.05f);
    if (pool > 32768) pool = 32768;
    typedef struct
    {
        float stored[3], center[3], dsq;
    } voxel;
    voxel *vox = (voxel *) malloc((pool ? pool : 1) * sizeof(voxel));
    size_t n_vox = 0;
    /* x -> y-table, (y-table, y) -> z-table, (z-table, z) -> voxel; tables in creation order */
    int x_to_y[255];
    for (int i = 0; i < 255; ++i) x_to_y[i] = -1;
    size_t n_y = 0, n_z = 0, cap_z = 64;
    int(*y_to_z)[255] = (int(*)[255]) malloc(255 * sizeof(*y_to_z));
    size_t *y_first_z = NULL; /* z tables are listed per y table: keep (owner, order) */
    (void) y_first_z;
    int(*z_to_v)[255] = (int(*)[255]) malloc(cap_z * sizeof(*z_to_v));
    int *z_owner = (int *) malloc(cap_z * sizeof(int));
    size_t *z_count = (size_t *) malloc(cap_z * sizeof(size_t));
    int(*z_list)[255] = (int(*)[255]) malloc(cap_z * sizeof(*z_list)); /* voxels of a z table in creation order */
    int failed = 0;
    for (size_t i = 0; i < n && !failed; ++i)
    {
        const float *p = pc + 3 * i;
        const float dx = p[0] - origin[0], dy = p[1] - origin[1], dz = p[2] - origin[2];
        if (dx * dx + dy * dy + dz * dz >= max_range_sq) continue;
        if (p[0] < ws_min[0] || p[0] > ws_max[0] || p[1] < ws_min[1] || p[1] > ws_max[1] || p[2] < ws_min[2] ||
            p[2] > ws_max[2])
            continue;
        int v[3];
        for (int k = 0; k < 3; ++k)
        {
            int c = (int) ((p[k] - ws_min[k]) * isf);
            v[k] = c < 0 ? 0 : (c > 254 ? 254 : c);
        }
        if (x_to_y[v[0]] < 0)
        {
            x_to_y[v[0]] = (int) n_y;
            for (int j = 0; j < 255; ++j) y_to_z[n_y][j] = -1;
            ++n_y;
        }
        const int yt = x_to_y[v[0]];
        if (y_to_z[yt][v[1]] < 0)
        {
            if (n_z == cap_z)
            {
                cap_z *= 2;
                z_to_v = (int(*)[255]) realloc(z_to_v, cap_z * sizeof(*z_to_v));
                z_owner = (int *) realloc(z_owner, cap_z * sizeof(int));
                z_count = (size_t *) realloc(z_count, cap_z * sizeof(size_t));
                z_list = (int(*)[255]) realloc(z_list, cap_z * sizeof(*z_list));
            }
            y_to_z[yt][v[1]] = (int) n_z;
            for (int j = 0; j < 255; ++j) z_to_v[n_z][j] = -1;
            z_owner[n_z] = yt;
            z_count[n_z] = 0;
            ++n_z;
        }
        const int zt = y_to_z[yt][v[1]];
        if (z_to_v[zt][v[2]] < 0)
        {
            if (n_vox >= pool)
            {
                failed = 1;
                break;
            }
            voxel *nv = &vox[n_vox];
            for (int k = 0; k < 3; ++k) nv->center[k] = ws_min[k] + ((float) v[k] + 0.5f) * voxel_size;
            nv->dsq = -1.0f; /* unoccupied */
            z_to_v[zt][v[2]] = (int) n_vox;
            z_list[zt][z_count[zt]++] = (int) n_vox;
            ++n_vox;
        }
        voxel *vx = &vox[z_to_v[zt][v[2]]];
        const float ex = p[0] - vx->center[0], ey = p[1] - vx->center[1], ez = p[2] - vx->center[2];
        const float nd = ex * ex + ey * ey + ez * ez;
        if (vx->dsq < 0.0f || nd < vx->dsq)
        {
            memcpy(vx->stored, p, sizeof(vx->stored));
            vx->dsq = nd;
        }
    }
    size_t out = 0;
    if (!failed)
    {
        /* y tables in creation order; within each, its z tables in creation order (they were appended to the owner's
         * vector in that order: global creation order filtered by owner) */
        for (size_t y = 0; y < n_y; ++y)
            for (size_t z = 0; z < n_z; ++z)
                if (z_owner[z] == (int) y)
                    for (size_t i = 0; i < z_count[z]; ++i)
                    {
                        memcpy(out_xyz + 3 * out, vox[z_list[z][i]].stored, 3 * sizeof(float));
                        ++out;
                    }
    }
    free(vox);
    free(y_to_z);
    free(z_to_v);
    free(z_owner);
    free(z_count);
    free(z_list);
    return failed ? (size_t) -1 : out;
}

/* bindings/robot_helper.hh:284-322: keep the points whose sphere neither overlaps a robot sphere at q nor the environment */
size_t vo_filter_self_from_pointcloud(int robot, const vo_env *e, const float *q, const float *pts, size_t n, float r,
                                      float *out)
{
    const vo_robot *R = &vo_robots[robot];
    float c[3 * 256];
    R->fk_fine(q, c);
    size_t m = 0;
    for (size_t i = 0; i < n; ++i)
    {
        const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        int valid = 1;
        for (size_t s = 0; s < R->n_spheres && valid; ++s)
            if (signbit_set(sphere_sphere_sql2(c[3 * s], c[3 * s + 1], c[3 * s + 2], R->radii[s], x, y, z, r)) ||
                sphere_environment_in_collision(e, &x, &y, &z, r, 1))
                valid = 0;
        if (valid)
        {
            out[3 * m] = x, out[3 * m + 1] = y, out[3 * m + 2] = z;
            ++m;
        }
    }
    return m;
}

/* ------------------------------------------------------------------------- */
/* AVX2 build of the same restatement (cpu_baseline of bench.py)             */
/* ------------------------------------------------------------------------- */
/* What the reference's SIMD layer (vector/avx.hh: FloatVector<8> = __m256) can do on a batch: one AVX2 lane per
 * configuration, 8 DISTINCT configurations per rake, per-lane validity masks (the reference's fkcc<8> folds the 8
 * lanes into one boolean; a batch API needs one boolean per configuration).  Every lane executes exactly the
 * operations of the scalar port above (one rounding per written operation; -ffp-contract=off; exact vsqrtps), so the
 * answers are bit-identical to vo_validate_batch - checked by tests/test_oracle_pins.py and again by bench.py before it
 * times this.  Primitive environments only (no point clouds, heightfields, attachments): the bench workload. */
#if defined(__x86_64__) && defined(__GNUC__)
#pragma GCC push_options
#pragma GCC target("avx2")
#include <immintrin.h>

typedef __m256 v8f;
#define V8_SET1(x) _mm256_set1_ps(x)

static inline v8f v8_and(v8f a, v8f b) { return _mm256_and_ps(a, b); }
static inline unsigned v8_neg_mask(v8f v) { return (unsigned) _mm256_movemask_ps(v); } /* sign bits: `not test_zero` */
static inline v8f v8_abs(v8f v) { return v8_and(v, _mm256_castsi256_ps(_mm256_set1_epi32(0x7fffffff))); }
static inline v8f v8_max(v8f a, v8f b) { return _mm256_max_ps(a, b); } /* a > b ? a : b (second operand on NaN) */
static inline v8f v8_min(v8f a, v8f b) { return _mm256_min_ps(a, b); }
static inline v8f v8_clamp(v8f v, v8f lo, v8f hi) { return v8_min(v8_max(v, lo), hi); }

/* vector/avx.hh:455-548, the statements of vo_sinf on 8 lanes */
static inline v8f v8_sin(v8f x)
{
    const v8f sign_mask = _mm256_castsi256_ps(_mm256_set1_epi32((int) 0x80000000u));
    v8f sign_bit = v8_and(x, sign_mask);
    x = v8_abs(x);
    v8f y = x * V8_SET1(1.27323954473516f);
    __m256i j = _mm256_cvtps_epi32(y); /* round to nearest even; out of range -> 0x80000000 */
    j = _mm256_add_epi32(j, _mm256_set1_epi32(1));
    j = _mm256_and_si256(j, _mm256_set1_epi32(~1));
    y = _mm256_cvtepi32_ps(j);
    const __m256i swap = _mm256_slli_epi32(_mm256_and_si256(j, _mm256_set1_epi32(4)), 29);
    const __m256i poly = _mm256_cmpeq_epi32(_mm256_and_si256(j, _mm256_set1_epi32(2)), _mm256_setzero_si256());
    sign_bit = _mm256_xor_ps(sign_bit, _mm256_castsi256_ps(swap));
    const v8f xmm1 = y * V8_SET1(-0.78515625f);
    const v8f xmm2 = y * V8_SET1(-2.4187564849853515625e-4f);
    const v8f xmm3 = y * V8_SET1(-3.77489497744594108e-8f);
    x = x + xmm1;
    x = x + xmm2;
    x = x + xmm3;
    const v8f z = x * x;
    v8f yc = V8_SET1(2.443315711809948E-005f);
    yc = yc * z;
    yc = yc + V8_SET1(-1.388731625493765E-003f);
    yc = yc * z;
    yc = yc + V8_SET1(4.166664568298827E-002f);
    yc = yc * z;
    yc = yc * z;
    const v8f tmp = z * V8_SET1(0.5f);
    yc = yc - tmp;
    yc = yc + V8_SET1(1.0f);
    v8f y2 = V8_SET1(-1.9515295891E-4f);
    y2 = y2 * z;
    y2 = y2 + V8_SET1(8.3321608736E-3f);
    y2 = y2 * z;
    y2 = y2 + V8_SET1(-1.6666654611E-1f);
    y2 = y2 * z;
    y2 = y2 * x;
    y2 = y2 + x;
    const v8f pm = _mm256_castsi256_ps(poly);
    const v8f sel = _mm256_and_ps(pm, y2) + _mm256_andnot_ps(pm, yc); /* and / andnot / add */
    return _mm256_xor_ps(sel, sign_bit);
}
static inline v8f v8_cos(v8f x)
{
    const float PI = 3.14159265359f;
    const v8f v_sq = x + V8_SET1((float) (PI / 2.));
    const v8f sub = v8_and(_mm256_cmp_ps(v_sq, V8_SET1(PI), _CMP_GE_OQ), V8_SET1((float) (2 * PI)));
    return v8_sin(v_sq - sub);
}

#include "gen/robots_fk_v8.inc"

static inline v8f v8_dot3(v8f ax, v8f ay, v8f az, v8f bx, v8f by, v8f bz) { return (ax * bx) + (ay * by) + (az * bz); }
static inline v8f v8_sql2(v8f ax, v8f ay, v8f az, v8f bx, v8f by, v8f bz)
{
    const v8f xs = ax - bx, ys = ay - by, zs = az - bz;
    return v8_dot3(xs, ys, zs, xs, ys, zs);
}

/* sphere_environment_in_collision for 8 independent lanes: bit l of the result = lane l's sphere hits something.
 * `act`: lanes whose answer is wanted.  Lists are sorted by min_distance, so a lane's early break (validity.hh:62-66)
 * is the per-primitive predicate neg(min_distance - max_extent): once false it stays false. */
static unsigned v8_env_hit(const vo_env *e, v8f x, v8f y, v8f z, float r, unsigned act)
{
    const v8f ext = _mm256_sqrt_ps(v8_dot3(x, y, z, x, y, z)) + V8_SET1(r);
    const v8f zero = _mm256_setzero_ps(), one = V8_SET1(1.F);
    unsigned hit = 0;
#define V8_LIST(COUNT, MD, TEST)                                                       \
    for (size_t i = 0; i < (COUNT); ++i)                                               \
    {                                                                                  \
        const unsigned live = v8_neg_mask(V8_SET1(MD) - ext) & act & ~hit;             \
        if (!live) break;                                                              \
        hit |= v8_neg_mask(TEST) & live;                                               \
    }
    V8_LIST(e->n_spheres, e->spheres[i].min_distance,
            ({
                const vo_sphere *s = &e->spheres[i];
                const v8f sum = v8_sql2(V8_SET1(s->x), V8_SET1(s->y), V8_SET1(s->z), x, y, z);
                const v8f rs = V8_SET1(s->r) + V8_SET1(r);
                sum - rs * rs;
            }))
    V8_LIST(e->n_capsules, e->capsules[i].min_distance,
            ({
                const vo_capsule *c = &e->capsules[i];
                const v8f dot = v8_dot3(x - V8_SET1(c->x1), y - V8_SET1(c->y1), z - V8_SET1(c->z1), V8_SET1(c->xv),
                                        V8_SET1(c->yv), V8_SET1(c->zv));
                const v8f cdf = v8_clamp(dot * V8_SET1(c->rdv), zero, one);
                const v8f sum = v8_sql2(x, y, z, V8_SET1(c->x1) + V8_SET1(c->xv) * cdf, V8_SET1(c->y1) + V8_SET1(c->yv) * cdf,
                                        V8_SET1(c->z1) + V8_SET1(c->zv) * cdf);
                const v8f rs = V8_SET1(r) + V8_SET1(c->r);
                sum - rs * rs;
            }))
    V8_LIST(e->n_z_capsules, e->z_capsules[i].min_distance,
            ({
                const vo_capsule *c = &e->z_capsules[i];
                const v8f dot = (z - V8_SET1(c->z1)) * V8_SET1(c->zv);
                const v8f cdf = v8_clamp(dot * V8_SET1(c->rdv), zero, one);
                const v8f sum = v8_sql2(x, y, z, V8_SET1(c->x1), V8_SET1(c->y1), V8_SET1(c->z1) + V8_SET1(c->zv) * cdf);
                const v8f rs = V8_SET1(r) + V8_SET1(c->r);
                sum - rs * rs;
            }))
    const v8f rsq = V8_SET1(r * r);
    V8_LIST(e->n_cuboids, e->cuboids[i].min_distance,
            ({
                const float *c = e->cuboids[i].p;
                const v8f xs = x - V8_SET1(c[0]), ys = y - V8_SET1(c[1]), zs = z - V8_SET1(c[2]);
                const v8f a1 = v8_max(v8_abs(v8_dot3(V8_SET1(c[3]), V8_SET1(c[4]), V8_SET1(c[5]), xs, ys, zs)) - V8_SET1(c[12]), zero);
                const v8f a2 = v8_max(v8_abs(v8_dot3(V8_SET1(c[6]), V8_SET1(c[7]), V8_SET1(c[8]), xs, ys, zs)) - V8_SET1(c[13]), zero);
                const v8f a3 = v8_max(v8_abs(v8_dot3(V8_SET1(c[9]), V8_SET1(c[10]), V8_SET1(c[11]), xs, ys, zs)) - V8_SET1(c[14]), zero);
                v8_dot3(a1, a2, a3, a1, a2, a3) - rsq;
            }))
    V8_LIST(e->n_z_cuboids, e->z_cuboids[i].min_distance,
            ({
                const float *c = e->z_cuboids[i].p;
                const v8f xs = x - V8_SET1(c[0]), ys = y - V8_SET1(c[1]), zs = z - V8_SET1(c[2]);
                const v8f a1 = v8_max(v8_abs((V8_SET1(c[3]) * xs) + (V8_SET1(c[4]) * ys)) - V8_SET1(c[12]), zero);
                const v8f a2 = v8_max(v8_abs((V8_SET1(c[6]) * xs) + (V8_SET1(c[7]) * ys)) - V8_SET1(c[13]), zero);
                const v8f a3 = v8_max(v8_abs(zs) - V8_SET1(c[14]), zero);
                v8_dot3(a1, a2, a3, a1, a2, a3) - rsq;
            }))
#undef V8_LIST
    return hit;
}

/* Robot::fkcc on a rake of 8 DISTINCT configurations -> bit l = configuration l is valid */
static unsigned v8_fkcc(int robot, const vo_env *e, const v8f *q, unsigned lanes)
{
    const vo_robot *R = &vo_robots[robot];
    v8f C[3 * 256];
    vo_fk_all_v8[robot](q, C);
    unsigned bad = 0;
    for (size_t g = 0; g < R->n_env_groups && (lanes & ~bad); ++g)
    {
        const vo_env_group *G = &R->env_groups[g];
        const unsigned gate = v8_env_hit(e, C[3 * G->bound], C[3 * G->bound + 1], C[3 * G->bound + 2], R->radii[G->bound], lanes & ~bad);
        for (size_t f = 0; f < G->n_fine && (gate & ~bad); ++f)
        {
            const unsigned s = R->env_fine[G->fine_offset + f];
            bad |= v8_env_hit(e, C[3 * s], C[3 * s + 1], C[3 * s + 2], R->radii[s], gate & ~bad);
        }
    }
    for (size_t g = 0; g < R->n_self_groups && (lanes & ~bad); ++g)
    {
        const vo_self_group *G = &R->self_groups[g];
#define V8_PAIR(A, B)                                                                                           \
    ({                                                                                                          \
        const v8f rs = V8_SET1(R->radii[A]) + V8_SET1(R->radii[B]);                                             \
        v8_neg_mask(v8_sql2(C[3 * (A)], C[3 * (A) + 1], C[3 * (A) + 2], C[3 * (B)], C[3 * (B) + 1], C[3 * (B) + 2]) - rs * rs); \
    })
        const unsigned gate = V8_PAIR(G->bound_a, G->bound_b) & lanes & ~bad;
        for (size_t p = 0; p < G->n_pairs && (gate & ~bad); ++p)
        {
            const unsigned a = R->self_pairs[G->pair_offset + p][0], b = R->self_pairs[G->pair_offset + p][1];
            bad |= V8_PAIR(a, b) & gate;
        }
#undef V8_PAIR
    }
    return lanes & ~bad;
}

int vo_has_avx2(void) { return __builtin_cpu_supports("avx2") ? 1 : 0; }

/* out[i] = 1 valid / 0 invalid, as vo_validate_batch; returns 0, or -1 for environments this build does not cover */
int vo_validate_batch_avx2(int robot, const vo_env *e, const float *q, size_t n, uint8_t *out)
{
    if (e->n_capts || e->n_mvts || e->n_heightfields || e->attached) return -1;
    const size_t dim = vo_robots[robot].dimension;
    for (size_t base = 0; base < n; base += 8)
    {
        const unsigned cnt = (n - base < 8) ? (unsigned) (n - base) : 8u;
        float block[16][8] __attribute__((aligned(32)));
        for (size_t j = 0; j < dim; ++j)
            for (unsigned l = 0; l < 8; ++l) block[j][l] = q[(base + (l < cnt ? l : 0)) * dim + j]; /* AoS -> rake */
        v8f rake[16];
        for (size_t j = 0; j < dim; ++j) rake[j] = _mm256_load_ps(block[j]);
        const unsigned valid = v8_fkcc(robot, e, rake, (1u << cnt) - 1u);
        for (unsigned l = 0; l < cnt; ++l) /* + the boundary rule for non-finite joints (vo_validate_batch) */
            out[base + l] = (uint8_t) (((valid >> l) & 1u) && all_finite(q + (base + l) * dim, dim));
    }
    return 0;
}
#pragma GCC pop_options

typedef struct
{
    int robot;
    const vo_env *e;
    const float *q;
    size_t n;
    uint8_t *out;
} avx_job;
static void *avx_run(void *p)
{
    avx_job *j = (avx_job *) p;
    vo_validate_batch_avx2(j->robot, j->e, j->q, j->n, j->out);
    return NULL;
}
int vo_validate_batch_avx2_mt(int robot, const vo_env *e, const float *q, size_t n, uint8_t *out, int threads)
{
    if (e->n_capts || e->n_mvts || e->n_heightfields || e->attached) return -1;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    const size_t dim = vo_robots[robot].dimension;
    pthread_t tid[256];
    avx_job jobs[256];
    const size_t per = (((n + (size_t) threads - 1) / (size_t) threads) + 7) & ~(size_t) 7; /* whole rakes per shard */
    int started = 0;
    for (int t = 0; t < threads; ++t)
    {
        const size_t b = per * (size_t) t;
        if (b >= n) break;
        jobs[t] = (avx_job){robot, e, q + b * dim, (b + per <= n) ? per : n - b, out + b};
        pthread_create(&tid[t], NULL, avx_run, &jobs[t]);
        started++;
    }
    for (int t = 0; t < started; ++t) pthread_join(tid[t], NULL);
    return 0;
}
#else
int vo_has_avx2(void) { return 0; }
int vo_validate_batch_avx2(int robot, const vo_env *e, const float *q, size_t n, uint8_t *out)
{
    (void) robot, (void) e, (void) q, (void) n, (void) out;
    return -1;
}
int vo_validate_batch_avx2_mt(int robot, const vo_env *e, const float *q, size_t n, uint8_t *out, int threads)
{
    (void) robot, (void) e, (void) q, (void) n, (void) out, (void) threads;
    return -1;
}
#endif
