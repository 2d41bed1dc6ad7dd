/*
 * rc_oracle.c -- CPU restatement of the reference ray caster (TEST INFRASTRUCTURE ONLY).
 * See rc_oracle.h for scope and parity status.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread rc_oracle.c -lm
 * -ffp-contract=off is REQUIRED: every product and sum below is a separately rounded
 * binary32 operation, as the source expressions of the reference read.
 */
#include "rc_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- vec3.h operators used on the path, same evaluation order ---------------------- */

static inline rco_vec3 v3(float x, float y, float z) { rco_vec3 r = {x, y, z}; return r; }
/* vec3.h:32-38,95-98 */
static inline rco_vec3 v_add(rco_vec3 a, rco_vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
/* vec3.h:40-46,100-103 */
static inline rco_vec3 v_sub(rco_vec3 a, rco_vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
/* vec3.h:48-54,105-108 */
static inline rco_vec3 v_mul(rco_vec3 a, rco_vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
/* vec3.h:81-87,126-129 */
static inline rco_vec3 v_scale(rco_vec3 a, float t) { return v3(a.x * t, a.y * t, a.z * t); }
/* vec3.h:65-71 */
static inline rco_vec3 v_addf(rco_vec3 a, float t) { return v3(a.x + t, a.y + t, a.z + t); }
/* vec3.h:153-156: x*u.x + y*u.y + z*u.z, left-associated */
static inline float v_dot(rco_vec3 a, rco_vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* vec3.h:158-161 */
static inline rco_vec3 v_cross(rco_vec3 a, rco_vec3 u)
{
    return v3(a.y * u.z - a.z * u.y, a.z * u.x - a.x * u.z, a.x * u.y - a.y * u.x);
}
/* vec3.h:173-176: sqrt of the left-associated sum of squares (float overload of sqrt) */
static inline float v_length(rco_vec3 a) { return sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z); }

/* vec3.h:168-171 -> operator/=(float) at vec3.h:89-92: ONE reciprocal, three multiplies */
rco_vec3 rco_norm(rco_vec3 v)
{
    float inv = 1.0f / v_length(v);
    return v_scale(v, inv);
}

/* ---- kernel.cu:44-50 ---------------------------------------------------------------- */
rco_vec3 rco_cast_screen_ray(const rco_camera *c, int x, int y, int width, int height)
{
    float xAngle = c->fov * (1.0f * (float)x / (float)width - 0.5f);
    float yAngle = c->fov * (1.0f * (float)y / (float)height - 0.5f);
    /* forward + (right * sin(xa)) * aspect + up * sin(ya), left to right */
    rco_vec3 t = v_scale(v_scale(c->right, sinf(xAngle)), c->aspect);
    rco_vec3 s = v_add(c->forward, t);
    rco_vec3 u = v_scale(c->up, sinf(yAngle));
    return rco_norm(v_add(s, u));
}

/* ---- kernel.cu:52-67 ---------------------------------------------------------------- */
float rco_find_intersection(rco_vec3 pos, float r, rco_vec3 o, rco_vec3 d)
{
    rco_vec3 v = v3(o.x - pos.x, o.y - pos.y, o.z - pos.z);
    float b = ((2.0f * v.x) * d.x + (2.0f * v.y) * d.y) + (2.0f * v.z) * d.z;
    float c = v_dot(v, v) - (r * r);

    float dist = -1.0f;
    float discriminant = b * b - 4.0f * c;
    if (discriminant >= 0.0f) {
        float tmp = sqrtf(discriminant);
        dist = ((-b - tmp) / 2.0f) - 0.000001f;
        if (dist < 0.0f)
            dist = ((-b + tmp) / 2.0f) - 0.000001f;
    }
    return dist;
}

/* ---- kernel.cu:21-42 ---------------------------------------------------------------- */
rco_vec3 rco_sphere_color(rco_vec3 position, rco_vec3 color, rco_vec3 point)
{
    rco_vec3 normal = v_sub(point, position);      /* not normalised (kernel.cu:25) */
    rco_vec3 c = v3(0.2f, 0.2f, 0.2f);             /* AMBIENT 0.2 -> float */
    rco_vec3 toLight;
    float diff;

    toLight = rco_norm(v3(1.0f, 0.0f, -0.5f));
    diff = fmaxf(v_dot(toLight, normal), 0.0f);
    c = v_addf(c, (float)((double)diff * 0.5));    /* DIFFUSE is a double constant */

    toLight = rco_norm(v3(-1.0f, 0.0f, -0.5f));
    diff = fmaxf(v_dot(toLight, normal), 0.0f);
    c = v_addf(c, (float)((double)diff * 0.5));

    toLight = rco_norm(v3(0.0f, 1.0f, -0.5f));
    diff = fmaxf(v_dot(toLight, normal), 0.0f);
    c = v_addf(c, (float)((double)diff * 0.5));

    return v_mul(color, c);
}

/* ---- kernel.cu:69-112 --------------------------------------------------------------- */
int rco_draw_pixel(const rco_screen *screen, const rco_scene *scene, int x, int y,
                   int y_offset, int full_height)
{
    float *pixel;
    int row = y - y_offset;
    if (x >= screen->width || row >= screen->height || row < 0 || x < 0) return -1;

    pixel = (float *)((char *)screen->surface + (size_t)row * screen->pitch) + 4 * x;

    rco_vec3 rayDir = rco_cast_screen_ray(&scene->cam, x, y, screen->width, full_height);
    rco_vec3 rayOrigin = scene->cam.position;

    pixel[3] = 1.0f;

    float minDist = 10000.0f;
    int idx = -1;

    for (int i = 0; i < scene->sphere_num; i++) {
        float dist = rco_find_intersection(scene->position[i], scene->radius[i], rayOrigin, rayDir);
        if (dist > 0.0f && dist < minDist) {
            minDist = dist;
            idx = i;
        }
    }

    if (idx >= 0) {
        rco_vec3 p = v_add(rayOrigin, v_scale(rayDir, minDist));
        rco_vec3 col = rco_sphere_color(scene->position[idx], scene->color[idx], p);
        pixel[0] = col.x;
        pixel[1] = col.y;
        pixel[2] = col.z;
    } else {
        pixel[0] = 0.0f;
        pixel[1] = 0.0f;
        pixel[2] = 0.0f;
    }
    return idx;
}

/* ---- EXT mode (definition in rc_oracle.h; unpinned by the reference) ------------------ */

/* vec3.h:163-166: u * this.dot(u) / u.length() / u.length() * 2 - this, left to right;
 * operator/(float t) multiplies by (1 / t) (vec3.h:131-134) */
static inline rco_vec3 v_reflect(rco_vec3 self, rco_vec3 u)
{
    rco_vec3 t = v_scale(u, v_dot(self, u));
    t = v_scale(t, 1.0f / v_length(u));
    t = v_scale(t, 1.0f / v_length(u));
    t = v_scale(t, 2.0f);
    return v_sub(t, self);
}

int rco_draw_pixel_ext(const rco_screen *screen, const rco_scene *scene, const rco_ext *ext,
                       int x, int y, int y_offset, int full_height)
{
    int row = y - y_offset;
    if (x >= screen->width || row >= screen->height || row < 0 || x < 0) return -1;
    float *pixel = (float *)((char *)screen->surface + (size_t)row * screen->pitch) + 4 * x;

    rco_vec3 d = rco_cast_screen_ray(&scene->cam, x, y, screen->width, full_height);
    rco_vec3 o = scene->cam.position;
    pixel[3] = 1.0f;

    float minDist = 10000.0f;
    int idx = -1;
    for (int i = 0; i < scene->sphere_num; i++) {
        float dist = rco_find_intersection(scene->position[i], scene->radius[i], o, d);
        if (dist > 0.0f && dist < minDist) { minDist = dist; idx = i; }
    }
    float denom = 0.0f;
    if (ext->has_plane) {
        denom = v_dot(ext->plane_normal, d);
        if (denom != 0.0f) {
            float t = ((ext->plane_h - v_dot(ext->plane_normal, o)) / denom) - 0.000001f;
            if (t > 0.0f && t < minDist) { minDist = t; idx = -2; }
        }
    }
    if (idx == -1) {
        pixel[0] = 0.0f; pixel[1] = 0.0f; pixel[2] = 0.0f;
        return idx;
    }

    rco_vec3 P = v_add(o, v_scale(d, minDist));
    rco_vec3 N, base;
    if (idx >= 0) {
        N = rco_norm(v_sub(P, scene->position[idx]));
        base = scene->color[idx];
    } else {
        N = rco_norm(ext->plane_normal);
        if (denom > 0.0f) N = v_scale(N, -1.0f);
        base = ext->plane_color;
    }
    rco_vec3 Lv = v_sub(ext->light_pos, P);
    float lightDist = v_length(Lv);
    float invL = 1.0f / lightDist;
    rco_vec3 Ldir = v_scale(Lv, invL);

    int shadowed = 0;
    if (ext->shadows) {
        rco_vec3 e = v_scale(v_sub(P, ext->light_pos), invL);
        float limit = lightDist - 0.001f;
        for (int i = 0; i < scene->sphere_num; i++) {
            if (i == idx) continue;
            float dist = rco_find_intersection(scene->position[i], scene->radius[i], ext->light_pos, e);
            if (dist > 0.0f && dist < limit) shadowed = 1;
        }
        if (ext->has_plane && idx != -2) {
            float de = v_dot(ext->plane_normal, e);
            if (de != 0.0f) {
                float t = ((ext->plane_h - v_dot(ext->plane_normal, ext->light_pos)) / de) - 0.000001f;
                if (t > 0.0f && t < limit) shadowed = 1;
            }
        }
    }

    float c = 0.2f;
    float spec = 0.0f;
    if (!shadowed) {
        float diff = fmaxf(v_dot(N, Ldir), 0.0f);
        c = c + (float)((double)diff * 0.5);
        rco_vec3 R = v_reflect(Ldir, N);
        float s = fmaxf(v_dot(R, v_scale(d, -1.0f)), 0.0f);
        for (int k = 0; k < ext->shininess_log2; k++) s = s * s;
        spec = ext->specular * s;
    }
    pixel[0] = base.x * c + spec;
    pixel[1] = base.y * c + spec;
    pixel[2] = base.z * c + spec;
    return idx;
}

/* ---- row-split driver (the reference's grid, kernel.cu:120-123, minus the floor) ----- */
typedef struct {
    const rco_screen *screen;
    const rco_scene *scene;
    const rco_ext *ext;          /* NULL: REF mode */
    int y_offset, full_height, row_begin, row_end;
    int tid, threads;
} rco_job;

static void *rco_worker(void *arg)
{
    rco_job *j = (rco_job *)arg;
    /* interleaved rows: near-equal load whatever the hit distribution */
    for (int row = j->row_begin + j->tid; row < j->row_end; row += j->threads)
        for (int x = 0; x < j->screen->width; x++) {
            if (j->ext)
                rco_draw_pixel_ext(j->screen, j->scene, j->ext, x, row + j->y_offset, j->y_offset, j->full_height);
            else
                rco_draw_pixel(j->screen, j->scene, x, row + j->y_offset, j->y_offset, j->full_height);
        }
    return NULL;
}

void rco_render_rows_ext(const rco_screen *screen, const rco_scene *scene, const rco_ext *ext,
                         int y_offset, int full_height, int row_begin, int row_end, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    rco_job *jobs = (rco_job *)calloc((size_t)threads, sizeof(rco_job));
    pthread_t *tids = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    for (int t = 0; t < threads; t++) {
        rco_job j = {screen, scene, ext, y_offset, full_height, row_begin, row_end, t, threads};
        jobs[t] = j;
    }
    for (int t = 1; t < threads; t++) pthread_create(&tids[t], NULL, rco_worker, &jobs[t]);
    rco_worker(&jobs[0]);
    for (int t = 1; t < threads; t++) pthread_join(tids[t], NULL);
    free(jobs);
    free(tids);
}

void rco_render_rows(const rco_screen *screen, const rco_scene *scene, int y_offset,
                     int full_height, int row_begin, int row_end, int threads)
{
    rco_render_rows_ext(screen, scene, NULL, y_offset, full_height, row_begin, row_end, threads);
}

void rco_render(const rco_screen *screen, const rco_scene *scene, int threads)
{
    rco_render_rows(screen, scene, 0, screen->height, 0, screen->height, threads);
}

/* ---- Camera.h:43-55,71-91,101-112 --------------------------------------------------- */
rco_camera rco_camera_data(rco_vec3 position, rco_vec3 rot, int width, int height)
{
    rco_camera d;
    d.aspect = (float)(1.0 * width / height);                 /* Camera.h:53, double divide */
    d.fov = (90.f / 360.0f) * 3.1415926536f;                  /* Camera.h:6,54 */
    d.position = position;
    d.forward = v3(-cosf(rot.x) * sinf(rot.y), sinf(rot.x), -cosf(rot.x) * cosf(rot.y));
    d.up = v3(sinf(rot.x) * sinf(rot.y), cosf(rot.x), cosf(rot.y) * sinf(rot.x));
    d.right = rco_norm(v_cross(d.forward, d.up));             /* Camera.h:89-91 */
    return d;
}

/* ---- scene generator ---------------------------------------------------------------- */
typedef struct { uint32_t mt[624]; int idx; } rco_mt;

static void mt_seed(rco_mt *s, uint32_t seed)
{
    s->mt[0] = seed;
    for (int i = 1; i < 624; i++)
        s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
}

static uint32_t mt_next(rco_mt *s)
{
    if (s->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            uint32_t y = (s->mt[i] & 0x80000000u) | (s->mt[(i + 1) % 624] & 0x7fffffffu);
            uint32_t v = s->mt[(i + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            s->mt[i] = v;
        }
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

static float mt_uniform(rco_mt *s, float lo, float hi)
{
    float u = (float)(mt_next(s) >> 8) * 0x1p-24f;
    return lo + u * (hi - lo);
}

void rco_generate_scene(uint32_t seed, int n, rco_vec3 *position, float *radius, rco_vec3 *color)
{
    rco_mt s;
    mt_seed(&s, seed);
    for (int i = 0; i < n; i++) {                 /* SceneManager.cpp:35-45 draw order */
        radius[i] = mt_uniform(&s, 0.3f, 1.0f);
        color[i].x = mt_uniform(&s, 0.0f, 1.0f);
        color[i].y = mt_uniform(&s, 0.0f, 1.0f);
        color[i].z = mt_uniform(&s, 0.0f, 1.0f);
        position[i].x = mt_uniform(&s, -10.0f, 10.0f);
        position[i].y = mt_uniform(&s, -5.0f, 5.0f);
        position[i].z = mt_uniform(&s, -15.0f, -5.0f);
    }
}

/* ---- display conversion -------------------------------------------------------------- */
static inline uint8_t unorm8(float f)
{
    if (!(f == f)) return 0;        /* NaN -> 0 */
    if (f <= 0.0f) return 0;
    if (f >= 1.0f) return 255;
    return (uint8_t)(int)(f * 255.0f + 0.5f);
}

void rco_to_rgba8(const rco_screen *screen, uint8_t *out, size_t out_pitch, int flip)
{
    for (int y = 0; y < screen->height; y++) {
        const float *src = (const float *)((const char *)screen->surface + (size_t)y * screen->pitch);
        int oy = flip ? screen->height - 1 - y : y;
        uint8_t *dst = out + (size_t)oy * out_pitch;
        for (int x = 0; x < screen->width * 4; x++) dst[x] = unorm8(src[x]);
    }
}

uint64_t rco_fnv1a_surface(const rco_screen *screen)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (int y = 0; y < screen->height; y++) {
        const unsigned char *p = (const unsigned char *)screen->surface + (size_t)y * screen->pitch;
        size_t n = (size_t)screen->width * 16;
        for (size_t i = 0; i < n; i++) {
            h ^= p[i];
            h *= 0x100000001b3ull;
        }
    }
    return h;
}

uint64_t rco_fnv1a64_words(const rco_screen *screen, uint64_t basis)
{
    uint64_t h = basis;
    for (int y = 0; y < screen->height; y++) {
        const uint32_t *p = (const uint32_t *)((const char *)screen->surface + (size_t)y * screen->pitch);
        size_t n = (size_t)screen->width * 4;
        for (size_t i = 0; i < n; i++) {
            h ^= p[i];
            h *= 0x100000001b3ull;
        }
    }
    return h;
}

void rco_hit_mask(const rco_screen *screen, const rco_scene *scene, int y_offset,
                  int full_height, uint8_t *mask_bits, int32_t *index_out)
{
    size_t npx = (size_t)screen->width * (size_t)screen->height;
    if (mask_bits) memset(mask_bits, 0, (npx + 7) / 8);
    for (int row = 0; row < screen->height; row++) {
        for (int x = 0; x < screen->width; x++) {
            rco_vec3 d = rco_cast_screen_ray(&scene->cam, x, row + y_offset, screen->width, full_height);
            float minDist = 10000.0f;
            int idx = -1;
            for (int i = 0; i < scene->sphere_num; i++) {
                float dist = rco_find_intersection(scene->position[i], scene->radius[i],
                                                   scene->cam.position, d);
                if (dist > 0.0f && dist < minDist) { minDist = dist; idx = i; }
            }
            size_t k = (size_t)row * screen->width + x;
            if (index_out) index_out[k] = idx;
            if (mask_bits && idx >= 0) mask_bits[k >> 3] |= (uint8_t)(1u << (k & 7));
        }
    }
}
