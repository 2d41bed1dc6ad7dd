/*
 * filter_model.c -- CPU model of the tiled kernel's conservative filter, checked against the
 * oracle's exact test (test infrastructure; built and run by tests/test_filter_model.py).
 *
 * For random scenes and cameras it evaluates, for every (pixel, sphere) pair,
 *   exact :  rco_find_intersection(...) > 0           (the reference's arithmetic)
 *   filter:  T = fma(a, beta, fma(b, gamma, alpha)) >= len   with the record (alpha, beta, gamma)
 *            built as rc_prepass_sphere builds it, except that k is made SMALLER by `k_ulps`
 *            units in the last place -- the worst case of the hardware's 1-ulp sqrt and reciprocal,
 *            because a smaller k flags less.
 * and reports: pairs, exact hits, hits the filter would have missed (must be 0), spheres/lanes that
 * bypass the filter, and the smallest margin (T - len) / len seen over all exact hits.
 *
 * usage: filter_model SCENES WIDTH HEIGHT MAXSPHERES K_ULPS SEED
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rc_oracle.h"

static uint64_t rng_state;
static double rnd(void)
{   /* xorshift64* */
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (double)((rng_state * 2685821657736338717ull) >> 11) / 9007199254740992.0;
}
static double uni(double lo, double hi) { return lo + (hi - lo) * rnd(); }

static float ulp_down(float x, int n)
{
    uint32_t b;
    memcpy(&b, &x, 4);
    if (x > 0) b -= (uint32_t)n; else if (x < 0) b += (uint32_t)n;
    memcpy(&x, &b, 4);
    return x;
}

int main(int argc, char **argv)
{
    if (argc != 7) { fprintf(stderr, "usage: %s SCENES W H MAXN K_ULPS SEED\n", argv[0]); return 2; }
    int scenes = atoi(argv[1]), W = atoi(argv[2]), H = atoi(argv[3]), maxn = atoi(argv[4]), k_ulps = atoi(argv[5]);
    rng_state = 0x9E3779B97F4A7C15ull ^ (uint64_t)atoll(argv[6]);
    long long pairs = 0, hits = 0, missed = 0, always_pairs = 0, forced_pixels = 0;
    double min_margin = 1e30;

    rco_vec3 *pos = malloc(sizeof(rco_vec3) * (size_t)maxn);
    float *rad = malloc(sizeof(float) * (size_t)maxn);
    float *al = malloc(sizeof(float) * (size_t)maxn), *be = malloc(sizeof(float) * (size_t)maxn),
          *ga = malloc(sizeof(float) * (size_t)maxn);
    char *always = malloc((size_t)maxn);

    for (int sc = 0; sc < scenes; sc++) {
        int n = 1 + (int)(rnd() * (maxn - 1));
        double spread = pow(10.0, uni(-1, 2));
        for (int i = 0; i < n; i++) {
            pos[i].x = (float)uni(-spread, spread); pos[i].y = (float)uni(-spread, spread); pos[i].z = (float)uni(-spread, spread);
            rad[i] = (float)(spread * pow(10.0, uni(-3, 0)));
        }
        /* camera: at a random relative distance (down to 1e-7) from a random sphere's surface */
        int ks = (int)(rnd() * n);
        double dx = uni(-1, 1), dy = uni(-1, 1), dz = uni(-1, 1), dl = sqrt(dx * dx + dy * dy + dz * dz) + 1e-30;
        double off = 1.0 + (rnd() < 0.5 ? -1.0 : 1.0) * pow(10.0, uni(-7, 1));
        rco_vec3 cp = {(float)(pos[ks].x + dx / dl * rad[ks] * off), (float)(pos[ks].y + dy / dl * rad[ks] * off),
                       (float)(pos[ks].z + dz / dl * rad[ks] * off)};
        rco_vec3 rot = {(float)uni(-3.2, 3.2), (float)uni(-3.2, 3.2), 0.0f};
        if (sc % 2 == 0) {   /* look at the sphere */
            rot.x = (float)asin(fmax(-1.0, fmin(1.0, -dy / dl)));
            rot.y = (float)atan2(dx / dl, dz / dl);
        }
        rco_camera cam = rco_camera_data(cp, rot, W, H);
        const float ox = cam.position.x, oy = cam.position.y, oz = cam.position.z;
        const double nF = sqrt((double)cam.forward.x * cam.forward.x + (double)cam.forward.y * cam.forward.y + (double)cam.forward.z * cam.forward.z) * (1 + 1e-6);
        const double nR = sqrt((double)cam.right.x * cam.right.x + (double)cam.right.y * cam.right.y + (double)cam.right.z * cam.right.z) * (1 + 1e-6);
        const double nU = sqrt((double)cam.up.x * cam.up.x + (double)cam.up.y * cam.up.y + (double)cam.up.z * cam.up.z) * (1 + 1e-6);

        /* records, as rc_prepass_sphere (binary32, explicit fmaf), k lowered by k_ulps */
        for (int i = 0; i < n; i++) {
            float vx = ox - pos[i].x, vy = oy - pos[i].y, vz = oz - pos[i].z;
            float vv = (vx * vx + vy * vy) + vz * vz;
            float c = vv - rad[i] * rad[i];
            float thr = sqrtf(c) * (1.0f - 0x1p-19f) - 0x1p-17f * sqrtf(vv);
            int aw = !(c > 0.0f) || !(thr > 0.0f) || !(vv < 0x1p80f) || !(vv > 0x1p-80f);
            float k = ulp_down(1.0f / thr, k_ulps);
            float vF = fmaf(vz, cam.forward.z, fmaf(vy, cam.forward.y, vx * cam.forward.x));
            float vR = fmaf(vz, cam.right.z, fmaf(vy, cam.right.y, vx * cam.right.x));
            float vU = fmaf(vz, cam.up.z, fmaf(vy, cam.up.y, vx * cam.up.x));
            al[i] = -k * vF; be[i] = -k * vR; ga[i] = -k * vU;
            aw = aw || !(fabsf(al[i]) < 0x1p40f) || !(fabsf(be[i]) < 0x1p40f) || !(fabsf(ga[i]) < 0x1p40f);
            always[i] = (char)aw;
        }
        for (int y = 0; y < H; y++) {
            for (int x = 0; x < W; x++) {
                float sx = sinf(cam.fov * (1.0f * (float)x / (float)W - 0.5f));
                float sy = sinf(cam.fov * (1.0f * (float)y / (float)H - 0.5f));
                rco_vec3 d = rco_cast_screen_ray(&cam, x, y, W, H);
                /* |q| as the kernel keeps it (the reference's own statements) */
                float tx = cam.right.x * sx, ty = cam.right.y * sx, tz = cam.right.z * sx;
                tx = tx * cam.aspect; ty = ty * cam.aspect; tz = tz * cam.aspect;
                float qx = cam.forward.x + tx, qy = cam.forward.y + ty, qz = cam.forward.z + tz;
                qx = qx + cam.up.x * sy; qy = qy + cam.up.y * sy; qz = qz + cam.up.z * sy;
                float len = sqrtf((qx * qx + qy * qy) + qz * qz);
                float fa = sx * cam.aspect, fb = sy;
                float Q = (float)nF + fabsf(fa) * (float)nR + fabsf(fb) * (float)nU;
                int forced = !(Q <= 4.0f * len) || !(len > 0x1p-20f) || !(len < 0x1p20f);
                forced_pixels += forced;
                for (int i = 0; i < n; i++) {
                    pairs++;
                    float dist = rco_find_intersection(pos[i], rad[i], cam.position, d);
                    int hit = dist > 0.0f;
                    if (always[i]) { always_pairs++; continue; }
                    if (forced) continue;
                    float T = fmaf(fa, be[i], fmaf(fb, ga[i], al[i]));
                    if (hit) {
                        hits++;
                        double m = ((double)T - (double)len) / (double)len;
                        if (m < min_margin) min_margin = m;
                        if (!(T >= len)) missed++;
                    }
                }
            }
        }
    }
    printf("pairs=%lld hits=%lld missed=%lld always_pairs=%lld forced_pixels=%lld min_margin=%.3e\n",
           pairs, hits, missed, always_pairs, forced_pixels, min_margin);
    return missed == 0 ? 0 : 1;
}
