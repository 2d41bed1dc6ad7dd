/*
 * bound_model.c -- CPU models of two more conservative devices of the tiled kernels, checked against the
 * oracle's exact test (test infrastructure; built and run by tests/test_filter_model.py).
 *
 *   near   rc_near_bound: "every positive distance kernel.cu:52-67 returns for this sphere is >= bound".
 *          Random spheres, origins and reference-normalised rays (half of them through the sphere's centre
 *          or grazing it, radii from 1e-4 to 1 of the distance); the bound is evaluated as the kernel does,
 *          with the hardware's 1-ulp sqrt and reciprocal modelled by their worst case (sqrt one ulp up,
 *          reciprocal one ulp down: both make the bound larger).  Reports violations (must be 0), the
 *          smallest slack (dist - bound) / |v| and, as a check of the derivation's constant, the largest
 *          observed shortfall of dist against |v| - r - 1e-6 in units of u vv / r (derived <= 12.25 + 2.01 r/|v|).
 *
 *   axis   the dominant-axis form of the EXT shadow filter (rc_shadow_blocks<P, true>): records
 *          (s h_m, h_j, h_k), pixel (u, v, 1/g) with g = s e_m >= 1/2, flagged iff
 *          fma(u, R1, fma(v, R2, R0)) >= il (1 - 2^-21).  For random lights, spheres and shadow rays built
 *          the way rc_ext_ray builds them, every ray whose exact test returns a positive distance must be
 *          flagged for EVERY admissible axis, with the reciprocal one ulp up and one ulp down and k lowered
 *          by K_ULPS ulps.  Reports misses (must be 0) and the smallest margin (T - threshold) / |h|.
 *
 * usage: bound_model near SAMPLES SEED | bound_model axis SAMPLES K_ULPS SEED
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

static float ulps(float x, int n)       /* x moved n ulps away from zero (n < 0: towards zero) */
{
    uint32_t b;
    memcpy(&b, &x, 4);
    b = (uint32_t)((int32_t)b + n);
    memcpy(&x, &b, 4);
    return x;
}

static void unit(double *x, double *y, double *z)
{
    double l;
    do { *x = uni(-1, 1); *y = uni(-1, 1); *z = uni(-1, 1); l = *x * *x + *y * *y + *z * *z; } while (l < 1e-3 || l > 1);
    l = sqrt(l); *x /= l; *y /= l; *z /= l;
}

/* the kernel's rc_near_bound with worst-case hardware sqrt / reciprocal */
static float near_bound_worst(float vv, float r)
{
    const float L = ulps(sqrtf(vv), 1);
    const float ar = fabsf(r);
    const float rc = ulps(1.0f / ar, -1);
    return ((L - ar) - 0x1p-19f * (vv * rc)) - (0x1p-18f * L + 2.0e-6f);
}

static int run_near(long samples)
{
    long hits = 0, violations = 0, usable = 0;
    double min_slack = 1e30, max_shortfall = -1e30;
    const double u = ldexp(1.0, -24);
    for (long s = 0; s < samples; s++) {
        const double scale = pow(10.0, uni(-1, 3));
        rco_vec3 o = {(float)uni(-scale, scale), (float)uni(-scale, scale), (float)uni(-scale, scale)};
        double dx, dy, dz;
        unit(&dx, &dy, &dz);
        const double dist_c = scale * pow(10.0, uni(-1.5, 0.5));
        rco_vec3 pos = {(float)(o.x + dx * dist_c), (float)(o.y + dy * dist_c), (float)(o.z + dz * dist_c)};
        const float r = (float)(dist_c * pow(10.0, uni(-4, -0.02)));
        /* the reference's v and the ray: through the centre, grazing, or anywhere inside the silhouette */
        const float vx = o.x - pos.x, vy = o.y - pos.y, vz = o.z - pos.z;
        const float vv = (vx * vx + vy * vy) + vz * vz;
        double tx, ty, tz;
        unit(&tx, &ty, &tz);
        const int kind = (int)(rnd() * 4);
        const double ang = asin(fmin(1.0, r / sqrt((double)vv)));
        const double off = kind == 0 ? 0.0 : (kind == 1 ? ang * (1.0 - pow(10.0, uni(-7, -2))) : ang * rnd());
        /* direction = centre direction tilted by `off` towards a random perpendicular */
        double cx = -(double)vx, cy = -(double)vy, cz = -(double)vz, cl = sqrt(cx * cx + cy * cy + cz * cz);
        cx /= cl; cy /= cl; cz /= cl;
        double px = ty * cz - tz * cy, py = tz * cx - tx * cz, pz = tx * cy - ty * cx, pl = sqrt(px * px + py * py + pz * pz);
        if (pl < 1e-6) continue;
        px /= pl; py /= pl; pz /= pl;
        const float stretch = (float)uni(0.5, 2.0);                        /* an un-normalised ray ... */
        rco_vec3 q = {(float)(cos(off) * cx + sin(off) * px) * stretch, (float)(cos(off) * cy + sin(off) * py) * stretch,
                      (float)(cos(off) * cz + sin(off) * pz) * stretch};
        const rco_vec3 d = rco_norm(q);                                    /* ... normalised as vec3::norm does */
        const float dist = rco_find_intersection(pos, r, o, d);
        if (!(dist > 0.0f)) continue;
        hits++;
        const float nb = near_bound_worst(vv, r);
        const double L = sqrt((double)vv);
        if (nb > 0.0f) usable++;
        if (dist < nb) violations++;
        const double slack = ((double)dist - (double)nb) / L;
        if (slack < min_slack) min_slack = slack;
        const double shortfall = ((L - (double)r - 1.0000001e-6) - (double)dist) / (u * (double)vv / (double)r);
        if (shortfall > max_shortfall) max_shortfall = shortfall;
    }
    printf("near: samples=%ld hits=%ld usable=%ld violations=%ld min_slack=%.3e max_shortfall_in_u_vv_over_r=%.3f\n",
           samples, hits, usable, violations, min_slack, max_shortfall);
    return violations == 0 ? 0 : 1;
}

static int run_axis(long samples, int k_ulps)
{
    long hits = 0, missed = 0, checks = 0, always = 0, forced = 0;
    double min_margin = 1e30;
    for (long s = 0; s < samples; s++) {
        const double scale = pow(10.0, uni(-1, 2));
        rco_vec3 light = {(float)uni(-scale, scale), (float)uni(-scale, scale), (float)uni(-scale, scale)};
        double dx, dy, dz;
        unit(&dx, &dy, &dz);
        const double dist_c = scale * pow(10.0, uni(-1.5, 0.5));
        rco_vec3 pos = {(float)(light.x + dx * dist_c), (float)(light.y + dy * dist_c), (float)(light.z + dz * dist_c)};
        /* the light from 1e-6 radii off the surface to far away */
        const float r = (float)(dist_c / (1.0 + pow(10.0, uni(-6, 2))));
        /* rc_prepass_light */
        const float vx = light.x - pos.x, vy = light.y - pos.y, vz = light.z - pos.z;
        const float vv = (vx * vx + vy * vy) + vz * vz;
        const float c = vv - r * r;
        const float thr = sqrtf(c) * (1.0f - 0x1p-19f) - 0x1p-17f * sqrtf(vv);
        if (!(c > 0.0f) || !(thr > 0.0f) || !(vv < 0x1p80f) || !(vv > 0x1p-80f)) { always++; continue; }
        const float k = ulps(1.0f / thr, -k_ulps);
        const float h[3] = {-k * vx, -k * vy, -k * vz};
        if (!(fabsf(h[0]) < 0x1p40f) || !(fabsf(h[1]) < 0x1p40f) || !(fabsf(h[2]) < 0x1p40f)) { always++; continue; }
        const double hl = sqrt((double)h[0] * h[0] + (double)h[1] * h[1] + (double)h[2] * h[2]);
        /* a hit point beyond the sphere, seen from the light inside / at the edge of its silhouette */
        const double ang = asin(fmin(1.0, (double)r / sqrt((double)vv)));
        const int kind = (int)(rnd() * 3);
        const double off = kind == 0 ? ang * rnd() : ang * (1.0 + (kind == 1 ? -1.0 : 1.0) * pow(10.0, uni(-7, -2)));
        double tx, ty, tz;
        unit(&tx, &ty, &tz);
        double px = ty * dz - tz * dy, py = tz * dx - tx * dz, pz = tx * dy - ty * dx, pl = sqrt(px * px + py * py + pz * pz);
        if (pl < 1e-6) continue;
        px /= pl; py /= pl; pz /= pl;
        const double far = dist_c * uni(0.2, 4.0);
        rco_vec3 P = {(float)(light.x + far * (cos(off) * dx + sin(off) * px)), (float)(light.y + far * (cos(off) * dy + sin(off) * py)),
                      (float)(light.z + far * (cos(off) * dz + sin(off) * pz))};
        /* rc_ext_ray */
        const float lvx = light.x - P.x, lvy = light.y - P.y, lvz = light.z - P.z;
        const float lightDist = sqrtf((lvx * lvx + lvy * lvy) + lvz * lvz);
        const float invL = 1.0f / lightDist;
        const float e[3] = {(P.x - light.x) * invL, (P.y - light.y) * invL, (P.z - light.z) * invL};
        const float ee = (e[0] * e[0] + e[1] * e[1]) + e[2] * e[2];
        if (!(ee < 1.0f + 0x1p-19f) || !(ee > 1.0f - 0x1p-19f)) { forced++; continue; }
        rco_vec3 ev = {e[0], e[1], e[2]};
        const float dist = rco_find_intersection(pos, r, light, ev);
        if (!(dist > 0.0f)) continue;
        hits++;
        for (int m = 0; m < 3; m++) {
            for (int sg = 0; sg < 2; sg++) {
                const float sign = sg ? -1.0f : 1.0f;
                const float g = sign * e[m];
                if (!(g >= 0.5f)) continue;
                const int j = (m + 1) % 3, kk = (m + 2) % 3;
                for (int du = -1; du <= 1; du += 2) {          /* hardware reciprocal: one ulp either way */
                    const float il = ulps(1.0f / g, du);
                    const float fu = e[j] * il, fv = e[kk] * il, flen = il * (1.0f - 0x1p-21f);
                    const float T = fmaf(fu, h[j], fmaf(fv, h[kk], sign * h[m]));
                    checks++;
                    if (!(T >= flen)) missed++;
                    const double margin = ((double)T - (double)flen) / hl;
                    if (margin < min_margin) min_margin = margin;
                }
            }
        }
    }
    printf("axis: samples=%ld hits=%ld checks=%ld missed=%ld always=%ld forced=%ld min_margin=%.3e\n",
           samples, hits, checks, missed, always, forced, min_margin);
    return missed == 0 ? 0 : 1;
}

int main(int argc, char **argv)
{
    if (argc == 4 && strcmp(argv[1], "near") == 0) {
        rng_state = 0x9E3779B97F4A7C15ull ^ (uint64_t)atoll(argv[3]);
        return run_near(atol(argv[2]));
    }
    if (argc == 5 && strcmp(argv[1], "axis") == 0) {
        rng_state = 0x9E3779B97F4A7C15ull ^ (uint64_t)atoll(argv[4]);
        return run_axis(atol(argv[2]), atoi(argv[3]));
    }
    fprintf(stderr, "usage: %s near SAMPLES SEED | %s axis SAMPLES K_ULPS SEED\n", argv[0], argv[0]);
    return 2;
}
