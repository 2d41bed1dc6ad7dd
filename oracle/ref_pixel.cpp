// ref_pixel.cpp -- the oracle's hot functions against the REFERENCE'S OWN vec3 / Camera classes.
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/ref_pixel with
//     -I/root/reference/RayCasting/RayCasting
// so vec3.h, Camera.h, Scene.h and Screen.h are compiled where they lie, untouched; nothing of the
// reference is copied into this repository and no stand-in header is written.
//
// kernel.cu cannot be compiled here (it includes cuda_runtime.h / device_launch_parameters.h, which
// this image lacks).  What CAN be done without it: evaluate the statements of its three device
// functions and of the draw loop through the reference's own operator overloads, constructors and
// implicit conversions -- vec3::dot, operator-, operator+=(float) fed a double product,
// operator*(const vec3&), vec3(...).norm(), operator*(float), operator+, Camera::GetData -- so that
// every conversion and evaluation order those overloads imply is the reference's, not a reading of
// it.  The namespace via_ref below does that in this file's own words, one function per reference
// function, each citing the lines it follows.
//
//   ref_pixel check            >= 10^5 random inputs per function (grazing discriminants included)
//                              and whole pixels of random scenes, bit for bit against the oracle
//   ref_pixel kat FILE N       640x480 frame of the first N spheres of a 7-float-per-sphere scene
//                              file (radius, colour, position) from the reference's default camera:
//                              prints the hit count and the word-wise FNV-1a hash of the surface
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "Camera.h"
#include "Scene.h"
#include "Screen.h"
#include "vec3.h"

#include "rc_oracle.h"

namespace via_ref {

// kernel.cu:52-67: quadratic of a unit-direction ray against one sphere; nearest root in front of
// the origin minus 1e-6, or -1.  Integer literals 2 and 4 as the reference writes them.
static float intersect(vec3 &center, float &radius, vec3 &origin, vec3 &dir)
{
    vec3 v(origin.x - center.x, origin.y - center.y, origin.z - center.z);
    float b = (2 * v.x * dir.x) + (2 * v.y * dir.y) + (2 * v.z * dir.z);
    float c = v.dot(v) - (radius * radius);
    float t = -1;
    float disc = b * b - 4 * c;
    if (disc >= 0) {
        float root = sqrtf(disc);
        t = ((-b - root) / 2) - 0.000001f;
        if (t < 0) t = ((-b + root) / 2) - 0.000001f;
    }
    return t;
}

// kernel.cu:21-42: ambient 0.2 plus three directional Lambert terms with a DOUBLE coefficient 0.5,
// un-normalised normal, colour multiplied componentwise.
static vec3 shade(vec3 &center, vec3 &albedo, vec3 &hit)
{
    static const float lights[3][3] = {{1, 0, -0.5f}, {-1, 0, -0.5f}, {0, 1, -0.5f}};
    vec3 n = hit - center;
    vec3 acc = vec3(0.2, 0.2, 0.2);
    for (const auto &l : lights) {
        vec3 towards = (vec3(l[0], l[1], l[2])).norm();
        float diff = fmaxf(towards.dot(n), 0);
        acc += (diff * 0.5);                    // double product -> operator+=(float)
    }
    return albedo * acc;
}

// kernel.cu:44-50
static vec3 screen_ray(CameraData &c, int &x, int &y, int &width, int &height)
{
    float xa = c.fov * (1.0f * x / width - 0.5f);
    float ya = c.fov * (1.0f * y / height - 0.5f);
    return vec3(c.forward + c.right * sinf(xa) * c.aspect + c.up * sinf(ya)).norm();
}

// kernel.cu:69-112 for one pixel (the block/thread arithmetic replaced by x, y)
static int pixel(Screen &screen, Scene &scene, int x, int y)
{
    if (x >= screen.width || y >= screen.height) return -1;
    float *px = (float *)((char *)screen.surface + y * screen.pitch) + 4 * x;
    vec3 dir = screen_ray(scene.cam, x, y, screen.width, screen.height);
    vec3 origin = scene.cam.position;
    px[3] = 1.0f;
    float best = 10000;
    int winner = -1;
    for (int i = 0; i < scene.sphere_num; i++) {
        float t = intersect(scene.position[i], scene.radius[i], origin, dir);
        if (t > 0 && t < best) {
            best = t;
            winner = i;
        }
    }
    if (winner >= 0) {
        vec3 p = origin + dir * best;
        vec3 col = shade(scene.position[winner], scene.color[winner], p);
        px[0] = col.x; px[1] = col.y; px[2] = col.z;
    } else {
        px[0] = 0.0; px[1] = 0.0f; px[2] = 0.0f;
    }
    return winner;
}

} // namespace via_ref

static bool same_bits(float a, float b)
{
    uint32_t x, y;
    std::memcpy(&x, &a, 4);
    std::memcpy(&y, &b, 4);
    return x == y || (a != a && b != b);
}
static rco_vec3 P(const vec3 &v) { rco_vec3 r = {v.x, v.y, v.z}; return r; }
static bool same3(const vec3 &a, const rco_vec3 &b) { return same_bits(a.x, b.x) && same_bits(a.y, b.y) && same_bits(a.z, b.z); }

static int check()
{
    std::mt19937 gen(20240611);
    std::uniform_real_distribution<float> big(-30.f, 30.f), unit(0.f, 1.f), rad(0.05f, 3.f), ang(-3.2f, 3.2f);
    long checks = 0, bad = 0;

    // getSphereColor: random hit points around random spheres, plus far / tiny normals
    for (int i = 0; i < 200000; i++) {
        vec3 c(big(gen), big(gen), big(gen)), col(unit(gen), unit(gen), unit(gen));
        float s = (i % 5 == 0) ? 1e-4f : (i % 7 == 0 ? 1e3f : 1.0f);
        vec3 hit(c.x + s * (unit(gen) - 0.5f), c.y + s * (unit(gen) - 0.5f), c.z + s * (unit(gen) - 0.5f));
        if (!same3(via_ref::shade(c, col, hit), rco_sphere_color(P(c), P(col), P(hit)))) bad++;
        checks++;
    }

    // findIntersection: random rays, and rays grazing the silhouette (discriminant around zero)
    long hits = 0, grazing_hits = 0, grazing_miss = 0;
    for (int i = 0; i < 400000; i++) {
        vec3 c(big(gen), big(gen), big(gen)), o(big(gen), big(gen), big(gen));
        float r = rad(gen);
        vec3 d;
        if (i & 1) {
            // direction at the tangent angle asin(r/|v|) from the centre, off by a relative epsilon
            double vx = (double)c.x - o.x, vy = (double)c.y - o.y, vz = (double)c.z - o.z;
            double L = std::sqrt(vx * vx + vy * vy + vz * vz);
            if (L <= r) { d = vec3(big(gen), big(gen), big(gen)).norm(); }
            else {
                static const double eps[] = {0, 1e-8, -1e-8, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-3, -1e-3};
                double th = std::asin(r / L) * (1.0 + eps[(i >> 1) % 11]);
                double ax = vx / L, ay = vy / L, az = vz / L;
                // any unit vector perpendicular to a, rotated by a random angle about a
                double px = -ay, py = ax, pz = 0;
                if (std::fabs(ax) < 1e-3 && std::fabs(ay) < 1e-3) { px = 0; py = -az; pz = ay; }
                double pl = std::sqrt(px * px + py * py + pz * pz);
                px /= pl; py /= pl; pz /= pl;
                double qx = ay * pz - az * py, qy = az * px - ax * pz, qz = ax * py - ay * px;
                double phi = 6.283185307179586 * unit(gen);
                double wx = std::cos(phi) * px + std::sin(phi) * qx, wy = std::cos(phi) * py + std::sin(phi) * qy,
                       wz = std::cos(phi) * pz + std::sin(phi) * qz;
                d = vec3((float)(std::cos(th) * ax + std::sin(th) * wx), (float)(std::cos(th) * ay + std::sin(th) * wy),
                         (float)(std::cos(th) * az + std::sin(th) * wz)).norm();
            }
        } else {
            d = vec3(big(gen), big(gen), big(gen)).norm();
            if (i % 4 == 0) d = (c - o).norm();              // straight at the centre
        }
        float a = via_ref::intersect(c, r, o, d);
        float b = rco_find_intersection(P(c), r, P(o), P(d));
        if (!same_bits(a, b)) bad++;
        if (a > 0) { hits++; if (i & 1) grazing_hits++; } else if (i & 1) grazing_miss++;
        checks++;
    }

    // castScreenRay through CameraData of the reference's Camera, arbitrary rotations
    for (int i = 0; i < 100000; i++) {
        int w = 16 + (int)(gen() % 4000), h = 16 + (int)(gen() % 2200);
        Camera cam(w, h);
        cam.position = vec3(big(gen), big(gen), big(gen));
        cam.SetRotation(vec3(ang(gen), ang(gen), 0.f));
        CameraData cd = cam.GetData();
        rco_camera oc;
        std::memcpy(&oc, &cd, sizeof(oc));
        int x = (int)(gen() % (unsigned)w), y = (int)(gen() % (unsigned)h);
        if (!same3(via_ref::screen_ray(cd, x, y, w, h), rco_cast_screen_ray(&oc, x, y, w, h))) bad++;
        checks++;
    }

    // whole pixels: random scenes (overlapping spheres, ties from duplicated spheres, camera inside one)
    long px_hits = 0;
    for (int s = 0; s < 60; s++) {
        const int n = 1 + (int)(gen() % 96), w = 96, h = 64;
        std::vector<vec3> pos(n), col(n);
        std::vector<float> r(n);
        for (int i = 0; i < n; i++) {
            pos[i] = vec3(big(gen) / 3, big(gen) / 6, -5.f - 10.f * unit(gen));
            col[i] = vec3(unit(gen), unit(gen), unit(gen));
            r[i] = 0.3f + 0.7f * unit(gen);
            if (i > 0 && i % 9 == 0) { pos[i] = pos[i - 1]; r[i] = r[i - 1]; }     // exact tie: lowest index wins
        }
        Camera cam(w, h);
        cam.position = (s % 10 == 9) ? vec3(pos[0].x, pos[0].y, pos[0].z + 0.5f * r[0]) : vec3(big(gen) / 10, big(gen) / 10, 20.f);
        cam.SetRotation(vec3(ang(gen) / 20, ang(gen) / 10, 0.f));
        Scene scene;
        scene.sphere_num = n; scene.position = pos.data(); scene.radius = r.data(); scene.color = col.data();
        scene.cam = cam.GetData();
        rco_scene oscene;
        static_assert(sizeof(Scene) == sizeof(rco_scene), "Scene layout");
        std::memcpy((void *)&oscene, (const void *)&scene, sizeof(oscene));
        std::vector<float> a((size_t)w * h * 4, -7.f), b((size_t)w * h * 4, -7.f);
        Screen sa; sa.surface = a.data(); sa.width = w; sa.height = h; sa.pitch = (size_t)w * 16;
        rco_screen sb = {b.data(), w, h, (size_t)w * 16};
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int ia = via_ref::pixel(sa, scene, x, y);
                int ib = rco_draw_pixel(&sb, &oscene, x, y, 0, h);
                if (ia != ib) bad++;
                if (ia >= 0) px_hits++;
                checks++;
            }
        if (std::memcmp(a.data(), b.data(), a.size() * 4) != 0) bad++;
        checks++;
    }
    std::printf("ref_pixel: %ld checks, %ld mismatches (%ld ray hits, grazing rays: %ld hit / %ld miss, %ld hit pixels)\n",
                checks, bad, hits, grazing_hits, grazing_miss, px_hits);
    return bad == 0 ? 0 : 1;
}

static int kat(const char *path, int n)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) { std::fprintf(stderr, "cannot open %s\n", path); return 2; }
    std::vector<float> rows((size_t)n * 7);
    size_t got = std::fread(rows.data(), 4, rows.size(), f);
    std::fclose(f);
    if (got != rows.size()) { std::fprintf(stderr, "%s holds fewer than %d spheres\n", path, n); return 2; }
    std::vector<vec3> pos(n), col(n);
    std::vector<float> r(n);
    for (int i = 0; i < n; i++) {
        const float *q = &rows[(size_t)i * 7];
        r[i] = q[0];
        col[i] = vec3(q[1], q[2], q[3]);
        pos[i] = vec3(q[4], q[5], q[6]);
    }
    const int w = 640, h = 480;
    Camera cam(w, h);                               // SceneManager.cpp:4-5
    cam.position = vec3(0, 0, 20);
    Scene scene;
    scene.sphere_num = n; scene.position = pos.data(); scene.radius = r.data(); scene.color = col.data();
    scene.cam = cam.GetData();
    std::vector<float> surf((size_t)w * h * 4, 0.f);
    Screen screen; screen.surface = surf.data(); screen.width = w; screen.height = h; screen.pitch = (size_t)w * 16;
    long hits = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            if (via_ref::pixel(screen, scene, x, y) >= 0) hits++;
    uint64_t hash = 1469598103934665603ull;
    for (size_t i = 0; i < surf.size(); i++) {
        uint32_t word;
        std::memcpy(&word, &surf[i], 4);
        hash ^= word;
        hash *= 0x100000001b3ull;
    }
    std::printf("n=%d hits=%ld fnv=%016llx\n", n, hits, (unsigned long long)hash);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && std::strcmp(argv[1], "check") == 0) return check();
    if (argc >= 4 && std::strcmp(argv[1], "kat") == 0) return kat(argv[2], std::atoi(argv[3]));
    std::fprintf(stderr, "usage: ref_pixel check | ref_pixel kat SCENE.f32 N\n");
    return 2;
}
