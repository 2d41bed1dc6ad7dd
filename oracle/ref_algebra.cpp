// ref_algebra.cpp -- cross-check of the oracle against the REFERENCE'S OWN headers.
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/ref_algebra with
//     -I/root/reference/RayCasting/RayCasting
// so vec3.h, Camera.h, Ray.h, Scene.h and Screen.h are compiled where they lie, untouched
// (their CUDA macro is empty off nvcc, CudaCallableMember.h:3-7).  Nothing of the reference
// is copied into this repository; only the resulting binary travels to the GPU box.
//
// Checked bit for bit on random inputs:
//   * layout of vec3 / CameraData / Scene / Screen against the oracle's (and the ABI's) PODs
//   * vec3::norm                          vs rco_norm
//   * Camera::GetData (Euler -> basis)    vs rco_camera_data
//   * Camera::CastScreenRay               vs rco_cast_screen_ray   (same expression as the
//     device castScreenRay, kernel.cu:44-50; the host version passes forward x up as `right`)
// kernel.cu itself needs the CUDA toolkit headers; its functions (findIntersection, getSphereColor,
// draw) are cross-checked through the reference's own classes by ref_pixel.cpp.

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

#include "Camera.h"
#include "Scene.h"
#include "Screen.h"
#include "vec3.h"

#include "rc_oracle.h"

static_assert(sizeof(vec3) == sizeof(rco_vec3), "vec3 layout");
static_assert(sizeof(CameraData) == sizeof(rco_camera) && sizeof(CameraData) == 56, "CameraData layout");
static_assert(sizeof(Scene) == sizeof(rco_scene) && sizeof(Scene) == 88, "Scene layout");
static_assert(sizeof(Screen) == sizeof(rco_screen) && sizeof(Screen) == 24, "Screen layout");
static_assert(offsetof(Scene, cam) == offsetof(rco_scene, cam), "Scene::cam offset");
static_assert(offsetof(Scene, color) == offsetof(rco_scene, color), "Scene::color offset");
static_assert(offsetof(Screen, pitch) == offsetof(rco_screen, pitch), "Screen::pitch offset");

static bool same_bits(float a, float b)
{
    uint32_t x, y;
    std::memcpy(&x, &a, 4);
    std::memcpy(&y, &b, 4);
    return x == y || (a != a && b != b);
}

static bool same3(const vec3 &a, const rco_vec3 &b)
{
    return same_bits(a.x, b.x) && same_bits(a.y, b.y) && same_bits(a.z, b.z);
}

int main()
{
    std::mt19937 gen(12345);
    std::uniform_real_distribution<float> big(-50.f, 50.f), ang(-3.2f, 3.2f);
    long checks = 0, bad = 0;

    for (int i = 0; i < 200000; i++) {
        vec3 v(big(gen), big(gen), big(gen));
        rco_vec3 w = {v.x, v.y, v.z};
        if (!same3(v.norm(), rco_norm(w))) bad++;
        checks++;
    }

    const int sizes[][2] = {{640, 480}, {1280, 768}, {1920, 1080}, {3840, 2160}, {7680, 4320}, {333, 77}};
    for (auto &wh : sizes) {
        for (int i = 0; i < 2000; i++) {
            vec3 rot = (i == 0) ? vec3(0, 0, 0) : vec3(ang(gen), ang(gen), 0.f);
            vec3 pos(big(gen), big(gen), big(gen));
            Camera cam(wh[0], wh[1]);
            cam.position = pos;
            cam.SetRotation(rot);
            CameraData d = cam.GetData();
            rco_vec3 p = {pos.x, pos.y, pos.z}, r = {rot.x, rot.y, rot.z};
            rco_camera o = rco_camera_data(p, r, wh[0], wh[1]);
            bool ok = same_bits(d.aspect, o.aspect) && same_bits(d.fov, o.fov) && same3(d.position, o.position) &&
                      same3(d.forward, o.forward) && same3(d.up, o.up) && same3(d.right, o.right);
            if (!ok) bad++;
            checks++;

            // Camera::CastScreenRay (Camera.h:57-70) uses right = forward x up, not normalised
            rco_camera c2 = o;
            vec3 cr = d.forward.cross(d.up);
            c2.right.x = cr.x; c2.right.y = cr.y; c2.right.z = cr.z;
            for (int k = 0; k < 16; k++) {
                int x = (int)(gen() % (unsigned)wh[0]), y = (int)(gen() % (unsigned)wh[1]);
                Ray ray = cam.CastScreenRay(x, y);
                rco_vec3 dir = rco_cast_screen_ray(&c2, x, y, wh[0], wh[1]);
                if (!same3(ray.direction, dir) || !same3(ray.origin, o.position)) bad++;
                checks++;
            }
        }
    }
    std::printf("ref_algebra: %ld checks, %ld mismatches\n", checks, bad);
    return bad == 0 ? 0 : 1;
}
