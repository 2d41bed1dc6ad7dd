// dropin_caller.cpp -- calls librc_hip.so's execute_kernel() exactly as the reference's own
// call site does (RayTracingKernel.cpp:3-6,17): the argument types come from the
// REFERENCE'S headers (Screen.h, Scene.h, Camera.h, compiled where they lie under
// /root/reference, untouched), the declaration is the hand-written one of the reference.
// TEST INFRASTRUCTURE ONLY: built by oracle/Makefile into oracle/_ref/dropin_caller; only the
// binary travels to the GPU box, where a gpu test runs it and compares the dumped surface
// with the oracle.  Proves the by-value calling convention of the drop-in symbol.
//
// usage: dropin_caller WIDTH HEIGHT SPHERES SEED OUT.raw   (OUT = tightly packed float RGBA)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "Camera.h"
#include "Scene.h"
#include "Screen.h"

#include "rc_abi.h"

extern "C"
{
	void execute_kernel(Screen screen, Scene scene);
}

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, rc_last_error_string()); return 2; } } while (0)

int main(int argc, char **argv)
{
    if (argc != 6) { std::fprintf(stderr, "usage: %s W H N SEED OUT.raw\n", argv[0]); return 2; }
    int w = std::atoi(argv[1]), h = std::atoi(argv[2]), n = std::atoi(argv[3]);
    unsigned seed = (unsigned)std::strtoul(argv[4], nullptr, 10);

    std::vector<rc_vec3> pos(n), col(n);
    std::vector<float> rad(n);
    rc_generate_scene(seed, n, pos.data(), rad.data(), col.data());

    Scene scene;
    scene.sphere_num = n;
    CHECK(rc_malloc((void **)&scene.position, sizeof(vec3) * n));
    CHECK(rc_malloc((void **)&scene.radius, sizeof(float) * n));
    CHECK(rc_malloc((void **)&scene.color, sizeof(vec3) * n));
    CHECK(rc_upload(scene.position, pos.data(), sizeof(vec3) * n, nullptr));
    CHECK(rc_upload(scene.radius, rad.data(), sizeof(float) * n, nullptr));
    CHECK(rc_upload(scene.color, col.data(), sizeof(vec3) * n, nullptr));

    Camera cam(w, h);                      // the reference's own host camera
    cam.position = vec3(0, 0, 20);
    scene.cam = cam.GetData();

    Screen screen;
    screen.width = w;
    screen.height = h;
    CHECK(rc_malloc_pitch(&screen.surface, &screen.pitch, (size_t)w * 16, (size_t)h));
    CHECK(rc_memset(screen.surface, 1, screen.pitch * h, nullptr));

    execute_kernel(screen, scene);
    CHECK(rc_sync(nullptr));

    std::vector<float> host((size_t)w * h * 4);
    CHECK(rc_download_2d(host.data(), (size_t)w * 16, screen.surface, screen.pitch, (size_t)w * 16, (size_t)h, nullptr));
    FILE *f = std::fopen(argv[5], "wb");
    if (!f) return 2;
    std::fwrite(host.data(), sizeof(float), host.size(), f);
    std::fclose(f);
    rc_free(screen.surface); rc_free(scene.position); rc_free(scene.radius); rc_free(scene.color);
    std::printf("dropin_caller: %dx%d, %d spheres -> %s\n", w, h, n, argv[5]);
    return 0;
}
