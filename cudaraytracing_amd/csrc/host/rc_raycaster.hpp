// rc_raycaster.hpp -- counterpart of the reference's RayTracingKernel (RayTracingKernel.{h,cpp})
// without the D3D11 interop: the frame lives in a plain pitched device buffer that is read back
// to host memory or written as a PPM instead of being copied into an ID3D11Texture2D.
#pragma once
#include <cstdio>
#include <vector>

#include "rc_abi.h"
#include "rc_scene_manager.hpp"

namespace rc {

class RayTracingKernel {
public:
    rc_screen screen{};
    SceneManager sm;
    rc_opts opts{};

    // RayTracingKernel.cpp:34-45
    bool Init(int width, int height, int sphere_num = 32 * 16, unsigned seed = 0)
    {
        screen.width = width;
        screen.height = height;
        if (!FindDevice()) return false;
        sm.InitScene(width, height, sphere_num, seed);
        return true;
    }

    // RayTracingKernel.cpp:47-82
    bool FindDevice(bool print = true)
    {
        int count = 0;
        check(rc_device_count(&count), "rc_device_count");
        if (count == 0) {
            if (print) std::printf("> There are no HIP devices\n");
            return false;
        }
        if (print) std::printf("> Found %d HIP device(s)\n", count);
        for (int d = 0; d < count && print; d++) {
            char name[256];
            check(rc_device_name(d, name, sizeof(name)), "rc_device_name");
            std::printf("> GPU %d: %s\n", d, name);
        }
        return true;
    }

    // RayTracingKernel.cpp:84-95 minus the D3D registration: pitched float-RGBA surface,
    // filled with the byte 1 as the reference does.
    void RegisterSurface()
    {
        ReleaseSurface();
        check(rc_malloc_pitch(&screen.surface, &screen.pitch, (size_t)screen.width * 16, (size_t)screen.height),
              "rc_malloc_pitch");
        check(rc_memset(screen.surface, 1, screen.pitch * screen.height, nullptr), "rc_memset");
    }

    // RayTracingKernel.cpp:8-32: launch, then wait (the interop map/copy/unmap is gone).
    void Run(bool sync = true)
    {
        check(rc_render(&screen, &sm.scene, &opts), "rc_render");
        if (sync) check(rc_sync(opts.stream), "rc_sync");
    }

    // Replaces cudaMemcpy2DToArray into the D3D texture (RayTracingKernel.cpp:22-27).
    std::vector<float> ReadSurface() const
    {
        std::vector<float> host((size_t)screen.width * screen.height * 4);
        check(rc_download_2d(host.data(), (size_t)screen.width * 16, screen.surface, screen.pitch,
                             (size_t)screen.width * 16, (size_t)screen.height, nullptr), "rc_download_2d");
        return host;
    }

    // Display conversion (DXManager.cpp:134,328: float texture drawn into R8G8B8A8_UNORM) on
    // the device, top row first, then a binary PPM.
    void WritePPM(const char *path) const
    {
        void *d8 = nullptr;
        const size_t pitch8 = (size_t)screen.width * 4;
        check(rc_malloc(&d8, pitch8 * screen.height), "rc_malloc(rgba8)");
        std::vector<uint8_t> host(pitch8 * screen.height);
        int rc = rc_to_rgba8(&screen, d8, pitch8, 1, nullptr);
        if (rc == RC_OK) rc = rc_download_2d(host.data(), pitch8, d8, pitch8, pitch8, (size_t)screen.height, nullptr);
        rc_free(d8);                                      // also when a step above failed
        check(rc, "rc_to_rgba8 / rc_download_2d");
        check(rc_write_ppm(path, host.data(), pitch8, screen.width, screen.height), "rc_write_ppm");
    }

    void Cleanup()                                        // RayTracingKernel.cpp:97-102
    {
        ReleaseSurface();
        sm.Cleanup();
    }

    ~RayTracingKernel() { ReleaseSurface(); }

private:
    void ReleaseSurface()
    {
        if (screen.surface) rc_free(screen.surface);
        screen.surface = nullptr;
    }
};

} // namespace rc
