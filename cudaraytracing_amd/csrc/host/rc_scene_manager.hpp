// rc_scene_manager.hpp -- counterpart of the reference's SceneManager (SceneManager.{h,cpp}):
// owns the camera and the device-resident sphere arrays of an rc_scene.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "rc_abi.h"
#include "rc_camera.hpp"

namespace rc {

// A failed rc_* call as an exception that keeps the call's return code.
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &msg) : std::runtime_error(msg), code(c) {}
};

inline void check(int rc, const char *what)
{
    if (rc != RC_OK) throw Error(rc, std::string(what) + ": " + rc_last_error_string());
}

class SceneManager {
public:
    rc_scene scene{};
    Camera cam;
    std::vector<rc_vec3> h_position, h_color;   // host copies (the reference frees its own)
    std::vector<float> h_radius;

    // SceneManager.cpp:3-29.  The reference hard-codes 32*16 spheres (SceneManager.cpp:8).
    void InitScene(int screen_width, int screen_height, int sphere_num = 32 * 16, unsigned seed = 0)
    {
        Cleanup();
        cam = Camera(screen_width, screen_height);
        cam.position = rc_vec3{0.0f, 0.0f, 20.0f};
        UpdateCamera();
        h_position.resize(sphere_num);
        h_radius.resize(sphere_num);
        h_color.resize(sphere_num);
        rc_generate_scene(seed, sphere_num, h_position.data(), h_radius.data(), h_color.data());
        Upload();
    }

    // Caller-provided spheres instead of the random scene.
    void SetScene(const std::vector<rc_vec3> &position, const std::vector<float> &radius,
                  const std::vector<rc_vec3> &color)
    {
        if (position.size() != radius.size() || color.size() != radius.size())
            throw std::invalid_argument("SetScene: array sizes differ");
        h_position = position; h_radius = radius; h_color = color;
        Upload();
    }

    void UpdateCamera() { scene.cam = cam.GetData(); }   // SceneManager.cpp:55-57

    void Cleanup()                                        // SceneManager.cpp:48-53
    {
        FreeArrays();
        scene.sphere_num = 0;
    }

    ~SceneManager() { FreeArrays(); }

private:
    // SceneManager.cpp:17-24.  The host arrays go to fresh device arrays first; the scene switches to
    // them only once all three are allocated and filled.  A failure part-way frees what it made and
    // leaves an EMPTY scene (sphere_num 0, no arrays), never a half-built one.
    void Upload()
    {
        const size_t n = h_radius.size();
        void *d[3] = {nullptr, nullptr, nullptr};
        const void *src[3] = {h_position.data(), h_radius.data(), h_color.data()};
        const size_t bytes[3] = {sizeof(rc_vec3) * n, sizeof(float) * n, sizeof(rc_vec3) * n};
        static const char *const what[3] = {"position", "radius", "color"};
        FreeArrays();
        scene.sphere_num = 0;
        for (int i = 0; i < 3; i++) {
            int rc = rc_malloc(&d[i], bytes[i]);
            if (rc == RC_OK) rc = rc_upload(d[i], src[i], bytes[i], nullptr);
            if (rc != RC_OK) {
                const std::string msg = std::string("scene upload (") + what[i] + "): " + rc_last_error_string();
                for (int k = 0; k <= i; k++)
                    if (d[k]) rc_free(d[k]);
                h_position.clear(); h_radius.clear(); h_color.clear();
                throw Error(rc, msg);
            }
        }
        scene.position = static_cast<rc_vec3 *>(d[0]);
        scene.radius = static_cast<float *>(d[1]);
        scene.color = static_cast<rc_vec3 *>(d[2]);
        scene.sphere_num = (int)n;
    }
    void FreeArrays()
    {
        if (scene.position) rc_free(scene.position);
        if (scene.radius) rc_free(scene.radius);
        if (scene.color) rc_free(scene.color);
        scene.position = nullptr; scene.radius = nullptr; scene.color = nullptr;
    }
};

} // namespace rc
