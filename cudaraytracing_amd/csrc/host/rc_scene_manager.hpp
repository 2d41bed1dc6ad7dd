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
        scene.sphere_num = sphere_num;
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
        FreeArrays();
        h_position = position; h_radius = radius; h_color = color;
        scene.sphere_num = (int)radius.size();
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
    void Upload()                                         // SceneManager.cpp:17-24
    {
        const size_t n = (size_t)scene.sphere_num;
        check(rc_malloc((void **)&scene.position, sizeof(rc_vec3) * n), "rc_malloc(position)");
        check(rc_malloc((void **)&scene.radius, sizeof(float) * n), "rc_malloc(radius)");
        check(rc_malloc((void **)&scene.color, sizeof(rc_vec3) * n), "rc_malloc(color)");
        check(rc_upload(scene.position, h_position.data(), sizeof(rc_vec3) * n, nullptr), "rc_upload(position)");
        check(rc_upload(scene.radius, h_radius.data(), sizeof(float) * n, nullptr), "rc_upload(radius)");
        check(rc_upload(scene.color, h_color.data(), sizeof(rc_vec3) * n, nullptr), "rc_upload(color)");
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
