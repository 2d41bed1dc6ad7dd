// rc_camera.hpp -- host camera, the counterpart of the reference's Camera (Camera.h:21-113).
// Position + Euler rotation (z unused, Camera.h:25); the basis and the CameraData block are
// produced by rc_camera_data() of the C ABI, so there is one implementation of the
// Euler -> basis arithmetic (Camera.h:71-91) for C++, Python and the tests.
#pragma once
#include "rc_abi.h"

namespace rc {

class Camera {
public:
    rc_vec3 position{0.0f, 0.0f, 0.0f};
    rc_vec3 rotation{0.0f, 0.0f, 0.0f};
    float movement_speed = 100.0f;   // Camera.h:48
    float rotation_speed = 10.0f;    // Camera.h:49

    Camera() = default;
    Camera(int screen_width, int screen_height) : width_(screen_width), height_(screen_height) {}

    void SetRotation(const rc_vec3 &r) { rotation = r; }
    rc_vec3 GetRotation() const { return rotation; }

    // Camera.h:101-112
    rc_camera GetData() const
    {
        rc_camera d;
        rc_camera_data(&position, &rotation, width_, height_, &d);
        return d;
    }
    rc_vec3 GetForward() const { return GetData().forward; }
    rc_vec3 GetUp() const { return GetData().up; }
    rc_vec3 GetRight() const { return GetData().right; }

    // One poll of the interactive controls (main.cpp:53-128) from a scripted input state, in
    // the reference's arithmetic; dt in milliseconds (main.cpp:137).  True if the camera
    // changed (the caller then runs SceneManager::UpdateCamera, main.cpp:126-127).
    bool HandleInput(const rc_input &in, float dt)
    {
        return rc_handle_input(&position, &rotation, movement_speed, rotation_speed, dt, &in) == 1;
    }

    // Free-form scripted moves: along the camera axes by world units / rotate by radians.
    void Move(float forward, float right, float up)
    {
        const rc_camera d = GetData();
        position.x += d.forward.x * forward + d.right.x * right + d.up.x * up;
        position.y += d.forward.y * forward + d.right.y * right + d.up.y * up;
        position.z += d.forward.z * forward + d.right.z * right + d.up.z * up;
    }
    void Rotate(float pitch, float yaw) { rotation.x += pitch; rotation.y += yaw; }

    int width() const { return width_; }
    int height() const { return height_; }

private:
    int width_ = 1, height_ = 1;
};

} // namespace rc
