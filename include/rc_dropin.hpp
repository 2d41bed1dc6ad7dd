/*
 * rc_dropin.hpp -- the reference's launch entry point for C++ callers that do NOT use the reference's headers.
 *
 *     extern "C" void execute_kernel(Screen screen, Scene scene);      // RayCasting/RayCasting/kernel.cu:114-115
 *
 * librc_hip.so exports that symbol with C linkage and the C++ by-value convention of the reference's argument
 * classes.  The reference re-declares it by hand at its one call site (RayTracingKernel.cpp:3-6) over its own
 * Screen.h / Scene.h; a caller built against those headers needs nothing from this file.  For any other C++
 * caller the types below are layout- and calling-convention-identical stand-ins:
 *
 *   rc_ref_abi::Vec3        vec3.h:6-18    three floats; the copy constructor is USER-PROVIDED, as at vec3.h:13-18,
 *                                          which is what makes a by-value Scene non-trivial for the purposes of
 *                                          calls (passed by invisible reference under the Itanium C++ ABI)
 *   rc_ref_abi::CameraData  Camera.h:8-19  aspect, fov, position, forward, up, right            (56 bytes)
 *   rc_ref_abi::Scene       Scene.h:4-13   sphere_num, position, radius, color, cam             (88 bytes)
 *   rc_ref_abi::Screen      Screen.h:3-9   surface, width, height, pitch                        (24 bytes, trivially copyable)
 *
 * Behaviour as in the reference: enqueue only on the default stream, void return, a failed launch prints
 * "cuda_kernel_texture_2d() failed to launch error = %d" and returns (kernel.cu:125-130).  The pointers inside
 * Scene and Screen are DEVICE pointers owned by the caller.  C callers, and callers that want row bands, streams,
 * EXT shading or error codes, use rc_render (rc_abi.h) instead.
 */
#ifndef RC_DROPIN_HPP
#define RC_DROPIN_HPP

#include <stddef.h>

namespace rc_ref_abi {

struct Vec3 {
    float x, y, z;
    Vec3() : x(0), y(0), z(0) {}
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    Vec3(const Vec3 &u) { x = u.x; y = u.y; z = u.z; }      // user-provided on purpose (vec3.h:13-18)
    Vec3 &operator=(const Vec3 &u) { x = u.x; y = u.y; z = u.z; return *this; }
};

struct CameraData {
    float aspect, fov;
    Vec3 position, forward, up, right;
};

struct Scene {
    int sphere_num;
    Vec3 *position;
    float *radius;
    Vec3 *color;
    CameraData cam;
};

struct Screen {
    void *surface;
    int width;
    int height;
    size_t pitch;
};

static_assert(sizeof(Vec3) == 12 && sizeof(CameraData) == 56, "vec3.h / Camera.h layouts");
static_assert(sizeof(Scene) == 88 && offsetof(Scene, cam) == 32, "Scene.h layout");
static_assert(sizeof(Screen) == 24 && offsetof(Screen, pitch) == 16, "Screen.h layout");

} // namespace rc_ref_abi

extern "C" void execute_kernel(rc_ref_abi::Screen screen, rc_ref_abi::Scene scene);

#endif /* RC_DROPIN_HPP */
