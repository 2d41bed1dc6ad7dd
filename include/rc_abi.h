/*
 * rc_abi.h -- C ABI of librc_hip.so, the MI355X (gfx950) ray-casting hot path.
 *
 * Plain C: pointers, sizes and POD structs only.  Every entry point cites the reference
 * interface it replaces (paths relative to RayCasting/RayCasting/ of McThrok/CUDARayTracing).
 *
 * Conventions
 *   - All rc_* functions return 0 on success or a negative value: -(hipError_t) for a HIP
 *     failure, RC_EINVAL for a rejected argument.  They never exit() and never throw.
 *     rc_last_error_string() describes the calling thread's last failure.
 *   - rc_render()/execute_kernel() only ENQUEUE work (the reference launch at kernel.cu:123
 *     is asynchronous and its caller synchronises at RayTracingKernel.cpp:18).
 *   - The caller owns every buffer it passes, as in the reference (scene arrays:
 *     SceneManager.cpp:17-24,49-51; surface: RayTracingKernel.cpp:93,100).
 *   - Surface format: float RGBA, 16 bytes per pixel, row pitch >= 16*width bytes and a
 *     multiple of 16 (kernel.cu:78,83,102-109; DXGI_FORMAT_R32G32B32A32_FLOAT,
 *     RayTracingKernel.cpp:92-93).  Row 0 is the BOTTOM of the picture (kernel.cu:47).
 */
#ifndef RC_ABI_H
#define RC_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct layout, an enum value or the meaning of an argument changes; bindings compare
 * it with rc_abi_version() of the library they loaded and refuse a mismatch. */
#define RC_ABI_VERSION 3

#define RC_OK 0
#define RC_EINVAL (-100001)
#define RC_ENOTBUILT (-100002)

/* ---- argument structs: layout-identical PODs of the reference classes ---------------- */

/* vec3.h:6-10 (12 bytes) */
typedef struct rc_vec3 { float x, y, z; } rc_vec3;

/* Camera.h:8-19 CameraData (56 bytes) */
typedef struct rc_camera {
    float aspect;
    float fov;
    rc_vec3 position;
    rc_vec3 forward;
    rc_vec3 up;
    rc_vec3 right;
} rc_camera;

/* Scene.h:4-13 (88 bytes; position/radius/color are DEVICE pointers) */
typedef struct rc_scene {
    int sphere_num;
    rc_vec3 *position;
    float *radius;
    rc_vec3 *color;
    rc_camera cam;
} rc_scene;

/* Screen.h:3-9 (24 bytes; surface is a DEVICE pointer) */
typedef struct rc_screen {
    void *surface;
    int width;
    int height;
    size_t pitch;
} rc_screen;

/* Kernel variants.  All produce bit-identical surfaces. */
enum {
    RC_VARIANT_AUTO = 0,     /* library picks (the tiled kernel)                          */
    RC_VARIANT_LITERAL = 1,  /* one pixel per lane, every test evaluated as kernel.cu:52-67 */
    RC_VARIANT_TILED = 2,    /* LDS-staged scene, FMA filter + exact re-test (default);
                                pixels per lane chosen from the frame size                 */
    RC_VARIANT_TILED_P2 = 3, /* tuning: the tiled kernel forced to 2 pixels per lane       */
    RC_VARIANT_TILED_P8 = 4, /* tuning: forced to 8 pixels per lane (REF mode; EXT uses 4)  */
    RC_VARIANT_TILED_P4 = 5, /* tuning: forced to 4 pixels per lane                         */
    RC_VARIANT_CULLED = 6    /* opt-in: interval tests per workgroup tile, then per wave tile, in
                                place of the per-pixel loop; spheres that cannot touch a tile
                                are never tested against its pixels.  Same surfaces, but no
                                longer one test per (pixel, sphere) pair.                     */
};

/* rc_opts.tuning: overrides of launcher heuristics, for A/B measurements and for tests that must reach
 * both sides of a heuristic on small scenes.  Every combination produces the same surfaces. */
enum {
    RC_TUNE_SHADOW_AXIS_ON = 1,   /* EXT: use the dominant-axis shadow filter whatever the sphere count  */
    RC_TUNE_SHADOW_AXIS_OFF = 2,  /* EXT: never use it (default: from 1024 spheres up)                    */
    RC_TUNE_NO_LDS_EXACT = 4,     /* do not stage the exact-test terms in LDS beside the filter records  */
    RC_TUNE_NO_NEAR_SKIP = 8      /* run the exact test also for pixels whose nearest hit so far is already
                                     closer than the sphere's nearest point                               */
};

/* Shading modes. */
enum {
    RC_MODE_REF = 0,   /* the reference's math: spheres, ambient + three directional Lambert
                          lights (kernel.cu:21-42).  Parity is graded on this mode.          */
    RC_MODE_EXT = 1    /* + one plane, one point light, Phong, hard shadows.  The reference
                          has no such code: this mode is defined by this library (and its
                          CPU oracle) and is NOT pinned by the reference.                    */
};

/* EXT-mode scene additions (no counterpart in the reference; 56 bytes).
 *   primary search  spheres as in REF, then the plane n.x = h last:
 *                   t = ((h - n.o) / (n.d)) - 1e-6f, accepted if n.d != 0 and 0 < t < minDist
 *   shadow          ray from the light to the hit point; shadowed if a sphere other than the
 *                   one hit (or the plane, unless the plane was hit) lies closer than
 *                   |light - point| - 0.001
 *   colour          base * (0.2 + 0.5 max(N.L, 0)) + specular * max(R.V, 0)^(2^shininess_log2),
 *                   R = L.reflect(N) as vec3.h:163-166; ambient only when shadowed        */
typedef struct rc_ext {
    rc_vec3 plane_normal;   /* need not be a unit vector */
    float plane_h;
    rc_vec3 plane_color;
    rc_vec3 light_pos;
    float specular;
    int shininess_log2;     /* 0..30 */
    int has_plane;
    int shadows;
} rc_ext;

/* What the reference launch lacks (kernel.cu:114-131 has no options). Zero-initialise. */
typedef struct rc_opts {
    int y_offset;        /* frame row stored in surface row 0 (row-band rendering)        */
    int full_height;     /* frame height used in the y angle (kernel.cu:47); 0 = height   */
    int variant;         /* RC_VARIANT_*                                                   */
    int mode;            /* RC_MODE_*                                                      */
    void *stream;        /* hipStream_t; NULL = the default stream as at kernel.cu:123     */
    void *rgba8;         /* optional DEVICE R8G8B8A8 surface written with the same frame   */
    size_t rgba8_pitch;  /* bytes per rgba8 row (>= 4*width)                               */
    int rgba8_flip;      /* != 0: surface row 0 goes to the LAST rgba8 row (top-down image)*/
    int tuning;          /* RC_TUNE_* bits; 0 = the library's own choices.  Never changes a pixel.  */
    const rc_ext *ext;   /* HOST pointer, required when mode == RC_MODE_EXT                */
    void *workspace;     /* optional DEVICE scratch, 16-byte aligned, rc_workspace_bytes(n) or more:
                            RC_VARIANT_CULLED then builds the sphere records once per frame in it
                            (a small kernel ahead of the draw kernel on the same stream) instead of
                            once per workgroup.  Ignored by the other variants.  Frames in flight
                            at the same time need a workspace each.                              */
    size_t workspace_bytes;
} rc_opts;

/* bytes of rc_opts.workspace needed for a scene of sphere_num spheres (16 per sphere) */
size_t rc_workspace_bytes(int sphere_num);
/* bytes that additionally let RC_VARIANT_CULLED list, per bin of 4 x 4 workgroup tiles, the spheres
 * that can touch the bin (a third small kernel), so that a workgroup walks its bin's list instead of
 * the whole scene; for frames of this size.  A workspace between the two sizes gets the records only. */
size_t rc_workspace_bytes_frame(int sphere_num, int width, int height);

/* ---- the hot path ---------------------------------------------------------------------- */

/* Replaces execute_kernel (kernel.cu:114-131, declared by hand at RayTracingKernel.cpp:3-6,
 * called at RayTracingKernel.cpp:17).  Renders EVERY pixel of the screen (the reference's
 * floor-divided grid at kernel.cu:121 drops the remainder rows/columns).  Enqueue only.
 * Width, height and full_height may be at most 65536; sphere_num is unlimited. */
int rc_render(const rc_screen *screen, const rc_scene *scene, const rc_opts *opts);

/* Builds, for the current device, the cached sine tables rc_render needs for frames of this
 * width / full height / field of view (kernel.cu:46-47), so that later rc_render calls with the
 * same three values allocate and copy nothing (e.g. before capturing them into a HIP graph).
 * No counterpart in the reference, which evaluates sinf per pixel on the device. */
int rc_prepare(int width, int full_height, float fov);

/* rc_render for callers that cannot allocate device memory: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  It has NO counterpart in the reference.  Same
 * arguments, same frames as rc_render, but the device scratch the library needs -- the sine tables
 * and, for RC_VARIANT_CULLED, the per-frame sphere records -- lives in a fixed array inside the
 * library's code object (256 KiB, held by every GPU the library runs on until the process ends), so
 * the library itself calls no hipMalloc / hipFree and no allocating library routine (what the HIP runtime does
 * internally to stage a table upload from pageable memory on a format change is the runtime's business).
 * Limits (RC_EINVAL beyond them): width + full_height <= 32768; RC_VARIANT_CULLED: sphere_num <= 8192;
 * device index <= 15.  opts->workspace is ignored.
 * All calls on one GPU share that GPU's array, which keeps the last call's contents; nothing in it has
 * to start at zero (tables and records are rewritten in full before they are read).  When the frame
 * format (width, full_height, fov) differs from the previous call's on that GPU, the new tables are
 * uploaded on opts->stream and the call waits for that stream once (so such a call cannot be captured
 * into a graph: render one frame of the format first).  A captured call is replayed with the tables' address,
 * not their contents, so from the first captured call on, the captured frame format stays the only one this
 * entry point serves on that GPU for the life of the process: a call with another format returns RC_EINVAL
 * instead of overwriting the tables under the graph (rc_render, whose cache pins captured tables, serves any
 * number of formats).  Calls that overlap on one GPU -- different
 * streams or threads without ordering between them -- race on the array unless they use the same
 * frame format and none uses RC_VARIANT_CULLED. */
int rc_render_static(const rc_screen *screen, const rc_scene *scene, const rc_opts *opts);

/* Display step after the path (DXManager.cpp:134,299,328: the float texture drawn 1:1 into
 * an R8G8B8A8_UNORM target): NaN->0, clamp [0,1], *255 + 0.5, truncate.  Device to device. */
int rc_to_rgba8(const rc_screen *screen, void *rgba8, size_t rgba8_pitch, int flip, void *stream);

/* ---- host-side helpers that mirror what RayTracingKernel / SceneManager do -------------- */

/* RayTracingKernel.cpp:47-82 findCUDADevice (count + names, no printing) */
int rc_device_count(int *count);
int rc_device_name(int device, char *name, size_t name_len);
int rc_set_device(int device);
/* number of compute units and max clock of `device` (for roofline reporting) */
int rc_device_info(int device, int *compute_units, int *clock_khz, size_t *lds_per_block);

/* RayTracingKernel.cpp:93 cudaMallocPitch: pitch = width_bytes rounded up to 256 */
int rc_malloc_pitch(void **ptr, size_t *pitch, size_t width_bytes, size_t height);
/* SceneManager.cpp:17,20,23 cudaMalloc */
int rc_malloc(void **ptr, size_t bytes);
/* SceneManager.cpp:49-51, RayTracingKernel.cpp:100 cudaFree */
int rc_free(void *ptr);
/* RayTracingKernel.cpp:94 cudaMemset */
int rc_memset(void *ptr, int value, size_t bytes, void *stream);
/* SceneManager.cpp:18,21,24 cudaMemcpy host->device */
int rc_upload(void *dst_device, const void *src_host, size_t bytes, void *stream);
/* replaces the D3D interop copy (RayTracingKernel.cpp:22-27): pitched device->host */
int rc_download_2d(void *dst_host, size_t dst_pitch, const void *src_device, size_t src_pitch,
                   size_t width_bytes, size_t height, void *stream);
/* RayTracingKernel.cpp:18 cudaDeviceSynchronize (stream == NULL: whole device) */
int rc_sync(void *stream);

/* ---- frame pipeline: replaces the reference's render -> device sync -> full-frame copy ->
 * present sequence (RayTracingKernel.cpp:17-27, main.cpp:47-48) with double-buffered
 * surfaces whose read-back overlaps the next frame's render.  None of these has a
 * counterpart in the reference, which uses the default stream only (kernel.cu:123). */
int rc_stream_create(void **stream);
int rc_stream_destroy(void *stream);
/* work submitted to `stream` after this call waits for `event` (recorded on another stream) */
int rc_stream_wait_event(void *stream, void *event);
/* page-locked host memory, so that rc_download_2d_async really is asynchronous */
int rc_host_alloc(void **ptr, size_t bytes);
int rc_host_free(void *ptr);
/* rc_download_2d without the wait: returns once the copy is enqueued on `stream` */
int rc_download_2d_async(void *dst_host, size_t dst_pitch, const void *src_device, size_t src_pitch,
                         size_t width_bytes, size_t height, void *stream);

/* HIP graphs for launch-bound sequences (small frames: a 640x480 x 8-sphere frame renders in less
 * time than one launch costs).  Everything enqueued on `stream` between rc_graph_begin and
 * rc_graph_end -- rc_render with any scene and camera (the arguments are baked into each node),
 * rc_to_rgba8, rc_download_2d_async -- is recorded instead of executed; rc_graph_launch replays the
 * whole sequence with one submission.  `stream` must be a stream from rc_stream_create (the default
 * stream cannot be captured), and rc_prepare (or one earlier rc_render) must have built the sine
 * tables of every frame format used inside the capture, because a capture cannot allocate.  A graph
 * node keeps the address of its format's tables, so a format used inside a capture stays in the table
 * cache (at most 32 formats per device) for the rest of the process, also after rc_graph_destroy.  No
 * counterpart in the reference, which launches frame by frame (RayTracingKernel.cpp:17). */
int rc_graph_begin(void *stream);
int rc_graph_end(void *stream, void **graph_exec);
int rc_graph_launch(void *graph_exec, void *stream);
int rc_graph_destroy(void *graph_exec);

/* HIP events on the stream the kernels run on (the reference only has a wall clock,
 * main.cpp:45-50) */
int rc_event_create(void **event);
int rc_event_record(void *event, void *stream);
int rc_event_elapsed_ms(void *start, void *stop, float *ms); /* waits for `stop` */
int rc_event_wait(void *event);                              /* blocks the host until `event` has happened */
int rc_event_destroy(void *event);

/* Camera.h:43-55,71-91,101-112: position + Euler rotation -> CameraData */
void rc_camera_data(const rc_vec3 *position, const rc_vec3 *rotation, int width, int height,
                    rc_camera *out);

/* One poll of the interactive controls, scripted (main.cpp:53-128 reads the same state from
 * GetKeyState/GetCursorPos).  Non-zero = held down. */
typedef struct rc_input {
    int rbutton;             /* VK_RBUTTON: mouse look (main.cpp:64)                          */
    int lshift;              /* VK_LSHIFT: speed x5 instead of x50 (main.cpp:75-80)           */
    int key_q, key_z;        /* up / down along world y (main.cpp:82-92)                      */
    int key_w, key_a, key_s, key_d; /* forward / left / back / right in the xz plane (:94-124) */
    int cursor_dx, cursor_dy;/* cursor movement since the previous poll (main.cpp:67,71)      */
} rc_input;

/* main.cpp:53-128 HandleInput: applies one poll to a camera's position and Euler rotation, in
 * the reference's order and arithmetic (rotation first, then q, z, w, a, s, d; every move is
 * norm(axis with y = 0) * speed * dt).  dt is the frame time in MILLISECONDS, as the reference
 * passes it (main.cpp:137).  Returns 1 if the camera changed (the reference then calls
 * SceneManager::UpdateCamera, main.cpp:126-127), 0 if not, RC_EINVAL for a null pointer. */
int rc_handle_input(rc_vec3 *position, rc_vec3 *rotation, float movement_speed, float rotation_speed,
                    float dt, const rc_input *input);

/* SceneManager.cpp:14-15,31-46 with a portable float mapping: std::mt19937(seed)
 * (SceneManager.h:22), u = (r >> 8) * 2^-24, value = lo + u * (hi - lo); per sphere
 * radius [0.3,1], colour [0,1]^3, x [-10,10], y [-5,5], z [-15,-5].  HOST arrays. */
void rc_generate_scene(uint32_t seed, int n, rc_vec3 *position, float *radius, rc_vec3 *color);

/* Binary PPM (P6) of a host RGBA8 image whose first row is the top of the picture. */
int rc_write_ppm(const char *path, const uint8_t *rgba8, size_t pitch, int width, int height);

/* ---- context: what the reference's RayTracingKernel + SceneManager objects do on the host ----
 * One context owns a device, a camera, the device-resident sphere arrays and a pitched float-RGBA
 * surface, as RayTracingKernel owns `screen` and `sm` (RayTracingKernel.h).  A convenience layer over
 * the calls above for C and FFI callers; the C++ classes in csrc/host/ are the same layer in C++. */
typedef struct rc_ctx rc_ctx;
/* Every rc_ctx_* call runs with the context's device current and puts the caller's current device
 * back before it returns.  A failed rc_scene_generate / rc_scene_upload leaves the context with an
 * empty scene (0 spheres), never a partly uploaded one. */
/* RayTracingKernel::Init (RayTracingKernel.cpp:34-45) without the scene: picks `device`, records the
 * frame size, camera at (0,0,20) looking down -z (SceneManager.cpp:4-6). */
int rc_ctx_create(rc_ctx **ctx, int device, int width, int height);
/* RayTracingKernel::Cleanup + SceneManager::Cleanup (RayTracingKernel.cpp:97-102, SceneManager.cpp:48-53) */
int rc_ctx_destroy(rc_ctx *ctx);
/* SceneManager::InitScene's random spheres + upload (SceneManager.cpp:8-29), portable generator */
int rc_scene_generate(rc_ctx *ctx, int sphere_num, uint32_t seed);
/* the same upload (SceneManager.cpp:17-24) with the caller's HOST arrays */
int rc_scene_upload(rc_ctx *ctx, int sphere_num, const rc_vec3 *position, const float *radius, const rc_vec3 *color);
/* Camera position / rotation + SceneManager::UpdateCamera (SceneManager.cpp:55-57) */
int rc_camera_set(rc_ctx *ctx, const rc_vec3 *position, const rc_vec3 *rotation);
/* one HandleInput poll (main.cpp:53-128) on the context's camera; returns 1 if it moved */
int rc_camera_input(rc_ctx *ctx, const rc_input *input, float dt);
/* RegisterTexture's cudaMallocPitch + cudaMemset(1) (RayTracingKernel.cpp:93-94); pitch may be NULL */
int rc_surface_alloc(rc_ctx *ctx, size_t *pitch);
/* RayTracingKernel::Run's launch (RayTracingKernel.cpp:17); enqueue only, opts may be NULL.  With
 * opts->variant == RC_VARIANT_CULLED and no opts->workspace the context lends one of its own, so
 * renders of one context must then not overlap on different streams. */
int rc_ctx_render(rc_ctx *ctx, const rc_opts *opts);
/* replaces the interop copy of Run (RayTracingKernel.cpp:18-27): waits, then copies the float frame
 * (row 0 = bottom) / its RGBA8 conversion (flip != 0: top row first) to HOST memory */
int rc_readback(rc_ctx *ctx, float *dst_host, size_t dst_pitch);
int rc_readback_rgba8(rc_ctx *ctx, uint8_t *dst_host, size_t dst_pitch, int flip);
/* the frame as a binary PPM, top row first */
int rc_ctx_write_ppm(rc_ctx *ctx, const char *path);
/* the structs the context would pass to rc_render (device pointers inside; owned by the context) */
const rc_screen *rc_ctx_screen(const rc_ctx *ctx);
const rc_scene *rc_ctx_scene(const rc_ctx *ctx);

/* Pixels per lane (1, 2, 4 or 8) of the kernel rc_render launches for `rows` rows rendered in one call (a band's
 * rows; full_height: the whole frame's, as in rc_opts, 0 = rows), or a negative code.  The reference's launch shape is fixed, one pixel per
 * thread in 32 x 16 blocks (kernel.cu:120-123); here it depends on the frame, the sphere count and the CU count
 * (0: 256), and callers that price the executed work or name the kernel in a profile need to know it.  Host only: no
 * device is touched. */
int rc_pixels_per_lane(int width, int rows, int full_height, int sphere_num, int mode, int variant, int compute_units);

/* Diagnostic, no counterpart in the reference.  The kernels evaluate some correctly rounded square roots and
 * reciprocals (vec3::norm, vec3.h:168-176,89-92) with sequences shorter than the compiler's general ones, valid for
 * normal operands with 2^-96 <= |x| < 2^96 (others take the general sequence).  This runs both on the `count`
 * operands whose bit patterns are first_bits, first_bits + 1, ... and ADDS the number of operands on which they
 * differ to *mismatches_dev (device memory, 8 bytes, zeroed by the caller); the two result arrays (device, `count`
 * floats each) are optional.  which: 0 = square root, 1 = reciprocal.  Enqueue only. */
int rc_selftest_math(int which, uint32_t first_bits, uint32_t count, unsigned long long *mismatches_dev,
                     float *out_fast_dev, float *out_general_dev, void *stream);

const char *rc_last_error_string(void);
/* RC_ABI_VERSION the library was built with */
int rc_abi_version(void);
/* "rc_hip <version> gfx950" */
const char *rc_version(void);

#ifdef __cplusplus
} /* extern "C" */
#endif

/*
 * Source-level drop-in for the reference entry point:
 *
 *     extern "C" void execute_kernel(Screen screen, Scene scene);     // kernel.cu:114-115
 *
 * is exported by librc_hip.so with C linkage and the C++ by-value convention of the
 * reference types (Scene embeds vec3, whose copy constructor is user-provided at
 * vec3.h:13-18, so a by-value Scene is passed by invisible reference).  It is declared in
 * include/rc_dropin.hpp for C++ callers that do not use the reference's own headers; a caller
 * that does (RayTracingKernel.cpp:3-6) needs no header from this library at all.
 */
#endif /* RC_ABI_H */
