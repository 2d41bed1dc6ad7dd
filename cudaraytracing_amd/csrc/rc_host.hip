// rc_host.hip -- C ABI of librc_hip.so (declared in include/rc_abi.h) and the source-level
// drop-in execute_kernel().  Host code only; the kernels live in rc_kernels.hip.
//
// This file is built with -ffp-contract=off: the sine tables below must be evaluated
// exactly as the reference writes the angle (kernel.cu:46-47), one rounding per operation.

#include <hip/hip_runtime.h>

#include <math.h>
#include <cmath>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <shared_mutex>
#include <random>
#include <vector>

#include "rc_abi.h"
#include "rc_dropin.hpp"
#include "rc_kernels.h"

#pragma clang fp contract(off)

static_assert(sizeof(rc_vec3) == 12, "vec3.h:6-10 layout");
static_assert(sizeof(rc_camera) == 56, "Camera.h:8-19 layout");
static_assert(sizeof(rc_scene) == 88, "Scene.h:4-13 layout");
static_assert(sizeof(rc_screen) == 24, "Screen.h:3-9 layout");
static_assert(sizeof(rc_ext) == 56, "rc_ext layout");
static_assert(sizeof(rc_opts) == 72 && offsetof(rc_opts, stream) == 16 && offsetof(rc_opts, ext) == 48 &&
              offsetof(rc_opts, workspace) == 56, "rc_opts layout (mirrored by the ctypes binding)");
static_assert(sizeof(rc_input) == 40, "rc_input layout");
static_assert(RC_TUNE_SHADOW_AXIS_ON == RC_KTUNE_SHADOW_AXIS_ON && RC_TUNE_SHADOW_AXIS_OFF == RC_KTUNE_SHADOW_AXIS_OFF &&
              RC_TUNE_NO_LDS_EXACT == RC_KTUNE_NO_LDS_EXACT && RC_TUNE_NO_NEAR_SKIP == RC_KTUNE_NO_NEAR_SKIP, "tuning bits");
static_assert(offsetof(rc_scene, position) == 8 && offsetof(rc_scene, radius) == 16 &&
              offsetof(rc_scene, color) == 24 && offsetof(rc_scene, cam) == 32, "Scene.h field offsets");
static_assert(offsetof(rc_screen, width) == 8 && offsetof(rc_screen, height) == 12 &&
              offsetof(rc_screen, pitch) == 16, "Screen.h field offsets");

namespace {

thread_local char g_err[256] = "";
constexpr int kMinBinnedSpheres = 2048;  // below this the binning launch costs more than it saves (measured)
constexpr int kMaxDim = 65536;      // frame width / height limit (keeps every grid within range)

int rc_fail_hip(hipError_t e, const char *what)
{
    snprintf(g_err, sizeof(g_err), "%s: %s (hipError %d)", what, hipGetErrorString(e), (int)e);
    return -(int)e;
}

int rc_fail_arg(const char *what)
{
    snprintf(g_err, sizeof(g_err), "invalid argument: %s", what);
    return RC_EINVAL;
}

#define RC_HIP(call)                                      \
    do {                                                  \
        hipError_t e_ = (call);                           \
        if (e_ != hipSuccess) return rc_fail_hip(e_, #call); \
    } while (0)

// ---- sine tables (kernel.cu:46-49) ---------------------------------------------------
// sin_x[x] = sinf(fov * (1.0f * x / width - 0.5f)), sin_y likewise over full_height.
// Entries are immutable once built.  A render holds the cache's shared lock from the lookup
// until its kernel is enqueued; building and evicting take the exclusive lock, and eviction
// waits for the device before freeing, so no frame ever reads a freed table.
struct SinTables {
    int device;
    uint32_t fov_bits;
    int width, full_height;
    float *d_base;   // [width + full_height]
    uint64_t last_use;
    int pinned;      // looked up while a stream was being captured: a graph node holds d_base, never evicted
};

std::shared_mutex g_mu;
std::vector<SinTables> g_tables;
uint64_t g_tick = 0;
constexpr size_t kMaxTables = 32;   // per device

// caller holds g_mu (shared or exclusive)
bool rc_find_tables(int dev, uint32_t fb, int width, int full_height, const float **sx, const float **sy, bool pin = false)
{
    for (auto &t : g_tables) {
        if (t.device == dev && t.fov_bits == fb && t.width == width && t.full_height == full_height) {
            __atomic_store_n(&t.last_use, __atomic_add_fetch(&g_tick, 1, __ATOMIC_RELAXED), __ATOMIC_RELAXED);
            if (pin) __atomic_store_n(&t.pinned, 1, __ATOMIC_RELAXED);
            *sx = t.d_base;
            *sy = t.d_base + width;
            return true;
        }
    }
    return false;
}

// caller holds g_mu exclusively
int rc_build_tables(int dev, uint32_t fb, float fov, int width, int full_height)
{
    const float *sx, *sy;
    if (rc_find_tables(dev, fb, width, full_height, &sx, &sy)) return RC_OK;   // another thread was first
    size_t mine = 0;
    for (const auto &t : g_tables) mine += (t.device == dev);
    if (mine >= kMaxTables) {
        // evict this device's least recently used entry once the frames that read it have finished;
        // tables a captured graph may replay at any later time are never evicted
        size_t victim = g_tables.size();
        for (size_t i = 0; i < g_tables.size(); i++)
            if (g_tables[i].device == dev && !g_tables[i].pinned &&
                (victim == g_tables.size() || g_tables[i].last_use < g_tables[victim].last_use))
                victim = i;
        if (victim == g_tables.size()) {
            snprintf(g_err, sizeof(g_err), "all %zu cached frame formats of this device are held by captured graphs", kMaxTables);
            return RC_EINVAL;
        }
        RC_HIP(hipDeviceSynchronize());
        RC_HIP(hipFree(g_tables[victim].d_base));
        g_tables.erase(g_tables.begin() + (long)victim);
    }
    std::vector<float> h((size_t)width + (size_t)full_height);
    for (int x = 0; x < width; x++)
        h[(size_t)x] = sinf(fov * (1.0f * (float)x / (float)width - 0.5f));
    for (int y = 0; y < full_height; y++)
        h[(size_t)width + (size_t)y] = sinf(fov * (1.0f * (float)y / (float)full_height - 0.5f));
    SinTables t{dev, fb, width, full_height, nullptr, ++g_tick, 0};
    RC_HIP(hipMalloc(reinterpret_cast<void **>(&t.d_base), h.size() * sizeof(float)));
    hipError_t e = hipMemcpy(t.d_base, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(t.d_base);
        return rc_fail_hip(e, "hipMemcpy(sin tables)");
    }
    g_tables.push_back(t);
    return RC_OK;
}

// Looks the tables up, building them first if needed.  On success `lease` holds the shared
// lock: the caller releases it after the kernel that reads the tables has been enqueued.
int rc_get_tables(float fov, int width, int full_height, const float **sx, const float **sy,
                  std::shared_lock<std::shared_mutex> &lease, hipStream_t stream)
{
    int dev = 0;
    RC_HIP(hipGetDevice(&dev));
    uint32_t fb;
    memcpy(&fb, &fov, 4);
    // A kernel node recorded into a graph keeps the table's address for as long as the graph lives,
    // so a table looked up during a capture is pinned (exempt from eviction) from then on.
    bool capturing = false;
    if (stream) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        capturing = hipStreamIsCapturing(stream, &st) == hipSuccess && st == hipStreamCaptureStatusActive;
    }
    for (int attempt = 0; attempt < 8; attempt++) {
        lease = std::shared_lock<std::shared_mutex>(g_mu);
        if (rc_find_tables(dev, fb, width, full_height, sx, sy, capturing)) return RC_OK;
        lease.unlock();
        if (capturing)      // building allocates and copies: not possible while the stream is being captured
            return rc_fail_arg("frame format not prepared: call rc_prepare(width, full_height, fov) before capturing");
        std::unique_lock<std::shared_mutex> build(g_mu);
        int rc = rc_build_tables(dev, fb, fov, width, full_height);
        if (rc != RC_OK) return rc;
    }
    snprintf(g_err, sizeof(g_err), "sine table cache thrashing: more than %zu frame formats in use at once", kMaxTables);
    return RC_EINVAL;
}

// compute units per device, queried once (sizes the tiled kernel's grid)
int rc_compute_units(int dev, int *cus)
{
    static int cache[64];
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 0 && dev < 64 && cache[dev] > 0) {
        *cus = cache[dev];
        return RC_OK;
    }
    int v = 0;
    RC_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    if (dev >= 0 && dev < 64) cache[dev] = v;
    *cus = v;
    return RC_OK;
}

} // namespace

// ---- hot path ------------------------------------------------------------------------------

namespace {

// Argument checks shared by rc_render and rc_render_static; fills the effective options.
int rc_check_render_args(const rc_screen *screen, const rc_scene *scene, const rc_opts *opts, rc_opts &o,
                         int &full_height, int &kv)
{
    if (!screen || !scene) return rc_fail_arg("null screen or scene");
    if (!screen->surface) return rc_fail_arg("null surface");
    if (screen->width <= 0 || screen->height <= 0) return rc_fail_arg("width/height must be positive");
    if (screen->width > kMaxDim || screen->height > kMaxDim || (opts && opts->full_height > kMaxDim))
        return rc_fail_arg("width/height/full_height above 65536");
    if (screen->pitch < (size_t)screen->width * 16 || (screen->pitch & 15) != 0)
        return rc_fail_arg("pitch must be >= 16*width and a multiple of 16");
    if ((reinterpret_cast<uintptr_t>(screen->surface) & 15) != 0) return rc_fail_arg("surface must be 16-byte aligned");
    if (scene->sphere_num < 0) return rc_fail_arg("sphere_num < 0");
    if (scene->sphere_num > 0 && (!scene->position || !scene->radius || !scene->color))
        return rc_fail_arg("null scene array");

    memset(&o, 0, sizeof(o));
    if (opts) o = *opts;
    full_height = o.full_height > 0 ? o.full_height : screen->height;
    if (o.y_offset < 0 || (long long)o.y_offset + screen->height > full_height)
        return rc_fail_arg("row band [y_offset, y_offset+height) exceeds full_height");
    if (o.rgba8 && (o.rgba8_pitch < (size_t)screen->width * 4 || (o.rgba8_pitch & 3) != 0 ||
                    (reinterpret_cast<uintptr_t>(o.rgba8) & 3) != 0))
        return rc_fail_arg("rgba8 surface must be 4-byte aligned with pitch >= 4*width, multiple of 4");

    if (o.mode != RC_MODE_REF && o.mode != RC_MODE_EXT) return rc_fail_arg("unknown mode");
    if (o.tuning & ~(RC_TUNE_SHADOW_AXIS_ON | RC_TUNE_SHADOW_AXIS_OFF | RC_TUNE_NO_LDS_EXACT | RC_TUNE_NO_NEAR_SKIP))
        return rc_fail_arg("unknown tuning bits");
    if (o.mode == RC_MODE_EXT) {
        if (!o.ext) return rc_fail_arg("mode EXT needs opts->ext");
        if (o.ext->shininess_log2 < 0 || o.ext->shininess_log2 > 30) return rc_fail_arg("shininess_log2 must be 0..30");
    }

    switch (o.variant) {
    case RC_VARIANT_AUTO:
    case RC_VARIANT_TILED: kv = RC_KVARIANT_AUTO; break;
    case RC_VARIANT_LITERAL: kv = RC_KVARIANT_LITERAL; break;
    case RC_VARIANT_TILED_P2: kv = RC_KVARIANT_TILED2; break;
    case RC_VARIANT_TILED_P8: kv = RC_KVARIANT_TILED8; break;
    case RC_VARIANT_TILED_P4: kv = RC_KVARIANT_TILED4; break;
    case RC_VARIANT_CULLED: kv = RC_KVARIANT_AUTO; break;
    default: return rc_fail_arg("unknown variant");
    }
    return RC_OK;
}

// Fills the kernel argument block and enqueues the frame.  sx / sy: device sine tables of this
// frame format; workspace (device, may be null): scratch of the culled variant.
int rc_enqueue_frame(const rc_screen *screen, const rc_scene *scene, const rc_opts &o, int kv,
                     const float *sx, const float *sy, void *workspace, size_t workspace_bytes)
{
    RcKernArgs a;
    memset(&a, 0, sizeof(a));
    a.surface = screen->surface;
    a.pitch = screen->pitch;
    a.width = screen->width;
    a.rows = screen->height;
    a.y_offset = o.y_offset;
    a.n = scene->sphere_num;
    a.sin_x = sx;
    a.sin_y = sy;
    a.position = reinterpret_cast<const float *>(scene->position);
    a.radius = scene->radius;
    a.color = reinterpret_cast<const float *>(scene->color);
    a.rgba8 = o.rgba8;
    a.rgba8_pitch = o.rgba8_pitch;
    a.rgba8_flip = o.rgba8_flip;
    a.cull = (o.variant == RC_VARIANT_CULLED) ? 1 : 0;
    a.tuning = o.tuning;
    a.aspect = scene->cam.aspect;
    a.ox = scene->cam.position.x; a.oy = scene->cam.position.y; a.oz = scene->cam.position.z;
    a.fx = scene->cam.forward.x;  a.fy = scene->cam.forward.y;  a.fz = scene->cam.forward.z;
    a.ux = scene->cam.up.x;       a.uy = scene->cam.up.y;       a.uz = scene->cam.up.z;
    a.rx = scene->cam.right.x;    a.ry = scene->cam.right.y;    a.rz = scene->cam.right.z;

    if (o.mode == RC_MODE_EXT) {
        const rc_ext &x = *o.ext;
        a.ext_mode = 1;
        a.has_plane = x.has_plane ? 1 : 0;
        a.shadows = x.shadows ? 1 : 0;
        a.shininess_log2 = x.shininess_log2;
        a.pnx = x.plane_normal.x; a.pny = x.plane_normal.y; a.pnz = x.plane_normal.z; a.plane_h = x.plane_h;
        a.pcx = x.plane_color.x; a.pcy = x.plane_color.y; a.pcz = x.plane_color.z;
        a.lgx = x.light_pos.x; a.lgy = x.light_pos.y; a.lgz = x.light_pos.z;
        a.specular = x.specular;
        // per-frame constants, with the per-pixel code's own statements (this file is built without contraction;
        // sqrtf and / are correctly rounded on the host as they are on the device)
        const float inv = 1.0f / sqrtf((a.pnx * a.pnx + a.pny * a.pny) + a.pnz * a.pnz);
        a.pux = a.pnx * inv; a.puy = a.pny * inv; a.puz = a.pnz * inv;
        a.pu_il = 1.0f / sqrtf((a.pux * a.pux + a.puy * a.puy) + a.puz * a.puz);
        a.plane_num = a.plane_h - ((a.pnx * a.ox + a.pny * a.oy) + a.pnz * a.oz);
        a.plane_num_light = a.plane_h - ((a.pnx * a.lgx + a.pny * a.lgy) + a.pnz * a.lgz);
        // which pixels can hit the plane at all (see the tiled kernel): only for a normal with ONE non-zero, finite
        // component and a finite, non-zero numerator
        const float pn[3] = {a.pnx, a.pny, a.pnz};
        int nonzero = 0, axis = 0;
        for (int k = 0; k < 3; k++)
            if (!(pn[k] == 0.0f)) { nonzero++; axis = k; }           // NaN counts as non-zero
        a.plane_axis = 0;
        a.plane_side = 0.0f;
        if (a.has_plane && nonzero == 1 && std::isfinite(pn[axis]) && std::isfinite(a.plane_num) && a.plane_num != 0.0f) {
            a.plane_axis = axis;
            a.plane_side = ((a.plane_num > 0.0f) == (pn[axis] > 0.0f)) ? 1.0f : -1.0f;
        }
    }

    {
        // range in which the tiled kernel's filter bounds are proven; outside it every lane
        // takes the exact path (same results, slower)
        auto norm_up = [](float x, float y, float z) {
            return (float)(sqrt((double)x * x + (double)y * y + (double)z * z) * (1.0 + 1e-6));
        };
        a.nF = norm_up(a.fx, a.fy, a.fz);
        a.nR = norm_up(a.rx, a.ry, a.rz);
        a.nU = norm_up(a.ux, a.uy, a.uz);
        auto in_range = [](float v) { return v > 0x1p-20f && v < 0x1p20f; };
        a.cam_ok = (in_range(a.nF) && in_range(a.nR) && in_range(a.nU) && fabsf(a.aspect) < 0x1p20f &&
                    fabsf(a.ox) < 0x1p60f && fabsf(a.oy) < 0x1p60f && fabsf(a.oz) < 0x1p60f) ? 1 : 0;
        // Near-orthogonal basis (the reference's Camera always produces one): every lane's own check
        //     Q = |F| + |a||R| + |b||U| <= 4 len,   2^-20 < len < 2^20,   len = |F + aR + bU| as rounded,
        // holds for all a, b the tables can supply, so the kernels skip it.  With pairwise |cos| <= d = 2^-10:
        //     len^2 >= (F^2 + a^2 R^2 + b^2 U^2) - d (2|a||F||R| + 2|b||F||U| + 2|a||b||R||U|) >= (1 - 2d)(F^2 + a^2 R^2 + b^2 U^2)
        // (2xy <= x^2 + y^2 on each pair) and Q <= sqrt(3) sqrt(F^2 + a^2 R^2 + b^2 U^2), hence Q <= 1.74 len in real
        // numbers; the float evaluations of Q (inputs rounded up by 1e-6) and of len (error <= 8 u Q) move that by
        // parts in 10^6.  len >= 0.99 |F| and len <= Q <= |F| + |aspect||R| + |U| because |sin| <= 1 for a finite fov.
        if (a.cam_ok) {
            auto dotd = [](float x0, float y0, float z0, float x1, float y1, float z1) {
                return fabs((double)x0 * x1 + (double)y0 * y1 + (double)z0 * z1);
            };
            const double d = 0x1p-10;
            const double qmax = (double)a.nF + fabs((double)a.aspect) * a.nR + (double)a.nU;
            if (std::isfinite(scene->cam.fov) && a.nF > 0x1p-18f && qmax < 0x1p18 &&
                dotd(a.fx, a.fy, a.fz, a.rx, a.ry, a.rz) <= d * a.nF * a.nR &&
                dotd(a.fx, a.fy, a.fz, a.ux, a.uy, a.uz) <= d * a.nF * a.nU &&
                dotd(a.rx, a.ry, a.rz, a.ux, a.uy, a.uz) <= d * a.nR * a.nU)
                a.cam_ok = 2;
        }
    }

    int dev = 0, cus = 0;
    RC_HIP(hipGetDevice(&dev));
    int rc = rc_compute_units(dev, &cus);
    if (rc != RC_OK) return rc;
    if (a.cull && workspace && a.n > 0) {
        if ((reinterpret_cast<uintptr_t>(workspace) & 15) != 0 || workspace_bytes < rc_workspace_bytes(a.n))
            return rc_fail_arg("workspace must be 16-byte aligned and hold rc_workspace_bytes(sphere_num) bytes");
        float4 *records = static_cast<float4 *>(workspace);
        hipError_t er = rc_launch_records(a, records, static_cast<hipStream_t>(o.stream));
        if (er != hipSuccess) return rc_fail_hip(er, "record kernel launch");
        a.records = records;
        // level 0 (per-bin sphere lists) when the workspace also has room for it
        const size_t rec_bytes = (rc_workspace_bytes(a.n) + 255) & ~(size_t)255;
        if (workspace_bytes > rec_bytes && a.n >= kMinBinnedSpheres) {
            a.bins = static_cast<const unsigned char *>(workspace) + rec_bytes;
            a.bins_bytes = workspace_bytes - rec_bytes;
        }
    }
    hipError_t e = rc_launch_draw(a, kv, o.full_height > 0 ? o.full_height : screen->height, cus,
                                  static_cast<hipStream_t>(o.stream));
    if (e != hipSuccess) return rc_fail_hip(e, "draw kernel launch");
    return RC_OK;
}

} // namespace

extern "C" int rc_render(const rc_screen *screen, const rc_scene *scene, const rc_opts *opts)
{
    rc_opts o;
    int full_height = 0, kv = 0;
    int rc = rc_check_render_args(screen, scene, opts, o, full_height, kv);
    if (rc != RC_OK) return rc;
    const float *sx = nullptr, *sy = nullptr;
    std::shared_lock<std::shared_mutex> lease;     // held until the kernel is enqueued
    rc = rc_get_tables(scene->cam.fov, screen->width, full_height, &sx, &sy, lease, static_cast<hipStream_t>(o.stream));
    if (rc != RC_OK) return rc;
    return rc_enqueue_frame(screen, scene, o, kv, sx, sy, o.workspace, o.workspace_bytes);
}

// ---- rc_render_static: the same frame with no device allocation at all ----------------------
// The library's device scratch -- the sine tables rc_render keeps in hipMalloc'ed cache entries and,
// for RC_VARIANT_CULLED, the per-frame sphere records a caller workspace would hold -- lives in one
// file-scope device array.  Every GPU has its own copy of the array at its own address
// (hipGetSymbolAddress, once per device, with that device current); host code only ever uses that
// address, never the array's name.
constexpr int kStaticTableFloats = 32768;                 // width + full_height of the frame format
constexpr int kStaticRecords = 8192;                      // spheres, RC_VARIANT_CULLED only
constexpr size_t kStaticTableBytes = (size_t)kStaticTableFloats * sizeof(float);
constexpr size_t kStaticScratchBytes = kStaticTableBytes + (size_t)kStaticRecords * 16;
constexpr int kStaticDevices = 16;

static __device__ __align__(256) unsigned char rc_static_scratch[kStaticScratchBytes];

namespace {
struct StaticScratch {
    std::mutex mu;
    unsigned char *ptr = nullptr;      // this device's copy of rc_static_scratch
    bool loaded = false;               // tables of (fov_bits, width, full_height) are in ptr[0, 4 (width + full_height))
    uint32_t fov_bits = 0;
    int width = 0, full_height = 0;
    bool captured = false;             // a call was recorded into a graph with these tables: they must never change again
};
StaticScratch g_static[kStaticDevices];
float g_static_stage[kStaticDevices][kStaticTableFloats];   // host side of the table upload, one per device
} // namespace

extern "C" int rc_render_static(const rc_screen *screen, const rc_scene *scene, const rc_opts *opts)
{
    rc_opts o;
    int full_height = 0, kv = 0;
    int rc = rc_check_render_args(screen, scene, opts, o, full_height, kv);
    if (rc != RC_OK) return rc;
    if ((long long)screen->width + full_height > kStaticTableFloats) {
        snprintf(g_err, sizeof(g_err), "rc_render_static: width + full_height = %lld exceeds the static table of %d entries (use rc_render)",
                 (long long)screen->width + full_height, kStaticTableFloats);
        return RC_EINVAL;
    }
    if (o.variant == RC_VARIANT_CULLED && scene->sphere_num > kStaticRecords) {
        snprintf(g_err, sizeof(g_err), "rc_render_static: RC_VARIANT_CULLED with %d spheres exceeds the static record area of %d (use rc_render with a workspace)",
                 scene->sphere_num, kStaticRecords);
        return RC_EINVAL;
    }
    int dev = 0;
    RC_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kStaticDevices) return rc_fail_arg("rc_render_static: device index above 15");
    StaticScratch &st = g_static[dev];
    std::lock_guard<std::mutex> lock(st.mu);
    if (!st.ptr) {
        void *p = nullptr;
        RC_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(rc_static_scratch)));
        st.ptr = static_cast<unsigned char *>(p);
    }
    hipStream_t stream = static_cast<hipStream_t>(o.stream);
    uint32_t fb;
    memcpy(&fb, &scene->cam.fov, 4);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
    if (!st.loaded || st.fov_bits != fb || st.width != screen->width || st.full_height != full_height) {
        // New frame format: tabulate on the host (kernel.cu:46-47, libm sinf) and upload on the caller's
        // stream, behind the frames already enqueued there that read the previous tables.  The wait makes
        // the staging array free for the next format change, whichever stream that comes from.
        if (capturing)
            return rc_fail_arg("rc_render_static: this frame format is not loaded; render one frame of it before capturing");
        // A graph that recorded a call replays it at any later time with the tables' address, not their
        // contents: once a format has been captured it stays the only one this entry point serves on the GPU.
        if (st.captured) {
            snprintf(g_err, sizeof(g_err), "rc_render_static: a captured graph on device %d uses the tables of %dx%d; "
                     "another frame format would overwrite them under it (use rc_render for other formats)",
                     dev, st.width, st.full_height);
            return RC_EINVAL;
        }
        const float fov = scene->cam.fov;
        const int width = screen->width;
        float *h = g_static_stage[dev];
        for (int x = 0; x < width; x++)
            h[x] = sinf(fov * (1.0f * (float)x / (float)width - 0.5f));
        for (int y = 0; y < full_height; y++)
            h[width + y] = sinf(fov * (1.0f * (float)y / (float)full_height - 0.5f));
        st.loaded = false;
        RC_HIP(hipMemcpyAsync(st.ptr, h, ((size_t)width + (size_t)full_height) * sizeof(float), hipMemcpyHostToDevice, stream));
        RC_HIP(hipStreamSynchronize(stream));
        st.loaded = true;
        st.fov_bits = fb;
        st.width = width;
        st.full_height = full_height;
    }
    if (capturing) st.captured = true;
    const float *sx = reinterpret_cast<const float *>(st.ptr);
    void *records = (o.variant == RC_VARIANT_CULLED) ? st.ptr + kStaticTableBytes : nullptr;
    // exactly the records' size: the per-bin lists of the culled variant are not kept in the static array
    return rc_enqueue_frame(screen, scene, o, kv, sx, sx + screen->width, records,
                            records ? rc_workspace_bytes(scene->sphere_num) : 0);
}

extern "C" size_t rc_workspace_bytes(int sphere_num)
{
    return sphere_num > 0 ? (size_t)sphere_num * 16 : 0;
}

extern "C" size_t rc_workspace_bytes_frame(int sphere_num, int width, int height)
{
    if (sphere_num <= 0 || width <= 0 || height <= 0) return 0;
    const size_t rec_bytes = (rc_workspace_bytes(sphere_num) + 255) & ~(size_t)255;
    if (sphere_num > 65535) return rec_bytes;                   // lists hold 16-bit indices
    // smallest workgroup tile the culled variant uses is 32 x 16 pixels; a bin is 4 x 4 of them
    const size_t bins = (size_t)((width + 127) / 128) * (size_t)((height + 63) / 64);
    const size_t q = (((size_t)sphere_num + 3) / 4 + 7) & ~(size_t)7;
    return rec_bytes + bins * (16 + 8 * q);
}

extern "C" int rc_prepare(int width, int full_height, float fov)
{
    if (width <= 0 || full_height <= 0) return rc_fail_arg("width/full_height must be positive");
    if (width > kMaxDim || full_height > kMaxDim) return rc_fail_arg("width/full_height above 65536");
    const float *sx = nullptr, *sy = nullptr;
    std::shared_lock<std::shared_mutex> lease;
    return rc_get_tables(fov, width, full_height, &sx, &sy, lease, nullptr);
}

extern "C" int rc_to_rgba8(const rc_screen *screen, void *rgba8, size_t rgba8_pitch, int flip, void *stream)
{
    if (!screen || !screen->surface || !rgba8) return rc_fail_arg("null pointer");
    if (screen->width <= 0 || screen->height <= 0) return rc_fail_arg("width/height must be positive");
    if (screen->pitch < (size_t)screen->width * 16 || (screen->pitch & 15) != 0) return rc_fail_arg("bad surface pitch");
    if (rgba8_pitch < (size_t)screen->width * 4 || (rgba8_pitch & 3) != 0) return rc_fail_arg("bad rgba8 pitch");
    if ((reinterpret_cast<uintptr_t>(screen->surface) & 15) != 0 || (reinterpret_cast<uintptr_t>(rgba8) & 3) != 0)
        return rc_fail_arg("surface must be 16-byte and rgba8 4-byte aligned");
    if (screen->width > kMaxDim || screen->height > kMaxDim) return rc_fail_arg("width/height above 65536");
    hipError_t e = rc_launch_rgba8(screen->surface, screen->pitch, screen->width, screen->height, rgba8,
                                   rgba8_pitch, flip, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return rc_fail_hip(e, "rgba8 kernel launch");
    return RC_OK;
}

// ---- source-level drop-in ------------------------------------------------------------------
// include/rc_dropin.hpp declares layout- and calling-convention-identical stand-ins for the reference's argument
// classes: the copy constructor of Vec3 is user-provided exactly as at vec3.h:13-18, which makes Scene
// non-trivial for the purposes of calls (passed by invisible reference), while Screen (Screen.h:3-9) stays a
// trivially copyable 24-byte aggregate passed in memory.
static_assert(sizeof(rc_ref_abi::Scene) == sizeof(rc_scene) && sizeof(rc_ref_abi::Screen) == sizeof(rc_screen), "drop-in layout");

// kernel.cu:114-131.  void return; a failed launch only prints (kernel.cu:127-130).
extern "C" void execute_kernel(rc_ref_abi::Screen screen, rc_ref_abi::Scene scene)
{
    rc_screen s;
    rc_scene c;
    memcpy(&s, &screen, sizeof(s));
    memcpy(&c, static_cast<const void *>(&scene), sizeof(c));
    int rc = rc_render(&s, &c, nullptr);
    if (rc != RC_OK) printf("cuda_kernel_texture_2d() failed to launch error = %d\n", rc < 0 ? -rc : rc);
}

// ---- host helpers ----------------------------------------------------------------------------

extern "C" int rc_device_count(int *count)
{
    if (!count) return rc_fail_arg("null count");
    RC_HIP(hipGetDeviceCount(count));
    return RC_OK;
}

extern "C" int rc_device_name(int device, char *name, size_t name_len)
{
    if (!name || name_len == 0) return rc_fail_arg("null name buffer");
    hipDeviceProp_t p;
    RC_HIP(hipGetDeviceProperties(&p, device));
    snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
    return RC_OK;
}

extern "C" int rc_set_device(int device)
{
    RC_HIP(hipSetDevice(device));
    return RC_OK;
}

extern "C" int rc_device_info(int device, int *compute_units, int *clock_khz, size_t *lds_per_block)
{
    hipDeviceProp_t p;
    RC_HIP(hipGetDeviceProperties(&p, device));
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (clock_khz) *clock_khz = p.clockRate;
    if (lds_per_block) *lds_per_block = p.sharedMemPerBlock;
    return RC_OK;
}

extern "C" int rc_malloc_pitch(void **ptr, size_t *pitch, size_t width_bytes, size_t height)
{
    if (!ptr || !pitch || width_bytes == 0 || height == 0) return rc_fail_arg("rc_malloc_pitch arguments");
    size_t p = (width_bytes + 255) & ~(size_t)255;
    RC_HIP(hipMalloc(ptr, p * height));
    *pitch = p;
    return RC_OK;
}

extern "C" int rc_malloc(void **ptr, size_t bytes)
{
    if (!ptr) return rc_fail_arg("null ptr");
    RC_HIP(hipMalloc(ptr, bytes ? bytes : 1));
    return RC_OK;
}

extern "C" int rc_free(void *ptr)
{
    RC_HIP(hipFree(ptr));
    return RC_OK;
}

extern "C" int rc_memset(void *ptr, int value, size_t bytes, void *stream)
{
    RC_HIP(hipMemsetAsync(ptr, value, bytes, static_cast<hipStream_t>(stream)));
    return RC_OK;
}

extern "C" int rc_upload(void *dst_device, const void *src_host, size_t bytes, void *stream)
{
    if (bytes == 0) return RC_OK;
    if (!dst_device || !src_host) return rc_fail_arg("null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    RC_HIP(hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, s));
    RC_HIP(hipStreamSynchronize(s));   // pageable source may be reused by the caller at once
    return RC_OK;
}

extern "C" int rc_download_2d(void *dst_host, size_t dst_pitch, const void *src_device, size_t src_pitch,
                              size_t width_bytes, size_t height, void *stream)
{
    if (!dst_host || !src_device) return rc_fail_arg("null pointer");
    if (dst_pitch < width_bytes || src_pitch < width_bytes) return rc_fail_arg("pitch < width_bytes");
    hipStream_t s = static_cast<hipStream_t>(stream);
    RC_HIP(hipMemcpy2DAsync(dst_host, dst_pitch, src_device, src_pitch, width_bytes, height,
                            hipMemcpyDeviceToHost, s));
    RC_HIP(hipStreamSynchronize(s));
    return RC_OK;
}

extern "C" int rc_download_2d_async(void *dst_host, size_t dst_pitch, const void *src_device, size_t src_pitch,
                                    size_t width_bytes, size_t height, void *stream)
{
    if (!dst_host || !src_device) return rc_fail_arg("null pointer");
    if (dst_pitch < width_bytes || src_pitch < width_bytes) return rc_fail_arg("pitch < width_bytes");
    RC_HIP(hipMemcpy2DAsync(dst_host, dst_pitch, src_device, src_pitch, width_bytes, height,
                            hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    return RC_OK;
}

extern "C" int rc_stream_create(void **stream)
{
    if (!stream) return rc_fail_arg("null stream");
    hipStream_t s;
    RC_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return RC_OK;
}

extern "C" int rc_stream_destroy(void *stream)
{
    RC_HIP(hipStreamDestroy(static_cast<hipStream_t>(stream)));
    return RC_OK;
}

extern "C" int rc_stream_wait_event(void *stream, void *event)
{
    RC_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0));
    return RC_OK;
}

extern "C" int rc_host_alloc(void **ptr, size_t bytes)
{
    if (!ptr) return rc_fail_arg("null ptr");
    RC_HIP(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return RC_OK;
}

extern "C" int rc_host_free(void *ptr)
{
    RC_HIP(hipHostFree(ptr));
    return RC_OK;
}

extern "C" int rc_sync(void *stream)
{
    if (stream) RC_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    else RC_HIP(hipDeviceSynchronize());
    return RC_OK;
}

extern "C" int rc_graph_begin(void *stream)
{
    if (!stream) return rc_fail_arg("the default stream cannot be captured: pass a stream from rc_stream_create");
    RC_HIP(hipStreamBeginCapture(static_cast<hipStream_t>(stream), hipStreamCaptureModeThreadLocal));
    return RC_OK;
}

extern "C" int rc_graph_end(void *stream, void **graph_exec)
{
    if (!stream || !graph_exec) return rc_fail_arg("null stream or graph_exec");
    hipGraph_t graph = nullptr;
    RC_HIP(hipStreamEndCapture(static_cast<hipStream_t>(stream), &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return rc_fail_hip(e, "hipGraphInstantiate");
    *graph_exec = exec;
    return RC_OK;
}

extern "C" int rc_graph_launch(void *graph_exec, void *stream)
{
    if (!graph_exec) return rc_fail_arg("null graph_exec");
    RC_HIP(hipGraphLaunch(static_cast<hipGraphExec_t>(graph_exec), static_cast<hipStream_t>(stream)));
    return RC_OK;
}

extern "C" int rc_graph_destroy(void *graph_exec)
{
    if (!graph_exec) return RC_OK;
    RC_HIP(hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph_exec)));
    return RC_OK;
}

extern "C" int rc_event_create(void **event)
{
    if (!event) return rc_fail_arg("null event");
    hipEvent_t e;
    RC_HIP(hipEventCreate(&e));
    *event = e;
    return RC_OK;
}

extern "C" int rc_event_record(void *event, void *stream)
{
    RC_HIP(hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)));
    return RC_OK;
}

extern "C" int rc_event_elapsed_ms(void *start, void *stop, float *ms)
{
    if (!ms) return rc_fail_arg("null ms");
    RC_HIP(hipEventSynchronize(static_cast<hipEvent_t>(stop)));
    RC_HIP(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
    return RC_OK;
}

extern "C" int rc_event_wait(void *event)
{
    RC_HIP(hipEventSynchronize(static_cast<hipEvent_t>(event)));
    return RC_OK;
}

extern "C" int rc_event_destroy(void *event)
{
    RC_HIP(hipEventDestroy(static_cast<hipEvent_t>(event)));
    return RC_OK;
}

// vec3.h:168-176,89-92
static rc_vec3 rc_norm3(rc_vec3 v)
{
    float inv = 1.0f / sqrtf((v.x * v.x + v.y * v.y) + v.z * v.z);
    rc_vec3 r = {v.x * inv, v.y * inv, v.z * inv};
    return r;
}

extern "C" void rc_camera_data(const rc_vec3 *position, const rc_vec3 *rotation, int width, int height,
                               rc_camera *out)
{
    const rc_vec3 r = *rotation;
    out->aspect = (float)(1.0 * width / height);            // Camera.h:53 (double divide)
    out->fov = (90.f / 360.0f) * 3.1415926536f;             // Camera.h:6,54
    out->position = *position;
    rc_vec3 f = {-cosf(r.x) * sinf(r.y), sinf(r.x), -cosf(r.x) * cosf(r.y)};   // Camera.h:73
    rc_vec3 u = {sinf(r.x) * sinf(r.y), cosf(r.x), cosf(r.y) * sinf(r.x)};     // Camera.h:80-84
    rc_vec3 c = {f.y * u.z - f.z * u.y, f.z * u.x - f.x * u.z, f.x * u.y - f.y * u.x}; // vec3.h:158-161
    out->forward = f;
    out->up = u;
    out->right = rc_norm3(c);                               // Camera.h:89-91
}

// main.cpp:53-128.  The camera axes are those of Camera::GetForward / GetRight
// (Camera.h:71-91) for the rotation AFTER the mouse-look update, as SetRotation invalidates
// the cached basis (Camera.h:95-99) before the movement keys read it.
extern "C" int rc_handle_input(rc_vec3 *position, rc_vec3 *rotation, float movement_speed, float rotation_speed,
                               float dt, const rc_input *input)
{
    if (!position || !rotation || !input) return rc_fail_arg("rc_handle_input: null pointer");
    int update = 0;
    if (input->rbutton) {                                      // main.cpp:64-70
        const float angle = dt * rotation_speed;
        rotation->x = rotation->x + -(float)input->cursor_dy * angle;
        rotation->y = rotation->y + -(float)input->cursor_dx * angle;
        rotation->z = rotation->z + 0.0f;
        update = 1;
    }
    float speed = movement_speed;                              // main.cpp:73-80
    if (input->lshift) speed *= 5;
    else speed *= 50;

    if (input->key_q) { position->y += dt * speed; update = 1; }   // main.cpp:82-86
    if (input->key_z) { position->y -= dt * speed; update = 1; }   // main.cpp:88-92

    rc_camera basis;
    const rc_vec3 origin = {0.0f, 0.0f, 0.0f};
    rc_camera_data(&origin, rotation, 1, 1, &basis);           // forward / right for this rotation
    // norm(axis with y = 0) * speed * dt, each product rounded on its own (vec3.h:126-129,168-171)
    auto step = [&](rc_vec3 axis, float sign) {
        axis.y = 0;
        rc_vec3 m = rc_norm3(axis);
        m.x *= speed; m.y *= speed; m.z *= speed;
        m.x *= dt; m.y *= dt; m.z *= dt;
        if (sign > 0) { position->x += m.x; position->y += m.y; position->z += m.z; }
        else { position->x -= m.x; position->y -= m.y; position->z -= m.z; }
        update = 1;
    };
    if (input->key_w) step(basis.forward, +1.0f);              // main.cpp:94-100
    if (input->key_a) step(basis.right, -1.0f);                // main.cpp:102-108
    if (input->key_s) step(basis.forward, -1.0f);              // main.cpp:110-116
    if (input->key_d) step(basis.right, +1.0f);                // main.cpp:118-124
    return update;
}

extern "C" void rc_generate_scene(uint32_t seed, int n, rc_vec3 *position, float *radius, rc_vec3 *color)
{
    std::mt19937 gen{seed};                                 // SceneManager.h:22
    auto uni = [&gen](float lo, float hi) {
        float u = (float)(gen() >> 8) * 0x1p-24f;
        return lo + u * (hi - lo);
    };
    for (int i = 0; i < n; i++) {                           // SceneManager.cpp:35-45
        radius[i] = uni(0.3f, 1.0f);
        color[i].x = uni(0.0f, 1.0f);
        color[i].y = uni(0.0f, 1.0f);
        color[i].z = uni(0.0f, 1.0f);
        position[i].x = uni(-10.0f, 10.0f);
        position[i].y = uni(-5.0f, 5.0f);
        position[i].z = uni(-15.0f, -5.0f);
    }
}

extern "C" int rc_write_ppm(const char *path, const uint8_t *rgba8, size_t pitch, int width, int height)
{
    if (!path || !rgba8 || width <= 0 || height <= 0 || pitch < (size_t)width * 4) return rc_fail_arg("rc_write_ppm arguments");
    FILE *f = fopen(path, "wb");
    if (!f) {
        snprintf(g_err, sizeof(g_err), "cannot open %s for writing", path);
        return RC_EINVAL;
    }
    fprintf(f, "P6\n%d %d\n255\n", width, height);
    std::vector<uint8_t> line((size_t)width * 3);
    for (int y = 0; y < height; y++) {
        const uint8_t *src = rgba8 + (size_t)y * pitch;
        for (int x = 0; x < width; x++) {
            line[3 * (size_t)x + 0] = src[4 * (size_t)x + 0];
            line[3 * (size_t)x + 1] = src[4 * (size_t)x + 1];
            line[3 * (size_t)x + 2] = src[4 * (size_t)x + 2];
        }
        if (fwrite(line.data(), 1, line.size(), f) != line.size()) {
            fclose(f);
            snprintf(g_err, sizeof(g_err), "short write to %s", path);
            return RC_EINVAL;
        }
    }
    if (fclose(f) != 0) {
        snprintf(g_err, sizeof(g_err), "cannot finish writing %s", path);
        return RC_EINVAL;
    }
    return RC_OK;
}

extern "C" int rc_selftest_math(int which, uint32_t first_bits, uint32_t count, unsigned long long *mismatches_dev,
                                float *out_fast_dev, float *out_general_dev, void *stream)
{
    if (which != 0 && which != 1) return rc_fail_arg("rc_selftest_math: which must be 0 (sqrt) or 1 (reciprocal)");
    if (!mismatches_dev) return rc_fail_arg("rc_selftest_math: null counter");
    if ((uint64_t)first_bits + count > 0x100000000ull) return rc_fail_arg("rc_selftest_math: operand range wraps");
    hipError_t e = rc_launch_selftest_math(which, first_bits, count, mismatches_dev, out_fast_dev, out_general_dev,
                                           static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return rc_fail_hip(e, "self-test kernel launch");
    return RC_OK;
}

extern "C" int rc_pixels_per_lane(int width, int rows, int full_height, int sphere_num, int mode, int variant,
                                  int compute_units)
{
    if (width <= 0 || rows <= 0 || sphere_num < 0) return rc_fail_arg("rc_pixels_per_lane: bad frame or sphere count");
    if (full_height != 0 && full_height < rows) return rc_fail_arg("rc_pixels_per_lane: full_height below rows");
    if (mode != RC_MODE_REF && mode != RC_MODE_EXT) return rc_fail_arg("rc_pixels_per_lane: unknown mode");
    const bool ext = mode == RC_MODE_EXT;
    switch (variant) {
    case RC_VARIANT_LITERAL: return 1;
    case RC_VARIANT_TILED_P2: return 2;
    case RC_VARIANT_TILED_P4: return 4;
    case RC_VARIANT_TILED_P8: return ext ? 4 : 8;
    case RC_VARIANT_AUTO: case RC_VARIANT_TILED: case RC_VARIANT_CULLED: break;
    default: return rc_fail_arg("rc_pixels_per_lane: unknown variant");
    }
    RcKernArgs a{};
    a.width = width; a.rows = rows; a.n = sphere_num;
    a.ext_mode = ext ? 1 : 0;
    a.cull = variant == RC_VARIANT_CULLED ? 1 : 0;
    return rc_auto_pixels_per_lane(a, full_height, compute_units);
}

extern "C" const char *rc_last_error_string(void) { return g_err; }

extern "C" const char *rc_version(void) { return "rc_hip 0.2 gfx950"; }

extern "C" int rc_abi_version(void) { return RC_ABI_VERSION; }

// ---- context API: the C++ host classes (csrc/host/*.hpp) behind C entry points ----------------
#include "host/rc_raycaster.hpp"

struct rc_ctx {
    int device = 0;
    rc::RayTracingKernel rtk;
    void *workspace = nullptr;       // for RC_VARIANT_CULLED: records + bin lists of the current scene / frame size
    size_t workspace_bytes = 0;
};

namespace {
// runs `body` with the context's device current; C++ exceptions of the host classes become codes
template <class F>
int rc_ctx_call(rc_ctx *ctx, F body)
{
    if (!ctx) return rc_fail_arg("null context");
    // the caller's current device is put back on every way out (a multi-GPU caller's own raw
    // rc_malloc / rc_render calls must not be silently retargeted by a context call)
    struct Restore {
        int prev = -1;
        ~Restore() { if (prev >= 0) (void)hipSetDevice(prev); }
    } restore;
    int prev = -1;
    RC_HIP(hipGetDevice(&prev));
    RC_HIP(hipSetDevice(ctx->device));
    if (prev != ctx->device) restore.prev = prev;
    try {
        body(ctx->rtk);
        return RC_OK;
    } catch (const rc::Error &e) {
        return e.code;                 // g_err was set by the failing rc_* call
    } catch (const std::exception &e) {
        return rc_fail_arg(e.what());
    }
}
} // namespace

extern "C" int rc_ctx_create(rc_ctx **out, int device, int width, int height)
{
    if (!out) return rc_fail_arg("null ctx pointer");
    if (width <= 0 || height <= 0 || width > kMaxDim || height > kMaxDim) return rc_fail_arg("width/height out of range");
    int count = 0;
    RC_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return rc_fail_arg("no such device");
    rc_ctx *ctx = new (std::nothrow) rc_ctx;
    if (!ctx) return rc_fail_arg("out of host memory");
    ctx->device = device;
    ctx->rtk.screen.width = width;
    ctx->rtk.screen.height = height;
    ctx->rtk.sm.cam = rc::Camera(width, height);
    ctx->rtk.sm.cam.position = rc_vec3{0.0f, 0.0f, 20.0f};       // SceneManager.cpp:5
    ctx->rtk.sm.UpdateCamera();
    *out = ctx;
    return RC_OK;
}

extern "C" int rc_ctx_destroy(rc_ctx *ctx)
{
    if (!ctx) return RC_OK;
    int rc = rc_ctx_call(ctx, [&](rc::RayTracingKernel &k) {
        k.Cleanup();
        if (ctx->workspace) rc_free(ctx->workspace);
        ctx->workspace = nullptr;
    });
    delete ctx;
    return rc;
}

extern "C" int rc_scene_generate(rc_ctx *ctx, int sphere_num, uint32_t seed)
{
    if (sphere_num < 0) return rc_fail_arg("sphere_num < 0");
    return rc_ctx_call(ctx, [&](rc::RayTracingKernel &k) {
        std::vector<rc_vec3> p((size_t)sphere_num), c((size_t)sphere_num);
        std::vector<float> r((size_t)sphere_num);
        rc_generate_scene(seed, sphere_num, p.data(), r.data(), c.data());
        k.sm.SetScene(p, r, c);
    });
}

extern "C" int rc_scene_upload(rc_ctx *ctx, int sphere_num, const rc_vec3 *position, const float *radius, const rc_vec3 *color)
{
    if (sphere_num < 0 || (sphere_num > 0 && (!position || !radius || !color))) return rc_fail_arg("rc_scene_upload arguments");
    return rc_ctx_call(ctx, [&](rc::RayTracingKernel &k) {
        k.sm.SetScene(std::vector<rc_vec3>(position, position + sphere_num), std::vector<float>(radius, radius + sphere_num),
                      std::vector<rc_vec3>(color, color + sphere_num));
    });
}

extern "C" int rc_camera_set(rc_ctx *ctx, const rc_vec3 *position, const rc_vec3 *rotation)
{
    if (!position || !rotation) return rc_fail_arg("null position or rotation");
    return rc_ctx_call(ctx, [&](rc::RayTracingKernel &k) {
        k.sm.cam.position = *position;
        k.sm.cam.SetRotation(*rotation);
        k.sm.UpdateCamera();
    });
}

extern "C" int rc_camera_input(rc_ctx *ctx, const rc_input *input, float dt)
{
    if (!input) return rc_fail_arg("null input");
    bool moved = false;
    int rc = rc_ctx_call(ctx, [&](rc::RayTracingKernel &k) {
        moved = k.sm.cam.HandleInput(*input, dt);
        if (moved) k.sm.UpdateCamera();                             // main.cpp:126-127
    });
    return rc != RC_OK ? rc : (moved ? 1 : 0);
}

extern "C" int rc_surface_alloc(rc_ctx *ctx, size_t *pitch)
{
    return rc_ctx_call(ctx, [&](rc::RayTracingKernel &k) {
        k.RegisterSurface();
        if (pitch) *pitch = k.screen.pitch;
    });
}

extern "C" int rc_ctx_render(rc_ctx *ctx, const rc_opts *opts)
{
    return rc_ctx_call(ctx, [&](rc::RayTracingKernel &k) {
        if (!k.screen.surface) throw std::runtime_error("rc_ctx_render: no surface (call rc_surface_alloc)");
        if (opts && opts->variant == RC_VARIANT_CULLED && !opts->workspace) {
            // the context lends its own workspace (grown when the scene or the frame got bigger;
            // the device is idle-waited before the old one is freed, frames may still read it)
            const size_t need = rc_workspace_bytes_frame(k.sm.scene.sphere_num, k.screen.width, k.screen.height);
            if (need > ctx->workspace_bytes) {
                rc::check(rc_sync(nullptr), "rc_sync");
                if (ctx->workspace) rc_free(ctx->workspace);
                ctx->workspace = nullptr;
                ctx->workspace_bytes = 0;
                rc::check(rc_malloc(&ctx->workspace, need), "rc_malloc(workspace)");
                ctx->workspace_bytes = need;
            }
            rc_opts o = *opts;
            o.workspace = ctx->workspace;
            o.workspace_bytes = ctx->workspace_bytes;
            rc::check(rc_render(&k.screen, &k.sm.scene, &o), "rc_render");
            return;
        }
        rc::check(rc_render(&k.screen, &k.sm.scene, opts), "rc_render");
    });
}

extern "C" int rc_readback(rc_ctx *ctx, float *dst_host, size_t dst_pitch)
{
    if (!dst_host) return rc_fail_arg("null destination");
    return rc_ctx_call(ctx, [&](rc::RayTracingKernel &k) {
        if (!k.screen.surface) throw std::runtime_error("rc_readback: no surface");
        rc::check(rc_sync(nullptr), "rc_sync");                      // RayTracingKernel.cpp:18
        rc::check(rc_download_2d(dst_host, dst_pitch, k.screen.surface, k.screen.pitch, (size_t)k.screen.width * 16,
                                 (size_t)k.screen.height, nullptr), "rc_download_2d");
    });
}

extern "C" int rc_readback_rgba8(rc_ctx *ctx, uint8_t *dst_host, size_t dst_pitch, int flip)
{
    if (!dst_host) return rc_fail_arg("null destination");
    return rc_ctx_call(ctx, [&](rc::RayTracingKernel &k) {
        if (!k.screen.surface) throw std::runtime_error("rc_readback_rgba8: no surface");
        const size_t pitch8 = (size_t)k.screen.width * 4;
        if (dst_pitch < pitch8) throw std::runtime_error("rc_readback_rgba8: pitch < 4*width");
        void *d8 = nullptr;
        rc::check(rc_malloc(&d8, pitch8 * (size_t)k.screen.height), "rc_malloc(rgba8)");
        int rc = rc_sync(nullptr);
        if (rc == RC_OK) rc = rc_to_rgba8(&k.screen, d8, pitch8, flip, nullptr);
        if (rc == RC_OK) rc = rc_download_2d(dst_host, dst_pitch, d8, pitch8, pitch8, (size_t)k.screen.height, nullptr);
        rc_free(d8);
        rc::check(rc, "rc_readback_rgba8");
    });
}

extern "C" int rc_ctx_write_ppm(rc_ctx *ctx, const char *path)
{
    if (!path) return rc_fail_arg("null path");
    return rc_ctx_call(ctx, [&](rc::RayTracingKernel &k) {
        if (!k.screen.surface) throw std::runtime_error("rc_ctx_write_ppm: no surface");
        rc::check(rc_sync(nullptr), "rc_sync");
        k.WritePPM(path);
    });
}

extern "C" const rc_screen *rc_ctx_screen(const rc_ctx *ctx) { return ctx ? &ctx->rtk.screen : nullptr; }
extern "C" const rc_scene *rc_ctx_scene(const rc_ctx *ctx) { return ctx ? &ctx->rtk.sm.scene : nullptr; }
