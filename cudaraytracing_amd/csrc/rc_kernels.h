// rc_kernels.h -- kernel argument block and launchers shared by rc_kernels.hip / rc_host.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define RC_CULL_CHUNK_MAX 1024  // culled variant: spheres per pass (see rc_launch_tiled)
#define RC_LDS_CHUNK_MAX 4096   // most spheres staged in LDS per pass (13 bytes each: 52 KiB)
#define RC_SHADOW_FAST_MIN 1024  // EXT: from this many spheres the shadow pass uses its dominant-axis filter
#define RC_LDS_EXACT_MAX 1024   // up to this chunk size the exact-test terms are staged too (+16 bytes each)

// rc_opts.tuning bits as the launchers see them (same values as RC_TUNE_* in rc_abi.h)
enum { RC_KTUNE_SHADOW_AXIS_ON = 1, RC_KTUNE_SHADOW_AXIS_OFF = 2, RC_KTUNE_NO_LDS_EXACT = 4, RC_KTUNE_NO_NEAR_SKIP = 8 };
enum { RC_KVARIANT_AUTO = 0, RC_KVARIANT_LITERAL = 1, RC_KVARIANT_TILED4 = 2, RC_KVARIANT_TILED2 = 3, RC_KVARIANT_TILED8 = 4 };

// Passed by value as the kernel parameter, as the reference passes Screen and Scene
// (kernel.cu:69).  `rows` is the number of surface rows (a band of the frame); frame row of
// surface row r is r + y_offset; sin_y is indexed by FRAME row.
struct RcKernArgs {
    void *surface;
    size_t pitch;
    int width;
    int rows;
    int y_offset;
    int n;
    const float *sin_x;      // [width]        sinf(fov * (x / width - 0.5))
    const float *sin_y;      // [full_height]  sinf(fov * (y / full_height - 0.5))
    const float *position;   // [3n]
    const float *radius;     // [n]
    const float *color;      // [3n]
    void *rgba8;             // optional
    size_t rgba8_pitch;
    int rgba8_flip;
    float aspect;
    float ox, oy, oz;        // cam.position
    float fx, fy, fz;        // cam.forward
    float ux, uy, uz;        // cam.up
    float rx, ry, rz;        // cam.right
    float nF, nR, nU;        // upper bounds of |forward|, |right|, |up| (filter range check)
    int cam_ok;              // 0: camera basis outside the range the filter is proven for (every lane takes the exact
                             // path); 1: in range, each lane checks its own ray; 2: in range and so close to orthogonal
                             // that no lane's ray can fail that check (see rc_enqueue_frame), which is then skipped
    // EXT mode (plane + point light + Phong + hard shadows; no counterpart in the reference)
    int tuning;              // RC_KTUNE_* bits
    int cull;                // 1: tile-culling variant of the tiled kernel
    int ext_mode;            // 0: the reference's shading; 1: EXT
    int has_plane, shadows, shininess_log2;
    float pnx, pny, pnz, plane_h;
    float pcx, pcy, pcz;
    float lgx, lgy, lgz;     // point light position
    float specular;
    // per-frame constants of EXT mode, evaluated on the host with the statements the per-pixel code would use
    // (IEEE binary32, one rounding per operation: the same bits as on the device)
    float pux, puy, puz;     // plane normal normalised: pn * (1 / sqrt(pn.pn))
    float pu_il;             // 1 / sqrt(pu.pu), the reciprocal length vec3::reflect divides by twice
    float plane_num;         // plane_h - pn.o      numerator of the primary plane test
    float plane_num_light;   // plane_h - pn.light  numerator of the plane's shadow test
    int plane_axis;          // the plane normal's single non-zero component (0 if plane_side is 0)
    float plane_side;        // sign a ray's un-normalised component along plane_axis must have to hit the plane: +1 / -1;
                             // 0 = unknown, every pixel takes the plane test (normal with several non-zero components,
                             // plane through the camera, non-finite data)
    // culled variant: sphere records built once per frame by rc_launch_records (else NULL)
    const float4 *records;   // [n] {alpha, gamma, beta, always ? 1 : 0}
    const unsigned char *bins; // culled variant, level 0: per-bin sphere lists (else NULL); bins_bytes available
    size_t bins_bytes;
    // filled in by the tiled launcher
    int lds_cap;             // spheres per LDS chunk (multiple of 32)
    int lds_exact;           // 1: the chunk's exact-test terms {2v, 4c} are staged in LDS as well
    int near_skip;           // 1: exact tests are skipped for pixels already nearer than the sphere's near bound
    int shadow_fast;         // 1: launch the kernel whose EXT shadow pass may use the dominant-axis filter (set by the launcher)
    int tiles_x, tiles_y;    // workgroup tiles covering the surface
    int bins_x, bin_q, bin_stride; // bins per row, entries per list segment, bytes per bin
};

// full_height: rows of the whole frame (a.rows of them are rendered by this launch)
hipError_t rc_launch_draw(const RcKernArgs &a, int variant, int full_height, int compute_units, hipStream_t stream);
// pixels per lane RC_KVARIANT_AUTO resolves to (reads a.width, a.rows, a.n, a.ext_mode, a.cull only)
int rc_auto_pixels_per_lane(const RcKernArgs &a, int full_height, int compute_units);
// Builds the n filter records of the scene for this camera into `records` (16 bytes each).
hipError_t rc_launch_records(const RcKernArgs &a, float4 *records, hipStream_t stream);
hipError_t rc_launch_rgba8(const void *surface, size_t pitch, int width, int rows, void *out,
                           size_t out_pitch, int flip, hipStream_t stream);
// Self-test of the short correctly-rounded sqrt (which = 0) / reciprocal (1) sequences against the general ones.
hipError_t rc_launch_selftest_math(int which, uint32_t first_bits, uint32_t count, unsigned long long *mismatches,
                                   float *out_fast, float *out_general, hipStream_t stream);
