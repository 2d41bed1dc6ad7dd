/* render_frame.c -- the reference's frame (1280x768 window, 32*16 random spheres, camera at
 * (0,0,20): main.cpp:32-33, SceneManager.cpp:5,8) through the C context API, written as a PPM.
 *
 *   gcc -std=c99 -Iinclude examples/render_frame.c -Lcudaraytracing_amd -lrc_hip \
 *       -Wl,-rpath,$PWD/cudaraytracing_amd -o render_frame && ./render_frame out.ppm
 */
#include <stdio.h>
#include <stdlib.h>

#include "rc_abi.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != RC_OK) {                                                      \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, rc_last_error_string()); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "frame.ppm";
    const int width = 1280, height = 768;
    rc_ctx *ctx = NULL;
    size_t pitch = 0;
    float *frame;
    size_t lit = 0, i;

    CHECK(rc_ctx_create(&ctx, 0, width, height));
    CHECK(rc_scene_generate(ctx, 32 * 16, 0));
    CHECK(rc_surface_alloc(ctx, &pitch));
    CHECK(rc_ctx_render(ctx, NULL));

    frame = (float *)malloc((size_t)width * height * 16);
    if (!frame) return 1;
    CHECK(rc_readback(ctx, frame, (size_t)width * 16));
    for (i = 0; i < (size_t)width * height; i++)
        lit += (frame[4 * i] != 0.0f || frame[4 * i + 1] != 0.0f || frame[4 * i + 2] != 0.0f);
    printf("%s: %dx%d, pitch %zu, %zu lit pixels\n", rc_version(), width, height, pitch, lit);
    CHECK(rc_ctx_write_ppm(ctx, path));
    free(frame);
    CHECK(rc_ctx_destroy(ctx));
    return 0;
}
