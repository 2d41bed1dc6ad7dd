// rc_frame_pipeline.hpp -- double-buffered frame pipeline.
//
// The reference renders, synchronises the device, copies the whole frame device-to-device
// into the D3D texture and presents it, one frame at a time on the default stream
// (RayTracingKernel.cpp:17-27, main.cpp:47-48).  Here a frame is rendered on one stream into
// one of two device surfaces while the previous frame is still being read back to pinned host
// memory on a second stream; the host only waits when it actually collects a frame.
//
//   FramePipeline pipe(width, height, /*rgba8=*/true);
//   for (f ...) { scene.cam = ...; pipe.Submit(scene); if (f) use(pipe.Collect()); }
//   use(pipe.Collect());
#pragma once
#include <cstdint>
#include <stdexcept>

#include "rc_abi.h"
#include "rc_scene_manager.hpp"   // rc::check

namespace rc {

class FramePipeline {
public:
    // rgba8 = true: the fused display conversion writes a top-down RGBA8 image and only that
    // (4 B/pixel) crosses PCIe; false: the float surface (16 B/pixel) is read back.
    FramePipeline(int width, int height, bool rgba8) : w_(width), h_(height), rgba8_(rgba8)
    {
        try {
            check(rc_stream_create(&render_), "rc_stream_create");
            check(rc_stream_create(&copy_), "rc_stream_create");
            for (int i = 0; i < 2; i++) {
                check(rc_malloc_pitch(&surf_[i], &pitch_, (size_t)w_ * 16, (size_t)h_), "rc_malloc_pitch");
                if (rgba8_) check(rc_malloc(&dev8_[i], (size_t)w_ * 4 * h_), "rc_malloc(rgba8)");
                check(rc_host_alloc(&host_[i], frame_bytes()), "rc_host_alloc");
                check(rc_event_create(&rendered_[i]), "rc_event_create");
                check(rc_event_create(&copied_[i]), "rc_event_create");
            }
        } catch (...) {
            Release();          // a constructor that throws runs no destructor
            throw;
        }
    }
    ~FramePipeline() { Release(); }
    FramePipeline(const FramePipeline &) = delete;
    FramePipeline &operator=(const FramePipeline &) = delete;

    size_t frame_bytes() const { return (size_t)w_ * h_ * (rgba8_ ? 4 : 16); }

    // Enqueue render + read-back of one frame; never blocks on the GPU unless both buffers
    // are still in flight.  `opts` carries mode/variant/ext; stream and rgba8 are overwritten.
    void Submit(const rc_scene &scene, rc_opts opts = rc_opts{})
    {
        if (in_flight_ == 2) throw std::logic_error("FramePipeline: Collect() a frame before submitting a third");
        const int b = next_;
        // the buffer's previous read-back must have finished before it is rendered into again
        if (used_[b]) check(rc_stream_wait_event(render_, copied_[b]), "rc_stream_wait_event");
        rc_screen screen{surf_[b], w_, h_, pitch_};
        opts.stream = render_;
        opts.rgba8 = rgba8_ ? dev8_[b] : nullptr;
        opts.rgba8_pitch = (size_t)w_ * 4;
        opts.rgba8_flip = 1;
        check(rc_render(&screen, &scene, &opts), "rc_render");
        check(rc_event_record(rendered_[b], render_), "rc_event_record");
        check(rc_stream_wait_event(copy_, rendered_[b]), "rc_stream_wait_event");
        if (rgba8_)
            check(rc_download_2d_async(host_[b], (size_t)w_ * 4, dev8_[b], (size_t)w_ * 4, (size_t)w_ * 4, (size_t)h_, copy_),
                  "rc_download_2d_async");
        else
            check(rc_download_2d_async(host_[b], (size_t)w_ * 16, surf_[b], pitch_, (size_t)w_ * 16, (size_t)h_, copy_),
                  "rc_download_2d_async");
        check(rc_event_record(copied_[b], copy_), "rc_event_record");
        used_[b] = true;
        next_ ^= 1;
        in_flight_++;
    }

    // Oldest submitted frame, in pinned host memory (valid until the second Submit after this).
    const void *Collect()
    {
        if (in_flight_ == 0) throw std::logic_error("FramePipeline: nothing to collect");
        const int b = (in_flight_ == 2) ? next_ : (next_ ^ 1);
        check(rc_event_wait(copied_[b]), "rc_event_wait");
        in_flight_--;
        return host_[b];
    }

private:
    void Release()
    {
        rc_sync(nullptr);
        for (int i = 0; i < 2; i++) {
            if (surf_[i]) rc_free(surf_[i]);
            if (dev8_[i]) rc_free(dev8_[i]);
            if (host_[i]) rc_host_free(host_[i]);
            if (rendered_[i]) rc_event_destroy(rendered_[i]);
            if (copied_[i]) rc_event_destroy(copied_[i]);
            surf_[i] = dev8_[i] = host_[i] = rendered_[i] = copied_[i] = nullptr;
        }
        if (render_) rc_stream_destroy(render_);
        if (copy_) rc_stream_destroy(copy_);
        render_ = copy_ = nullptr;
    }

    int w_, h_;
    bool rgba8_;
    size_t pitch_ = 0;
    void *render_ = nullptr, *copy_ = nullptr;
    void *surf_[2] = {nullptr, nullptr}, *dev8_[2] = {nullptr, nullptr}, *host_[2] = {nullptr, nullptr};
    void *rendered_[2] = {nullptr, nullptr}, *copied_[2] = {nullptr, nullptr};
    bool used_[2] = {false, false};
    int next_ = 0, in_flight_ = 0;
};

} // namespace rc
