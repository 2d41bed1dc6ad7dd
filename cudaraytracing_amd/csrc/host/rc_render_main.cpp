// rc_render -- command-line front end of the ray caster (replaces the Win32/D3D11 shell of
// main.cpp and DXManager.cpp: no window, the frame goes to a PPM file).
//
//   rc_render [--width W] [--height H] [--spheres N] [--seed S] [--frames F] [--variant V]
//             [--pos x,y,z] [--rot pitch,yaw] [--out frame.ppm] [--out-pattern dir/f_%03d.ppm]
//             [--script input.txt]
//
// --script replays the interactive controls of main.cpp:53-128 from a text file, one poll and
// one rendered frame per line:  <dt in ms> <keys> <cursor dx> <cursor dy>, keys being '-' or any
// of r (right mouse button), shift, q, z, w, a, s, d joined by '+', e.g. "16.6 w+d+r 3 -2".
//
// With --frames F > 1 the camera follows the scripted 300-frame path
// (theta = 2 pi f / 300: rotation (0.15 sin 2theta, 0.5 sin theta, 0), position
// (6 sin theta, 1.5 sin 2theta, 20)) and the per-frame time is printed the way the reference
// prints it (main.cpp:50), here measured with HIP events around the launch only.  With
// --out-pattern every frame of the sequence goes through the double-buffered FramePipeline
// (render of frame f+1 overlaps the RGBA8 read-back of frame f) and is written as a PPM.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "rc_frame_pipeline.hpp"
#include "rc_raycaster.hpp"

static bool parse3(const char *s, float *a, float *b, float *c)
{
    return std::sscanf(s, "%f,%f,%f", a, b, c) == 3;
}

struct ScriptPoll { float dt; rc_input in; };

static bool load_script(const std::string &path, std::vector<ScriptPoll> *polls)
{
    std::ifstream f(path);
    if (!f) return false;
    std::string line;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ls(line);
        ScriptPoll p{};
        std::string keys;
        if (!(ls >> p.dt >> keys >> p.in.cursor_dx >> p.in.cursor_dy)) return false;
        std::istringstream ks(keys);
        std::string k;
        while (std::getline(ks, k, '+')) {
            if (k == "-" || k.empty()) continue;
            else if (k == "r") p.in.rbutton = 1;
            else if (k == "shift") p.in.lshift = 1;
            else if (k == "q") p.in.key_q = 1;
            else if (k == "z") p.in.key_z = 1;
            else if (k == "w") p.in.key_w = 1;
            else if (k == "a") p.in.key_a = 1;
            else if (k == "s") p.in.key_s = 1;
            else if (k == "d") p.in.key_d = 1;
            else return false;
        }
        polls->push_back(p);
    }
    return true;
}

int main(int argc, char **argv)
{
    int w = 1280, h = 768, n = 32 * 16, frames = 1, variant = RC_VARIANT_AUTO;   // main.cpp:32-33, SceneManager.cpp:8
    unsigned seed = 0;
    rc_vec3 pos{0.0f, 0.0f, 20.0f}, rot{0.0f, 0.0f, 0.0f};
    std::string out = "frame.ppm", pattern, script;
    for (int i = 1; i < argc; i++) {
        auto need = [&](const char *flag) -> const char * {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", flag); std::exit(2); }
            return argv[++i];
        };
        if (!std::strcmp(argv[i], "--width")) w = std::atoi(need("--width"));
        else if (!std::strcmp(argv[i], "--height")) h = std::atoi(need("--height"));
        else if (!std::strcmp(argv[i], "--spheres")) n = std::atoi(need("--spheres"));
        else if (!std::strcmp(argv[i], "--seed")) seed = (unsigned)std::strtoul(need("--seed"), nullptr, 10);
        else if (!std::strcmp(argv[i], "--frames")) frames = std::atoi(need("--frames"));
        else if (!std::strcmp(argv[i], "--variant")) variant = std::atoi(need("--variant"));
        else if (!std::strcmp(argv[i], "--out")) out = need("--out");
        else if (!std::strcmp(argv[i], "--out-pattern")) pattern = need("--out-pattern");
        else if (!std::strcmp(argv[i], "--script")) script = need("--script");
        else if (!std::strcmp(argv[i], "--pos")) { if (!parse3(need("--pos"), &pos.x, &pos.y, &pos.z)) return 2; }
        else if (!std::strcmp(argv[i], "--rot")) { float z; const char *v = need("--rot");
            if (std::sscanf(v, "%f,%f", &rot.x, &rot.y) != 2 && !parse3(v, &rot.x, &rot.y, &z)) return 2; }
        else { std::fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    std::vector<ScriptPoll> polls;
    if (!script.empty()) {
        if (!load_script(script, &polls) || polls.empty()) { std::fprintf(stderr, "cannot read script %s\n", script.c_str()); return 2; }
        frames = (int)polls.size();
    }
    try {
        rc::RayTracingKernel rtk;
        if (!rtk.Init(w, h, n, seed)) return 1;
        rtk.opts.variant = variant;
        rtk.RegisterSurface();
        rtk.sm.cam.position = pos;
        rtk.sm.cam.SetRotation(rot);
        rtk.sm.UpdateCamera();
        auto path_camera = [&](int f) {
            if (!polls.empty()) {                          // scripted HandleInput (main.cpp:53-128)
                if (rtk.sm.cam.HandleInput(polls[(size_t)f].in, polls[(size_t)f].dt)) rtk.sm.UpdateCamera();
                return;
            }
            const double th = 2.0 * 3.14159265358979323846 * f / 300.0;
            rtk.sm.cam.position = rc_vec3{(float)(6.0 * std::sin(th)), (float)(1.5 * std::sin(2 * th)), 20.0f};
            rtk.sm.cam.SetRotation(rc_vec3{(float)(0.15 * std::sin(2 * th)), (float)(0.5 * std::sin(th)), 0.0f});
            rtk.sm.UpdateCamera();
        };
        if (!pattern.empty()) {
            rc::FramePipeline pipe(w, h, true);
            auto write = [&](int f, const void *img) {
                char name[1024];
                std::snprintf(name, sizeof(name), pattern.c_str(), f);
                rc::check(rc_write_ppm(name, static_cast<const uint8_t *>(img), (size_t)w * 4, w, h), "rc_write_ppm");
            };
            for (int f = 0; f < frames; f++) {
                path_camera(f);
                pipe.Submit(rtk.sm.scene, rtk.opts);
                if (f > 0) write(f - 1, pipe.Collect());
            }
            write(frames - 1, pipe.Collect());
            std::printf("wrote %d pipelined frames (%dx%d, %d spheres)\n", frames, w, h, n);
            rtk.Cleanup();
            return 0;
        }
        void *e0 = nullptr, *e1 = nullptr;
        rc::check(rc_event_create(&e0), "rc_event_create");
        rc::check(rc_event_create(&e1), "rc_event_create");
        for (int f = 0; f < frames; f++) {
            if (frames > 1 || !polls.empty()) path_camera(f);
            rc::check(rc_event_record(e0, nullptr), "rc_event_record");
            rtk.Run(false);
            rc::check(rc_event_record(e1, nullptr), "rc_event_record");
            float ms = 0.0f;
            rc::check(rc_event_elapsed_ms(e0, e1, &ms), "rc_event_elapsed_ms");
            std::printf("frame: %f\n", ms);
        }
        rtk.WritePPM(out.c_str());
        std::printf("wrote %s (%dx%d, %d spheres)\n", out.c_str(), w, h, n);
        if (!polls.empty())
            std::printf("camera: position %.9g %.9g %.9g rotation %.9g %.9g\n", rtk.sm.cam.position.x, rtk.sm.cam.position.y,
                        rtk.sm.cam.position.z, rtk.sm.cam.rotation.x, rtk.sm.cam.rotation.y);
        rc_event_destroy(e0);
        rc_event_destroy(e1);
        rtk.Cleanup();
    } catch (const std::exception &e) {
        std::fprintf(stderr, "rc_render: %s\n", e.what());
        return 1;
    }
    return 0;
}
