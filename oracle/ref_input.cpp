// ref_input.cpp -- cross-check of rc_handle_input() against the REFERENCE'S OWN Camera / vec3.
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/ref_input with
//     -I/root/reference/RayCasting/RayCasting
// so Camera.h and vec3.h are compiled where they lie, untouched.
//
// The reference's HandleInput (main.cpp:53-128) reads the keyboard and the cursor through Win32
// calls, so it cannot be compiled here; what it DOES with that state is restated below against
// the reference's own Camera and vec3 types (ref_poll), so every vec3 operator, norm(), the
// cached GetForward/GetUp and SetRotation are the reference's code.  rc_handle_input (the
// library's scripted replacement) must leave position and rotation bit-identical after every
// poll of a random script.
//
//   ref_input                 run the comparison (exit status 0 = identical)
//   ref_input --emit FILE     also write a 96-poll script with the expected camera after every
//                             poll as JSON (committed as tests/golden/input_script.json)

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

#include "Camera.h"
#include "vec3.h"

#include "rc_abi.h"

// main.cpp:53-128 with the Win32 reads replaced by `in`; dt in milliseconds (main.cpp:137).
static bool ref_poll(Camera *cam, const rc_input &in, float dt)
{
    bool update = false;
    if (in.rbutton) {                                                   // :64-70
        float angle = dt * cam->rotation_speed;
        vec3 change = {-(float)(in.cursor_dy) * angle, -(float)(in.cursor_dx) * angle, 0};
        cam->SetRotation(cam->GetRotation() + change);
        update = true;
    }
    float speed = cam->movement_speed;                                  // :73-80
    if (in.lshift) speed *= 5;
    else speed *= 50;
    if (in.key_q) { cam->position.y += dt * speed; update = true; }     // :82-86
    if (in.key_z) { cam->position.y -= dt * speed; update = true; }     // :88-92
    if (in.key_w) { vec3 f = cam->GetForward(); f.y = 0; cam->position += f.norm() * speed * dt; update = true; }
    if (in.key_a) { vec3 r = cam->GetRight(); r.y = 0; cam->position -= r.norm() * speed * dt; update = true; }
    if (in.key_s) { vec3 f = cam->GetForward(); f.y = 0; cam->position -= f.norm() * speed * dt; update = true; }
    if (in.key_d) { vec3 r = cam->GetRight(); r.y = 0; cam->position += r.norm() * speed * dt; update = true; }
    return update;
}

static uint32_t bits(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

static bool same(float a, float b) { return bits(a) == bits(b) || (a != a && b != b); }

static rc_input random_poll(std::mt19937 &gen)
{
    const uint32_t k = gen();
    rc_input in;
    in.rbutton = (k >> 0) & 1;
    in.lshift = (k >> 1) & 1;
    in.key_q = ((k >> 2) & 3) == 0;
    in.key_z = ((k >> 4) & 3) == 0;
    in.key_w = (k >> 6) & 1;
    in.key_a = ((k >> 7) & 3) == 0;
    in.key_s = ((k >> 9) & 3) == 0;
    in.key_d = (k >> 11) & 1;
    in.cursor_dx = (int)((k >> 12) % 41u) - 20;
    in.cursor_dy = (int)((k >> 20) % 41u) - 20;
    return in;
}

int main(int argc, char **argv)
{
    const char *emit = (argc == 3 && !std::strcmp(argv[1], "--emit")) ? argv[2] : nullptr;
    std::mt19937 gen(2024);
    long polls = 0, bad = 0;
    FILE *js = emit ? std::fopen(emit, "w") : nullptr;
    if (emit && !js) { std::perror(emit); return 2; }
    if (js) std::fprintf(js, "{\"comment\": \"input script and the camera the reference's Camera/vec3 types give after every poll "
                             "(oracle/ref_input.cpp --emit); floats as raw binary32 bit patterns\",\n \"movement_speed\": 100, "
                             "\"rotation_speed\": 10, \"start_position\": [0, 0, 20], \"polls\": [\n");

    for (int run = 0; run < 2000; run++) {
        Camera cam(1280, 768);                                          // main.cpp:32-33
        cam.position = vec3(0, 0, 20);                                  // SceneManager.cpp:5
        rc_vec3 pos = {0.0f, 0.0f, 20.0f}, rot = {0.0f, 0.0f, 0.0f};
        const int steps = (run == 0) ? 96 : 50;
        for (int i = 0; i < steps; i++) {
            const rc_input in = random_poll(gen);
            // frame times from 0.05 ms up to 0.55 ms or 40 ms
            const float dt = 0.05f + (float)(gen() >> 8) * (1.0f / 16777216.0f) * (run % 3 == 0 ? 0.5f : 40.0f);
            const bool u_ref = ref_poll(&cam, in, dt);
            const int u_lib = rc_handle_input(&pos, &rot, cam.movement_speed, cam.rotation_speed, dt, &in);
            const vec3 r = cam.GetRotation();
            const bool ok = same(cam.position.x, pos.x) && same(cam.position.y, pos.y) && same(cam.position.z, pos.z) &&
                            same(r.x, rot.x) && same(r.y, rot.y) && same(r.z, rot.z) && (u_lib == (u_ref ? 1 : 0));
            if (!ok) bad++;
            polls++;
            if (js && run == 0)
                std::fprintf(js, "  {\"dt_bits\": %u, \"keys\": [%d,%d,%d,%d,%d,%d,%d,%d], \"cursor\": [%d,%d], \"update\": %d, "
                                 "\"position_bits\": [%u,%u,%u], \"rotation_bits\": [%u,%u,%u]}%s\n",
                             bits(dt), in.rbutton, in.lshift, in.key_q, in.key_z, in.key_w, in.key_a, in.key_s, in.key_d,
                             in.cursor_dx, in.cursor_dy, u_ref ? 1 : 0, bits(cam.position.x), bits(cam.position.y),
                             bits(cam.position.z), bits(r.x), bits(r.y), bits(r.z), i + 1 < steps ? "," : "");
        }
    }
    if (js) {
        std::fprintf(js, " ]}\n");
        std::fclose(js);
    }
    std::printf("ref_input: %ld polls, %ld mismatches\n", polls, bad);
    return bad == 0 ? 0 : 1;
}
