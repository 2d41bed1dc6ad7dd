// gen_kat_scene.cpp -- writes the scene the reference known-answer vectors were taken on.
//
// The vectors in ref_kat.json were produced by the reference's device code with the scene of
// SceneManager.cpp:31-45: std::mt19937{0} (SceneManager.h:22) through libstdc++'s
// std::uniform_real_distribution<float>, whose output is implementation-defined.  This tool
// uses only <random> (nothing of the reference) and dumps the first N spheres as raw little-
// endian float32 rows {radius, r, g, b, x, y, z}; the 8- and 64-sphere scenes are prefixes.
//
//   g++ -O2 gen_kat_scene.cpp -o /tmp/gen_kat_scene && /tmp/gen_kat_scene 1024 kat_scene_libstdcxx_1024.f32
#include <cstdio>
#include <cstdlib>
#include <random>

int main(int argc, char **argv)
{
    if (argc != 3) return 2;
    int n = std::atoi(argv[1]);
    std::mt19937 gen{0};
    auto rf = [&gen](float lo, float hi) { return std::uniform_real_distribution<float>{lo, hi}(gen); };
    FILE *f = std::fopen(argv[2], "wb");
    if (!f) return 2;
    for (int i = 0; i < n; i++) {
        float row[7];
        row[0] = rf(0.3, 1);
        row[1] = rf(0, 1); row[2] = rf(0, 1); row[3] = rf(0, 1);
        row[4] = rf(-10, 10); row[5] = rf(-5, 5); row[6] = rf(-15, -5);
        std::fwrite(row, sizeof(float), 7, f);
    }
    std::fclose(f);
    return 0;
}
