// Test harness (host only): builds the acceleration structure for a synthetic triangle soup and prints a hash of the emitted nodes +
// leaf-ordered triangles. tests/test_bvh_build.py runs it with KJ_BVH_THREADS=1 and with worker threads: the output must be identical.
//   bvh_build_check <n> <kind>   kind: 0 uniform soup, 1 all centroids coincident, 2 a long thin strip (degenerate axes), 3 clustered
#include "../kajiya_amd/csrc/kj_bvh_build.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>
using namespace kj;
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 100000, kind = argc > 2 ? atoi(argv[2]) : 0;
    std::mt19937 rng(7u + unsigned(kind));
    std::uniform_real_distribution<float> U(-50.f, 50.f), S(-0.3f, 0.3f);
    std::vector<BvhTri> tris;
    tris.resize(size_t(n));
    for (int i = 0; i < n; ++i) {
        float c[3] = {U(rng), U(rng) * 0.1f, U(rng)};
        if (kind == 1) c[0] = c[1] = c[2] = 1.0f;
        if (kind == 2) { c[1] = 0.0f; c[2] = 0.0f; }
        if (kind == 3) { const int k = i % 17; c[0] = float(k) * 5.0f + S(rng); c[1] = S(rng); c[2] = float(k * k % 7) + S(rng); }
        for (int k = 0; k < 3; ++k) {
            const float sa = kind == 1 ? (k == i % 3 ? 0.5f : -0.5f) : S(rng);
            tris[i].v0[k] = c[k] + (kind == 1 ? sa : S(rng)); tris[i].v1[k] = c[k] + (kind == 1 ? -sa : S(rng)); tris[i].v2[k] = c[k] + (kind == 1 ? 0.0f : S(rng));
        }
        tris[i].world_id = uint32_t(i); tris[i].inst = 0; tris[i].prim = uint32_t(i);
    }
    BuiltBvh b;
    build_bvh4(tris, b);
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t bytes) { const unsigned char* q = (const unsigned char*)p; for (size_t i = 0; i < bytes; ++i) { h ^= q[i]; h *= 1099511628211ull; } };
    mix(b.nodes.data(), b.nodes.size() * sizeof(BvhNode));
    mix(b.tris.data(), b.tris.size() * sizeof(BvhTri));
    // every triangle is referenced exactly once
    std::vector<unsigned char> seen;
    seen.assign(size_t(n), 0);
    size_t dup = 0;
    for (const BvhTri& t : b.tris) { if (seen[t.world_id]++) ++dup; }
    printf("%zu %zu %u %zu %016llx\n", b.nodes.size(), b.tris.size(), b.max_stack, dup, h);
    return 0;
}
