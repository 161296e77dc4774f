// TEST INFRASTRUCTURE: the post-processing kernels (kajiya_amd/csrc/post.hip through tests/post_emu.cpp) over exact-size heap buffers at awkward extents,
// built with -fsanitize=address,undefined,float-cast-overflow by tests/test_post_emulation.py: an out-of-bounds access a GPU would turn into a memory
// fault (or silently read), a NaN -> int cast, a signed overflow or a bad shift stops the program here.
#include "post_emu.cpp"
#include <random>
int main() {
    std::mt19937 rng(1);
    std::vector<uint8_t> bn(256 * 256 * 4);
    for (auto& b : bn) b = uint8_t(rng());
    KjDevice* d = emu_device_create(bn.data());
    KjFrameConstants fc{};
    fc.pre_exposure = 1.0f; fc.frame_index = 5; fc.view_constants.clip_to_view[11] = 100.0f;
    emu_frame_begin(d, &fc);
    std::vector<uint16_t> lut(128, 0x2000);
    KjPost* p = nullptr; KjMotionBlur* m = nullptr;
    if (kj_post_create(d, lut.data(), &p) || kj_motion_blur_create(d, &m)) return 1;
    const int sizes[][4] = {{1, 1, 1, 1}, {2, 3, 2, 3}, {17, 9, 17, 9}, {64, 64, 64, 64}, {65, 33, 65, 33}, {129, 70, 129, 70}, {200, 120, 100, 60}, {31, 257, 31, 257}};
    for (auto& sz : sizes) {
        const int W = sz[0], H = sz[1], DW = sz[2], DH = sz[3];
        // exact-size heap buffers: ASan flags any read past the end
        std::vector<uint16_t> in(size_t(W) * H * 4);
        for (auto& v : in) v = (rng() % 11 == 0) ? uint16_t(0) : uint16_t(0x3000 + (rng() & 0x0fff));
        for (size_t i = 0; i + 3 < in.size(); i += 44) in[i] = in[i + 1] = in[i + 2] = 0;
        std::vector<float> in32(size_t(W) * H * 4);
        for (auto& v : in32) v = float(rng() & 1023) / 64.0f;
        std::vector<float> depth(size_t(DW) * DH);
        for (auto& v : depth) v = (rng() & 7) ? 0.001f + float(rng() & 255) * 1e-5f : 0.0f;
        std::vector<int16_t> rm(size_t(DW) * DH * 4);
        for (auto& v : rm) v = int16_t(int(rng() % 8001) - 4000);
        const void* out = nullptr;
        if (kj_post_render(p, in.data(), KJ_POST_INPUT_RGBA16F, W, H, 1.0f, 1.0f, &out, nullptr)) { printf("post failed %s\n", emu_last_error()); return 1; }
        if (kj_post_render(p, in32.data(), KJ_POST_INPUT_RGBA32F, W, H, 1.0f, 1.0f, &out, nullptr)) return 1;
        if (kj_motion_blur_render(m, in.data(), W, H, depth.data(), rm.data(), DW, DH, &out, nullptr)) { printf("mb failed %s\n", emu_last_error()); return 1; }
        float lum; uint32_t hist[256];
        kj_post_read_back_histogram(p, 0.1f, 0.1f, &lum, hist);
        printf("%dx%d ok lum %.3f\n", W, H, lum);
    }
    kj_post_destroy(p); kj_motion_blur_destroy(m); emu_device_destroy(d);
    return 0;
}
