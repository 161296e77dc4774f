// TEST INFRASTRUCTURE: the product's post-processing kernels + host sequencing (kajiya_amd/csrc/post.hip, compiled from where it lies)
// executed on the CPU through tests/hip_emu (see hip/hip_runtime.h there). Built by tests/test_post_emulation.py into tests/_build/;
// never part of libkajiya_amd.so. The KjDevice a real kj_device_create / kj_frame_begin would set up is filled in by hand.
#include <cstdarg>
#include "../kajiya_amd/csrc/post.hip"

namespace kj {
static thread_local char g_err[512];
void set_last_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace kj

extern "C" {
const char* emu_last_error() { return kj::g_err; }
KjDevice* emu_device_create(const uint8_t* blue_noise_rgba8_256) {
    KjDevice* d = new KjDevice();
    d->blue_noise.upload(blue_noise_rgba8_256, 256 * 256 * 4);
    return d;
}
void emu_device_destroy(KjDevice* d) { delete d; }
void emu_frame_begin(KjDevice* d, const KjFrameConstants* fc) {
    d->fc_host = *fc;
    d->fc_dev = &d->fc_host;
}
uint16_t emu_f32_to_f16(float f) { return kj::f32_to_f16(f); }
float emu_f16_to_f32(uint16_t h) { return kj::f16_to_f32(h); }
}
