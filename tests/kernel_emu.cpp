// TEST INFRASTRUCTURE: one product kernel file (-DKJ_EMU_SOURCE="<path to a .hip under kajiya_amd/csrc>") compiled against the CPU stand-in
// for HIP (tests/hip_emu) together with the little a KjDevice needs. tests/test_kernel_sanitizers.py builds one library per kernel file
// with -fsanitize=address,undefined and drives the real kj_* entry points with the oracle's inputs: the shipped kernel source must
// reproduce the oracle and must not touch a byte outside its buffers. Never part of libkajiya_amd.so.
#include <cstdarg>
#include KJ_EMU_SOURCE

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
}
