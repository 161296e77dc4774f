// Writes the KjFrameConstants the C++ host mirror (include/kajiya_amd.hpp: FrameState::prepare_frame_constants) produces for an orbiting
// camera, frame by frame, as raw bytes on stdout — tests/test_cpp_host.py compares them with kajiya_amd/frame.py. Host-only.
//   dump_frame_constants <W> <H> <frames> <cx> <cy> <cz> <radius> <height> <rate>
#include <cstdio>
#include "../include/kajiya_amd.hpp"

int main(int argc, char** argv) {
    if (argc < 10) return 2;
    const uint32_t W = uint32_t(atoi(argv[1])), H = uint32_t(atoi(argv[2]));
    const int frames = atoi(argv[3]);
    double c[6];
    for (int i = 0; i < 6; ++i) c[i] = atof(argv[4 + i]);
    kajiya_amd::FrameState fs(W, H);
    for (int i = 0; i < frames; ++i) {
        const double ang = c[5] * double(i);
        const double eye[3] = {c[0] + c[3] * std::sin(ang), c[1] + c[4], c[2] + c[3] * std::cos(ang)};
        const double target[3] = {c[0], c[1], c[2]};
        const KjFrameConstants fc = fs.prepare_frame_constants(kajiya_amd::CameraMatrices::look_at(eye, target, 52.0, double(W) / double(H)));
        fwrite(&fc, sizeof(fc), 1, stdout);
        fs.retire_frame();
    }
    return 0;
}
