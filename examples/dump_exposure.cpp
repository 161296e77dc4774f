// Runs the C++ host mirror's exposure state (include/kajiya_amd.hpp: DynamicExposureState, ExposureState, update_pre_exposure) over a
// sequence of image_log2_lum values read from argv and prints pre_mult post_mult pre_mult_prev pre_mult_delta ev_fast ev_slow per frame
// as raw f32 on stdout — tests/test_cpp_host.py compares them with kajiya_amd/exposure.py. Host-only.
//   dump_exposure <enabled 0|1> <speed_log2> <ev_shift> <mode 0|1> <lum0> <lum1> ...
#include <cstdio>
#include <cstdlib>
#include "../include/kajiya_amd.hpp"

int main(int argc, char** argv) {
    if (argc < 6) return 2;
    kajiya_amd::DynamicExposureState dyn;
    dyn.enabled = atoi(argv[1]) != 0;
    dyn.speed_log2 = float(atof(argv[2]));
    const float ev_shift = float(atof(argv[3]));
    const kajiya_amd::RenderMode mode = atoi(argv[4]) ? kajiya_amd::RenderMode::Reference : kajiya_amd::RenderMode::Standard;
    kajiya_amd::ExposureState st;
    for (int i = 5; i < argc; ++i) {
        kajiya_amd::update_pre_exposure(st, dyn, ev_shift, float(atof(argv[i])), mode);
        const float row[6] = {st.pre_mult, st.post_mult, st.pre_mult_prev, st.pre_mult_delta, dyn.ev_fast, dyn.ev_slow};
        fwrite(row, sizeof(row), 1, stdout);
    }
    return 0;
}
