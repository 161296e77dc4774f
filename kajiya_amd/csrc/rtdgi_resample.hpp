// Launchers of the screen-space resampling kernels (rtdgi_resample.hip), called by RtdgiRenderer's host side (rtdgi.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdint>
#include "../../include/kajiya_amd.h"

namespace kj {

struct SpatialLaunch {
    const KjFrameConstants* fc;
    const void *reservoir_input, *half_gbuf, *half_depth, *temporal_reservoir_packed;
    void* reservoir_output;
    int W, H, hw, hh;
    uint32_t pass_idx, perform_occlusion_raymarch, occlusion_raymarch_importance_only;
    int row0, row1;
    // 2 (default): G-buffer records gathered per tap; 0: staged in LDS by 32x32 (first pass) / 16x16 workgroups; 1: LDS, 16x16 workgroups.
    // Measured at 1080p (profiles/r02_spatial_variants.md): first pass 37.6 us (2) / 49.7 (0) / 58.0 (1), second pass 44.7 / 48.0 / 47.9:
    // a tap window of 72x72 texels per 8x8 block is ten times the bytes the taps read, and the fill serialises the workgroup.
    int variant;
};
hipError_t launch_restir_spatial(const SpatialLaunch& L, hipStream_t s);

struct ResolveLaunch {
    const KjFrameConstants* fc;
    const void *radiance, *reservoir_input, *gbuffer, *depth, *half_gbuf, *ssao, *candidate_radiance, *candidate_hit, *temporal_reservoir_packed, *blue_noise;
    void* irradiance_output;
    int W, H, hw, hh, row0, row1;   // rows in full-res pixels, row0 a multiple of 16
};
hipError_t launch_restir_resolve(const ResolveLaunch& L, hipStream_t s);
hipError_t launch_temporal_filter(const KjFrameConstants* fc, const void* input, const void* history, const void* variance_history, const void* reprojection, const void* rt_history_invalidity,
                                  void* output, void* history_output, void* variance_history_output, int W, int H, int row0, int row1, hipStream_t s);
hipError_t launch_spatial_filter(const KjFrameConstants* fc, const void* input, const void* depth, const void* ssao, const void* geometric_normal, void* output,
                                 int W, int H, int row0, int row1, hipStream_t s);

}  // namespace kj
