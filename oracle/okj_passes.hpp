// ORACLE (test infrastructure). Screen-space passes of the ReSTIR-GI path,
// restated from the reference HLSL (file:line cited per function). Surfaces are
// flat row-major arrays in the reference's texel formats; out-of-bounds loads
// return 0 and out-of-bounds stores are dropped.
#pragma once
#include "okj_scene.hpp"
#include <map>
#include <string>

namespace okj {

template <typename T> struct Img {
    T* p = nullptr;
    int w = 0, h = 0;
    Img() {}
    Img(void* ptr, int w_, int h_) : p((T*)ptr), w(w_), h(h_) {}
    bool inb(int x, int y) const { return x >= 0 && y >= 0 && x < w && y < h; }
    T ld(int x, int y) const { if (!inb(x, y)) { T z; memset(&z, 0, sizeof(T)); return z; } return p[size_t(y) * w + x]; }
    void st(int x, int y, const T& v) const { if (inb(x, y)) p[size_t(y) * w + x] = v; }
};
typedef Img<h4> ImgRGBA16F;
typedef Img<float> ImgR32F;
typedef Img<uint32_t> ImgU32;
typedef Img<u4> ImgU4;
typedef Img<u2> ImgU2;
typedef Img<uint8_t> ImgR8;
typedef Img<int8_t> ImgR8S;
struct s4 { int16_t x, y, z, w; };
typedef Img<s4> ImgRGBA16S;
struct h2 { uint16_t x, y; };
typedef Img<h2> ImgRG16F;

static inline f4 ld4(const ImgRGBA16F& i, int x, int y) { return unpack_rgba16f(i.ld(x, y)); }
static inline void st4(const ImgRGBA16F& i, int x, int y, f4 v) { i.st(x, y, pack_rgba16f(v)); }
static inline f4 ld_reproj(const ImgRGBA16S& i, int x, int y) {
    s4 v = i.ld(x, y);
    return f4{from_snorm16(v.x), from_snorm16(v.y), from_snorm16(v.z), from_snorm16(v.w)};
}
static inline f2 ld2(const ImgRG16F& i, int x, int y) { h2 v = i.ld(x, y); return f2{f16_to_f32(v.x), f16_to_f32(v.y)}; }
static inline void st2(const ImgRG16F& i, int x, int y, f2 v) { i.st(x, y, h2{f32_to_f16(v.x), f32_to_f16(v.y)}); }
static inline f3 ld_nrm_snorm8(const ImgU32& i, int x, int y) { return xyz(unpack_rgba8_snorm(i.ld(x, y))); }

// nearest, clamp-to-edge (sampler_nnc): texel = floor(uv*size), clamped
template <typename T> static inline T sample_nearest_clamp(const Img<T>& i, f2 uv) {
    int x = int(floorf(uv.x * float(i.w))), y = int(floorf(uv.y * float(i.h)));
    x = std::min(std::max(x, 0), i.w - 1);
    y = std::min(std::max(y, 0), i.h - 1);
    return i.p[size_t(y) * i.w + x];
}
static inline f4 sample_bilinear_clamp(const ImgRGBA16F& i, f2 uv) { return sample_bilinear_clamp_rgba16f(i.p, i.w, i.h, uv); }
static inline f2 sample_bilinear_clamp(const ImgRG16F& i, f2 uv) {
    float fx = uv.x * float(i.w) - 0.5f, fy = uv.y * float(i.h) - 0.5f;
    float x0f = floorf(fx), y0f = floorf(fy);
    float tx = fx - x0f, ty = fy - y0f;
    int x0 = int(x0f), y0 = int(y0f);
    auto cl = [](int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); };
    int xa = cl(x0, i.w), xb = cl(x0 + 1, i.w), ya = cl(y0, i.h), yb = cl(y0 + 1, i.h);
    f2 s00 = ld2(i, xa, ya), s10 = ld2(i, xb, ya), s01 = ld2(i, xa, yb), s11 = ld2(i, xb, yb);
    f2 a = s00 * (1.0f - tx) + s10 * tx;
    f2 b = s01 * (1.0f - tx) + s11 * tx;
    return a * (1.0f - ty) + b * ty;
}

// ------------------------------------------------------------------ G-buffer stand-in
// Primary rays through the jittered camera; packing as raster_simple_ps.hlsl:126-137.
static inline void raster_gbuffer(const Scene& sc, const FrameConstants& fc, int W, int H,
                                  ImgU32 geometric_normal, ImgU4 gbuffer, ImgR32F depth, ImgRGBA16F velocity) {
    const f4 ts = tex_size4(W, H);
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            f2 uv = get_uv(float(x), float(y), ts);
            ViewRayContext vrc = ViewRayContext::from_uv(fc, uv);
            Ray ray{vrc.ray_origin_ws(), 0.0f, vrc.ray_dir_ws(), FLT_MAX};
            GbufferPathVertex pv = gbuffer_raytrace(sc, fc, ray, 0, false, pixel_ray_cone_from_image_height(fc, float(H)));   // stand-in for the raster pass's implicit LOD
            if (!pv.is_hit) {
                geometric_normal.st(x, y, 0);
                gbuffer.st(x, y, u4{0, 0, 0, 0});
                depth.st(x, y, 0.0f);
                st4(velocity, x, y, mk4(0.0f));
                continue;
            }
            Hit h = sc.trace_closest(ray, false);
            const WorldTri& wt = sc.tris[h.tri];
            f3 gn_ws = normalize(cross(wt.v1 - wt.v0, wt.v2 - wt.v0));
            if (dot(gn_ws, ray.d) > 0) gn_ws = -gn_ws;
            f3 gn_vs = normalize(direction_world_to_view(fc, gn_ws));
            f3 cs = position_world_to_sample(fc, pv.position);
            geometric_normal.st(x, y, pack_a2r10g10b10(gn_vs * 0.5f + 0.5f));
            gbuffer.st(x, y, pv.gbuffer_packed);
            depth.st(x, y, cs.z);
            st4(velocity, x, y, mk4(0.0f));
        }
}

// ------------------------------------------------------------------ calculate_reprojection_map.hlsl:17-142
struct Bilinear { f2 origin, weights; };
static inline Bilinear get_bilinear_filter(f2 uv, f2 tex_size) {
    Bilinear r;
    f2 p = uv * tex_size - 0.5f;
    r.origin = f2{truncf(p.x), truncf(p.y)};
    r.weights = f2{frac(p.x), frac(p.y)};
    return r;
}
static inline void calculate_reprojection_map(const FrameConstants& fc, int W, int H, ImgR32F depth_tex, ImgU32 geometric_normal_tex,
                                              ImgR32F prev_depth_tex, ImgRGBA16F velocity_tex, ImgRGBA16S output_tex) {
    const KjViewConstants& vc = fc.view_constants;
    const f4 ts = tex_size4(W, H);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            f2 uv = get_uv(float(x), float(y), ts);
            auto store = [&](f4 v) { output_tex.st(x, y, s4{to_snorm16(v.x), to_snorm16(v.y), to_snorm16(v.z), to_snorm16(v.w)}); };
            if (depth_tex.ld(x, y) == 0.0f) {
                f2 cs = uv_to_cs(uv);
                f4 pos_cs{cs.x, cs.y, 0.0f, 1.0f};
                f4 pos_vs = mul44(vc.clip_to_view, pos_cs);
                f4 prev_cs = mul44(vc.view_to_clip, pos_vs);
                f4 prev_pcs = mul44(vc.clip_to_prev_clip, prev_cs);
                f2 prev_uv = cs_to_uv(f2{prev_pcs.x, prev_pcs.y});
                f2 uv_diff = prev_uv - uv;
                store(f4{uv_diff.x, uv_diff.y, 0, 0});
                continue;
            }
            float depth = 0.0f;
            {
                float s = depth_tex.ld(x, y);
                if (s != 0.0f) depth = fmaxf(depth, s);
            }
            f3 normal_vs = unpack_a2r10g10b10(geometric_normal_tex.ld(x, y)) * 2.0f - 1.0f;
            f3 normal_pvs = xyz(mul44(vc.prev_clip_to_prev_view, mul44(vc.clip_to_prev_clip, mul44(vc.view_to_clip, mk4(normal_vs, 0)))));
            f2 cs = uv_to_cs(uv);
            f4 pos_cs{cs.x, cs.y, depth, 1.0f};
            f4 pos_vs = mul44(vc.clip_to_view, pos_cs);
            float dist_to_point = -(pos_vs.z / pos_vs.w);
            f4 prev_vs = pos_vs / pos_vs.w;
            f4 vel = ld4(velocity_tex, x, y);
            prev_vs.x += vel.x; prev_vs.y += vel.y; prev_vs.z += vel.z;
            f4 prev_cs = mul44(vc.view_to_clip, prev_vs);
            f4 prev_pcs = mul44(vc.clip_to_prev_clip, prev_cs);
            f2 prev_uv = cs_to_uv(f2{prev_pcs.x / prev_pcs.w, prev_pcs.y / prev_pcs.w});
            f2 uv_diff = prev_uv - uv;
            uv_diff = f2{floorf(uv_diff.x * 32767.0f + 0.5f) / 32767.0f, floorf(uv_diff.y * 32767.0f + 0.5f) / 32767.0f};
            prev_uv = uv + uv_diff;
            f4 prev_pvs = mul44(vc.prev_clip_to_prev_view, prev_pcs);
            prev_pvs = prev_pvs / prev_pvs.w;
            float plane_dist_prev_dz = fminf(-0.2f, normal_vs.z);
            const Bilinear bl = get_bilinear_filter(prev_uv, f2{float(W), float(H)});
            // GatherRed(...).wzxy at (origin+1)/size: x=(0,0) y=(1,0) z=(0,1) w=(1,1) relative to origin
            int ox = int(bl.origin.x), oy = int(bl.origin.y);
            auto pd = [&](int dx, int dy) {
                int sx = std::min(std::max(ox + dx, 0), W - 1), sy = std::min(std::max(oy + dy, 0), H - 1);
                return prev_depth_tex.p[size_t(sy) * W + sx];
            };
            f4 prev_depth{pd(0, 0), pd(1, 0), pd(0, 1), pd(1, 1)};
            const float m43 = -vc.prev_clip_to_prev_view[11];
            f4 prev_view_z{1.0f / (prev_depth.x * m43), 1.0f / (prev_depth.y * m43), 1.0f / (prev_depth.z * m43), 1.0f / (prev_depth.w * m43)};
            f4 quad_dists{fabsf(plane_dist_prev_dz * (prev_view_z.x - prev_pvs.z)), fabsf(plane_dist_prev_dz * (prev_view_z.y - prev_pvs.z)),
                          fabsf(plane_dist_prev_dz * (prev_view_z.z - prev_pvs.z)), fabsf(plane_dist_prev_dz * (prev_view_z.w - prev_pvs.z))};
            const float acceptance_threshold = 0.001f * (1080.0f / float(H));
            const f3 pos_vs_norm = normalize(xyz(pos_vs) / pos_vs.w);
            const float ndotv = dot(normal_vs, pos_vs_norm);
            const float prev_ndotv = dot(normal_pvs, normalize(xyz(prev_pvs)));
            const float thr = acceptance_threshold * dist_to_point / -ndotv;
            f4 qv{step(quad_dists.x, thr), step(quad_dists.y, thr), step(quad_dists.z, thr), step(quad_dists.w, thr)};
            auto inb = [&](int dx, int dy) { int sx = ox + dx, sy = oy + dy; return (sx >= 0 && sy >= 0 && sx < W && sy < H) ? 1.0f : 0.0f; };
            qv.x *= inb(0, 0); qv.y *= inb(1, 0); qv.z *= inb(0, 1); qv.w *= inb(1, 1);
            float validity = dot(qv, f4{1, 2, 4, 8}) / 15.0f;
            float accuracy = 1;
            accuracy *= smoothstep(0.8f, 0.95f, prev_ndotv / ndotv);
            if (saturate(prev_uv.x) != prev_uv.x || saturate(prev_uv.y) != prev_uv.y) accuracy = -1;
            store(f4{uv_diff.x, uv_diff.y, validity, accuracy});
        }
}

// ------------------------------------------------------------------ extract_half_res_*.hlsl
static inline void extract_half_res(const FrameConstants& fc, int W, int H, ImgU4 gbuffer, ImgR32F depth, ImgR8 ssao,
                                    ImgU32 half_view_normal, ImgR32F half_depth, ImgR8S half_ssao) {
    const int hw = half_depth.w, hh = half_depth.h;
    const i2 off = halfres_subsample_offset(fc);
    (void)W; (void)H;
    for (int y = 0; y < hh; ++y)
        for (int x = 0; x < hw; ++x) {
            int sx = x * 2 + off.x, sy = y * 2 + off.y;
            f3 normal_ws = unpack_normal_11_10_11_no_normalize(gbuffer.ld(sx, sy).y);
            f3 normal_vs = normalize(xyz(mul44(fc.view_constants.world_to_view, mk4(normal_ws, 0))));
            half_view_normal.st(x, y, pack_rgba8_snorm(mk4(normal_vs, 1.0f)));
            half_depth.st(x, y, depth.ld(sx, sy));
            half_ssao.st(x, y, to_snorm8(from_unorm8(ssao.ld(sx, sy))));
        }
}

} // namespace okj
