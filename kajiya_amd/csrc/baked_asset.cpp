// Reader for the reference's baked asset files: `cache/<name>.mesh` = PackedTriMesh::Flat and
// `cache/<identity:08x>.image` = GpuImage::Flat (kajiya-asset/src/mesh.rs:787-807), the files `bin/bake` writes
// (kajiya-asset-pipe/src/lib.rs:38-60) and WorldRenderer::add_mesh / load_gpu_image_asset mmap (world_renderer.rs:297-322,604-700).
//
// The format (mesh.rs:460-632): a struct of FlatVec headers {u64 len; u64 offset}, `offset` counted in bytes from the
// address of the offset field itself to the first element; nested vectors are again FlatVec headers. Everything is packed
// (no padding), native endian. The views below are zero-copy: pointers into the caller's bytes, every range bounds-checked
// (the reference trusts the file; a truncated file here is KJ_ERR_INVALID_ARGUMENT, not a fault).
//
// Texel decode to RGBA8 for every format the baker emits: R8G8B8A8 (copy), BC1, BC3, BC4, BC5 and BC7 (image.rs:131-213,289-336) — the
// job of the GPU's fixed-function texture unit in the reference.
#include "kj_host.hpp"
#include <cstring>
#include <utility>

using namespace kj;

namespace {

struct FlatVecRaw { uint64_t len, offset; };

// Resolves the FlatVec whose header starts at `hdr_off`; returns false if header or payload leave [0, size).
bool resolve(const uint8_t* base, uint64_t size, uint64_t hdr_off, uint64_t elem_size, const uint8_t** data, uint64_t* len) {
    if (hdr_off > size || size - hdr_off < 16) return false;
    FlatVecRaw h;
    memcpy(&h, base + hdr_off, 16);
    uint64_t field = hdr_off + 8;                         // offsets are relative to the `offset` field (mesh.rs:498-501)
    if (h.offset > size - field) return false;
    uint64_t start = field + h.offset;
    if (elem_size && h.len > (size - start) / elem_size) return false;
    *data = h.len ? base + start : nullptr;
    *len = h.len;
    return true;
}

enum : uint32_t { // ash::vk::Format values the baker can emit (image.rs:133-134,190-199,312-325)
    VK_R8G8B8A8_UNORM = 37, VK_R8G8B8A8_SRGB = 43,
    VK_BC1_RGB_UNORM = 131, VK_BC1_RGB_SRGB = 132, VK_BC1_RGBA_UNORM = 133, VK_BC1_RGBA_SRGB = 134,
    VK_BC3_UNORM = 137, VK_BC3_SRGB = 138, VK_BC4_UNORM = 139, VK_BC5_UNORM = 141,
    VK_BC7_UNORM = 145, VK_BC7_SRGB = 146,
};

void bc1_colors(const uint8_t* b, uint8_t pal[4][4], bool punch_through) {
    uint32_t c0 = b[0] | (b[1] << 8), c1 = b[2] | (b[3] << 8);
    auto expand = [](uint32_t c, uint8_t* o) {
        uint32_t r = (c >> 11) & 31, g = (c >> 5) & 63, bl = c & 31;
        o[0] = uint8_t((r << 3) | (r >> 2)); o[1] = uint8_t((g << 2) | (g >> 4)); o[2] = uint8_t((bl << 3) | (bl >> 2)); o[3] = 255;
    };
    expand(c0, pal[0]); expand(c1, pal[1]);
    for (int k = 0; k < 3; ++k) {
        if (c0 > c1 || !punch_through) {
            pal[2][k] = uint8_t((2 * pal[0][k] + pal[1][k]) / 3);
            pal[3][k] = uint8_t((pal[0][k] + 2 * pal[1][k]) / 3);
        } else {
            pal[2][k] = uint8_t((pal[0][k] + pal[1][k]) / 2);
            pal[3][k] = 0;
        }
    }
    pal[2][3] = 255;
    pal[3][3] = (c0 > c1 || !punch_through) ? 255 : 0;
}

void bc4_values(const uint8_t* b, uint8_t v[8]) {
    uint32_t a0 = b[0], a1 = b[1];
    v[0] = uint8_t(a0); v[1] = uint8_t(a1);
    if (a0 > a1) for (uint32_t i = 1; i < 7; ++i) v[i + 1] = uint8_t(((7 - i) * a0 + i * a1) / 7);
    else { for (uint32_t i = 1; i < 5; ++i) v[i + 1] = uint8_t(((5 - i) * a0 + i * a1) / 5); v[6] = 0; v[7] = 255; }
}

// out: 16 texels x 4 bytes in row-major block order
void decode_block(uint32_t fmt, const uint8_t* b, uint8_t out[16][4]) {
    const bool bc1 = fmt >= VK_BC1_RGB_UNORM && fmt <= VK_BC1_RGBA_SRGB;
    if (bc1 || fmt == VK_BC3_UNORM || fmt == VK_BC3_SRGB) {
        const uint8_t* cb = bc1 ? b : b + 8;
        uint8_t pal[4][4];
        bc1_colors(cb, pal, bc1);
        uint32_t idx = cb[4] | (cb[5] << 8) | (cb[6] << 16) | (uint32_t(cb[7]) << 24);
        for (int i = 0; i < 16; ++i) memcpy(out[i], pal[(idx >> (2 * i)) & 3], 4);
        if (fmt == VK_BC1_RGB_UNORM || fmt == VK_BC1_RGB_SRGB) for (int i = 0; i < 16; ++i) out[i][3] = 255;
        if (!bc1) {
            uint8_t v[8];
            bc4_values(b, v);
            uint64_t bits = 0;
            for (int k = 0; k < 6; ++k) bits |= uint64_t(b[2 + k]) << (8 * k);
            for (int i = 0; i < 16; ++i) out[i][3] = v[(bits >> (3 * i)) & 7];
        }
        return;
    }
    // BC4 (r,0,0,1) / BC5 (r,g,0,1)
    const int channels = fmt == VK_BC5_UNORM ? 2 : 1;
    for (int i = 0; i < 16; ++i) { out[i][0] = out[i][1] = out[i][2] = 0; out[i][3] = 255; }
    for (int c = 0; c < channels; ++c) {
        const uint8_t* cb = b + 8 * c;
        uint8_t v[8];
        bc4_values(cb, v);
        uint64_t bits = 0;
        for (int k = 0; k < 6; ++k) bits |= uint64_t(cb[2 + k]) << (8 * k);
        for (int i = 0; i < 16; ++i) out[i][c] = v[(bits >> (3 * i)) & 7];
    }
}

#include "bc7_tables.inc"

// BC7 (BPTC) block -> 16 RGBA8 texels. Mode table, bit order, endpoint expansion and interpolation weights per the format specification;
// partition / anchor tables in bc7_tables.inc (scripts/derive_bc7_tables.py). Checked against Pillow's decoder on random blocks of every
// mode (tests/test_baked_assets.py).
struct Bc7Mode { uint8_t ns, pb, rb, isb, cb, ab, epb, spb, ib, ib2; };
static const Bc7Mode BC7_MODES[8] = {{3, 4, 0, 0, 4, 0, 1, 0, 3, 0}, {2, 6, 0, 0, 6, 0, 0, 1, 3, 0}, {3, 6, 0, 0, 5, 0, 0, 0, 2, 0}, {2, 6, 0, 0, 7, 0, 1, 0, 2, 0},
                                     {1, 0, 2, 1, 5, 6, 0, 0, 2, 3}, {1, 0, 2, 0, 7, 8, 0, 0, 2, 2}, {1, 0, 0, 0, 7, 7, 1, 0, 4, 0}, {2, 6, 0, 0, 5, 5, 1, 0, 2, 0}};
static const uint8_t BC7_W2[4] = {0, 21, 43, 64}, BC7_W3[8] = {0, 9, 18, 27, 37, 46, 55, 64}, BC7_W4[16] = {0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64};

void decode_bc7_block(const uint8_t* b, uint8_t out[16][4]) {
    uint32_t pos = 0;
    auto get = [&](uint32_t n) -> uint32_t {
        uint32_t v = 0;
        for (uint32_t i = 0; i < n; ++i, ++pos) v |= uint32_t((b[pos >> 3] >> (pos & 7)) & 1u) << i;
        return v;
    };
    uint32_t mode = 0;
    while (mode < 8 && get(1) == 0) ++mode;
    if (mode >= 8) { memset(out, 0, 64); return; }                 // reserved: decodes to transparent black
    const Bc7Mode m = BC7_MODES[mode];
    const uint32_t partition = get(m.pb), rotation = get(m.rb), idx_mode = get(m.isb);
    const uint32_t ne = 2u * m.ns;
    uint32_t ep[6][4];
    for (int ch = 0; ch < 3; ++ch) for (uint32_t e = 0; e < ne; ++e) ep[e][ch] = get(m.cb);
    for (uint32_t e = 0; e < ne; ++e) ep[e][3] = m.ab ? get(m.ab) : 255u;
    uint32_t cbits = m.cb, abits = m.ab;
    if (m.epb) {
        for (uint32_t e = 0; e < ne; ++e) {
            const uint32_t pbit = get(1);
            for (int ch = 0; ch < 3; ++ch) ep[e][ch] = (ep[e][ch] << 1) | pbit;
            if (m.ab) ep[e][3] = (ep[e][3] << 1) | pbit;
        }
        ++cbits; if (m.ab) ++abits;
    } else if (m.spb) {
        for (uint32_t sub = 0; sub < m.ns; ++sub) {
            const uint32_t pbit = get(1);
            for (uint32_t e = 2 * sub; e < 2 * sub + 2; ++e) for (int ch = 0; ch < 3; ++ch) ep[e][ch] = (ep[e][ch] << 1) | pbit;
        }
        ++cbits;
    }
    for (uint32_t e = 0; e < ne; ++e) {
        for (int ch = 0; ch < 3; ++ch) { uint32_t v = ep[e][ch] << (8 - cbits); ep[e][ch] = v | (v >> cbits); }
        if (m.ab) { uint32_t v = ep[e][3] << (8 - abits); ep[e][3] = v | (v >> abits); }
    }
    const uint8_t* part = m.ns == 1 ? nullptr : (m.ns == 2 ? BC7_PARTITION2[partition] : BC7_PARTITION3[partition]);
    const uint32_t anchor1 = m.ns == 2 ? BC7_ANCHOR2[partition] : (m.ns == 3 ? BC7_ANCHOR3A[partition] : 0xffu);
    const uint32_t anchor2 = m.ns == 3 ? BC7_ANCHOR3B[partition] : 0xffu;
    uint32_t idx0[16], idx1[16];
    for (uint32_t i = 0; i < 16; ++i) idx0[i] = get((i == 0 || i == anchor1 || i == anchor2) ? m.ib - 1u : m.ib);
    for (uint32_t i = 0; i < 16; ++i) idx1[i] = m.ib2 ? get(i == 0 ? m.ib2 - 1u : m.ib2) : 0u;
    auto weight = [](uint32_t bits, uint32_t i) -> uint32_t { return bits == 2 ? BC7_W2[i] : (bits == 3 ? BC7_W3[i] : BC7_W4[i]); };
    for (uint32_t i = 0; i < 16; ++i) {
        const uint32_t sub = part ? part[i] : 0u;
        const uint32_t* e0 = ep[2 * sub];
        const uint32_t* e1 = ep[2 * sub + 1];
        uint32_t cw, aw;
        if (!m.ib2) { cw = aw = weight(m.ib, idx0[i]); }
        else if (!idx_mode) { cw = weight(m.ib, idx0[i]); aw = weight(m.ib2, idx1[i]); }
        else { cw = weight(m.ib2, idx1[i]); aw = weight(m.ib, idx0[i]); }
        uint8_t px[4];
        for (int ch = 0; ch < 3; ++ch) px[ch] = uint8_t(((64 - cw) * e0[ch] + cw * e1[ch] + 32) >> 6);
        px[3] = uint8_t(((64 - aw) * e0[3] + aw * e1[3] + 32) >> 6);
        if (rotation == 1) std::swap(px[3], px[0]);
        else if (rotation == 2) std::swap(px[3], px[1]);
        else if (rotation == 3) std::swap(px[3], px[2]);
        memcpy(out[i], px, 4);
    }
}

uint32_t block_bytes(uint32_t fmt) {
    switch (fmt) {
        case VK_BC1_RGB_UNORM: case VK_BC1_RGB_SRGB: case VK_BC1_RGBA_UNORM: case VK_BC1_RGBA_SRGB: case VK_BC4_UNORM: return 8;
        case VK_BC3_UNORM: case VK_BC3_SRGB: case VK_BC5_UNORM: case VK_BC7_UNORM: case VK_BC7_SRGB: return 16;
        default: return 0;
    }
}

} // namespace

extern "C" {

KjStatus kj_baked_mesh_view(const void* bytes, uint64_t size, KjBakedMeshView* out) {
    KJ_REQUIRE(bytes && out, "null argument");
    const uint8_t* base = (const uint8_t*)bytes;
    memset(out, 0, sizeof(*out));
    // PackedTriMesh::Flat (mesh.rs:796-807): verts, uvs, tangents, colors, indices, material_ids, materials, maps
    static const uint64_t elem[8] = {sizeof(KjPackedVertex), 8, 16, 16, 4, 4, sizeof(KjMeshMaterial), 8};
    const uint8_t* p[8];
    uint64_t n[8];
    for (int i = 0; i < 8; ++i) KJ_REQUIRE(resolve(base, size, 16u * i, elem[i], &p[i], &n[i]), "baked mesh: vector out of bounds (truncated or foreign file)");
    for (int i = 0; i < 8; ++i) KJ_REQUIRE(n[i] <= 0xffffffffull, "baked mesh: vector too long");
    KJ_REQUIRE(n[1] == 0 || n[1] == n[0], "baked mesh: uv count differs from the vertex count");
    KJ_REQUIRE(n[2] == 0 || n[2] == n[0], "baked mesh: tangent count differs from the vertex count");
    KJ_REQUIRE(n[3] == 0 || n[3] == n[0], "baked mesh: colour count differs from the vertex count");
    KJ_REQUIRE(n[5] == 0 || n[5] == n[0], "baked mesh: material id count differs from the vertex count");
    KJ_REQUIRE(n[4] % 3 == 0, "baked mesh: index count is not a multiple of 3");
    out->verts = (const KjPackedVertex*)p[0];  out->vertex_count = uint32_t(n[0]);
    out->uvs = (const float*)p[1];
    out->tangents = (const float*)p[2];
    out->colors = (const float*)p[3];
    out->indices = (const uint32_t*)p[4];      out->index_count = uint32_t(n[4]);
    out->material_ids = (const uint32_t*)p[5];
    out->materials = (const KjMeshMaterial*)p[6]; out->material_count = uint32_t(n[6]);
    out->map_identities = (const uint64_t*)p[7];  out->map_count = uint32_t(n[7]);
    return KJ_OK;
}

KjStatus kj_baked_image_view(const void* bytes, uint64_t size, KjBakedImageView* out) {
    KJ_REQUIRE(bytes && out, "null argument");
    KJ_REQUIRE(size >= 32, "baked image: shorter than the GpuImage::Flat header");
    const uint8_t* base = (const uint8_t*)bytes;
    // GpuImage::Flat (mesh.rs:787-793): format i32, extent [u32; 3], mips FlatVec<FlatVec<u8>>
    memcpy(&out->vk_format, base, 4);
    memcpy(out->extent, base + 4, 12);
    const uint8_t* mips; uint64_t n;
    KJ_REQUIRE(resolve(base, size, 16, 16, &mips, &n), "baked image: mip table out of bounds");
    KJ_REQUIRE(n >= 1 && n <= 32, "baked image: bad mip count");
    out->mip_count = uint32_t(n);
    return KJ_OK;
}

KjStatus kj_baked_image_mip(const void* bytes, uint64_t size, uint32_t level, const uint8_t** out_data, uint64_t* out_len) {
    KJ_REQUIRE(bytes && out_data && out_len, "null argument");
    const uint8_t* base = (const uint8_t*)bytes;
    const uint8_t* mips; uint64_t n;
    KJ_REQUIRE(size >= 32 && resolve(base, size, 16, 16, &mips, &n), "baked image: mip table out of bounds");
    KJ_REQUIRE(level < n, "baked image: mip level out of range");
    KJ_REQUIRE(resolve(base, size, uint64_t(mips - base) + 16ull * level, 1, out_data, out_len), "baked image: mip data out of bounds");
    return KJ_OK;
}

KjStatus kj_baked_image_decode_rgba8(uint32_t vk_format, const uint8_t* mip_data, uint64_t mip_len, uint32_t width, uint32_t height, uint8_t* out_rgba8) {
    KJ_REQUIRE(mip_data && out_rgba8 && width && height, "null argument");
    if (vk_format == VK_R8G8B8A8_UNORM || vk_format == VK_R8G8B8A8_SRGB) {
        KJ_REQUIRE(mip_len >= uint64_t(width) * height * 4, "baked image: mip shorter than width*height*4");
        memcpy(out_rgba8, mip_data, size_t(width) * height * 4);
        return KJ_OK;
    }
    const uint32_t bb = block_bytes(vk_format);
    if (!bb) {
        set_last_error("baked image: vk::Format %u is not decoded natively", vk_format);
        return KJ_ERR_UNSUPPORTED;
    }
    const uint32_t bw = (width + 3) / 4, bh = (height + 3) / 4;
    KJ_REQUIRE(mip_len >= uint64_t(bw) * bh * bb, "baked image: mip shorter than its block count");
    for (uint32_t by = 0; by < bh; ++by)
        for (uint32_t bx = 0; bx < bw; ++bx) {
            uint8_t texels[16][4];
            if (vk_format == VK_BC7_UNORM || vk_format == VK_BC7_SRGB) decode_bc7_block(mip_data + (uint64_t(by) * bw + bx) * bb, texels);
            else decode_block(vk_format, mip_data + (uint64_t(by) * bw + bx) * bb, texels);
            for (uint32_t y = 0; y < 4 && by * 4 + y < height; ++y)
                for (uint32_t x = 0; x < 4 && bx * 4 + x < width; ++x)
                    memcpy(out_rgba8 + (uint64_t(by * 4 + y) * width + bx * 4 + x) * 4, texels[y * 4 + x], 4);
        }
    return KJ_OK;
}

} // extern "C"
