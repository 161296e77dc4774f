// A compiled-code host for the hot path (what kajiya's Rust `WorldRenderer` does, through include/kajiya_amd.hpp): maps baked `.mesh` /
// `.image` files (the reference's `bin/bake` output format), builds the scene, runs N frames of prepare_render_graph_standard with an
// orbiting camera and dumps the last GI, reflection and anti-aliased images. No Python, no torch.
//   world_render_passes <blue_noise_256_rgba8.bin> <scene_dir> <W> <H> <frames> <out_prefix>
// <scene_dir>/scene.txt:  `mesh <file>` | `instance <mesh index> <12 floats, row-major 3x4>` | `camera <cx cy cz radius height rate>`
// <scene_dir>/rtr_tables.bin: ranking (128*128*8 u32) | scrambling (128*128*8 u32) | sobol (256*256 u32) | spatial resolve offsets (512 x int4)
#include <chrono>
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include "../include/kajiya_amd.hpp"

using namespace kajiya_amd;

static std::vector<uint8_t> read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error("cannot open " + path);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void write_device(const std::string& path, const void* dev, size_t bytes) {
    std::vector<uint8_t> h(bytes);
    check_hip(hipMemcpy(h.data(), dev, bytes, hipMemcpyDeviceToHost), "hipMemcpy");
    std::ofstream(path, std::ios::binary).write((const char*)h.data(), std::streamsize(bytes));
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: %s <blue_noise.bin> <scene_dir> <W> <H> <frames> <out_prefix> [post]\n", argv[0]); return 2; }
    try {
        const std::string dir = argv[2], out = argv[6];
        const uint32_t W = uint32_t(atoi(argv[3])), H = uint32_t(atoi(argv[4]));
        const int frames = atoi(argv[5]);
        const std::vector<uint8_t> blue_noise = read_file(argv[1]);
        if (blue_noise.size() != 256 * 256 * 4) throw Error("blue noise must be 256x256 RGBA8");
        Device device(0, blue_noise.data());
        Scene scene(device);
        hipStream_t stream;
        check_hip(hipStreamCreate(&stream), "hipStreamCreate");

        std::map<uint64_t, std::vector<uint8_t>> images;          // cache/<identity>.image, loaded once per identity (world_renderer.rs:610-631)
        auto image_bytes = [&](uint64_t id) -> const std::vector<uint8_t>& {
            auto it = images.find(id);
            if (it == images.end()) {
                char name[64]; snprintf(name, sizeof(name), "%8llx.image", (unsigned long long)id);
                it = images.emplace(id, read_file(dir + "/" + name)).first;
            }
            return it->second;
        };
        std::vector<std::vector<uint8_t>> mesh_files;             // the mapped files must outlive add_mesh only (the scene copies what it needs)
        std::vector<MeshHandle> meshes;
        double cam[6] = {0.0, 1.0, 0.0, 9.0, 3.0, 0.01};
        std::ifstream sf(dir + "/scene.txt");
        if (!sf) throw Error("cannot open " + dir + "/scene.txt");
        std::string line;
        while (std::getline(sf, line)) {
            std::istringstream ls(line);
            std::string kind; ls >> kind;
            if (kind == "mesh") {
                std::string file; ls >> file;
                mesh_files.push_back(read_file(dir + "/" + file));
                meshes.push_back(scene.add_baked_mesh(mesh_files.back().data(), mesh_files.back().size(), image_bytes));
            } else if (kind == "instance") {
                uint32_t mi; float xf[12]; ls >> mi;
                for (float& v : xf) ls >> v;
                if (mi >= meshes.size()) throw Error("scene.txt: instance of an unknown mesh");
                scene.add_instance(meshes[mi], xf);
            } else if (kind == "camera") {
                for (double& v : cam) ls >> v;
            }
        }
        scene.build_ray_tracing_top_level_acceleration(stream);

        const std::vector<uint8_t> tb = read_file(dir + "/rtr_tables.bin");
        const size_t n_tile = 128 * 128 * 8, n_sobol = 256 * 256, n_off = 16 * 4 * 8 * 4;
        if (tb.size() != (2 * n_tile + n_sobol + n_off) * 4) throw Error("rtr_tables.bin has the wrong size");
        const uint32_t* t32 = (const uint32_t*)tb.data();
        KjRtrTables tables{t32, t32 + n_tile, t32 + 2 * n_tile, (const int32_t*)(t32 + 2 * n_tile + n_sobol)};
        WorldRenderer world(device, scene, W, H, tables);
        const bool with_post = argc > 7 && std::string(argv[7]) == "post";     // the frame's tail: motion blur + PostProcessRenderer + the exposure loop
        const std::vector<uint16_t> no_hue_shift(128, 0);                        // Bezold-Brucke LUT: caller data; zeros = no shift (kajiya_amd/post_tables.py)
        if (with_post) world.enable_post(no_hue_shift.data());

        hipEvent_t e0, e1;
        check_hip(hipEventCreate(&e0), "hipEventCreate"); check_hip(hipEventCreate(&e1), "hipEventCreate");
        FrameOutput last{};
        float gpu_ms = 0.0f; int timed = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < frames; ++i) {
            const double ang = cam[5] * double(i);
            const double eye[3] = {cam[0] + cam[3] * std::sin(ang), cam[1] + cam[4], cam[2] + cam[3] * std::cos(ang)};
            const double target[3] = {cam[0], cam[1], cam[2]};
            const CameraMatrices camera = CameraMatrices::look_at(eye, target, 52.0, double(W) / double(H));
            const bool time_it = i >= frames / 2;
            if (time_it) check_hip(hipEventRecord(e0, stream), "hipEventRecord");
            last = world.prepare_render_graph_standard(camera, stream);
            if (time_it) {
                check_hip(hipEventRecord(e1, stream), "hipEventRecord");
                check_hip(hipEventSynchronize(e1), "hipEventSynchronize");
                float ms; check_hip(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
                gpu_ms += ms; ++timed;
            }
        }
        check_hip(hipStreamSynchronize(stream), "hipStreamSynchronize");
        const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        write_device(out + "_gi.bin", last.rtdgi.screen_irradiance_tex, size_t(W) * H * 8);
        write_device(out + "_rtr.bin", last.rtr, size_t(W) * H * 4);
        write_device(out + "_taa.bin", last.anti_aliased.this_frame_out, size_t(W) * H * 8);
        write_device(out + "_depth.bin", world.gbuffer_depth.depth.p, size_t(W) * H * 4);
        if (with_post) {
            write_device(out + "_final_post_input.bin", last.final_post_input, size_t(W) * H * 8);
            write_device(out + "_post.bin", last.post_processed, size_t(W) * H * 4);
        }
        printf("{\"host\": \"c++ (include/kajiya_amd.hpp)\", \"extent\": [%u, %u], \"frames\": %d, \"gpu_ms_per_frame\": %.4f, \"wall_ms_per_frame_incl_sync\": %.4f, \"triangle_lights\": %u}\n",
               W, H, frames, timed ? gpu_ms / float(timed) : 0.0f, wall_ms / double(frames), scene.triangle_light_count());
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
