// rt_render - command-line host of the render core: what the reference's UI does
// between "Start render" and the finished texture, without the UI.
//   job construction   src/job_constructing.rs:80-110 (scene factory -> build_bvh ->
//                      RenderSettings -> create_render_chunks -> worker threads)
//   running            src/job_running.rs (update() each frame, blit updates)
// over render_job.{h,cpp} and the two C ABIs.  Writes a binary PPM and prints one
// JSON line with wall time, ray counters and an FNV-1a checksum of the RGBA frame.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/rt_samples.h"
#include "render_job.h"

using namespace rtjob;

namespace {

struct Options {
    std::string scene = "Random Spheres", params, mesh_dir = "assets/meshes", out;
    uint32_t width = 400, height = 225, spp = 16, depth = 8, chunks_x = 4, chunks_y = 4, threads = 2;
    float lens_radius = 0.1f, fov = 45.0f;
    uint64_t seed = RNG_SEED;
    std::vector<int> devices{0};
    bool self_test = false, list = false;
    std::string multi;                 // "tiles" / "spp": one rt_render_frame_multi_host call instead of the chunk job
    int cancel_after_ms = -1;          // >= 0: stop the job after this many milliseconds (exercises the cancellation path)
};

int usage() {
    fprintf(stderr,
            "usage: rt_render [--scene NAME] [--params \"Key=v;Key=v\"] [--mesh-dir DIR] [--width W] [--height H] [--spp N]\n"
            "                 [--depth D] [--chunks CXxCY] [--threads T] [--seed S] [--devices G[,G...]] [--lens-radius R] [--fov F]\n"
            "                 [--out frame.ppm] [--list] [--self-test] [--multi tiles|spp] [--cancel-after-ms N]\n"
            "  --multi      the frame in ONE call on all --devices (rt_render_frame_multi_host) instead of the chunk job\n"
            "  --cancel-after-ms  stop the job after N ms: running chunk calls are cancelled on the GPU (rt_scene_cancel)\n"
            "  --self-test  exercises the job machinery with a synthetic per-chunk pattern instead of the GPU\n");
    return 2;
}

bool parse(int argc, char** argv, Options& o) {
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : nullptr; };
        const char* v = nullptr;
        if (a == "--list") o.list = true;
        else if (a == "--self-test") o.self_test = true;
        else if (!(v = next())) return false;
        else if (a == "--scene") o.scene = v;
        else if (a == "--params") o.params = v;
        else if (a == "--mesh-dir") o.mesh_dir = v;
        else if (a == "--out") o.out = v;
        else if (a == "--width") o.width = (uint32_t)atoi(v);
        else if (a == "--height") o.height = (uint32_t)atoi(v);
        else if (a == "--spp") o.spp = (uint32_t)atoi(v);
        else if (a == "--depth") o.depth = (uint32_t)atoi(v);
        else if (a == "--threads") o.threads = (uint32_t)atoi(v);
        else if (a == "--seed") o.seed = strtoull(v, nullptr, 10);
        else if (a == "--device" || a == "--devices") {           // "0" or "0,1,2,3": one scene replica per entry
            o.devices.clear();
            for (const char* p = v; *p;) { o.devices.push_back(atoi(p)); while (*p && *p != ',') p++; if (*p == ',') p++; }
            if (o.devices.empty()) return false;
        }
        else if (a == "--multi") { o.multi = v; if (o.multi != "tiles" && o.multi != "spp") return false; }
        else if (a == "--cancel-after-ms") o.cancel_after_ms = atoi(v);
        else if (a == "--lens-radius") o.lens_radius = (float)atof(v);
        else if (a == "--fov") o.fov = (float)atof(v);
        else if (a == "--chunks") { if (sscanf(v, "%ux%u", &o.chunks_x, &o.chunks_y) != 2) return false; }
        else return false;
    }
    return o.width && o.height && o.spp && o.threads && o.chunks_x && o.chunks_y;
}

uint64_t fnv1a(const std::vector<uint8_t>& d) {
    uint64_t h = 1469598103934665603ull;
    for (uint8_t b : d) { h ^= b; h *= 1099511628211ull; }
    return h;
}

}  // namespace

int main(int argc, char** argv) {
    Options o;
    if (!parse(argc, argv, o)) return usage();
    if (o.list) { for (int i = 0; i < rt_samples_count(); i++) printf("%s\n", rt_samples_name(i)); return 0; }

    RtSettings settings{o.width, o.height, o.depth, o.spp};                          // job_constructing.rs:96-101
    std::vector<RtRenderChunk> chunks((size_t)o.chunks_x * o.chunks_y);
    const int n_chunks = rt_create_render_chunks(o.width, o.height, o.chunks_x, o.chunks_y, chunks.data(), (uint32_t)chunks.size());
    if (n_chunks < 0) { fprintf(stderr, "rt_render: %s\n", rt_last_error()); return 1; }
    chunks.resize((size_t)n_chunks);

    RtSceneBuilder* builder = nullptr;
    std::vector<RtScene*> scenes;
    ChunkRenderer renderer;
    if (o.self_test) {
        // pattern keyed by the absolute pixel position: any lost, doubled or misplaced chunk shows
        renderer = [](const RtRenderChunk& c, RgbaBuffer& out, RtRayCounters& rays) -> int {
            for (uint32_t y = 0; y < c.height; y++)
                for (uint32_t x = 0; x < c.width; x++) {
                    const uint32_t ax = c.left + x, ay = c.top + y;
                    const uint8_t px[4] = {(uint8_t)(ax * 7u), (uint8_t)(ay * 13u), (uint8_t)((ax ^ ay) * 3u), 255};
                    out.put_pixel(x, y, px);
                }
            rays.samples += (uint64_t)c.width * c.height;
            return RT_OK;
        };
    } else {
        RtCameraConfig cam{(float)o.width, (float)o.height, o.lens_radius, o.fov, 0.0f, 0.0f, 0.0f};
        builder = rt_samples_create(o.scene.c_str(), &cam, o.mesh_dir.c_str(), o.params.empty() ? nullptr : o.params.c_str());
        if (!builder) { fprintf(stderr, "rt_render: %s\n", rt_samples_last_error()); return 1; }   // CreateSceneError, job_constructing.rs:45-53
        for (int dev : o.devices) {
            RtScene* scene = rt_scene_create(rt_builder_desc(builder), dev);                       // replaces scene.build_bvh(), :92
            if (!scene) {
                fprintf(stderr, "rt_render: %s\n", rt_last_error());
                for (RtScene* sc : scenes) rt_scene_destroy(sc);
                rt_builder_destroy(builder); return 1;
            }
            scenes.push_back(scene);
        }
        renderer = hip_chunk_renderer(scenes, settings, o.seed);
    }

    if (!o.multi.empty() && !o.self_test) {
        // the whole frame in one call on every device: the worker-thread pool of src/render.rs:186-224 across GPUs
        RgbaBuffer frame(o.width, o.height);
        RtRayCounters rays{};
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = rt_render_frame_multi_host(scenes.data(), (uint32_t)scenes.size(), &settings, o.multi == "tiles" ? RT_MULTI_TILES : RT_MULTI_SPP, o.seed,
                                                  nullptr, frame.data.data(), &rays);
        const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        int status = 0;
        if (rc != RT_OK) { fprintf(stderr, "rt_render: %s\n", rt_last_error()); status = 1; }
        if (status == 0 && !o.out.empty()) {
            FILE* f = fopen(o.out.c_str(), "wb");
            if (!f) { fprintf(stderr, "rt_render: cannot write %s\n", o.out.c_str()); status = 1; }
            else {
                fprintf(f, "P6\n%u %u\n255\n", o.width, o.height);
                for (size_t i = 0; i < (size_t)o.width * o.height; i++) fwrite(frame.data.data() + i * 4, 1, 3, f);
                fclose(f);
            }
        }
        if (status == 0) {
            const uint64_t total = rays.primary + rays.secondary + rays.shadow;
            printf("{\"scene\": \"%s\", \"width\": %u, \"height\": %u, \"spp\": %u, \"max_reflections\": %u, \"multi\": \"%s\", \"replicas\": %zu, "
                   "\"wall_ms\": %.3f, \"rays\": {\"primary\": %llu, \"secondary\": %llu, \"shadow\": %llu, \"samples\": %llu}, "
                   "\"mrays_per_s\": %.3f, \"fnv1a\": \"%016llx\"}\n",
                   o.scene.c_str(), o.width, o.height, o.spp, o.depth, o.multi.c_str(), scenes.size(), wall_ms, (unsigned long long)rays.primary,
                   (unsigned long long)rays.secondary, (unsigned long long)rays.shadow, (unsigned long long)rays.samples,
                   wall_ms > 0 ? (double)total / wall_ms / 1e3 : 0.0, (unsigned long long)fnv1a(frame.data));
        }
        for (RtScene* sc : scenes) rt_scene_destroy(sc);
        if (builder) rt_builder_destroy(builder);
        return status;
    }

    RenderJob job;
    job.render_args = std::make_shared<const ChunkRenderer>(std::move(renderer));
    job.chunks = chunks;
    job.worker_handle = start_background_render_threads(o.threads, scenes);
    job.started = std::chrono::steady_clock::now();

    RgbaBuffer frame(o.width, o.height);
    int status = 0;
    bool cancelled = false; double stop_ms = 0.0;
    while (!job.is_work_completed()) {                                               // the UI's per-frame poll, job_running.rs
        if (o.cancel_after_ms >= 0 &&
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - job.started).count() >= (double)o.cancel_after_ms) {
            // the user pressed "cancel" (src/app.rs drops the job): token, running GPU calls, join
            uint64_t done = 0, total = 0;
            if (!scenes.empty()) (void)rt_scene_progress(scenes[0], &done, &total);
            const auto t0 = std::chrono::steady_clock::now();
            job.worker_handle->stop();
            stop_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            job.update();
            printf("{\"cancelled\": true, \"after_ms\": %d, \"stop_ms\": %.3f, \"chunks_completed\": %zu, \"chunks\": %zu, \"blocks_done\": %llu, \"blocks_total\": %llu}\n",
                   o.cancel_after_ms, stop_ms, job.completed_chunk_count, job.chunks.size(), (unsigned long long)done, (unsigned long long)total);
            cancelled = true; break;
        }
        if (job.update() == RenderJobUpdateResult::ErrorRenderThreadsStopped) {
            fprintf(stderr, "rt_render: render threads stopped: %s\n", job.first_error.c_str());
            status = 1; break;
        }
        for (auto& u : job.updates) frame.blit(u.first.left, u.first.top, u.second);
        job.updates.clear();
        if (!job.is_work_completed()) std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    if (cancelled) {
        for (RtScene* sc : scenes) rt_scene_destroy(sc);
        if (builder) rt_builder_destroy(builder);
        return 0;
    }
    if (status == 0) {
        job.update();                                                                // drain the last FrameUpdated messages
        for (auto& u : job.updates) frame.blit(u.first.left, u.first.top, u.second);
        job.updates.clear();
    }
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - job.started).count();
    job.worker_handle->stop();

    if (status == 0 && !o.out.empty()) {
        FILE* f = fopen(o.out.c_str(), "wb");
        if (!f) { fprintf(stderr, "rt_render: cannot write %s\n", o.out.c_str()); status = 1; }
        else {
            fprintf(f, "P6\n%u %u\n255\n", o.width, o.height);
            for (size_t i = 0; i < (size_t)o.width * o.height; i++) fwrite(frame.data.data() + i * 4, 1, 3, f);
            fclose(f);
        }
    }
    if (status == 0) {
        const uint64_t rays = job.rays.primary + job.rays.secondary + job.rays.shadow;
        printf("{\"scene\": \"%s\", \"width\": %u, \"height\": %u, \"spp\": %u, \"max_reflections\": %u, \"chunks\": %zu, \"threads\": %u, "
               "\"wall_ms\": %.3f, \"rays\": {\"primary\": %llu, \"secondary\": %llu, \"shadow\": %llu, \"samples\": %llu}, "
               "\"mrays_per_s\": %.3f, \"fnv1a\": \"%016llx\", \"chunks_per_thread\": [",
               o.self_test ? "self-test" : o.scene.c_str(), o.width, o.height, o.spp, o.depth, job.chunks.size(), o.threads, wall_ms,
               (unsigned long long)job.rays.primary, (unsigned long long)job.rays.secondary, (unsigned long long)job.rays.shadow,
               (unsigned long long)job.rays.samples, wall_ms > 0 ? (double)rays / wall_ms / 1e3 : 0.0, (unsigned long long)fnv1a(frame.data));
        for (size_t i = 0; i < job.worker_handle->thread_handles.size(); i++)
            printf("%s%u", i ? ", " : "", job.worker_handle->thread_handles[i].total_chunks_rendered);
        printf("]}\n");
    }
    for (RtScene* sc : scenes) rt_scene_destroy(sc);
    if (builder) rt_builder_destroy(builder);
    return status;
}
