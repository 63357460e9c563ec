// render_job.cpp - see render_job.h.
#include "render_job.h"

#include <cstring>

namespace rtjob {

void RgbaBuffer::blit(uint32_t left, uint32_t top, const RgbaBuffer& src) {
    for (uint32_t y = 0; y < src.height && top + y < height; y++) {
        const uint32_t w = left + src.width <= width ? src.width : (left < width ? width - left : 0);
        if (w) memcpy(data.data() + ((size_t)(top + y) * width + left) * 4, src.data.data() + (size_t)y * src.width * 4, (size_t)w * 4);
    }
}

ChunkRenderer hip_chunk_renderer(RtScene* scene, RtSettings settings, uint64_t seed) {
    return [scene, settings, seed](const RtRenderChunk& chunk, RgbaBuffer& out, RtRayCounters& rays) -> int {
        RtRayCounters c{};
        const int rc = rt_render_chunk(scene, &settings, &chunk, seed, nullptr, out.data.data(), &c);
        rays.primary += c.primary; rays.secondary += c.secondary; rays.shadow += c.shadow; rays.samples += c.samples;
        return rc;
    };
}

ChunkRenderer hip_chunk_renderer(std::vector<RtScene*> scenes, RtSettings settings, uint64_t seed) {
    auto next = std::make_shared<std::atomic<size_t>>(0);
    return [scenes, settings, seed, next](const RtRenderChunk& chunk, RgbaBuffer& out, RtRayCounters& rays) -> int {
        thread_local const void* bound_to = nullptr; thread_local size_t mine = 0;
        if (bound_to != next.get()) { mine = next->fetch_add(1) % scenes.size(); bound_to = next.get(); }      // first chunk of this thread in this job
        RtRayCounters c{};
        const int rc = rt_render_chunk(scenes[mine], &settings, &chunk, seed, nullptr, out.data.data(), &c);
        rays.primary += c.primary; rays.secondary += c.secondary; rays.shadow += c.shadow; rays.samples += c.samples;
        return rc;
    };
}

namespace {

// render.rs:117-167
void render_thread(uint32_t id, std::shared_ptr<std::atomic<bool>> cancel, std::shared_ptr<Channel<RenderWork>> work_receiver,
                   std::shared_ptr<Channel<RenderThreadMessage>> result_sender) {
    std::string error;
    auto body = [&]() -> bool {
        { RenderThreadMessage m; m.kind = RenderThreadMessage::Ready; if (!result_sender->send(std::move(m))) return false; }
        RenderWork work;
        while (work_receiver->recv(work)) {
            if (cancel->load()) return true;
            const RtRenderChunk chunk = work.chunk;
            RgbaBuffer buffer(chunk.width, chunk.height);
            // paint the chunk black to start (render.rs:135-140): (255*0) as u8 with alpha 255
            uint8_t black[4]; const float zero[3] = {0.0f, 0.0f, 0.0f}; rt_v3_to_rgba(zero, 1, black);
            for (uint32_t y = 0; y < chunk.height; y++) for (uint32_t x = 0; x < chunk.width; x++) buffer.put_pixel(x, y, black);
            { RenderThreadMessage m; m.kind = RenderThreadMessage::FrameUpdated; m.chunk = chunk; m.buffer = buffer; if (!result_sender->send(std::move(m))) return false; }
            const auto frame_time = std::chrono::steady_clock::now();
            RtRayCounters rays{};
            const int rc = (*work.render)(chunk, buffer, rays);
            if (rc == RT_E_CANCELLED && cancel->load()) return true;      // the token reached into the running call: a normal end (render.rs:149)
            if (rc != RT_OK) { error = std::string("chunk render failed: ") + rt_last_error(); return false; }
            const auto elapsed = std::chrono::steady_clock::now() - frame_time;
            { RenderThreadMessage m; m.kind = RenderThreadMessage::FrameUpdated; m.chunk = chunk; m.buffer = std::move(buffer); if (!result_sender->send(std::move(m))) return false; }
            { RenderThreadMessage m; m.kind = RenderThreadMessage::FrameCompleted; m.thread_id = id; m.rays = rays;
              m.elapsed = std::chrono::duration_cast<std::chrono::nanoseconds>(elapsed); if (!result_sender->send(std::move(m))) return false; }
        }
        return true;
    };
    body();
    RenderThreadMessage m; m.kind = RenderThreadMessage::Terminated; m.thread_id = id; m.error = error;
    result_sender->send(std::move(m));           // there may be nobody listening (render.rs:200-202)
}

}  // namespace

std::unique_ptr<RenderJobWorkerHandle> start_background_render_threads(uint32_t render_thread_count, std::vector<RtScene*> scenes) {
    auto h = std::make_unique<RenderJobWorkerHandle>();
    h->scenes = std::move(scenes);
    h->cancel = std::make_shared<std::atomic<bool>>(false);
    h->work_sender = std::make_shared<Channel<RenderWork>>((size_t)render_thread_count * 3);   // flume::bounded(n * 3)
    h->result_receiver = std::make_shared<Channel<RenderThreadMessage>>();                     // flume::unbounded()
    h->thread_handles.resize(render_thread_count);
    for (uint32_t id = 0; id < render_thread_count; id++) {
        h->thread_handles[id].id = id;
        h->thread_handles[id].handle = std::thread(render_thread, id, h->cancel, h->work_sender, h->result_receiver);
    }
    return h;
}

void RenderJobWorkerHandle::stop() {
    if (cancel) cancel->store(true);
    for (RtScene* s : scenes) (void)rt_scene_cancel(s, 1);       // ends the chunk calls that are running on the GPU
    if (work_sender) work_sender->close();
    for (auto& t : thread_handles) if (t.handle.joinable()) t.handle.join();
    for (RtScene* s : scenes) (void)rt_scene_cancel(s, 0);       // every call has returned: the scenes are usable again
    scenes.clear();                                              // (stop() runs again from the destructor, when the scenes may be gone)
}

// render.rs:39-81
RenderJobUpdateResult RenderJob::update() {
    RenderThreadMessage r;
    while (worker_handle->result_receiver->try_recv(r)) {
        switch (r.kind) {
            case RenderThreadMessage::Ready: break;
            case RenderThreadMessage::FrameUpdated: updates.emplace_back(r.chunk, std::move(r.buffer)); break;
            case RenderThreadMessage::FrameCompleted: {
                RenderThread& t = worker_handle->thread_handles[r.thread_id];
                t.total_time += r.elapsed; t.total_chunks_rendered += 1;
                completed_chunk_count += 1;
                rays.primary += r.rays.primary; rays.secondary += r.rays.secondary; rays.shadow += r.rays.shadow; rays.samples += r.rays.samples;
                break;
            }
            case RenderThreadMessage::Terminated:
                terminated_threads += 1;
                if (!r.error.empty() && first_error.empty()) first_error = r.error;
                break;
        }
    }
    if (terminated_threads >= worker_handle->thread_handles.size() && !is_work_completed())
        return RenderJobUpdateResult::ErrorRenderThreadsStopped;
    while (next_chunk_index < chunks.size()) {
        RenderWork w{chunks[next_chunk_index], render_args};
        const auto s = worker_handle->work_sender->try_send(std::move(w));
        if (s == Channel<RenderWork>::TrySend::Full) break;                     // queue full, try again later
        if (s == Channel<RenderWork>::TrySend::Disconnected) return RenderJobUpdateResult::ErrorRenderThreadsStopped;
        next_chunk_index += 1;
    }
    return RenderJobUpdateResult::Updated;
}

}  // namespace rtjob
