// render_job.h - the host side of the drop-in boundary, in C++ (the reference's
// host is Rust; no Rust toolchain exists where this is built).
//
// Mirrors src/render.rs of the reference on top of the C ABI (include/rt_hip.h):
//   RenderJob / RenderJob::update              render.rs:18-97
//   RenderWork / RenderThreadMessage           render.rs:99-113
//   start_render_thread (the worker body)      render.rs:117-167
//   start_background_render_threads            render.rs:186-224
//   RgbaBuffer                                 src/rgba.rs:14-54
// The one deliberate difference: the worker's per-pixel loop (render.rs:148-159)
// is ONE call of rt_render_chunk per chunk.  The reference checks its cancellation
// token per pixel (render.rs:149): stop() therefore also cancels the scenes the job
// renders on (rt_scene_cancel), which ends a chunk call that is running within
// milliseconds; progress inside a chunk is rt_scene_progress.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_hip.h"

namespace rtjob {

constexpr uint64_t RNG_SEED = 12345;            // render.rs:16

// rgba.rs:14-54: one RGBA8 rectangle, row-major
struct RgbaBuffer {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> data;
    RgbaBuffer() = default;
    RgbaBuffer(uint32_t w, uint32_t h) : width(w), height(h), data((size_t)w * h * 4, 0) {}
    void put_pixel(uint32_t x, uint32_t y, const uint8_t rgba[4]) {
        uint8_t* p = data.data() + ((size_t)y * width + x) * 4;
        p[0] = rgba[0]; p[1] = rgba[1]; p[2] = rgba[2]; p[3] = rgba[3];
    }
    // copy_from_sub_buffer of the UI texture path (job_running.rs blits chunk buffers at [left, top])
    void blit(uint32_t left, uint32_t top, const RgbaBuffer& src);
};

// What a worker renders a chunk with.  The product binding is hip_chunk_renderer();
// tests of the job machinery may pass their own.
//   returns 0 or an RT_E_* code; fills `out` (chunk.width x chunk.height) and adds to `rays`.
using ChunkRenderer = std::function<int(const RtRenderChunk& chunk, RgbaBuffer& out, RtRayCounters& rays)>;
ChunkRenderer hip_chunk_renderer(RtScene* scene, RtSettings settings, uint64_t seed = RNG_SEED);
// Several GPUs in one process: `scenes` holds one replica of the scene per device; every worker thread
// is bound to one of them on its first chunk (round robin), so N x k threads keep N GPUs busy with the
// reference's own threading model.  The frame does not depend on which GPU rendered which chunk.
ChunkRenderer hip_chunk_renderer(std::vector<RtScene*> scenes, RtSettings settings, uint64_t seed = RNG_SEED);

struct RenderWork { RtRenderChunk chunk; std::shared_ptr<const ChunkRenderer> render; };

struct RenderThreadMessage {
    enum Kind { Ready, FrameUpdated, FrameCompleted, Terminated } kind = Ready;
    RtRenderChunk chunk{};                       // FrameUpdated
    RgbaBuffer buffer;                           // FrameUpdated
    uint32_t thread_id = 0;                      // FrameCompleted
    std::chrono::nanoseconds elapsed{0};         // FrameCompleted
    RtRayCounters rays{};                        // FrameCompleted (not in the reference: ray accounting)
    std::string error;                           // Terminated after a failure (render.rs:197-199 logs it)
};

// flume::bounded / flume::unbounded stand-in: MPMC queue, optional capacity, disconnect on close
template <class T>
class Channel {
public:
    explicit Channel(size_t capacity = 0) : cap_(capacity) {}
    enum class TrySend { Ok, Full, Disconnected };
    TrySend try_send(T&& v) {
        std::lock_guard<std::mutex> g(m_);
        if (closed_) return TrySend::Disconnected;
        if (cap_ && q_.size() >= cap_) return TrySend::Full;
        q_.push_back(std::move(v)); cv_.notify_one();
        return TrySend::Ok;
    }
    bool send(T&& v) {                           // unbounded use only
        std::lock_guard<std::mutex> g(m_);
        if (closed_) return false;
        q_.push_back(std::move(v)); cv_.notify_one();
        return true;
    }
    bool try_recv(T& out) {
        std::lock_guard<std::mutex> g(m_);
        if (q_.empty()) return false;
        out = std::move(q_.front()); q_.pop_front();
        return true;
    }
    bool recv(T& out) {                          // blocks; false once closed and drained
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return closed_ || !q_.empty(); });
        if (q_.empty()) return false;
        out = std::move(q_.front()); q_.pop_front();
        return true;
    }
    void close() { std::lock_guard<std::mutex> g(m_); closed_ = true; cv_.notify_all(); }
private:
    std::mutex m_; std::condition_variable cv_; std::deque<T> q_; size_t cap_; bool closed_ = false;
};

struct RenderThread {                            // render.rs:172-177
    uint32_t id = 0;
    std::thread handle;
    std::chrono::nanoseconds total_time{0};
    uint32_t total_chunks_rendered = 0;
};

struct RenderJobWorkerHandle {                   // render.rs:179-184
    std::shared_ptr<std::atomic<bool>> cancel;   // CancellationTokenSource
    std::shared_ptr<Channel<RenderWork>> work_sender;
    std::shared_ptr<Channel<RenderThreadMessage>> result_receiver;
    std::vector<RenderThread> thread_handles;
    std::vector<RtScene*> scenes;                // the scenes the chunk renderer uses: stop() cancels the calls running on them
    void stop();                                 // cancel (token + running GPU calls) + disconnect + join, then re-arm the scenes
    ~RenderJobWorkerHandle() { stop(); }
};

std::unique_ptr<RenderJobWorkerHandle> start_background_render_threads(uint32_t render_thread_count, std::vector<RtScene*> scenes = {});

enum class RenderJobUpdateResult { Updated, ErrorRenderThreadsStopped };

struct RenderJob {                               // render.rs:18-97
    std::shared_ptr<const ChunkRenderer> render_args;
    std::vector<RtRenderChunk> chunks;
    size_t next_chunk_index = 0;
    std::chrono::steady_clock::time_point started;
    size_t completed_chunk_count = 0;
    std::vector<std::pair<RtRenderChunk, RgbaBuffer>> updates;   // ([left, top], buffer) to blit
    std::unique_ptr<RenderJobWorkerHandle> worker_handle;
    RtRayCounters rays{};
    size_t terminated_threads = 0;
    std::string first_error;

    bool is_work_completed() const { return completed_chunk_count >= chunks.size(); }
    RenderJobUpdateResult update();
};

}  // namespace rtjob
