// az_host.h -- host-side helpers of the engine that own no game logic: the numpy-compatible RNG entry points
// (az_rng.cpp), the process-wide HIP stream pool and the producer that streams RNG tapes to the device ahead of the games.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace azrng {
void selfplay_tapes_parallel(uint64_t seed0, int g0, int count, int nn, double alpha, int max_plies, double *noise,
                             int64_t noise_stride, double *u, int threads);
void uniforms(uint64_t seed, int count, double *u);
struct Streams;
Streams *streams_new(uint64_t seed0, int count);
void streams_free(Streams *s);
void streams_plies(Streams *s, int i, int nn, double alpha, int m0, int m1, double *noise_row, double *u_row);
}

// Process-wide stream pool.  The runtime maps the streams of one priority level onto a small pool of hardware queues
// when they are created; engines that come and go (a new MultiEngine per configuration, the evaluator's engine, a tape
// producer per episode) would otherwise keep creating streams, and a later set of four engines can end up two to a
// hardware queue (measured: a 15x15 episode 11.6 -> 14.7 s after two create/destroy cycles).  Streams are therefore
// never destroyed: a released stream is synchronised and handed to the next engine on that device.
// Play streams are non-blocking and HIGH priority -- not for urgency: high-priority streams draw their hardware queues
// from a pool of their own, so next to a framework that has already created streams (PyTorch's context) the four
// engines of a GPU still get a queue each (measured: episode 15.3 -> 11.6 s through SelfPlayManager).
struct StreamPool {
    std::mutex mu;
    std::vector<hipStream_t> idle[16][2];      // [device][0 = normal priority copy streams, 1 = high priority play streams]
    bool primed[16] = {};
    static hipError_t create(bool high, hipStream_t *out)
    {
        int least = 0, greatest = 0;
        const char *sp = getenv("AZ_STREAM_PRIORITY");     // AZ_STREAM_PRIORITY=0: default priority for the play streams too
        if (high && !(sp && sp[0] == '0') && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest < least)
            return hipStreamCreateWithPriority(out, hipStreamNonBlocking, greatest);
        return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
    }
    hipError_t acquire(int device, bool high, hipStream_t *out)
    {
        std::lock_guard<std::mutex> lk(mu);
        std::vector<hipStream_t> &v = idle[device & 15][high ? 1 : 0];
        if (high && !primed[device & 15]) {
            // the first engine on a device creates the play streams of a whole MultiEngine back to back: the queues of
            // a set created in one go overlap well, a set pieced together around other queue creations may not
            // (measured: an engine that played before the four were created cost them 12 %)
            primed[device & 15] = true;
            for (int i = 0; i < 4; i++) {
                hipStream_t s = nullptr;
                hipError_t rc = create(true, &s);
                if (rc != hipSuccess) return rc;
                v.insert(v.begin(), s);
            }
        }
        if (!v.empty()) { *out = v.back(); v.pop_back(); return hipSuccess; }
        return create(high, out);
    }
    void release(int device, bool high, hipStream_t s)
    {
        if (!s) return;
        (void)hipStreamSynchronize(s);
        std::lock_guard<std::mutex> lk(mu);
        idle[device & 15][high ? 1 : 0].push_back(s);
    }
};
static StreamPool g_streams;

// Streams the self-play RNG tapes to the device a few plies ahead of the games instead of generating all n^2 plies of
// every game before the first move (26 M legacy-gamma draws for 1024 games at 15x15, ~4 s of host time, most of it for
// plies the games never reach).  The draws are trajectory-independent, so "wave" w = plies [wP, wP+P) of
// every unfinished game is produced by a few host threads into pinned staging, copied with two strided 2-D copies on a
// copy stream, and the play stream waits on the wave's event before the first ply that needs it.  A game's ply never
// exceeds the number of plies the episode has played, so waiting for wave floor(step / P) covers every slot, refilled
// ones included.  Same numbers in the same device layout as the bulk path (AZ_TAPE_STREAM=0).
struct TapeProducer {
    static constexpr int P = 2;          // plies per wave
    int device = 0, G = 0, nn = 0, plies = 0, waves = 0, threads = 1;
    double alpha = 0.3;
    int64_t tape_len = 0;
    double *noise_dev = nullptr, *u_dev = nullptr;
    std::vector<int64_t> off;            // off[m] = doubles before ply m in one game's noise tape
    azrng::Streams *streams = nullptr;
    double *hn[2] = {nullptr, nullptr}, *hu[2] = {nullptr, nullptr};    // pinned staging, double-buffered
    int *h_done = nullptr;               // pinned: g_nply as read back after every ply (non-zero = game over)
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> wave_event;  // recorded on copy_stream behind a wave's copies
    std::mutex mu;
    std::condition_variable cv;
    int waves_done = 0, want = 0;
    bool stop = false;
    hipError_t error = hipSuccess;
    std::thread th;

    hipError_t start(int dev, uint64_t seed0, int games, int cells, int max_plies, double a, int64_t tlen, double *nd,
                     double *ud, int nthreads)
    {
        device = dev; G = games; nn = cells; plies = max_plies; alpha = a; tape_len = tlen; noise_dev = nd; u_dev = ud;
        threads = nthreads < 1 ? 1 : nthreads;
        waves = (plies + P - 1) / P;
        off.assign((size_t)plies + 1, 0);
        for (int m = 0; m < plies; m++) off[m + 1] = off[m] + (nn - m);
        const size_t lmax = (size_t)P * nn;
        hipError_t rc;
        for (int b = 0; b < 2; b++) {
            if ((rc = hipHostMalloc((void **)&hn[b], (size_t)G * lmax * sizeof(double), hipHostMallocDefault))) return rc;
            if ((rc = hipHostMalloc((void **)&hu[b], (size_t)G * P * sizeof(double), hipHostMallocDefault))) return rc;
        }
        if ((rc = hipHostMalloc((void **)&h_done, (size_t)G * sizeof(int), hipHostMallocDefault))) return rc;
        memset(h_done, 0, (size_t)G * sizeof(int));
        if ((rc = g_streams.acquire(device, false, &copy_stream))) return rc;
        wave_event.assign((size_t)waves, nullptr);
        for (int w = 0; w < waves; w++)
            if ((rc = hipEventCreateWithFlags(&wave_event[w], hipEventDisableTiming))) return rc;
        streams = azrng::streams_new(seed0, G);
        want = 2;                        // waves 0 and 1 are produced straight away
        th = std::thread([this]() { run(); });
        return hipSuccess;
    }

    void run()
    {
        hipError_t rc = hipSetDevice(device);
        for (int w = 0; w < waves && rc == hipSuccess; w++) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || w < want; });
                if (stop) return;
            }
            const int b = w & 1, m0 = w * P, m1 = std::min(plies, m0 + P);
            if (w >= 2 && (rc = hipEventSynchronize(wave_event[w - 2])) != hipSuccess) break;   // staging buffer b is free again
            const int64_t L = off[m1] - off[m0];
            std::atomic<int> next{0};
            auto work = [&]() {
                const volatile int *done = h_done;
                for (;;) {
                    const int g0 = next.fetch_add(8);
                    if (g0 >= G) break;
                    for (int g = g0; g < std::min(G, g0 + 8); g++)
                        if (!done[g]) azrng::streams_plies(streams, g, nn, alpha, m0, m1, hn[b] + (size_t)g * L, hu[b] + (size_t)g * P);
                }
            };
            std::vector<std::thread> pool;
            for (int t = 1; t < threads; t++) pool.emplace_back(work);
            work();
            for (auto &t : pool) t.join();
            rc = hipMemcpy2DAsync(noise_dev + off[m0], (size_t)tape_len * 8, hn[b], (size_t)L * 8, (size_t)L * 8, G,
                                  hipMemcpyHostToDevice, copy_stream);
            if (rc == hipSuccess)
                rc = hipMemcpy2DAsync(u_dev + m0, (size_t)nn * 8, hu[b], (size_t)P * 8, (size_t)(m1 - m0) * 8, G,
                                      hipMemcpyHostToDevice, copy_stream);
            if (rc == hipSuccess) rc = hipEventRecord(wave_event[w], copy_stream);
            if (rc != hipSuccess) break;
            {
                std::lock_guard<std::mutex> lk(mu);
                waves_done = w + 1;
            }
            cv.notify_all();
        }
        if (rc != hipSuccess) {
            std::lock_guard<std::mutex> lk(mu);
            error = rc;
            cv.notify_all();
        }
    }

    // consumer side: the episode is about to play its ply number `step`; returns the event to wait on (or null)
    hipError_t need(int step, hipEvent_t *ev)
    {
        *ev = nullptr;
        const int w = step / P;
        if (w >= waves) return hipSuccess;
        std::unique_lock<std::mutex> lk(mu);
        if (want < w + 2) { want = w + 2; cv.notify_all(); }
        cv.wait(lk, [&] { return waves_done > w || error != hipSuccess; });
        if (error != hipSuccess) return error;
        *ev = wave_event[w];
        return hipSuccess;
    }

    void shutdown()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
        if (copy_stream) { g_streams.release(device, false, copy_stream); copy_stream = nullptr; }
        for (hipEvent_t ev : wave_event) if (ev) (void)hipEventDestroy(ev);
        wave_event.clear();
        for (int b = 0; b < 2; b++) {
            if (hn[b]) (void)hipHostFree(hn[b]);
            if (hu[b]) (void)hipHostFree(hu[b]);
            hn[b] = hu[b] = nullptr;
        }
        if (h_done) (void)hipHostFree(h_done);
        h_done = nullptr;
        if (streams) azrng::streams_free(streams);
        streams = nullptr;
    }
};
