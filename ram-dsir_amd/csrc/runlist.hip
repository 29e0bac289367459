// runlist.hip -- rd_run_list / rd_join_lanes (include/ramdsir.h): the step's launch list walked in C++, one call per segment,
// with the lane forks / joins as HIP events.  Host code only.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include "common.h"
#include "../../include/ramdsir.h"

thread_local hipEvent_t rd_tls_stop_event = nullptr;        // common.h rd_launch
thread_local int rd_tls_stop_used = 0;

namespace {

// Events for the cross-stream edges of the SINGLE-THREADED walk (record and wait are issued back to back by the caller).  An event can
// be recorded again as soon as the hipStreamWaitEvent that uses it has been enqueued (the wait captures the record it was called
// after), so a small ring per device is enough there.  Edges that pass through a lane worker thread use the worker's own pool
// (worker_event below), whose slots are reused only after the worker has executed the command that carries them.
constexpr int EV_RING = 64, EV_MAX_DEV = 16;
struct EvRing {
    hipEvent_t ev[EV_RING];
    int next = 0;
    bool ready = false;
};
EvRing g_rings[EV_MAX_DEV];

int next_event(hipEvent_t* out) {
    int dev = 0;
    RD_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= EV_MAX_DEV) return -1;
    EvRing& r = g_rings[dev];
    if (!r.ready) {
        for (int i = 0; i < EV_RING; ++i) RD_CHECK(hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming));
        r.ready = true;
    }
    *out = r.ev[r.next];
    r.next = (r.next + 1) % EV_RING;
    return 0;
}

// stream `to` waits for everything enqueued on `from` so far
int edge(hipStream_t from, hipStream_t to) {
    if (from == to) return 0;
    hipEvent_t e;
    const int err = next_event(&e);
    if (err) return err;
    RD_CHECK(hipEventRecord(e, from));
    RD_CHECK(hipStreamWaitEvent(to, e, 0));
    return 0;
}

inline float as_f(uint64_t v) {
    const uint32_t u = (uint32_t)v;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

#ifdef RD_DEBUG_SWITCHES
static int rd_switch_rl(const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; }
#endif

#define P(i) ((void*)(uintptr_t)o.a[i])
#define CP(T, i) ((const T*)(uintptr_t)o.a[i])
#define I(i) ((int)(int64_t)o.a[i])
#define L64(i) ((int64_t)o.a[i])
#define F(i) (as_f(o.a[i]))
#define NARGS(k) do { if (o.nargs != (k)) return -1; } while (0)

#ifdef RD_DEBUG_SWITCHES
// Debug library, RD_POISON_LDS=1: in front of EVERY launch of a list a kernel fills the LDS of every CU (and the registers of the waves
// it runs) with a NaN pattern -- 0x7fc07fc0 is a NaN as fp32 and as two bf16, and 1e307-sized as half an fp64.  LDS and registers are
// not cleared between kernels: a launch that reads either before writing it normally sees what the previous kernel of the same process
// left there (repeatable when the process is alone on the GPU, arbitrary when it is not); with the poison it sees NaNs, and the losses show it.
__global__ __launch_bounds__(1024) void rd_poison_lds_kernel(int words) {
    extern __shared__ unsigned poison_smem[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) poison_smem[i] = 0x7fc07fc0u;
    __syncthreads();
    if (poison_smem[(threadIdx.x * 7) % words] == 1u) poison_smem[0] = 2u;         // keep the stores alive
}
void poison_lds(void* st) {
    static int on = -1, cus = 0;
    if (on < 0) {
        on = rd_switch_rl("RD_POISON_LDS");
        hipDeviceProp_t pr;
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipGetDeviceProperties(&pr, dev);
        cus = pr.multiProcessorCount;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rd_poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (on != 1) return;
    // one 160 KB workgroup per CU (x2 so that every CU gets one even if some run two 80 KB... the second wave of workgroups overwrites again)
    rd_launch(rd_poison_lds_kernel, dim3(2 * cus), dim3(1024), 160 * 1024, (hipStream_t)st, 160 * 1024 / 4);
}
#endif

int call(const rd_launch_t& o, void* st) {
#ifdef RD_DEBUG_SWITCHES
    poison_lds(st);
#endif
    switch (o.op) {
    case RD_OP_CONV: NARGS(2); return rd_conv(CP(rd_conv_t, 0), I(1), st);
    case RD_OP_WGRAD: NARGS(2); return rd_wgrad(CP(rd_wgrad_t, 0), I(1), st);
    case RD_OP_CONV_BWD_FUSED: NARGS(3); return rd_conv_bwd_fused(CP(rd_conv_t, 0), CP(rd_wgrad_t, 1), I(2), st);
    case RD_OP_CONV_BWD_FUSED_REDUCE: NARGS(3); return rd_conv_bwd_fused_reduce(CP(rd_conv_t, 0), CP(rd_wgrad_t, 1), I(2), st);
    case RD_OP_PACK_WEIGHTS_BATCHED: NARGS(6); return rd_pack_weights_batched(CP(float, 0), P(1), CP(rd_pack_entry_t, 2), I(3), L64(4), I(5), st);
    case RD_OP_BN_FINALIZE_FWD: NARGS(1); return rd_bn_finalize_fwd(CP(rd_bn_fwd_t, 0), st);
    case RD_OP_BN_FINALIZE_BWD: NARGS(1); return rd_bn_finalize_bwd(CP(rd_bn_bwd_t, 0), st);
    case RD_OP_GN_FINALIZE_FWD: NARGS(1); return rd_gn_finalize_fwd(CP(rd_bn_fwd_t, 0), st);
    case RD_OP_GN_FINALIZE_BWD: NARGS(1); return rd_gn_finalize_bwd(CP(rd_bn_bwd_t, 0), st);
    case RD_OP_UP_STATS: NARGS(11); return rd_up_stats(P(0), (double*)P(1), P(2), I(3), I(4), I(5), I(6), I(7), CP(int32_t, 8), I(9), I(10), st);
    case RD_OP_BN_STATS: NARGS(9); return rd_bn_stats(P(0), (double*)P(1), I(2), I(3), I(4), I(5), I(6), CP(int32_t, 7), I(8), st);
    case RD_OP_UP_BWD:
        NARGS(15);
        return rd_up_bwd(P(0), P(1), P(2), CP(float, 3), CP(float, 4), CP(float, 5), I(6), I(7), I(8), I(9), I(10), CP(int32_t, 11), I(12), CP(rd_bn_bwd_t, 13), I(14), st);
    case RD_OP_POOL_FWD:
        NARGS(14);
        return rd_pool_fwd(P(0), CP(float, 1), CP(float, 2), F(3), P(4), I(5), I(6), I(7), I(8), I(9), CP(int32_t, 10), I(11), CP(rd_bn_fwd_t, 12), I(13), st);
    case RD_OP_POOL_BWD:
        NARGS(17);
        return rd_pool_bwd(P(0), P(1), CP(float, 2), CP(float, 3), F(4), I(5), P(6), I(7), (double*)P(8), I(9), I(10), I(11), I(12), I(13), CP(int32_t, 14), I(15), I(16), st);
    case RD_OP_BN_APPLY:
        NARGS(14);
        return rd_bn_apply(P(0), P(1), P(2), CP(float, 3), CP(float, 4), CP(float, 5), F(6), I(7), I(8), I(9), I(10), I(11), CP(int32_t, 12), I(13), st);
    case RD_OP_NCHW_TO_NHWC: NARGS(8); return rd_nchw_to_nhwc(CP(float, 0), P(1), I(2), I(3), I(4), I(5), I(6), I(7), st);
    case RD_OP_NHWC_TO_NCHW:
        NARGS(13);
        return rd_nhwc_to_nchw(P(0), (float*)P(1), CP(float, 2), CP(float, 3), I(4), F(5), I(6), I(7), I(8), I(9), I(10), CP(int32_t, 11), I(12), st);
    case RD_OP_GRAD_IN:
        NARGS(16);
        return rd_grad_in(CP(float, 0), P(1), P(2), CP(float, 3), CP(float, 4), (double*)P(5), I(6), F(7), I(8), I(9), I(10), I(11), I(12), I(13), CP(int32_t, 14), I(15), st);
    case RD_OP_COLSUM: NARGS(8); return rd_colsum(P(0), (float*)P(1), (float*)P(2), L64(3), I(4), I(5), F(6), I(7), st);
    case RD_OP_SEG_LOSS: NARGS(2); return rd_seg_loss(CP(rd_seg_loss_t, 0), I(1), st);
    case RD_OP_REC_LOSS:
        NARGS(15);
        return rd_rec_loss(P(0), P(1), P(2), (float*)P(3), (float*)P(4), I(5), I(6), I(7), I(8), I(9), I(10), I(11), CP(int32_t, 12), F(13), I(14), st);
    case RD_OP_ADAM_STEP: NARGS(1); return rd_adam_step(CP(rd_adam_t, 0), st);
    case RD_OP_ZERO: NARGS(3); return rd_zero((void* const*)P(0), CP(int64_t, 1), I(2), st);
    case RD_OP_RAM_MIX: NARGS(2); return rd_ram_mix(CP(rd_ram_t, 0), I(1), st);
    default: return -1;
    }
}

}  // namespace

namespace {

// ------------------------------------------------------------------------------------------------ lane worker threads
// A launch costs the host ~2.4 us inside the HIP runtime whoever calls it, so one thread cannot enqueue the step's ~305 kernels in less
// than ~0.85 ms.  With rd_run_list_threads(1) every lane k > 0 has a worker thread that enqueues that lane's entries while the calling
// thread enqueues lane 0's and records the events of the cross-lane edges:
//   fork / wait_main edge  the caller records an event at its position on the main stream, then hands the lane {wait for that event}:
//                          the event is recorded before the worker can see the command, so the worker never waits on the host side;
//   join                   the caller hands the lane {record an event}, waits (host side, bounded) until the worker has executed it and
//                          makes the main stream wait for the event.
// Stream order per lane is the list order, exactly as in the single-threaded walk: the GPU sees the same dependency graph.  Before
// rd_run_list returns every worker has drained its queue (everything is enqueued; descriptors may be changed by the caller afterwards).
struct Cmd {
    int kind;            // 0 launch ops[idx]; 1 stream waits for ev; 2 record ev on the stream; 3 new call: device / ops / stream
    int idx;
    hipEvent_t ev;
    const rd_launch_t* ops;
    hipStream_t stream;
};
constexpr unsigned RING = 4096;
// Events of the edges that go THROUGH a worker come from a pool of the worker's own (not from the per-device ring above): the caller
// records (or reserves) an event and the worker uses it later, asynchronously, so a slot may only be reused once the worker has
// executed the command that carries it.  ev_used counts those commands as the worker executes them; the caller hands out slot
// ev_issued % EV_POOL only while ev_issued - ev_used < EV_POOL (it waits otherwise: the worker is behind by a whole pool).
constexpr unsigned EV_POOL = 256;
struct Worker {
    Cmd ring[RING];
    std::atomic<unsigned> head{0}, tail{0};     // single producer (the caller), single consumer (the worker)
    std::atomic<int> err{0};
    std::atomic<int> asleep{0};
    std::atomic<int> poisoned{0};               // a push / drain timed out: the worker discards what it still holds and is never used again
    std::atomic<unsigned> ev_used{0};
    unsigned ev_issued = 0;                     // caller side only
    hipEvent_t evpool[EV_POOL];
    bool ev_ready = false;
    std::mutex m;
    std::condition_variable cv;
};
Worker* g_workers[EV_MAX_DEV][32] = {};          // keyed by (device, lane): two devices never share a single-producer ring
std::mutex g_workers_mutex;
std::atomic<int> g_threads_enabled{0};
std::atomic<int> g_bind_fork_events{1};                     // rd_run_list_bind_fork_events
std::atomic<long long> g_forks_bound{0}, g_forks_recorded{0};   // rd_run_list_fork_counts

void worker_main(Worker* w) {
    const rd_launch_t* ops = nullptr;
    hipStream_t stream = nullptr;
    for (;;) {
        unsigned h = w->head.load(std::memory_order_relaxed);
        int spins = 0;
        while (w->tail.load(std::memory_order_acquire) == h) {
            if (++spins < 20000) { __builtin_ia32_pause(); continue; }
            std::unique_lock<std::mutex> lk(w->m);                // nothing for ~100 us: sleep until the next call pushes
            w->asleep.store(1, std::memory_order_seq_cst);
            if (w->tail.load(std::memory_order_seq_cst) == h) w->cv.wait_for(lk, std::chrono::milliseconds(50));
            w->asleep.store(0, std::memory_order_seq_cst);
            spins = 0;
        }
        const Cmd c = w->ring[h % RING];
        int rc = 0;
        // a poisoned worker (the caller gave up on it and may have freed the list, its descriptors and streams) executes nothing:
        // it only consumes its ring
        if (!w->poisoned.load(std::memory_order_acquire)) {
            switch (c.kind) {
            case 0: rc = call(ops[c.idx], (void*)stream); break;
            case 1: rc = (int)hipStreamWaitEvent(stream, c.ev, 0); break;
            case 2: rc = (int)hipEventRecord(c.ev, stream); break;
            case 3: rc = (int)hipSetDevice(c.idx); ops = c.ops; stream = c.stream; break;
            }
        }
        if (c.kind == 1 || c.kind == 2) w->ev_used.fetch_add(1, std::memory_order_release);
        if (rc && !w->err.load()) w->err.store(rc);
        w->head.store(h + 1, std::memory_order_release);
    }
}

Worker* worker_for(int dev, int lane) {
    std::lock_guard<std::mutex> lk(g_workers_mutex);
    Worker*& slot = g_workers[dev][lane];
    if (!slot || slot->poisoned.load()) {
        Worker* w = new Worker();                                  // never freed: the thread outlives every static destructor
        std::thread(worker_main, w).detach();
        slot = w;                                                  // (a poisoned worker is abandoned with its thread)
    }
    return slot;
}

int push(Worker* w, const Cmd& c) {
    const unsigned t = w->tail.load(std::memory_order_relaxed);
    int spins = 0;
    while (t - w->head.load(std::memory_order_acquire) >= RING) {   // ring full: the worker is behind
        if (++spins > 200000000) { w->poisoned.store(1, std::memory_order_release); return -2; }
        __builtin_ia32_pause();
    }
    w->ring[t % RING] = c;
    w->tail.store(t + 1, std::memory_order_seq_cst);
    if (w->asleep.load(std::memory_order_seq_cst)) {
        std::lock_guard<std::mutex> lk(w->m);
        w->cv.notify_one();
    }
    return 0;
}

// the next event of the worker's pool, once the worker has executed the command that used the slot the last time round
int worker_event(Worker* w, hipEvent_t* out) {
    if (!w->ev_ready) {
        for (unsigned i = 0; i < EV_POOL; ++i) RD_CHECK(hipEventCreateWithFlags(&w->evpool[i], hipEventDisableTiming));
        w->ev_ready = true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (w->ev_issued - w->ev_used.load(std::memory_order_acquire) >= EV_POOL) {
        __builtin_ia32_pause();
        if ((++spins & 0xfffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) {
            w->poisoned.store(1, std::memory_order_release);
            return -2;
        }
    }
    *out = w->evpool[w->ev_issued % EV_POOL];
    ++w->ev_issued;
    return 0;
}

// wait until the worker has executed everything pushed so far (bounded: a stuck runtime call must not hang the caller for ever; the
// worker is poisoned then -- it still holds commands that point into the caller's list)
int drain(Worker* w) {
    const unsigned t = w->tail.load(std::memory_order_relaxed);
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while ((int)(w->head.load(std::memory_order_acquire) - t) < 0) {
        __builtin_ia32_pause();
        if ((++spins & 0xfffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) {
            w->poisoned.store(1, std::memory_order_release);
            return -2;
        }
    }
    return w->err.exchange(0);
}

// Does an entry that makes a lane wait for the main stream follow main-lane launch i before the main stream is given anything else?
// Then launch i carries the event of that fork on its own dispatch packet (common.h rd_launch) and the fork needs no record.
bool fork_follows(const rd_launch_t* ops, int n, int i) {
    for (int j = i + 1; j < n; ++j) {
        const rd_launch_t& q = ops[j];
        if (q.op == RD_OP_FORK) {
            if (q.lane > 0) return true;
            continue;
        }
        if (q.op == RD_OP_JOIN || q.lane == 0) return false;    // the main stream's position changes first
        if (q.wait_main) return true;
    }
    return false;
}
// main-lane launch with the fork's event bound to its (last) kernel; *bound = the event if a launch took it, else null
int call_with_stop_event(const rd_launch_t& o, void* st, hipEvent_t e, hipEvent_t* bound) {
    rd_tls_stop_event = e;
    rd_tls_stop_used = 0;
    const int rc = call(o, st);
    rd_tls_stop_event = nullptr;
    *bound = rd_tls_stop_used ? e : nullptr;
    return rc;
}

int run_list_threaded(const rd_launch_t* ops, int n, void* const* streams, int n_streams, uint32_t* open_lanes, int* bad_index) {
    uint32_t open = open_lanes ? *open_lanes : 0u;
    hipStream_t main_s = (hipStream_t)streams[0];
    int dev = 0;
    RD_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= EV_MAX_DEV) return -1;
    Worker* ws[32] = {};
    int rc = 0, i = 0;
    auto lane_worker = [&](int lane) -> Worker* {
        if (!ws[lane]) {
            ws[lane] = worker_for(dev, lane);
            Cmd c{3, dev, nullptr, ops, (hipStream_t)streams[lane]};
            if (push(ws[lane], c)) return nullptr;
        }
        return ws[lane];
    };
    auto edge_to_lane = [&](int lane) -> int {                     // streams[lane] waits for the main stream's current position
        Worker* w = lane_worker(lane);
        if (!w) return -2;
        if ((hipStream_t)streams[lane] == main_s) return 0;
        hipEvent_t e;
        const int err = worker_event(w, &e);
        if (err) return err;
        RD_CHECK(hipEventRecord(e, main_s));
        return push(w, Cmd{1, 0, e, nullptr, nullptr});
    };
    for (; i < n && rc == 0; ++i) {
        const rd_launch_t& o = ops[i];
        if (o.lane < 0 || o.lane >= n_streams || o.nargs < 0 || o.nargs > RD_LAUNCH_MAX_ARGS) { rc = -1; break; }
        if (o.op == RD_OP_FORK) {
            if (o.lane > 0) { rc = edge_to_lane(o.lane); open |= 1u << o.lane; }
            continue;
        }
        if (o.op == RD_OP_JOIN) {
            if (o.lane > 0 && (open & (1u << o.lane))) {
                open &= ~(1u << o.lane);
                Worker* w = lane_worker(o.lane);
                if (!w) { rc = -2; break; }
                if ((hipStream_t)streams[o.lane] != main_s) {
                    hipEvent_t e;
                    rc = worker_event(w, &e);
                    if (rc) break;
                    rc = push(w, Cmd{2, 0, e, nullptr, nullptr});
                    if (rc) break;
                    rc = drain(w);                                 // the record has been enqueued (and everything before it)
                    if (rc) break;
                    rc = (int)hipStreamWaitEvent(main_s, e, 0);
                } else {
                    rc = drain(w);
                }
            }
            continue;
        }
        if (o.lane == 0) {
            rc = call(o, (void*)main_s);
            continue;
        }
        if (o.wait_main) {
            rc = edge_to_lane(o.lane);
            open |= 1u << o.lane;
            if (rc) break;
        }
        Worker* w = lane_worker(o.lane);
        if (!w) { rc = -2; break; }
        rc = push(w, Cmd{0, i, nullptr, nullptr, nullptr});
    }
    if (rc && bad_index) *bad_index = i;
    for (int k = 1; k < n_streams; ++k)
        if (ws[k]) {
            const int e = drain(ws[k]);
            if (e && !rc) { rc = e; if (bad_index) *bad_index = -1; }
        }
    if (open_lanes) *open_lanes = open;
    return rc;
}

}  // namespace

// which main-lane entries of a list would carry a fork's event on their own dispatch packet (host logic only: no stream, no launch)
extern "C" int rd_run_list_fork_plan(const rd_launch_t* ops, int n, unsigned char* carries) {
    if (n < 0 || (n > 0 && (!ops || !carries))) return -1;
    for (int i = 0; i < n; ++i) {
        const rd_launch_t& o = ops[i];
        carries[i] = (o.op != RD_OP_FORK && o.op != RD_OP_JOIN && o.lane == 0 && fork_follows(ops, n, i)) ? 1 : 0;
    }
    return 0;
}

extern "C" void rd_run_list_fork_counts(long long* bound, long long* recorded) {
    if (bound) *bound = g_forks_bound.load();
    if (recorded) *recorded = g_forks_recorded.load();
}

extern "C" int rd_run_list_bind_fork_events(int enable) {
    const int old = g_bind_fork_events.exchange(enable ? 1 : 0);
    return old;
}

extern "C" int rd_run_list_threads(int enable) {
    const int old = g_threads_enabled.exchange(enable ? 1 : 0);
    return old;
}

extern "C" int rd_join_lanes(void* const* streams, int n_streams, uint32_t mask) {
    for (int k = 1; k < n_streams && k < 32; ++k)
        if (mask & (1u << k)) {
            const int err = edge((hipStream_t)streams[k], (hipStream_t)streams[0]);
            if (err) return err;
        }
    return 0;
}

extern "C" int rd_run_list(const rd_launch_t* ops, int n, void* const* streams, int n_streams, uint32_t* open_lanes, int* bad_index) {
    if (n_streams < 1 || n_streams > 32) return -1;
    hipStream_t main_s = (hipStream_t)streams[0];
    if (n_streams > 1 && g_threads_enabled.load()) {
        // worker threads only outside stream capture (a capture is recorded by the capturing thread)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(main_s, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone)
            return run_list_threaded(ops, n, streams, n_streams, open_lanes, bad_index);
    }
    uint32_t open = open_lanes ? *open_lanes : 0u;
    int rc = 0, i = 0;
    // at_main: an event bound to the LAST launch of the main stream (its dispatch packet's completion), valid until the main stream is
    // given anything else: forks at this position wait for it instead of recording one.  Not under stream capture (a captured graph
    // takes its edges from recorded events).
    hipEvent_t at_main = nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool bind = n_streams > 1 && g_bind_fork_events.load() && hipStreamIsCapturing(main_s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone;
    auto fork_edge = [&](hipStream_t to) -> int {
        if (to == main_s) return 0;
        if (at_main) {
            g_forks_bound.fetch_add(1, std::memory_order_relaxed);
            return (int)hipStreamWaitEvent(to, at_main, 0);
        }
        g_forks_recorded.fetch_add(1, std::memory_order_relaxed);
        return edge(main_s, to);
    };
    for (; i < n && rc == 0; ++i) {
        const rd_launch_t& o = ops[i];
        if (o.lane < 0 || o.lane >= n_streams || o.nargs < 0 || o.nargs > RD_LAUNCH_MAX_ARGS) { rc = -1; break; }
        hipStream_t st = (hipStream_t)streams[o.lane];
        if (o.op == RD_OP_FORK) {
            if (o.lane > 0) { rc = fork_edge(st); open |= 1u << o.lane; }
            continue;
        }
        if (o.op == RD_OP_JOIN) {
            if (o.lane > 0 && (open & (1u << o.lane))) {
                rc = edge(st, main_s);
                open &= ~(1u << o.lane);
                if (st != main_s) at_main = nullptr;              // the main stream now also waits for the lane
            }
            continue;
        }
        if (o.lane > 0 && o.wait_main) {
            rc = fork_edge(st);
            open |= 1u << o.lane;
            if (rc) break;
        }
        if (st == main_s) {
            at_main = nullptr;
            hipEvent_t e;
            if (bind && fork_follows(ops, n, i) && next_event(&e) == 0) rc = call_with_stop_event(o, (void*)st, e, &at_main);   // (an entry that
            // launched no kernel -- rd_zero's fallback to the runtime's fills -- leaves at_main null: the fork records)
            else rc = call(o, (void*)st);
        } else {
            rc = call(o, (void*)st);
        }
        if (rc) break;
    }
    if (rc && bad_index) *bad_index = i;
    if (open_lanes) *open_lanes = open;
    return rc;
}
