// amc_api.hip — host side of libamc.so: implements include/amc.h on top of the HIP kernels.
// No CPU fallback: every entry point that computes needs a gfx950 device.
#include <algorithm>
#include <random>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "amc_internal.h"
#include "slot_arena.h"
#include "scan_accept.h"
#include "camera_math.h"
#include "pose_math.h"  // median_angle_host

using namespace amc;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(AMC_E_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,            \
                        hipGetErrorString(e_));                                             \
    } while (0)

inline uint32_t round_up(uint32_t x, uint32_t m) { return (x + m - 1) / m * m; }

struct Slot {
    void* base = nullptr;  // one allocation: raw | prep | rs128
    ImageDev dev{};
    uint32_t maxsq = 0;    // max_r |raw[r]|^2
    bool valid = false;
    float* kp = nullptr;   // rows x 2 float32 keypoints (x, y)
    double* kp64 = nullptr;  // or rows x 2 float64 points (amc_upload_points_f64)
    double* kpn = nullptr;   // rows x 2 float64 CamFromImg of the points (cameras with distortion), see ensure_normalized
    bool kpn_valid = false;  // kpn matches the current points and camera
    uint32_t kp_rows = 0;
    bool has_kp = false, has_cam = false;
    CameraDev cam{};
    void* grid_base = nullptr;  // guided matching's keypoint grid: sxy | sidx | cell_start (one allocation)
    GridDev grid{};             // n == 0: none (no float32 keypoints, or non-finite coordinates)
};

// grow-only device / pinned-host scratch
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = std::max(n, (size_t)16);
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};
template <typename T>
struct PinBuf {
    T* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = std::max(n, (size_t)16);
        // default flags: mapped and coherent - kernels write it through the same pointer (the scan's copy parts,
        // launch_host_copy) and the host sees the data once the kernel's completion event has been waited for
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p), want * sizeof(T), 0);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

// The slots' device memory: amc::SlotArenaT (slot_arena.h) over the HIP allocator.
struct HipRaw {
    static int alloc(void** p, size_t bytes) { return (int)hipMalloc(p, bytes); }
    static void free(void* p) { (void)hipFree(p); }
    static void clear_error() { (void)hipGetLastError(); }
};
struct SlotArena : amc::SlotArenaT<HipRaw> {
    template <class T>
    hipError_t alloc(T** out, size_t bytes) {
        return (hipError_t)amc::SlotArenaT<HipRaw>::alloc(out, bytes);
    }
};

// Pinned host buffers behind amc_match_result.matches.  A result leases one (the D2H copies of a call land in it
// directly); amc_match_result_free returns it for the next call, so a pipeline allocates pinned memory once.  The
// pool is shared-owned: results may outlive their context.
struct PinnedPool {
    // Idle buffers are kept for the next call, but not without bound: at most three, and at most kMaxIdleBytes in
    // total (a dense 500 x 4096 call returns a 1 GiB table: one such buffer stays, a second one does not).  With one
    // context per device (gpu_index "-1") the bound holds per device.  amc_ctx_trim empties the pool.
    static constexpr size_t kMaxIdleBytes = (size_t)3 << 29;  // 1.5 GiB
    std::mutex mu;
    std::vector<PinBuf<uint32_t>> idle;
    // want (elements): the smallest idle buffer that holds it, else the largest (the caller grows it).  A call that leases
    // two buffers of different sizes (verification: records and masks) would otherwise hand the larger one to whichever
    // lease comes first and re-allocate the other - a hipHostMalloc of tens of MB in every early call of a run.
    PinBuf<uint32_t> acquire(size_t want = 0) {
        std::lock_guard<std::mutex> lock(mu);
        if (idle.empty()) return PinBuf<uint32_t>();
        auto better = [&](const PinBuf<uint32_t>& a, const PinBuf<uint32_t>& b) {
            const bool fa = a.cap >= want, fb = b.cap >= want;
            if (want && fa != fb) return fa;       // one that fits beats one that does not
            if (want && fa) return a.cap < b.cap;  // both fit: the smaller
            return a.cap > b.cap;                  // neither fits (or no wish): the larger
        };
        size_t best = 0;
        for (size_t i = 1; i < idle.size(); ++i)
            if (better(idle[i], idle[best])) best = i;
        PinBuf<uint32_t> b = idle[best];
        idle.erase(idle.begin() + best);
        return b;
    }
    void give_back(PinBuf<uint32_t> b) {
        if (!b.p) return;
        std::lock_guard<std::mutex> lock(mu);
        idle.push_back(b);
        auto total = [&] {
            size_t t = 0;
            for (auto& x : idle) t += x.cap * sizeof(uint32_t);
            return t;
        };
        // drop the smallest until the bounds hold (the largest is the one the next call of a pipeline wants); a single
        // buffer above the byte bound is dropped as well
        while (!idle.empty() && (idle.size() > 3 || total() > kMaxIdleBytes)) {
            size_t small = 0;
            for (size_t i = 1; i < idle.size(); ++i)
                if (idle[i].cap < idle[small].cap) small = i;
            idle[small].release();
            idle.erase(idle.begin() + small);
        }
    }
    void trim() {
        std::lock_guard<std::mutex> lock(mu);
        for (auto& b : idle) b.release();
        idle.clear();
    }
    ~PinnedPool() {
        for (auto& b : idle) b.release();
    }
};

// Verification runs as slices (VerifyRun below).  A slice owns its trial tables and mask buffers, and per size class its
// pair lists, workspaces and queue heads: slice k's F/H kernel runs beside slice k + 1's essential-matrix kernel, and
// the masks stay where they are until the call's packing step.  Grow-only, kept by the context across calls.
struct VerifyClassSlot {
    PinBuf<TvgPair> h_pairs, h_pairs_e;  // pinned staging of the two lists
    DevBuf<TvgPair> pairs, pairs_e;
    DevBuf<double> ws, ws_e;
    DevBuf<uint8_t> maskws;
    void release() { pairs.release(); pairs_e.release(); ws.release(); ws_e.release(); maskws.release(); h_pairs.release(); h_pairs_e.release(); }
};
struct VerifySliceBufs {
    PinBuf<uint32_t> h_tabs;
    DevBuf<uint32_t> tabs;
    DevBuf<uint8_t> outmask, emask;
    VerifyClassSlot cls[4];
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // E launches begin / end, F/H launches begin / end
    hipEvent_t ev_e_done = nullptr, ev_aux_done = nullptr;
    bool aux_pending = false;
    void release() {
        tabs.release(); outmask.release(); emask.release(); h_tabs.release();
        for (auto& k : cls) k.release();
    }
    ~VerifySliceBufs() {
        release();
        for (auto& e : ev)
            if (e) (void)hipEventDestroy(e);
        if (ev_e_done) (void)hipEventDestroy(ev_e_done);
        if (ev_aux_done) (void)hipEventDestroy(ev_aux_done);
    }
};
constexpr size_t kVScalarWords = 128;  // [0] bad match indices, [1] stream overruns, [2 + 8 slice + 2 class (+ 1)] queue heads
constexpr size_t kMaxStreamWords = (size_t)1 << 28;  // 1 GiB of words: max_num_trials ~ 1.6e7 at the default ratio

// key of a cached dyn_max_num_trials table
struct TrialTabKey {
    uint32_t M;
    double confidence, multiplier;
    bool operator<(const TrialTabKey& o) const {
        if (M != o.M) return M < o.M;
        if (confidence != o.confidence) return confidence < o.confidence;
        return multiplier < o.multiplier;
    }
};
constexpr size_t kTrialTabCacheWords = size_t(64) << 20;  // 256 MB of uint32

}  // namespace

struct amc_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;  // D2H of a batch's matches, beside the next batch's kernels
    hipEvent_t cev[2] = {nullptr, nullptr};  // batch k's matches are in place in d_keep
    // verification: the launches of the larger size classes (few pairs, each several milliseconds on one wave) run on this
    // stream beside the bulk class on `stream` instead of behind it, with their own pair lists and workspaces
    hipStream_t aux_stream = nullptr;
    hipEvent_t aev[2] = {nullptr, nullptr};
    std::vector<Slot> slots;
    bool table_dirty = true;
    DevBuf<ImageDev> d_imgs;
    DevBuf<GridDev> d_grids;   // guided matching's keypoint grids, by slot (uploaded with d_imgs)
    float* d_lut = nullptr;
    std::vector<float> h_lut;
    // the scan's accept-bit thresholds (scan_accept.h) for the last (max_ratio, max_distance) a match call used
    ScanAccept* d_accept = nullptr;
    ScanAccept h_accept{};
    float accept_ratio = 0.f, accept_distance = 0.f;
    bool accept_valid = false;
    uint32_t* d_scalars = nullptr;  // [0] cursor, [1] queue head, [2] maxsq scratch, [3] resolve errors, [4] stream overrun, [5] mfma items, [7] copy parts taken
    // per-batch scratch: TWO sets (round 6).  With one set a batch's cross-check chain (resolve, candidate selection,
    // reverse scan, finalize) has to finish before the next batch's forward scan may write the tables; with two the
    // next scan is launched right behind this one and the chain runs beside it on chain_stream (match_impl).  Set 0's
    // scalars are d_scalars itself.
    struct MatchScratch {
        uint32_t* scalars = nullptr;  // [0] cursor, [1] queue head, [3] resolve errors, [5] mfma items, [7] copy parts taken
        DevBuf<PairDev> d_pairs;
        DevBuf<Dot4Work> d_work;
        DevBuf<uint32_t> d_order, d_order2;
        // mfma work items: group cuts of the two queue orders, scratch of the packing kernels, the descriptors
        DevBuf<uint32_t> d_grp, d_grp2, d_seg_base, d_grp_segs, d_grp_item_base;
        DevBuf<SegDesc> d_segs;
        DevBuf<Top2> d_rowbuf, d_colbuf;
        DevBuf<uint32_t> d_accmask;  // one accept bit per row-table entry (mfma pairs)
        DevBuf<GuidedDev> d_guided;  // guided matching: one filter model per pair of the batch
        DevBuf<uint32_t> d_pair_off, d_pair_cnt, d_matches, d_cand_cnt, d_candbuf;
        void release_all() {
            d_pairs.release(); d_work.release(); d_order.release(); d_order2.release();
            d_grp.release(); d_grp2.release(); d_seg_base.release(); d_grp_segs.release();
            d_grp_item_base.release(); d_segs.release();
            d_rowbuf.release(); d_colbuf.release(); d_accmask.release(); d_guided.release();
            d_pair_off.release(); d_pair_cnt.release(); d_matches.release();
            d_cand_cnt.release(); d_candbuf.release();
        }
        void release_large() {  // (amc_ctx_trim)
            d_rowbuf.release(); d_colbuf.release(); d_accmask.release(); d_matches.release(); d_candbuf.release();
            d_segs.release(); d_seg_base.release();
        }
    };
    MatchScratch ms[2];
    uint32_t* d_scalars_alt = nullptr;  // set 1's scalars (16 words)
    hipStream_t chain_stream = nullptr; // the cross-check chain of batch k beside the forward scan of batch k + 1
    hipEvent_t sev[2] = {nullptr, nullptr};  // set k's tables are free again (chain and reorder of its batch are done)
    // amc_match_verify_pairs: the matches of every batch of the call stay here (appended batch after batch), so
    // that the verification kernel reads them where the matcher left them instead of from a host round trip
    DevBuf<uint32_t> d_keep;
    DevBuf<uint64_t> d_csr;                 // per batch: where each pair's matches go in d_keep (pair order)
    uint64_t resident_matches = 0;          // matches of the LAST match call, in its result's CSR order, at d_keep (amc_ctx_resident_matches)
    PinBuf<uint64_t> h_csr[2];
    std::shared_ptr<PinnedPool> result_pool = std::make_shared<PinnedPool>();
    SlotArena arena;  // the slots' device memory
    // host staging of a match batch, two sets: batch k+1 is prepared and enqueued while the results of
    // batch k are still being copied out and scattered (match_impl)
    PinBuf<PairDev> h_pairs[2];
    PinBuf<Dot4Work> h_work[2];
    PinBuf<uint32_t> h_order[2], h_order2[2], h_pair_off[2], h_pair_cnt[2], h_matches[2], h_bscalars[2];
    PinBuf<uint32_t> h_grp[2], h_grp2[2];  // where the streamed image changes in h_order / h_order2 (ngroups + 1 cuts)
    PinBuf<uint32_t> h_scalars;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t bev[2][5] = {{nullptr, nullptr, nullptr, nullptr, nullptr},
                            {nullptr, nullptr, nullptr, nullptr, nullptr}};  // scan start/end, cross end, small D2H, matches
    // verification scratch
    DevBuf<TvgImage> d_timgs;
    DevBuf<uint32_t> d_tmatches;
    DevBuf<TvgEState> d_estate;       // essential-matrix kernel -> F/H kernel hand-off, by pair
    // the slices of a verification run (lists, workspaces, tables, masks), its streams and events
    std::vector<std::unique_ptr<VerifySliceBufs>> vslices;
    hipStream_t vstream[2] = {nullptr, nullptr};
    hipEvent_t vev_setup = nullptr, vev_matches = nullptr;
    hipEvent_t kev[2] = {nullptr, nullptr};  // match batch k's rows are in place in d_keep (amc_match_verify_pairs)
    uint32_t* d_vscalars = nullptr;   // kVScalarWords
    std::vector<double> wm_cut_cache; // TvgParams::wm_cut for (wm_cut_conf, wm_cut_mult)
    double wm_cut_conf = 0.0, wm_cut_mult = 0.0;
    bool wm_cut_on_device = false;    // d_wmcut holds wm_cut_cache
    // Every upload of a verification call comes from pinned memory (round 6: the pageable ones - a few hundred KB each -
    // stalled a call by 10-20 ms once in ten to twenty calls, profiles/r06/pipeline_timeline_v1.txt), and the image table
    // is uploaded only when it changed.
    PinBuf<TvgImage> h_timgs;
    std::vector<TvgImage> timgs_on_device;
    PinBuf<TvgPair> h_tp;             // the call's pair records in the caller's order (VerifyRun::tp)
    PinBuf<uint64_t> h_moff;
    // tempered words of std::mt19937(seed): the sample stream every pair consumes (TvgParams::stream)
    DevBuf<uint32_t> d_stream;
    uint32_t stream_seed = 0;
    size_t stream_len = 0;
    DevBuf<double> d_wmcut;
    DevBuf<TvgOut> d_tout;
    // the call's results in the caller's layout (pack_verify_kernel): records without their counters, masks at the
    // input's CSR offsets - copied straight into the pinned buffers the result leases
    DevBuf<amc_tvg> d_tvg_packed;
    DevBuf<uint8_t> d_mask_packed;
    DevBuf<uint64_t> d_moff;
    DevBuf<TvgPair> d_tp_all;
    DevBuf<unsigned long long> d_worksum;
    amc::VerifyResident vres;  // the last verification call's results, where they lie (amc_internal.h)
    double timeline[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // amc_ctx_last_timeline
    double last_hook_ms = 0.0;
    std::shared_ptr<PinnedPool> verify_pool = std::make_shared<PinnedPool>();
    PinBuf<TvgOut> h_tout;    // where the records and masks of a verification call land (copied out before return)
    PinBuf<uint8_t> h_tmask;
    // dyn_max_num_trials tables by (match count, confidence, multiplier), see verify_impl
    std::map<TrialTabKey, std::vector<uint32_t>> trial_tabs;
    size_t trial_tab_words = 0;
    // relative-pose scratch
    DevBuf<PosePair> d_ppairs;
    DevBuf<uint32_t> d_pmatches;
    DevBuf<double> d_pcos;
    DevBuf<PoseOut> d_pout;
    PinBuf<PosePair> h_ppairs;  // pose_impl's staging (pinned, kept: 10^5 pairs are 23 + 32 MB; pageable vectors cost their
    PinBuf<PoseOut> h_pout;     //  first touch and a staged copy in every call)
};

namespace amc {
int api_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
CtxView ctx_view(amc_ctx* c) {
    return CtxView{c->device, c->stream, c->resident_matches ? c->d_keep.p : nullptr, c->resident_matches};
}
VerifyResident verify_resident(amc_ctx* c) { return c->vres; }
}  // namespace amc

extern "C" {

const char* amc_last_error(void) { return g_err.c_str(); }
int amc_abi_version(void) { return AMC_ABI_VERSION; }

int amc_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        if (e == hipErrorNoDevice) return 0;
        return fail(AMC_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    return n;
}

void amc_match_opts_default(amc_match_opts* o) {
    if (!o) return;
    o->max_ratio = 0.8;     // SiftMatchingOptions defaults, SURVEY.md A.2
    o->max_distance = 0.7;
    o->cross_check = 1;
    o->kernel = AMC_KERNEL_AUTO;
}

int amc_ctx_create(int device_id, amc_ctx** out) {
    if (!out) return fail(AMC_E_INVALID, "amc_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n)
        return fail(AMC_E_INVALID, "amc_ctx_create: device %d out of range (%d devices)",
                    device_id, n);
    HIPCHK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(AMC_E_HIP, "amc_ctx_create: device %d is %s; this library is gfx950-only",
                    device_id, prop.gcnArchName);
    amc_ctx* c = new (std::nothrow) amc_ctx();
    if (!c) return fail(AMC_E_NOMEM, "amc_ctx_create: out of host memory");
    c->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(AMC_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    c->stream = c->own_stream;
    e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(c->own_stream);
        delete c;
        return fail(AMC_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    bool ev_ok = true;  // (an event that was never created would fail every later record: fail here instead)
    for (auto& ev : c->cev) ev_ok &= hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
    for (auto& ev : c->sev) ev_ok &= hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
    if (hipStreamCreateWithFlags(&c->chain_stream, hipStreamNonBlocking) != hipSuccess) c->chain_stream = nullptr;
    {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // (0, 0 when it fails: the default priority)
        // (lowest priority; measured: the priority makes no difference here - what matters is that the aux launches are
        // issued first - so the one that can never be in the bulk class's way)
        if (hipStreamCreateWithPriority(&c->aux_stream, hipStreamNonBlocking, lo) != hipSuccess) c->aux_stream = nullptr;
        for (auto& ev : c->aev) ev_ok &= hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
    }
    for (auto& vs : c->vstream)
        if (hipStreamCreateWithFlags(&vs, hipStreamNonBlocking) != hipSuccess) vs = nullptr;
    if (!c->vstream[1]) c->vstream[1] = c->vstream[0];
    ev_ok &= hipEventCreateWithFlags(&c->vev_setup, hipEventDisableTiming) == hipSuccess;
    ev_ok &= hipEventCreateWithFlags(&c->vev_matches, hipEventDisableTiming) == hipSuccess;
    for (auto& ev : c->kev) ev_ok &= hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
    for (auto& ev : c->ev) ev_ok &= hipEventCreate(&ev) == hipSuccess;
    for (auto& set : c->bev)
        for (auto& ev : set) ev_ok &= hipEventCreate(&ev) == hipSuccess;
    if (!ev_ok) {
        amc_ctx_destroy(c);
        return fail(AMC_E_HIP, "amc_ctx_create: hipEventCreate failed");
    }
    // acos table with the HOST libm (the same one COLMAP's CPU path and the oracle call)
    c->h_lut.resize(kAcosLutSize);
    const float kDistNorm = 1.0f / (512.0f * 512.0f);
    for (int d = 0; d < kAcosLutSize; ++d)
        c->h_lut[d] = acosf(std::fmin(kDistNorm * (float)d, 1.0f));
    if (hipMalloc(reinterpret_cast<void**>(&c->d_lut), kAcosLutSize * sizeof(float)) !=
            hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_scalars), 16 * sizeof(uint32_t)) !=
            hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_scalars_alt), 16 * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_vscalars), kVScalarWords * sizeof(uint32_t)) != hipSuccess ||
        hipMemcpy(c->d_lut, c->h_lut.data(), kAcosLutSize * sizeof(float),
                  hipMemcpyHostToDevice) != hipSuccess ||
        c->h_scalars.ensure(16) != hipSuccess) {
        amc_ctx_destroy(c);
        return fail(AMC_E_HIP, "amc_ctx_create: device allocation failed");
    }
    c->ms[0].scalars = c->d_scalars;
    c->ms[1].scalars = c->d_scalars_alt;
    *out = c;
    return AMC_OK;
}

void amc_ctx_destroy(amc_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    const bool dprof = std::getenv("AMC_DESTROY_PROFILE") != nullptr;  // wall-clock of the teardown's parts on stderr
    auto tp0 = std::chrono::steady_clock::now();
    auto dlap = [&](const char* what) {
        if (!dprof) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[amc destroy] %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tp0).count());
        tp0 = now;
    };
    (void)hipDeviceSynchronize();
    dlap("sync");
    c->slots.clear();
    c->arena.release_all();
    dlap("slots");
    c->d_imgs.release();
    c->d_grids.release();
    if (c->d_lut) (void)hipFree(c->d_lut);
    if (c->d_accept) (void)hipFree(c->d_accept);
    if (c->d_scalars) (void)hipFree(c->d_scalars);
    for (auto& m : c->ms) m.release_all();
    if (c->d_scalars_alt) (void)hipFree(c->d_scalars_alt);
    c->d_keep.release(); c->d_csr.release();
    dlap("match device buffers");
    c->h_csr[0].release(); c->h_csr[1].release();
    c->h_tout.release(); c->h_tmask.release();
    for (int k = 0; k < 2; ++k) {
        c->h_pairs[k].release(); c->h_work[k].release(); c->h_order[k].release(); c->h_order2[k].release();
        c->h_grp[k].release(); c->h_grp2[k].release();
        c->h_pair_off[k].release(); c->h_pair_cnt[k].release(); c->h_matches[k].release();
        c->h_bscalars[k].release();
    }
    c->h_scalars.release();
    dlap("pinned host buffers");
    c->d_timgs.release(); c->d_tmatches.release();
    c->d_estate.release(); c->d_stream.release(); c->d_wmcut.release();
    c->d_tout.release();
    c->vslices.clear();
    c->h_timgs.release(); c->h_tp.release(); c->h_moff.release();
    if (c->d_vscalars) (void)hipFree(c->d_vscalars);
    c->d_tvg_packed.release(); c->d_mask_packed.release(); c->d_moff.release(); c->d_tp_all.release(); c->d_worksum.release();
    c->d_ppairs.release(); c->d_pmatches.release(); c->d_pcos.release(); c->d_pout.release();
    c->h_ppairs.release(); c->h_pout.release();
    dlap("verify device buffers");
    for (auto& ev : c->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& set : c->bev)
        for (auto& ev : set)
            if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : c->cev)
        if (ev) (void)hipEventDestroy(ev);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->chain_stream) (void)hipStreamDestroy(c->chain_stream);
    for (auto& ev : c->sev)
        if (ev) (void)hipEventDestroy(ev);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    for (auto& ev : c->aev)
        if (ev) (void)hipEventDestroy(ev);
    if (c->vstream[1] && c->vstream[1] != c->vstream[0]) (void)hipStreamDestroy(c->vstream[1]);
    if (c->vstream[0]) (void)hipStreamDestroy(c->vstream[0]);
    if (c->vev_setup) (void)hipEventDestroy(c->vev_setup);
    if (c->vev_matches) (void)hipEventDestroy(c->vev_matches);
    for (auto& ev : c->kev)
        if (ev) (void)hipEventDestroy(ev);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    dlap("events and streams");
    delete c;  // (the result pools' idle pinned buffers go with their last owner)
    dlap("delete (pools)");
}

int amc_ctx_set_stream(amc_ctx* c, void* hip_stream) {
    if (!c) return fail(AMC_E_INVALID, "amc_ctx_set_stream: ctx is NULL");
    c->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->own_stream;
    return AMC_OK;
}

int amc_ctx_trim(amc_ctx* c) {
    if (!c) return fail(AMC_E_INVALID, "amc_ctx_trim: NULL ctx");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->copy_stream) HIPCHK(hipStreamSynchronize(c->copy_stream));
    if (c->chain_stream) HIPCHK(hipStreamSynchronize(c->chain_stream));
    // per-call scratch and result staging: everything a later call re-allocates on demand (uploaded images, the acos
    // table, the sample stream and the trial tables stay)
    c->result_pool->trim();
    c->d_keep.release(); c->d_csr.release();
    c->resident_matches = 0;
    c->vres = amc::VerifyResident{};
    for (auto& m : c->ms) m.release_large();
    // the verification kernels' sample-stream table is rebuilt on demand (a host mt19937 run + one upload): keep the
    // default-sized one (~0.3 MB at COLMAP's default trial caps), drop one that a large max_num_trials blew up
    if (c->d_stream.cap * sizeof(uint32_t) > ((size_t)16 << 20)) {
        c->d_stream.release();
        c->stream_len = 0;
    }
    if (c->aux_stream) HIPCHK(hipStreamSynchronize(c->aux_stream));
    for (auto vs : c->vstream)
        if (vs) HIPCHK(hipStreamSynchronize(vs));
    for (auto& sl : c->vslices)
        if (sl) sl->release();
    c->h_tp.release(); c->h_moff.release(); c->h_ppairs.release(); c->h_pout.release();
    c->d_estate.release();
    c->d_tout.release(); c->d_tmatches.release(); c->d_pmatches.release(); c->d_pcos.release();
    c->d_tvg_packed.release(); c->d_mask_packed.release(); c->d_moff.release(); c->d_tp_all.release();
    c->verify_pool->trim();
    c->arena.release_idle_slabs();
    c->h_tout.release(); c->h_tmask.release();
    for (int k = 0; k < 2; ++k) {
        c->h_matches[k].release();
        c->h_csr[k].release();
    }
    return AMC_OK;
}

int amc_ctx_resident_matches(amc_ctx* c, const uint32_t** dev_matches, uint64_t* num_matches) {
    if (!c || !dev_matches || !num_matches) return fail(AMC_E_INVALID, "amc_ctx_resident_matches: NULL argument");
    *dev_matches = c->resident_matches ? c->d_keep.p : nullptr;
    *num_matches = c->resident_matches;
    return AMC_OK;
}

int amc_upload_matches(amc_ctx* c, const uint32_t* matches, uint64_t num_matches) {
    if (!c) return fail(AMC_E_INVALID, "amc_upload_matches: ctx is NULL");
    if (num_matches > 0 && !matches) return fail(AMC_E_INVALID, "amc_upload_matches: NULL rows");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));  // (nothing of an earlier call reads the table any more: every entry point blocks)
    c->resident_matches = 0;
    c->vres = amc::VerifyResident{};  // (a resident verification result indexes the table this call rewrites)
    if (num_matches == 0) return AMC_OK;
    HIPCHK(c->d_keep.ensure((size_t)(2 * num_matches)));
    HIPCHK(hipMemcpyAsync(c->d_keep.p, matches, (size_t)(2 * num_matches) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->resident_matches = num_matches;
    return AMC_OK;
}

int amc_ctx_reserve_slots(amc_ctx* c, uint32_t num_slots) {
    if (!c) return fail(AMC_E_INVALID, "amc_ctx_reserve_slots: ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->slots.assign(num_slots, Slot());
    c->arena.release_all();  // every slot is gone: the slabs go back to the driver
    c->table_dirty = true;
    return AMC_OK;
}

int amc_ctx_grow_slots(amc_ctx* c, uint32_t num_slots) {
    if (!c) return fail(AMC_E_INVALID, "amc_ctx_grow_slots: ctx is NULL");
    if (num_slots < c->slots.size())
        return fail(AMC_E_INVALID, "amc_ctx_grow_slots: %u < current %zu slots (use amc_ctx_reserve_slots to reset)",
                    num_slots, c->slots.size());
    c->slots.resize(num_slots);
    c->table_dirty = true;
    return AMC_OK;
}

static int upload_common(amc_ctx* c, uint32_t slot, const void* src, uint32_t rows,
                         hipMemcpyKind kind) {
    if (!c) return fail(AMC_E_INVALID, "upload_descriptors: ctx is NULL");
    if (slot >= c->slots.size())
        return fail(AMC_E_INVALID, "upload_descriptors: slot %u >= reserved %zu", slot,
                    c->slots.size());
    if (rows > 0 && !src) return fail(AMC_E_INVALID, "upload_descriptors: NULL data, rows=%u", rows);
    if (rows > (1u << 30)) return fail(AMC_E_INVALID, "upload_descriptors: rows=%u too large", rows);
    HIPCHK(hipSetDevice(c->device));
    Slot& s = c->slots[slot];
    if (s.base) {
        HIPCHK(hipStreamSynchronize(c->stream));
        c->arena.free(s.base);
        s.base = nullptr;
    }
    c->table_dirty = true;
    s.valid = true;
    s.dev.rows = rows;
    s.dev.rows_pad = round_up(rows, kRowPad);
    s.maxsq = 0;
    if (rows == 0) return AMC_OK;
    const size_t rp = s.dev.rows_pad;
    const size_t bytes = rp * kDim * 2 + rp * sizeof(int32_t);
    hipError_t e = c->arena.alloc(&s.base, bytes);
    if (e != hipSuccess) {
        s.valid = false;
        s.dev = ImageDev{};
        return fail(AMC_E_NOMEM, "upload_descriptors: hipMalloc(%zu): %s", bytes,
                    hipGetErrorString(e));
    }
    uint8_t* raw = static_cast<uint8_t*>(s.base);
    uint8_t* prep = raw + rp * kDim;
    int32_t* rs = reinterpret_cast<int32_t*>(prep + rp * kDim);
    s.dev.raw = raw;
    s.dev.prep = prep;
    s.dev.rs128 = rs;
    HIPCHK(hipMemcpyAsync(raw, src, (size_t)rows * kDim, kind, c->stream));
    if (rp > rows)
        HIPCHK(memset_async(raw + (size_t)rows * kDim, 0, (rp - rows) * kDim, c->stream));
    HIPCHK(memset_async(c->d_scalars + 2, 0, sizeof(uint32_t), c->stream));
    HIPCHK(launch_prep(raw, prep, rs, s.dev.rows_pad, c->d_scalars + 2, c->stream));
    HIPCHK(hipMemcpyAsync(c->h_scalars.p + 2, c->d_scalars + 2, sizeof(uint32_t),
                          hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    s.maxsq = c->h_scalars.p[2];
    return AMC_OK;
}

int amc_upload_descriptors(amc_ctx* c, uint32_t slot, const uint8_t* host_desc, uint32_t rows) {
    return upload_common(c, slot, host_desc, rows, hipMemcpyHostToDevice);
}
int amc_upload_descriptors_device(amc_ctx* c, uint32_t slot, const void* dev_desc,
                                  uint32_t rows) {
    return upload_common(c, slot, dev_desc, rows, hipMemcpyDeviceToDevice);
}

int amc_get_acos_lut(amc_ctx* c, float* out) {
    if (!c || !out) return fail(AMC_E_INVALID, "amc_get_acos_lut: NULL argument");
    HIPCHK(hipSetDevice(c->device));
    // read it back from the device so the test sees what the kernels see
    HIPCHK(hipMemcpy(out, c->d_lut, kAcosLutSize * sizeof(float), hipMemcpyDeviceToHost));
    return AMC_OK;
}

namespace {

struct ResultPriv {
    std::vector<uint64_t> offsets;
    PinBuf<uint32_t> matches;               // leased from the context's pool, returned by amc_match_result_free
    std::shared_ptr<PinnedPool> pool;
    ~ResultPriv() {
        if (pool) pool->give_back(matches);
        else matches.release();
    }
};

int ceil_log2(uint32_t x) {
    int b = 0;
    while ((1u << b) < x) ++b;
    return b;
}

// limits for one batch (bytes of device scratch)
// sized for 288 GB of HBM: few, large batches (each batch ends in a host synchronisation)
#ifndef AMC_MATCH_OVERLAP_DEFAULT
#define AMC_MATCH_OVERLAP_DEFAULT 0
#endif
constexpr size_t kFirstBatchDiv = 0;  // match_impl: a call's first batch as a fraction of a full one (0: a full one)
constexpr bool kMatchOverlapDefault = AMC_MATCH_OVERLAP_DEFAULT != 0;  // match_impl: a batch's chain beside the next batch's scan
constexpr size_t kMaxTop2Entries = (size_t)256 << 20;  // 256 Mi entries x 16 B = 4 GiB per side
constexpr size_t kMaxMatchCap = (size_t)256 << 20;     // worst-case matches of a batch: x 8 B = 2 GiB (device)

}  // namespace

// Whether a guided pair may take the candidate-generation kernel (match_guided.hip), and what that kernel needs
// beyond the float model: guided_region.h's guided_pair_setup on the two images' keypoint boxes.  Anything it turns
// down keeps the dense kernel, which evaluates the filter on all n1 x n2 pairings.
static void guided_grid_setup(GuidedDev& g, const GridDev& g1, const GridDev& g2, bool dense_only) {
    g.grid_ok = 0;
    g.bound[0] = g.bound[1] = 0.0;
    for (int k = 0; k < 9; ++k) g.minv[k] = 0.0;
    if (dense_only || g1.n == 0 || g2.n == 0) return;
    const float box1[4] = {g1.x0, g1.y0, g1.bx1, g1.by1}, box2[4] = {g2.x0, g2.y0, g2.bx1, g2.by1};
    g.grid_ok = guided::guided_pair_setup(g.kind, g.m, g.max_residual, box1, box2, g.bound, g.minv) ? 1 : 0;
}

// amc_match_pairs, and with `geoms` != nullptr guided matching (every pair then runs the dot4
// kernel with the pair's float32 filter; geoms[p] must have a configuration COLMAP guides on)
// keep_off != nullptr: the matches also stay on the device (c->d_keep) and keep_off[p] receives the position
// (in matches) of pair p's list there.
// batch_hook (amc_match_verify_pairs): called once per batch, in order, as soon as the batch's matches are in the
// resident table and the NEXT batch has been enqueued - with the pairs [begin, end) of the batch, the call's CSR offsets
// (valid up to `end`), where each pair's rows start in the resident table, and an event behind the reorder that put
// them there (null: the batch has no matches).  Its host work runs beside the next batch's scan.
// plan: called right after the batch's counts are on the host and BEFORE the next batch is enqueued, with the next
// batch's scan time as the host estimates it (0: there is no next batch) - returns how many CUs that scan shall leave
// free for the work the hook is about to launch; the same number comes back in `cus_free`.
struct BatchHook {
    std::function<int(size_t begin, size_t end, const uint64_t* offsets, double next_scan_ms)> plan;
    std::function<int(size_t begin, size_t end, const uint64_t* offsets, const uint64_t* keep_off, hipEvent_t ready, int cus_free)> submit;
};
static void verify_streams_sync(amc_ctx* c);

static int match_impl(amc_ctx* c, const uint32_t* slot1, const uint32_t* slot2, size_t npairs,
                      const amc_match_opts* opts_in, const amc_tvg* geoms, double max_error,
                      amc_match_result* out, std::vector<uint64_t>* keep_off = nullptr, const BatchHook* batch_hook = nullptr) {
    if (!c || !out) return fail(AMC_E_INVALID, "amc_match_pairs: NULL ctx/out");
    std::memset(out, 0, sizeof *out);
    if (npairs > 0 && (!slot1 || !slot2))
        return fail(AMC_E_INVALID, "amc_match_pairs: NULL pair arrays");
    // AMC_MATCH_PROFILE=1: wall-clock of the call's host phases on stderr
    const bool prof = std::getenv("AMC_MATCH_PROFILE") != nullptr;
    const auto wall0 = std::chrono::steady_clock::now();
    double t_prepare = 0.0, t_collect = 0.0, t_enqueue = 0.0, t_scatter = 0.0;
    auto since = [](std::chrono::steady_clock::time_point t) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
    };
    amc_match_opts o;
    if (opts_in) o = *opts_in; else amc_match_opts_default(&o);
    if (o.kernel != AMC_KERNEL_AUTO && o.kernel != AMC_KERNEL_MFMA && o.kernel != AMC_KERNEL_DOT4)
        return fail(AMC_E_INVALID, "amc_match_pairs: unknown kernel %d", o.kernel);
    uint64_t rows_total = 0;  // rows of image 1 (padded) over the call: what is left when a batch is carved
    for (size_t i = 0; i < npairs; ++i) {
        if (slot1[i] >= c->slots.size() || slot2[i] >= c->slots.size())
            return fail(AMC_E_INVALID, "amc_match_pairs: pair %zu references slot out of range", i);
        if (!c->slots[slot1[i]].valid || !c->slots[slot2[i]].valid)
            return fail(AMC_E_STATE, "amc_match_pairs: pair %zu references a slot with no "
                        "descriptors uploaded", i);
        rows_total += c->slots[slot1[i]].dev.rows_pad;  // (one pass over the pair list: a loop-closure call has 10^7 pairs)
    }
    std::vector<GuidedDev> h_guided;
    const bool guided_dense_only = std::getenv("AMC_GUIDED_DENSE") != nullptr;  // (test hook: the dense kernel for every pair)
    if (geoms) {
        h_guided.resize(npairs);
        for (size_t i = 0; i < npairs; ++i) {
            const int cfg = geoms[i].config;
            GuidedDev& g = h_guided[i];
            g.kind = (cfg == AMC_TVG_CALIBRATED || cfg == AMC_TVG_UNCALIBRATED) ? kGuidedF
                     : (cfg == AMC_TVG_PLANAR || cfg == AMC_TVG_PANORAMIC || cfg == AMC_TVG_PLANAR_OR_PANORAMIC)
                         ? kGuidedH : kGuidedNone;
            if (g.kind == kGuidedNone)
                return fail(AMC_E_INVALID, "amc_match_guided_pairs: pair %zu: configuration %d has no guided "
                            "matching (COLMAP keeps the inlier matches it has)", i, cfg);
            const double* m = g.kind == kGuidedF ? geoms[i].F : geoms[i].H;
            for (int k = 0; k < 9; ++k) g.m[k] = (float)m[k];
            g.max_residual = (float)(max_error * max_error);
            const Slot& a = c->slots[slot1[i]];
            const Slot& b = c->slots[slot2[i]];
            if (!a.kp || !b.kp || a.kp_rows < a.dev.rows || b.kp_rows < b.dev.rows)
                return fail(AMC_E_STATE, "amc_match_guided_pairs: pair %zu: float32 keypoints (one per descriptor) "
                            "must be uploaded for both images", i);
            guided_grid_setup(g, a.grid, b.grid, guided_dense_only);
        }
    }
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream;

    if (c->table_dirty) {
        HIPCHK(c->d_imgs.ensure(c->slots.size()));
        std::vector<ImageDev> t(c->slots.size());
        for (size_t i = 0; i < t.size(); ++i) t[i] = c->slots[i].dev;
        if (!t.empty())
            HIPCHK(hipMemcpy(c->d_imgs.p, t.data(), t.size() * sizeof(ImageDev),
                             hipMemcpyHostToDevice));
        HIPCHK(c->d_grids.ensure(c->slots.size()));
        std::vector<GridDev> gt(c->slots.size());
        for (size_t i = 0; i < gt.size(); ++i) gt[i] = c->slots[i].grid;
        if (!gt.empty())
            HIPCHK(hipMemcpy(c->d_grids.p, gt.data(), gt.size() * sizeof(GridDev), hipMemcpyHostToDevice));
        c->table_dirty = false;
    }

    const float max_ratio_f = (float)o.max_ratio;
    FinalizeParams fp;
    fp.max_ratio = max_ratio_f;
    fp.max_distance = (float)o.max_distance;
    fp.cross_check = o.cross_check ? 1 : 0;
    fp.reserved = 0;

    // the scan's accept thresholds for these options: built (and proven against the acos table) once per option pair
    if (!c->accept_valid || std::memcmp(&c->accept_ratio, &fp.max_ratio, sizeof(float)) != 0 ||
        std::memcmp(&c->accept_distance, &fp.max_distance, sizeof(float)) != 0) {
        c->h_accept = build_scan_accept(c->h_lut.data(), (uint32_t)c->h_lut.size(), fp.max_ratio, fp.max_distance);
        if (std::getenv("AMC_SCAN_ACCEPT_TRIVIAL")) c->h_accept.trivial = 1;  // (test hook: keep every row with best >= min_best)
        if (!c->d_accept) HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->d_accept), sizeof(ScanAccept)));
        HIPCHK(hipStreamSynchronize(st));  // a previous call's kernels may still read the old thresholds
        HIPCHK(hipMemcpy(c->d_accept, &c->h_accept, sizeof(ScanAccept), hipMemcpyHostToDevice));
        c->accept_ratio = fp.max_ratio;
        c->accept_distance = fp.max_distance;
        c->accept_valid = true;
    }
    HIPCHK(hipEventRecord(c->ev[0], st));
    // (everything above returns through HIPCHK: from here on errors go through rc / hc, which give the result's
    // pinned lease back)
    ResultPriv* priv = new (std::nothrow) ResultPriv();
    if (!priv) return fail(AMC_E_NOMEM, "amc_match_pairs: out of host memory");
    priv->offsets.assign(npairs + 1, 0);
    priv->pool = c->result_pool;
    priv->matches = c->result_pool->acquire();
    size_t keep_used = 0;  // matches of this call in c->d_keep so far (pair order: the result's CSR layout)
    c->resident_matches = 0;
    c->vres = amc::VerifyResident{};  // (a resident verification result indexes the table this call rewrites)
    if (keep_off) keep_off->assign(npairs, 0);

    const size_t mfma_max_cols = kSelectMaxCols;  // cross-check candidate bitmap (image 2 rows)

    uint64_t num_dist = 0, n_mfma = 0, n_dot4 = 0, n_grid = 0;
    double kernel_ms = 0.0, cross_ms = 0.0;
    uint32_t kernel_launches = 0;

    int rc = AMC_OK;
    auto hc = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == AMC_OK)
            rc = fail(AMC_E_HIP, "amc_match_pairs: %s: %s", what, hipGetErrorString(e));
        return e == hipSuccess;
    };
    // test hook: a smaller per-batch budget, so that small inputs exercise the multi-batch pipeline
    size_t max_entries = kMaxTop2Entries;
    if (const char* e = std::getenv("AMC_MATCH_BATCH_ENTRIES")) {
        const long long v = std::atoll(e);
        if (v > 0) max_entries = std::min<size_t>(kMaxTop2Entries, (size_t)v);
    }

    // A batch goes through four steps.  Steps of consecutive batches are interleaved so that the device
    // never waits for the host between them:
    //   prepare(k+1)   host only: route pairs to kernels, queue orders, staging set (k+1)&1   } while the device
    //   collect(k)     wait for batch k's counters, enqueue the copy of its matches            } runs batch k
    //   enqueue(k+1)   H2D + all kernels + the counters' D2H, behind that copy in stream order
    //   scatter(k)     wait for the matches, append them to the result                         (device runs k+1)
    struct Batch {
        size_t begin = 0, end = 0, nb = 0, top_rows = 0, top_cols = 0, cap = 0;
        size_t row_off = 0, nwork = 0, nord = 0, nwork_grid = 0;
        uint32_t max_cols = 0;  // largest image 2 among the batch's mfma pairs (select_candidates' bitmap)
        size_t ngrp = 0, ngrp2 = 0, seg_cap = 0;  // mfma: groups of the two queue orders, descriptors to provide for
        int set = 0;
        uint32_t total = 0;
        bool grouped_resolve = true;
    };
    // rows of image 1 (padded) from pair i to the end of the call: how much is left when a batch is carved
    // (a running total, not an array: a loop-closure call has 10^7 pairs, and 80 MB of suffix sums cost more than the
    // tail they shape)
    uint64_t rows_carved = 0, rows_collected = 0;
    bool even_batches = std::getenv("AMC_MATCH_EVEN_BATCHES") != nullptr;  // (A/B hook)
    if (batch_hook && batch_hook->plan) {
        // amc_match_verify_pairs with AMC_PIPELINE_INTERLEAVE=1: batch k's verification runs beside batch k + 1's scan, so (a) what is exposed is the LAST
        // batch's verification - more and equal batches make it small (the copy of the last batch's matches hides behind
        // it) - and (b) every batch but the first wants a scan long enough to hide a slice behind: up to six batches of
        // at least 32 Mi image-1 rows (an ~11 ms scan at 4,096 columns).  AMC_PIPELINE_BATCHES: A/B hook.
        size_t want = std::min<size_t>(6, std::max<size_t>(1, (size_t)(rows_total / ((uint64_t)32 << 20))));
        if (const char* e = std::getenv("AMC_PIPELINE_BATCHES")) want = (size_t)std::max(1, std::atoi(e));
        const size_t per = (size_t)((rows_total + want - 1) / want) + 4096;
        if (!std::getenv("AMC_MATCH_BATCH_ENTRIES")) max_entries = std::min(max_entries, std::max<size_t>(per, 1));
        even_batches = true;
    }
    // How a batch's matches reach the host.  The copy of batch k is handed to batch k + 1's forward scan, whose first
    // few workgroups carry it out (CopyJob, match_mfma.hip); the last batch's copy, and any the next launch cannot
    // take, goes to the copy stream as a small-grid kernel (launch_host_copy).  AMC_D2H=memcpy: hipMemcpyAsync for
    // all of them (A/B: its copy kernel takes every CU while PCIe moves the data, and the next scan waits);
    // AMC_D2H=stream: never fused.
    const char* d2h_env = std::getenv("AMC_D2H");
    const int d2h_mode = !d2h_env ? 0 : (std::strcmp(d2h_env, "memcpy") == 0 ? 2 : (std::strcmp(d2h_env, "stream") == 0 ? 1 : 0));
    struct PendingCopy {
        void* dst = nullptr;
        const void* src = nullptr;
        size_t bytes = 0;
        int set = -1;  // the batch set whose bev[set][4] marks the copy done; -1: nothing pending
    } pending;
    constexpr uint32_t kCopyParts = 8;
    // (smaller copies are not worth a scan's prologue; AMC_D2H_FUSE_MIN_BYTES lets the tests take the path with small inputs)
    const char* fmin_env = std::getenv("AMC_D2H_FUSE_MIN_BYTES");
    const size_t fuse_min_bytes = fmin_env ? (size_t)std::strtoull(fmin_env, nullptr, 10) : ((size_t)1 << 20);
    // The cross-check chain of batch k BESIDE the forward scan of batch k + 1 (round 6).  The scan's
    // workgroups take a CU's whole register file, so nothing shares a CU with them: the next scan is launched with
    // `chain_cus` workgroups fewer (a power-bound kernel: 8 of 256 CUs cost it ~1 %, profiles/r06/scan_grid_v1.txt) and
    // the chain's kernels - on chain_stream, behind the scan of their own batch by event - run on the CUs left over.
    // Needs the second set of batch tables (amc_ctx::ms).  AMC_MATCH_OVERLAP=0/1, AMC_CHAIN_CUS=n: A/B hooks.
    const char* ov_env = std::getenv("AMC_MATCH_OVERLAP");
    const bool overlap = c->chain_stream && !(batch_hook && batch_hook->plan) &&
                         (ov_env ? ov_env[0] == '1' : kMatchOverlapDefault);
    int chain_cus = 8;
    if (const char* e = std::getenv("AMC_CHAIN_CUS")) chain_cus = std::max(0, std::atoi(e));
    hipStream_t cs = overlap ? c->chain_stream : st;  // where a batch's chain, its reorder and its counters' download go
    auto sync_batch_streams = [&](const char* what) {
        return hc(hipStreamSynchronize(st), what) && (!overlap || hc(hipStreamSynchronize(cs), what));
    };
    const char* hs_env = std::getenv("AMC_HOOK_SPLIT");  // "0": off; "2": also calls of a few pairs (the tests' way to the path)
    const bool hook_split = batch_hook && batch_hook->submit && !batch_hook->plan && !(hs_env && hs_env[0] == '0');
    const size_t hook_split_min_pairs = (hs_env && hs_env[0] == '2') ? 2 : 4096;
    size_t first_div = kFirstBatchDiv;
    if (const char* e = std::getenv("AMC_MATCH_FIRST_DIV")) first_div = (size_t)std::max(0, std::atoi(e));
    auto carve = [&](size_t begin, int set) {
        Batch b;
        b.begin = b.end = begin;
        b.set = set;
        // The copy of a batch's matches to the host runs beside the NEXT batch's kernels; the last batch's copy has
        // nothing to hide behind.  So a call of several batches ends on a small one: when what is left would be the
        // last batch and is more than a quarter of a full one, this batch stops a quarter short of the end (on the
        // dense 500 x 4096 set the exposed copy is 530 MB otherwise).
        size_t limit = max_entries;
        const uint64_t rows_left = rows_total - rows_carved;  // (carve() is called for consecutive batches, in order)
        if (!even_batches && begin > 0 && rows_left <= max_entries && rows_left > max_entries / 4)
            limit = (size_t)(rows_left - max_entries / 4);
        // ... and begins on a small one: the device is idle while the host prepares the call's FIRST batch (queue orders
        // of 62 k pairs: 0.7 ms of a 177 ms step), the next batches' lists are made beside a scan.  AMC_MATCH_FIRST_DIV=d
        // (A/B hook): the first batch of a call of more than one full batch is 1 / d of a full one (0: off).
        if (!even_batches && begin == 0 && first_div > 1 && rows_total > max_entries) limit = max_entries / first_div;
        // amc_match_verify_pairs hands a batch's pairs to the verification's host side (checks, trial tables, class lists:
        // 3 - 25 ms for 33 k pairs) when the batch's counts are on the host - beside the NEXT batch's scan.  A call that
        // fits one batch has no next scan to hide that behind: it is cut in two (AMC_HOOK_SPLIT=0: A/B hook).
        if (!even_batches && begin == 0 && hook_split && rows_total <= max_entries && npairs >= hook_split_min_pairs)
            limit = (size_t)(rows_total * 6 / 10);
        while (b.end < npairs) {
            const Slot& x = c->slots[slot1[b.end]];
            const Slot& y = c->slots[slot2[b.end]];
            const size_t nr = x.dev.rows_pad, nc = y.dev.rows_pad;
            // cross-checked matches are one-to-one; without the cross check every row of image 1
            // may match (several rows may share a column)
            const size_t mc = o.cross_check ? std::min(x.dev.rows, y.dev.rows) : x.dev.rows;
            if (b.end > b.begin && (b.top_rows + nr > limit || b.top_cols + nc > max_entries ||
                                    b.cap + mc > kMaxMatchCap || b.end - b.begin >= (1u << 24)))
                break;
            b.top_rows += nr; b.top_cols += nc; b.cap += mc; ++b.end;
        }
        b.nb = b.end - b.begin;
        rows_carved += b.top_rows;
        return b;
    };
    // host side of a batch: which kernel takes each pair, the work queues (mfma: one item per pair, in
    // an order that keeps co-resident workgroups on the same streamed image; dot4: one item per 64 rows)
    auto prepare = [&](Batch& b) {
        const int k = b.set;
        const size_t nb = b.nb, begin = b.begin;
        if (!hc(c->h_pairs[k].ensure(nb), "pinned pairs") || !hc(c->h_order[k].ensure(nb), "pinned order") ||
            !hc(c->h_order2[k].ensure(nb), "pinned order2") || !hc(c->h_pair_off[k].ensure(nb), "pinned pair_off") ||
            !hc(c->h_pair_cnt[k].ensure(nb), "pinned pair_cnt") || !hc(c->h_bscalars[k].ensure(16), "pinned scalars"))
            return false;
        // mfma: exact for any u8 values and sizes; the lazy cross check's candidate bitmap
        // (select_candidates_kernel) holds kSelectMaxCols = 1 Mi image-2 rows, larger images take the dot4 path.
        std::vector<uint8_t> want_mfma(nb, 0);
        for (size_t i = 0; i < nb; ++i) {
            const Slot& x = c->slots[slot1[begin + i]];
            const Slot& y = c->slots[slot2[begin + i]];
            const bool nonempty = x.dev.rows > 0 && y.dev.rows > 0;
            want_mfma[i] = nonempty && o.kernel != AMC_KERNEL_DOT4 && !geoms &&
                           (!o.cross_check || y.dev.rows_pad <= mfma_max_cols);
        }
        PairDev* hp = c->h_pairs[k].p;
        size_t row_off = 0, col_off = 0, nwork = 0, nord = 0;
        b.grouped_resolve = std::getenv("AMC_RESOLVE_UNGROUPED") == nullptr;  // (test hook: the per-row kernel)
        for (size_t i = 0; i < nb; ++i) {
            const Slot& x = c->slots[slot1[begin + i]];
            const Slot& y = c->slots[slot2[begin + i]];
            const bool nonempty = x.dev.rows > 0 && y.dev.rows > 0;
            if (o.kernel == AMC_KERNEL_MFMA && nonempty && !want_mfma[i]) {
                rc = fail(AMC_E_INVALID,
                          "amc_match_pairs: kernel=MFMA forced but pair %zu is not eligible "
                          "(rows_pad=%u, cols_pad=%u > %zu)", begin + i, x.dev.rows_pad,
                          y.dev.rows_pad, mfma_max_cols);
                return false;
            }
            PairDev& pd = hp[i];
            pd.slot1 = slot1[begin + i];
            pd.slot2 = slot2[begin + i];
            pd.mode = want_mfma[i] ? 1u : 0u;
            pd.pad = 0;
            pd.row_off = row_off;
            pd.col_off = col_off;
            row_off += x.dev.rows_pad;
            col_off += y.dev.rows_pad;
            num_dist += (uint64_t)x.dev.rows * y.dev.rows;
            if (!nonempty) continue;
            if (want_mfma[i]) {
                c->h_order[k].p[nord++] = (uint32_t)i;
                ++n_mfma;
                b.max_cols = std::max(b.max_cols, y.dev.rows);
                // the tile-grouped resolve needs both images' tiles to fit its LDS histogram
                if (std::max(x.dev.rows_pad, y.dev.rows_pad) > resolve_grouped_max_rows()) b.grouped_resolve = false;
            } else {
                nwork += (x.dev.rows + 63) / 64;
                if (o.cross_check) nwork += (y.dev.rows + 63) / 64;
                if (geoms && h_guided[begin + i].grid_ok) ++n_grid; else ++n_dot4;
            }
        }
        // mfma queue orders: by (image 2, image 1) for the forward scan - co-resident workgroups stream the same Y - and
        // by (image 1, image 2) for the reverse scan.  Two stable counting sorts each (least significant key first):
        // O(pairs + slots) instead of a comparison sort through the pair array (this runs unhidden for the call's
        // first batch: 2.8 ms of a 180 ms call with std::stable_sort).
        if (nord) {
            const size_t nslots = c->slots.size();
            std::vector<uint32_t> cnt(nslots + 1), tmp(nord);
            auto by_slot = [&](const uint32_t* src, uint32_t* dst, bool key_is_slot2) {
                std::fill(cnt.begin(), cnt.end(), 0u);
                for (size_t q = 0; q < nord; ++q) ++cnt[(key_is_slot2 ? hp[src[q]].slot2 : hp[src[q]].slot1) + 1];
                for (size_t v = 0; v < nslots; ++v) cnt[v + 1] += cnt[v];
                for (size_t q = 0; q < nord; ++q) dst[cnt[key_is_slot2 ? hp[src[q]].slot2 : hp[src[q]].slot1]++] = src[q];
            };
            uint32_t* ord = c->h_order[k].p;
            by_slot(ord, tmp.data(), false);   // minor key: image 1
            by_slot(tmp.data(), ord, true);    // major key: image 2 (stable)
            if (o.cross_check) {
                uint32_t* ord2 = c->h_order2[k].p;
                by_slot(ord, tmp.data(), true);    // minor key: image 2
                by_slot(tmp.data(), ord2, false);  // major key: image 1
            }
        }
        // Cut both orders where the streamed image changes (the packing kernels fill whole items per image),
        // and bound the number of segment descriptors: ceil(rows / 128) per pair plus up to one item of padding
        // per group.  The reverse scan's X side is the candidate list, at most every row of image 2.
        b.ngrp = b.ngrp2 = 0;
        b.seg_cap = 0;
        if (nord) {
            if (!hc(c->h_grp[k].ensure(nord + 1), "pinned group cuts") ||
                (o.cross_check && !hc(c->h_grp2[k].ensure(nord + 1), "pinned group cuts")))
                return false;
            size_t seg1 = 0, seg2 = 0;
            for (size_t q = 0; q < nord; ++q) {
                const PairDev& pq = hp[c->h_order[k].p[q]];
                if (q == 0 || pq.slot2 != hp[c->h_order[k].p[q - 1]].slot2) c->h_grp[k].p[b.ngrp++] = (uint32_t)q;
                seg1 += (c->slots[pq.slot1].dev.rows + kSegRows - 1) / kSegRows;
                seg2 += (c->slots[pq.slot2].dev.rows + kSegRows - 1) / kSegRows;
            }
            c->h_grp[k].p[b.ngrp] = (uint32_t)nord;
            if (o.cross_check) {
                for (size_t q = 0; q < nord; ++q)
                    if (q == 0 || hp[c->h_order2[k].p[q]].slot1 != hp[c->h_order2[k].p[q - 1]].slot1)
                        c->h_grp2[k].p[b.ngrp2++] = (uint32_t)q;
                c->h_grp2[k].p[b.ngrp2] = (uint32_t)nord;
            }
            b.seg_cap = std::max(seg1 + kSegsPerItem * b.ngrp, o.cross_check ? seg2 + kSegsPerItem * b.ngrp2 : 0);
        }
        b.nwork_grid = 0;
        if (nwork) {
            if (!hc(c->h_work[k].ensure(nwork), "pinned work")) return false;
            size_t w = 0;
            // guided pairs the candidate-generation kernel takes come first: one launch per kernel over its part
            for (int pass = geoms ? 0 : 1; pass < 2; ++pass) {
                for (size_t i = 0; i < nb; ++i) {
                    if (hp[i].mode) continue;
                    if (geoms && (h_guided[begin + i].grid_ok != 0) != (pass == 0)) continue;
                    const Slot& x = c->slots[slot1[begin + i]];
                    const Slot& y = c->slots[slot2[begin + i]];
                    if (x.dev.rows == 0 || y.dev.rows == 0) continue;
                    for (uint32_t rb = 0; rb < (x.dev.rows + 63) / 64; ++rb)
                        c->h_work[k].p[w++] = Dot4Work{(uint32_t)i, 0u, rb};
                    if (o.cross_check)
                        for (uint32_t rb = 0; rb < (y.dev.rows + 63) / 64; ++rb)
                            c->h_work[k].p[w++] = Dot4Work{(uint32_t)i, 1u, rb};
                }
                if (pass == 0) b.nwork_grid = w;
            }
        }
        b.row_off = row_off;
        b.nwork = nwork;
        b.nord = nord;
        return true;
    };
    // a batch table's copy: the runtime's asynchronous copy, or - with the chain on its own stream - a kernel: the
    // runtime's copies of all streams share one in-order DMA queue, and a copy of the chain's stream waiting for its
    // kernels held the next scan's uploads behind it (launch_copy_words, match_common.hip)
    auto tcopy = [&](void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
        return overlap ? launch_copy_words(dst, src, bytes, s) : hipMemcpyAsync(dst, src, bytes, kind, s);
    };
    // device side of a batch, first part, on the stream: H2D of the queues, segment packing, the scans
    auto enqueue_scan = [&](Batch& b, int leave_cus = 0) {
        const int k = b.set;
        amc_ctx::MatchScratch& S = c->ms[overlap ? k : 0];
        const size_t nb = b.nb, nord = b.nord, nwork = b.nwork;
        // device scratch only ever grows; growing frees the old allocation, so drain the stream first
        const bool grow = S.d_pairs.cap < nb || S.d_order.cap < nb || S.d_order2.cap < nb ||
                          S.d_rowbuf.cap < b.top_rows || S.d_colbuf.cap < b.top_cols ||
                          S.d_accmask.cap < b.top_rows / 32 + 8 || S.d_pair_off.cap < nb || S.d_pair_cnt.cap < nb ||
                          S.d_matches.cap < 2 * b.cap || S.d_cand_cnt.cap < nb || S.d_candbuf.cap < b.top_cols ||
                          S.d_work.cap < nwork || (geoms && S.d_guided.cap < nb) || S.d_segs.cap < b.seg_cap ||
                          S.d_seg_base.cap < nord || S.d_grp.cap < b.ngrp + 1 || S.d_grp2.cap < b.ngrp2 + 1 ||
                          S.d_grp_segs.cap < std::max(b.ngrp, b.ngrp2) || S.d_grp_item_base.cap < std::max(b.ngrp, b.ngrp2);
        if (grow && !sync_batch_streams("sync before growing device scratch")) return false;
        // this set's tables are free when the chain and the reorder of the batch that used them last are done
        if (overlap && !hc(hipStreamWaitEvent(st, c->sev[k], 0), "stream wait")) return false;
        if (!hc(S.d_pairs.ensure(nb), "dev pairs") || !hc(S.d_order.ensure(nb), "dev order") ||
            !hc(S.d_order2.ensure(nb), "dev order2") ||
            !hc(S.d_rowbuf.ensure(b.top_rows), "row top2") || !hc(S.d_colbuf.ensure(b.top_cols), "col top2") ||
            !hc(S.d_accmask.ensure(b.top_rows / 32 + 8), "accept mask") ||
            !hc(S.d_pair_off.ensure(nb), "pair_off") || !hc(S.d_pair_cnt.ensure(nb), "pair_cnt") ||
            !hc(S.d_matches.ensure(2 * b.cap), "dev matches") ||
            !hc(S.d_cand_cnt.ensure(nb), "cand_cnt") || !hc(S.d_candbuf.ensure(b.top_cols), "candbuf") ||
            (nwork && !hc(S.d_work.ensure(nwork), "dev work")) || (geoms && !hc(S.d_guided.ensure(nb), "dev guided")) ||
            !hc(S.d_segs.ensure(b.seg_cap), "segment descriptors") || !hc(S.d_seg_base.ensure(nord), "segment bases") ||
            !hc(S.d_grp.ensure(b.ngrp + 1), "group cuts") || !hc(S.d_grp2.ensure(b.ngrp2 + 1), "group cuts") ||
            !hc(S.d_grp_segs.ensure(std::max(b.ngrp, b.ngrp2)), "group segments") ||
            !hc(S.d_grp_item_base.ensure(std::max(b.ngrp, b.ngrp2)), "group items"))
            return false;
        bool okq = hc(tcopy(S.d_pairs.p, c->h_pairs[k].p, nb * sizeof(PairDev),
                                     hipMemcpyHostToDevice, st), "H2D pairs") &&
                   hc(memset_async(S.scalars, 0, 2 * sizeof(uint32_t), st), "memset cursor") &&
                   hc(memset_async(S.scalars + 3, 0, sizeof(uint32_t), st), "memset errcount");
        if (okq && nord)
            okq = hc(tcopy(S.d_order.p, c->h_order[k].p, nord * sizeof(uint32_t),
                                    hipMemcpyHostToDevice, st), "H2D order") &&
                  hc(tcopy(S.d_grp.p, c->h_grp[k].p, (b.ngrp + 1) * sizeof(uint32_t),
                                    hipMemcpyHostToDevice, st), "H2D group cuts") &&
                  // segments no wave owns (beyond an image's last row) never write their words
                  hc(memset_async(S.d_accmask.p, 0, (b.row_off / 32 + 8) * sizeof(uint32_t), st), "memset accmask");
        if (okq && nwork)
            okq = hc(tcopy(S.d_work.p, c->h_work[k].p, nwork * sizeof(Dot4Work),
                                    hipMemcpyHostToDevice, st), "H2D work");
        if (okq && geoms)  // this batch's slice of the filter models (pageable source: the copy is staged)
            okq = hc(hipMemcpyAsync(S.d_guided.p, h_guided.data() + b.begin, nb * sizeof(GuidedDev),
                                    hipMemcpyHostToDevice, st), "H2D guided");
        if (!okq) return false;
        if (nord &&  // pack the pairs' 128-row segments into items (per streamed image) ...
            !hc(launch_build_segments(0, c->d_imgs.p, S.d_pairs.p, S.d_order.p, S.d_grp.p, (uint32_t)b.ngrp,
                                      S.d_cand_cnt.p, S.d_candbuf.p, S.d_rowbuf.p, S.d_seg_base.p, S.d_grp_segs.p,
                                      S.d_grp_item_base.p, S.d_segs.p, S.scalars + 5, st), "segment packing"))
            return false;
        if (!hc(hipEventRecord(c->bev[k][0], st), "event record")) return false;
        if (nord) {  // ... and scan them (the events bracket the scan kernel alone: bench.py's roofline leg)
            CopyJob job;
            const uintptr_t ps = reinterpret_cast<uintptr_t>(pending.src), pd = reinterpret_cast<uintptr_t>(pending.dst);
            const bool take = pending.set >= 0 && d2h_mode == 0 && b.seg_cap > 0 && pending.bytes >= fuse_min_bytes &&
                              (ps & 15) == (pd & 15);
            int done_set = -1;
            if (take) {  // the previous batch's matches ride in this launch; head / tail bytes around the 16-byte units first
                const size_t head = (ps & 15) ? 16 - (ps & 15) : 0, n16 = (pending.bytes - head) / 16;
                const size_t tail = pending.bytes - head - n16 * 16;
                if (head && !hc(memcpy_async(pending.dst, pending.src, head, hipMemcpyDeviceToHost, st), "D2H matches (head)"))
                    return false;
                if (tail && !hc(memcpy_async(static_cast<char*>(pending.dst) + head + n16 * 16,
                                             static_cast<const char*>(pending.src) + head + n16 * 16, tail,
                                             hipMemcpyDeviceToHost, st), "D2H matches (tail)"))
                    return false;
                job.src = static_cast<const char*>(pending.src) + head;
                job.dst = static_cast<char*>(pending.dst) + head;
                job.n16 = n16;
                job.parts = kCopyParts;
                done_set = pending.set;
                pending.set = -1;
            }
            if (!hc(launch_match_mfma(0, S.d_segs.p, S.scalars + 5, (uint32_t)std::min<size_t>(b.seg_cap, 0xFFFFFFFFu),
                                      S.scalars + 1, S.d_accmask.p, c->d_accept, st, job, S.scalars + 7, leave_cus), "forward scan"))
                return false;
            // that batch's matches are on the host when this scan is done
            if (done_set >= 0 && !hc(hipEventRecord(c->bev[done_set][4], st), "event record")) return false;
        }
        if (b.nwork_grid &&
            !hc(launch_match_guided_grid(c->d_imgs.p, c->d_grids.p, S.d_pairs.p, S.d_work.p, (uint32_t)b.nwork_grid,
                                         S.d_rowbuf.p, S.d_colbuf.p, S.d_guided.p, st), "guided scan"))
            return false;
        if (nwork > b.nwork_grid &&
            !hc(launch_match_dot4(c->d_imgs.p, S.d_pairs.p, S.d_work.p + b.nwork_grid, (uint32_t)(nwork - b.nwork_grid),
                                  S.d_rowbuf.p, S.d_colbuf.p, geoms ? S.d_guided.p : nullptr, st), "dot4 scan"))
            return false;
        kernel_launches += (nord ? 1 : 0) + (b.nwork_grid ? 1 : 0) + (nwork > b.nwork_grid ? 1 : 0);
        return hc(hipEventRecord(c->bev[k][1], st), "event record");
    };
    // second part, on the chain's stream (the same stream unless the chain runs beside the next scan): tile -> index,
    // lazy cross check, finalize, D2H of the counters
    auto enqueue_chain = [&](Batch& b) {
        const int k = b.set;
        amc_ctx::MatchScratch& S = c->ms[overlap ? k : 0];
        const size_t nb = b.nb, nord = b.nord;
        hipStream_t st = cs;  // (everything below is the chain)
        if (overlap && !hc(hipStreamWaitEvent(st, c->bev[k][1], 0), "stream wait")) return false;
        const bool use_order = std::getenv("AMC_RESOLVE_PAIR_ORDER") == nullptr;  // (A/B hook: workgroups in batch order)
        if (nord &&  // tile -> exact index for the accepted rows
            !hc(launch_resolve_index(0, c->d_imgs.p, S.d_pairs.p, (uint32_t)nb, S.d_rowbuf.p, S.d_accmask.p, c->d_lut,
                                     fp, S.d_cand_cnt.p, S.d_candbuf.p, S.scalars + 3, b.grouped_resolve,
                                     use_order ? S.d_order.p : nullptr, (uint32_t)nord, st), "resolve (rows)"))
            return false;
        if (nord && o.cross_check) {
            // lazy cross check: reverse scan only for the columns accepted rows point at
            if (!hc(launch_select_candidates(c->d_imgs.p, S.d_pairs.p, (uint32_t)nb, b.max_cols, S.d_rowbuf.p,
                                             S.d_accmask.p, c->d_lut, fp, S.d_cand_cnt.p, S.d_candbuf.p, st),
                    "candidate selection"))
                return false;
            if (!hc(tcopy(S.d_order2.p, c->h_order2[k].p, nord * sizeof(uint32_t),
                                   hipMemcpyHostToDevice, st), "H2D order2") ||
                !hc(tcopy(S.d_grp2.p, c->h_grp2[k].p, (b.ngrp2 + 1) * sizeof(uint32_t),
                                   hipMemcpyHostToDevice, st), "H2D group cuts"))
                return false;
            // the candidate counts exist only on the device: the packing kernels read them there
            if (!hc(launch_build_segments(1, c->d_imgs.p, S.d_pairs.p, S.d_order2.p, S.d_grp2.p, (uint32_t)b.ngrp2,
                                          S.d_cand_cnt.p, S.d_candbuf.p, S.d_colbuf.p, S.d_seg_base.p, S.d_grp_segs.p,
                                          S.d_grp_item_base.p, S.d_segs.p, S.scalars + 5, st), "segment packing (reverse)") ||
                !hc(launch_match_mfma(1, S.d_segs.p, S.scalars + 5, (uint32_t)std::min<size_t>(b.seg_cap, 0xFFFFFFFFu),
                                      S.scalars + 1, S.d_accmask.p, c->d_accept, st), "reverse scan") ||
                !hc(launch_resolve_index(1, c->d_imgs.p, S.d_pairs.p, (uint32_t)nb, S.d_colbuf.p, S.d_accmask.p, c->d_lut,
                                         fp, S.d_cand_cnt.p, S.d_candbuf.p, S.scalars + 3, b.grouped_resolve,
                                         use_order ? S.d_order2.p : nullptr, (uint32_t)nord, st), "resolve (columns)"))
                return false;
        }
        if (!hc(hipEventRecord(c->bev[k][2], st), "event record") ||
            !hc(launch_finalize(c->d_imgs.p, S.d_pairs.p, (uint32_t)nb, S.d_rowbuf.p, S.d_colbuf.p,
                                S.d_accmask.p, c->d_lut, fp, S.scalars, (uint32_t)std::min(b.cap, (size_t)0xFFFFFFFFu),
                                S.d_pair_off.p, S.d_pair_cnt.p, S.d_matches.p, st), "finalize"))
            return false;
        return hc(tcopy(c->h_bscalars[k].p, S.scalars, 4 * sizeof(uint32_t),
                                 hipMemcpyDeviceToHost, st), "D2H cursor") &&
               hc(tcopy(c->h_pair_off[k].p, S.d_pair_off.p, nb * sizeof(uint32_t),
                                 hipMemcpyDeviceToHost, st), "D2H pair_off") &&
               hc(tcopy(c->h_pair_cnt[k].p, S.d_pair_cnt.p, nb * sizeof(uint32_t),
                                 hipMemcpyDeviceToHost, st), "D2H pair_cnt") &&
               hc(hipEventRecord(c->bev[k][3], st), "event record");
    };
    // the batch's counters are on the host: check them, enqueue the copy of exactly `total` matches
    auto collect = [&](Batch& b) {
        const int k = b.set;
        amc_ctx::MatchScratch& S = c->ms[overlap ? k : 0];
        hipStream_t st = cs;  // (the reorder follows the chain)
        if (!hc(hipEventSynchronize(c->bev[k][3]), "wait for the batch")) return false;
        b.total = c->h_bscalars[k].p[0];
        if (c->h_bscalars[k].p[3] != 0) {
            rc = fail(AMC_E_HIP, "amc_match_pairs: internal: %u accepted rows could not be resolved "
                      "to an index (scan/recompute mismatch)", c->h_bscalars[k].p[3]);
            return false;
        }
        if (b.total > b.cap) {
            rc = fail(AMC_E_HIP, "amc_match_pairs: internal: %u matches exceed capacity %zu", b.total, b.cap);
            return false;
        }
        // The batch's matches lie in d_matches in the order the workgroups claimed space (atomic cursor).  Put them
        // in pair order behind the batches before it in d_keep - the CSR layout of the result - and copy that
        // straight into the result's pinned buffer: no per-pair scatter on the host, and amc_match_verify_pairs
        // reads the same table.
        if (!hc(c->h_csr[k].ensure(b.nb), "pinned csr")) return false;
        uint64_t run = keep_used;
        for (size_t i = 0; i < b.nb; ++i) {
            c->h_csr[k].p[i] = run;
            if (keep_off) (*keep_off)[b.begin + i] = run;
            run += c->h_pair_cnt[k].p[i];
            priv->offsets[b.begin + i + 1] = run;  // the result's CSR (offsets[0] = 0; batches are collected in order)
        }
        if (run - keep_used != b.total) {
            rc = fail(AMC_E_HIP, "amc_match_pairs: internal: pair counts (%llu) disagree with the cursor (%u)",
                      (unsigned long long)(run - keep_used), b.total);
            return false;
        }
        rows_collected += b.top_rows;
        if (b.total) {
            const auto tgrow = std::chrono::steady_clock::now();
            const size_t need = 2 * (keep_used + (size_t)b.total);
            // what the whole call will need if the batches to come match like the ones so far (+ 10 %): a table that has to
            // grow is sized for that at once - three batches otherwise pin (and copy) 2.4 times the final result
            // (at most four times what is needed now, and what is needed now if the larger request fails)
            size_t want = need;
            if (rows_collected > 0 && rows_collected < rows_total) {
                const double est = std::min((double)need * ((double)rows_total / (double)rows_collected) * 1.1, 4.0 * (double)need);
                want = std::max(need, (size_t)est / 2 * 2);
            }
            if (need > c->d_keep.cap) {
                DevBuf<uint32_t> bigger;
                if (bigger.ensure(std::max(want, 2 * c->d_keep.cap)) != hipSuccess) {
                    (void)hipGetLastError();
                    if (!hc(bigger.ensure(need), "resident match table")) return false;
                }
                if (keep_used &&
                    !hc(hipMemcpyAsync(bigger.p, c->d_keep.p, 2 * keep_used * sizeof(uint32_t), hipMemcpyDeviceToDevice, st),
                        "move resident match table"))
                    return false;
                if (!sync_batch_streams("sync before freeing the old resident table") ||
                    !hc(hipStreamSynchronize(c->copy_stream), "sync before freeing the old resident table"))
                    return false;
                if (batch_hook) verify_streams_sync(c);  // (verification slices of earlier batches read the old table)
                c->d_keep.release();
                c->d_keep = bigger;
            }
            if (need > priv->matches.cap) {  // grow the result buffer (first calls only: the pool keeps it)
                if (!sync_batch_streams("sync before growing the result buffer") ||
                    !hc(hipStreamSynchronize(c->copy_stream), "sync before growing the result buffer"))
                    return false;
                PinBuf<uint32_t> bigger;
                if (bigger.ensure(std::max(want, 2 * priv->matches.cap)) != hipSuccess) {
                    (void)hipGetLastError();
                    if (prof) std::fprintf(stderr, "[amc match profile] pinned result: %zu words refused, asking for %zu\n", want, need);
                    if (!hc(bigger.ensure(need), "pinned result")) return false;
                }
                if (keep_used) std::memcpy(bigger.p, priv->matches.p, 2 * keep_used * sizeof(uint32_t));
                priv->matches.release();
                priv->matches = bigger;
            }
            if (prof && since(tgrow) > 5.0)
                std::fprintf(stderr, "[amc match profile] batch of %zu pairs: %.1f ms growing the result tables to %zu words\n", b.nb,
                             since(tgrow), priv->matches.cap);
            if (!hc(c->d_csr.ensure(b.nb), "dev csr") ||
                !hc(tcopy(c->d_csr.p, c->h_csr[k].p, b.nb * sizeof(uint64_t), hipMemcpyHostToDevice, st), "H2D csr"))
                return false;
            // the copy to the host happens beside the next batch's kernels (which write d_matches and, later, d_keep
            // beyond this batch - never what is being copied): flush_copy() or the next enqueue() issues it
            if (!hc(launch_reorder_matches(S.d_pair_off.p, S.d_pair_cnt.p, c->d_csr.p, (uint32_t)b.nb, S.d_matches.p,
                                           c->d_keep.p, st), "reorder launch"))
                return false;
            if (batch_hook && !hc(hipEventRecord(c->kev[k], st), "event record")) return false;
            if (overlap && !hc(hipEventRecord(c->sev[k], st), "event record")) return false;
            pending.dst = priv->matches.p + 2 * keep_used;
            pending.src = c->d_keep.p + 2 * keep_used;
            pending.bytes = (size_t)b.total * 2 * sizeof(uint32_t);
            pending.set = k;
            keep_used += b.total;
            return true;
        }
        if (overlap && !hc(hipEventRecord(c->sev[k], st), "event record")) return false;
        return hc(hipEventRecord(c->bev[k][4], st), "event record");
    };
    double t_hook = 0.0;
    int cus_free = 0;  // what the scan being enqueued leaves to the hook's launches
    auto plan_hook = [&](const Batch& b, const Batch* next) {
        cus_free = 0;
        if (!batch_hook || !batch_hook->plan) return;
        double next_ms = 0.0;
        if (next) {  // the next batch's forward scan at the rate this kernel holds (~1.2e13 distances/s)
            double nd = 0.0;
            for (size_t i = next->begin; i < next->end; ++i) nd += (double)c->slots[slot1[i]].dev.rows * (double)c->slots[slot2[i]].dev.rows;
            next_ms = nd / 1.2e10;
        }
        cus_free = batch_hook->plan(b.begin, b.end, priv->offsets.data(), next_ms);
    };
    auto run_hook = [&](const Batch& b) {
        if (!batch_hook || !batch_hook->submit) return true;
        const auto th = std::chrono::steady_clock::now();
        const int hrc = batch_hook->submit(b.begin, b.end, priv->offsets.data(), keep_off ? keep_off->data() : nullptr,
                                           b.total ? c->kev[b.set] : nullptr, cus_free);
        t_hook += since(th);
        if (hrc != AMC_OK && rc == AMC_OK) rc = hrc;  // (the hook has set the message)
        return hrc == AMC_OK;
    };
    // the pending copy on the copy stream (the last batch's, or one the next launch does not take)
    auto flush_copy = [&]() {
        if (pending.set < 0) return true;
        const int k = pending.set;
        pending.set = -1;
        if (!hc(hipEventRecord(c->cev[k], cs), "event record") || !hc(hipStreamWaitEvent(c->copy_stream, c->cev[k], 0), "stream wait"))
            return false;
        if (d2h_mode == 2) {
            if (!hc(hipMemcpyAsync(pending.dst, pending.src, pending.bytes, hipMemcpyDeviceToHost, c->copy_stream), "D2H matches"))
                return false;
        } else {
            if (!hc(launch_host_copy(pending.dst, pending.src, pending.bytes, c->copy_stream), "D2H matches")) return false;
        }
        return hc(hipEventRecord(c->bev[k][4], c->copy_stream), "event record");
    };
    // append the batch's matches to the result CSR (pairs keep the caller's order)
    auto scatter = [&](Batch& b) {
        const int k = b.set;
        if (!hc(hipEventSynchronize(c->bev[k][4]), "wait for the matches")) return false;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->bev[k][0], c->bev[k][1]) == hipSuccess) kernel_ms += ms;
        if (hipEventElapsedTime(&ms, c->bev[k][1], c->bev[k][2]) == hipSuccess) cross_ms += ms;
        return true;  // (the offsets were filled by collect(): this set's pinned counts may be gone by now)
    };

    if (npairs > 0) {
        Batch cur = carve(0, 0);
        auto tp = std::chrono::steady_clock::now();
        bool ok = prepare(cur);
        t_prepare += since(tp);
        tp = std::chrono::steady_clock::now();
        ok = ok && enqueue_scan(cur) && enqueue_chain(cur);
        t_enqueue += since(tp);
        // The host runs one batch ahead of the device: while batch `cur` is scanned, the next batch's lists are
        // prepared; the matches of the batch BEFORE cur are read out (scatter) only after that - their copy rides in
        // cur's scan and is done when that scan is, and waiting for it earlier would leave the device idle while the
        // host prepares (a 10^7-pair loop-closure call: 8 ms of preparation per batch against a 3 ms cross-check stage).
        Batch prev;
        bool have_prev = false;
        while (ok) {
            Batch next;
            const bool have_next = cur.end < npairs;
            if (have_next) {
                tp = std::chrono::steady_clock::now();
                next = carve(cur.end, cur.set ^ 1);
                ok = prepare(next);
                t_prepare += since(tp);
            }
            tp = std::chrono::steady_clock::now();
            if (ok && have_prev) ok = scatter(prev);  // (before enqueue(next) re-records that set's events)
            t_scatter += since(tp);
            // (two sets: the next scan goes out BEFORE the wait for this batch's counters - it runs right behind this
            //  batch's scan, on all CUs but the few its chain gets)
            tp = std::chrono::steady_clock::now();
            if (ok && have_next && overlap) ok = enqueue_scan(next, chain_cus);
            t_enqueue += since(tp);
            tp = std::chrono::steady_clock::now();
            ok = ok && collect(cur);
            t_collect += since(tp);
            tp = std::chrono::steady_clock::now();
            if (ok) plan_hook(cur, have_next ? &next : nullptr);
            if (ok && have_next) ok = (overlap || enqueue_scan(next, cus_free)) && enqueue_chain(next);
            ok = ok && flush_copy();  // (not taken by a scan launch: the last batch's, a small one, dot4-only batches)
            t_enqueue += since(tp);
            ok = ok && run_hook(cur);  // (the device is busy with `next` - or, for the last batch, with the copy)
            if (!have_next) {
                tp = std::chrono::steady_clock::now();
                ok = ok && scatter(cur);
                t_scatter += since(tp);
                break;
            }
            prev = cur;
            have_prev = true;
            cur = next;
        }
        if (!ok && rc == AMC_OK) rc = fail(AMC_E_HIP, "amc_match_pairs: batch failed");
        if (rc != AMC_OK) {  // nothing of this call stays in flight
            (void)hipStreamSynchronize(st);
            if (overlap) (void)hipStreamSynchronize(cs);
            (void)hipStreamSynchronize(c->copy_stream);
        }
    }
    // device_ms ends with the last result byte on the host: the stream joins the copy stream first
    if (rc == AMC_OK && npairs > 0 &&
        hc(hipEventRecord(c->cev[0], c->copy_stream), "event record"))
        hc(hipStreamWaitEvent(st, c->cev[0], 0), "stream wait");
    if (rc == AMC_OK && npairs > 0 && overlap && hc(hipEventRecord(c->cev[1], cs), "event record"))
        hc(hipStreamWaitEvent(st, c->cev[1], 0), "stream wait");  // (and the chain's stream)
    if (rc == AMC_OK && hc(hipEventRecord(c->ev[1], st), "event record")) hc(hipEventSynchronize(c->ev[1]), "wait for the call");
    if (rc != AMC_OK) {
        delete priv;
        return rc;
    }
    float total_ms = 0.f;
    (void)hipEventElapsedTime(&total_ms, c->ev[0], c->ev[1]);

    out->npairs = npairs;
    c->resident_matches = keep_used;
    out->offsets = priv->offsets.data();
    out->matches = priv->offsets[npairs] ? priv->matches.p : nullptr;
    out->num_distances = num_dist;
    out->pairs_mfma = n_mfma;
    out->pairs_dot4 = n_dot4;
    out->pairs_guided_grid = n_grid;
    out->device_ms = total_ms;
    out->match_kernel_ms = kernel_ms;
    out->match_kernel_launches = kernel_launches;
    out->cross_kernel_ms = cross_ms;
    out->_priv = priv;
    c->last_hook_ms = t_hook;
    if (prof)
        std::fprintf(stderr, "[amc match profile] pairs=%zu wall=%.1f ms: prepare %.1f, enqueue %.1f, collect(wait+reorder+D2H enqueue) %.1f, "
                     "scatter(wait) %.1f, batch hook %.1f; device events %.1f ms (scan %.1f, cross %.1f)\n", npairs, since(wall0), t_prepare,
                     t_enqueue, t_collect, t_scatter, t_hook, (double)total_ms, kernel_ms, cross_ms);
    return AMC_OK;
}

int amc_match_pairs(amc_ctx* c, const uint32_t* slot1, const uint32_t* slot2, size_t npairs,
                    const amc_match_opts* opts_in, amc_match_result* out) {
    return match_impl(c, slot1, slot2, npairs, opts_in, nullptr, 0.0, out);
}

int amc_match_guided_pairs(amc_ctx* c, const uint32_t* slot1, const uint32_t* slot2, size_t npairs,
                           const amc_tvg* geoms, double max_error, const amc_match_opts* opts_in,
                           amc_match_result* out) {
    if (npairs > 0 && !geoms) {
        if (out) std::memset(out, 0, sizeof *out);
        return fail(AMC_E_INVALID, "amc_match_guided_pairs: NULL geometries");
    }
    if (!(max_error >= 0.0)) {
        if (out) std::memset(out, 0, sizeof *out);
        return fail(AMC_E_INVALID, "amc_match_guided_pairs: max_error must be >= 0");
    }
    static const amc_tvg kNone{};
    return match_impl(c, slot1, slot2, npairs, opts_in, npairs ? geoms : &kNone, max_error, out);
}

void amc_match_result_free(amc_match_result* r) {
    if (!r) return;
    delete static_cast<ResultPriv*>(r->_priv);
    std::memset(r, 0, sizeof *r);
}

// ------------------------------------------------------------------------------------------------
// two-view verification
// ------------------------------------------------------------------------------------------------
void amc_tvg_opts_default(amc_tvg_opts* o) {
    if (!o) return;
    o->min_num_inliers = 15;          // TwoViewGeometryOptions C++ defaults, SURVEY.md A.3
    o->detect_watermark = 1;
    o->multiple_ignore_watermark = 1;
    o->force_H_use = 0;
    o->compute_relative_pose = 0;
    o->multiple_models = 0;
    o->min_E_F_inlier_ratio = 0.95;
    o->max_H_inlier_ratio = 0.8;
    o->watermark_min_inlier_ratio = 0.7;
    o->watermark_border_size = 0.1;
    o->ransac.max_error = 4.0;
    o->ransac.min_inlier_ratio = 0.25;
    o->ransac.confidence = 0.999;
    o->ransac.dyn_num_trials_multiplier = 3.0;
    o->ransac.min_num_trials = 100;
    o->ransac.max_num_trials = 10000;
}

// Guided matching's candidate generation (match_guided.hip): bucket the image's keypoints on a kGridDim^2 grid
// over their bounding box.  No grid (grid.n = 0) when a coordinate is not finite: such pairs take the dense kernel.
static int build_keypoint_grid(amc_ctx* c, Slot& s, const float* xy, uint32_t rows) {
    guided::GridGeom gg;
    std::vector<uint32_t> sidx, start;
    if (!guided::build_grid(xy, rows, gg, sidx, start)) return AMC_OK;
    GridDev g{};
    g.x0 = gg.x0; g.y0 = gg.y0; g.cw = gg.cw; g.ch = gg.ch; g.inv_cw = gg.inv_cw; g.inv_ch = gg.inv_ch;
    g.bx1 = gg.bx1; g.by1 = gg.by1;
    g.n = rows;
    std::vector<float> sxy((size_t)rows * 2);
    for (uint32_t k = 0; k < rows; ++k) {
        sxy[2 * (size_t)k] = xy[2 * (size_t)sidx[k]];
        sxy[2 * (size_t)k + 1] = xy[2 * (size_t)sidx[k] + 1];
    }
    const size_t b_xy = sxy.size() * sizeof(float), b_idx = sidx.size() * sizeof(uint32_t), b_st = start.size() * sizeof(uint32_t);
    hipError_t e = c->arena.alloc(&s.grid_base, b_xy + b_idx + b_st);
    if (e != hipSuccess) return fail(AMC_E_NOMEM, "amc_upload_keypoints: hipMalloc (grid): %s", hipGetErrorString(e));
    char* base = static_cast<char*>(s.grid_base);
    HIPCHK(hipMemcpy(base, sxy.data(), b_xy, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(base + b_xy, sidx.data(), b_idx, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(base + b_xy + b_idx, start.data(), b_st, hipMemcpyHostToDevice));
    g.sxy = reinterpret_cast<const float*>(base);
    g.sidx = reinterpret_cast<const uint32_t*>(base + b_xy);
    g.cell_start = reinterpret_cast<const uint32_t*>(base + b_xy + b_idx);
    s.grid = g;
    return AMC_OK;
}

int amc_upload_keypoints(amc_ctx* c, uint32_t slot, const float* xy, uint32_t rows,
                         uint32_t stride_floats) {
    if (!c) return fail(AMC_E_INVALID, "amc_upload_keypoints: ctx is NULL");
    if (slot >= c->slots.size())
        return fail(AMC_E_INVALID, "amc_upload_keypoints: slot %u >= reserved %zu", slot, c->slots.size());
    if (rows > 0 && (!xy || stride_floats < 2))
        return fail(AMC_E_INVALID, "amc_upload_keypoints: need x,y columns (stride %u) and data", stride_floats);
    HIPCHK(hipSetDevice(c->device));
    Slot& s = c->slots[slot];
    if (s.kp || s.kp64 || s.kpn || s.grid_base) {
        HIPCHK(hipStreamSynchronize(c->stream));
        c->arena.free(s.kp);
        c->arena.free(s.kp64);
        c->arena.free(s.kpn);
        c->arena.free(s.grid_base);
        s.grid_base = nullptr;
        s.grid = GridDev{};
        s.kp = nullptr;
        s.kp64 = nullptr;
        s.kpn = nullptr;
        s.kpn_valid = false;
        s.dev.kp = nullptr;
        s.dev.kp_rows = 0;
        c->table_dirty = true;
    }
    s.kp_rows = rows;
    s.has_kp = true;
    if (rows == 0) return AMC_OK;
    std::vector<float> packed((size_t)rows * 2);
    for (uint32_t i = 0; i < rows; ++i) {
        packed[2 * (size_t)i] = xy[(size_t)i * stride_floats];
        packed[2 * (size_t)i + 1] = xy[(size_t)i * stride_floats + 1];
    }
    hipError_t e = c->arena.alloc(&s.kp, packed.size() * sizeof(float));
    if (e != hipSuccess) {
        s.has_kp = false;
        return fail(AMC_E_NOMEM, "amc_upload_keypoints: hipMalloc: %s", hipGetErrorString(e));
    }
    HIPCHK(hipMemcpy(s.kp, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
    s.dev.kp = s.kp;  // guided matching reads the float32 keypoints from the image table
    s.dev.kp_rows = rows;
    c->table_dirty = true;
    return build_keypoint_grid(c, s, packed.data(), rows);
}

int amc_upload_points_f64(amc_ctx* c, uint32_t slot, const double* xy, uint32_t rows) {
    if (!c) return fail(AMC_E_INVALID, "amc_upload_points_f64: ctx is NULL");
    if (slot >= c->slots.size())
        return fail(AMC_E_INVALID, "amc_upload_points_f64: slot %u >= reserved %zu", slot, c->slots.size());
    if (rows > 0 && !xy) return fail(AMC_E_INVALID, "amc_upload_points_f64: NULL data");
    HIPCHK(hipSetDevice(c->device));
    Slot& s = c->slots[slot];
    if (s.kp || s.kp64 || s.kpn || s.grid_base) {
        HIPCHK(hipStreamSynchronize(c->stream));
        c->arena.free(s.kp);
        c->arena.free(s.kp64);
        c->arena.free(s.kpn);
        c->arena.free(s.grid_base);
        s.grid_base = nullptr;
        s.grid = GridDev{};
        s.kp = nullptr;
        s.kp64 = nullptr;
        s.kpn = nullptr;
        s.kpn_valid = false;
        s.dev.kp = nullptr;
        s.dev.kp_rows = 0;
        c->table_dirty = true;
    }
    s.kp_rows = rows;
    s.has_kp = true;
    if (rows == 0) return AMC_OK;
    hipError_t e = c->arena.alloc(&s.kp64, (size_t)rows * 2 * sizeof(double));
    if (e != hipSuccess) {
        s.has_kp = false;
        return fail(AMC_E_NOMEM, "amc_upload_points_f64: hipMalloc: %s", hipGetErrorString(e));
    }
    HIPCHK(hipMemcpy(s.kp64, xy, (size_t)rows * 2 * sizeof(double), hipMemcpyHostToDevice));
    return AMC_OK;
}

int amc_upload_camera(amc_ctx* c, uint32_t slot, int32_t model_id, uint64_t width, uint64_t height,
                      const double* params, int32_t num_params, int32_t has_prior) {
    if (!c) return fail(AMC_E_INVALID, "amc_upload_camera: ctx is NULL");
    if (slot >= c->slots.size())
        return fail(AMC_E_INVALID, "amc_upload_camera: slot %u >= reserved %zu", slot, c->slots.size());
    if (num_params < 0 || (num_params > 0 && !params))
        return fail(AMC_E_INVALID, "amc_upload_camera: bad params");
    // Camera::VerifyParams: the parameter vector must have the model's length
    if (cam::num_params(model_id) < 0)
        return fail(AMC_E_INVALID, "amc_upload_camera: unknown camera model id %d", model_id);
    if (num_params != cam::num_params(model_id))
        return fail(AMC_E_INVALID, "amc_upload_camera: camera model %d takes %d parameters, got %d", model_id,
                    cam::num_params(model_id), num_params);
    Slot& s = c->slots[slot];
    s.cam = CameraDev{};
    s.cam.model_id = model_id;
    s.cam.has_prior = has_prior ? 1 : 0;
    s.cam.width = width;
    s.cam.height = height;
    for (int i = 0; i < num_params; ++i) s.cam.params[i] = params[i];
    s.has_cam = true;
    s.kpn_valid = false;  // the lifted keypoints belong to the previous camera
    return AMC_OK;
}

// Camera::CamFromImg of all keypoints of a slot, once per (points, camera): COLMAP lifts every matched point of
// every pair (EstimateCalibratedTwoViewGeometry, EstimateTwoViewGeometryPose); the lift depends on the keypoint
// only, so it is taken here per image and kept in HBM.  Pinhole cameras need nothing (two divisions, done where
// the points are gathered).  Polynomial distortion models run on the device (camera.hip); the fisheye family and
// FOV call atan / tan / sin / cos and are lifted with the host libm (camera_math.h).
static int ensure_normalized(amc_ctx* c, uint32_t slot) {
    Slot& s = c->slots[slot];
    if (!s.has_cam || !s.has_kp || cam::is_pinhole(s.cam.model_id) || s.kpn_valid) return AMC_OK;
    const uint32_t rows = s.kp_rows;
    if (rows == 0) {
        s.kpn_valid = true;
        return AMC_OK;
    }
    if (!s.kpn) {
        hipError_t e = c->arena.alloc(&s.kpn, (size_t)rows * 2 * sizeof(double));
        if (e != hipSuccess) return fail(AMC_E_NOMEM, "CamFromImg buffer: hipMalloc: %s", hipGetErrorString(e));
    }
    if (!cam::needs_libm(s.cam.model_id)) {
        HIPCHK(launch_undistort(s.kp, s.kp64, rows, s.cam, s.kpn, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    } else {
        std::vector<double> xy((size_t)rows * 2), uv((size_t)rows * 2);
        if (s.kp64) {
            HIPCHK(hipMemcpy(xy.data(), s.kp64, xy.size() * sizeof(double), hipMemcpyDeviceToHost));
        } else {
            std::vector<float> f((size_t)rows * 2);
            HIPCHK(hipMemcpy(f.data(), s.kp, f.size() * sizeof(float), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < f.size(); ++i) xy[i] = (double)f[i];
        }
        const CameraDev camd = s.cam;
        auto work = [&](uint32_t lo, uint32_t hi) {
            for (uint32_t i = lo; i < hi; ++i)
                cam::cam_from_img(camd.model_id, camd.params, xy[2 * (size_t)i], xy[2 * (size_t)i + 1], uv[2 * (size_t)i],
                                  uv[2 * (size_t)i + 1]);
        };
        const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        const unsigned nth = rows >= 2048 ? hw : 1;
        if (nth == 1) {
            work(0, rows);
        } else {
            std::vector<std::thread> th;
            const uint32_t per = (rows + nth - 1) / nth;
            for (unsigned t = 0; t < nth; ++t) {
                const uint32_t lo = std::min(rows, t * per), hi = std::min(rows, lo + per);
                if (lo < hi) th.emplace_back(work, lo, hi);
            }
            for (auto& t : th) t.join();
        }
        HIPCHK(hipMemcpy(s.kpn, uv.data(), uv.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    s.kpn_valid = true;
    return AMC_OK;
}

int amc_cam_from_img(amc_ctx* c, int32_t model_id, const double* params, int32_t num_params, const double* xy,
                     size_t n, double* uv) {
    if (!c) return fail(AMC_E_INVALID, "amc_cam_from_img: ctx is NULL");
    if (cam::num_params(model_id) < 0) return fail(AMC_E_INVALID, "amc_cam_from_img: unknown camera model id %d", model_id);
    if (num_params != cam::num_params(model_id) || !params)
        return fail(AMC_E_INVALID, "amc_cam_from_img: camera model %d takes %d parameters, got %d", model_id,
                    cam::num_params(model_id), num_params);
    if (n == 0) return AMC_OK;
    if (!xy || !uv) return fail(AMC_E_INVALID, "amc_cam_from_img: NULL points");
    if (n > 0x7FFFFFFFull) return fail(AMC_E_INVALID, "amc_cam_from_img: too many points");
    if (cam::needs_libm(model_id)) {
        for (size_t i = 0; i < n; ++i) cam::cam_from_img(model_id, params, xy[2 * i], xy[2 * i + 1], uv[2 * i], uv[2 * i + 1]);
        return AMC_OK;
    }
    HIPCHK(hipSetDevice(c->device));
    CameraDev cd{};
    cd.model_id = model_id;
    for (int i = 0; i < num_params; ++i) cd.params[i] = params[i];
    DevBuf<double> buf;
    HIPCHK(buf.ensure(4 * n));
    hipStream_t st = c->stream;
    int rc = AMC_OK;
    auto chk = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == AMC_OK) rc = fail(AMC_E_HIP, "amc_cam_from_img: %s: %s", what, hipGetErrorString(e));
    };
    chk(hipMemcpyAsync(buf.p, xy, 2 * n * sizeof(double), hipMemcpyHostToDevice, st), "copy in");
    if (rc == AMC_OK) chk(launch_undistort(nullptr, buf.p, (uint32_t)n, cd, buf.p + 2 * n, st), "launch");
    if (rc == AMC_OK) chk(hipMemcpyAsync(uv, buf.p + 2 * n, 2 * n * sizeof(double), hipMemcpyDeviceToHost, st), "copy out");
    chk(hipStreamSynchronize(st), "sync");
    buf.release();
    return rc;
}

int amc_img_from_cam(amc_ctx* c, int32_t model_id, const double* params, int32_t num_params, const double* uv,
                     size_t n, double* xy) {
    if (!c) return fail(AMC_E_INVALID, "amc_img_from_cam: ctx is NULL");
    if (cam::num_params(model_id) < 0) return fail(AMC_E_INVALID, "amc_img_from_cam: unknown camera model id %d", model_id);
    if (num_params != cam::num_params(model_id) || !params)
        return fail(AMC_E_INVALID, "amc_img_from_cam: camera model %d takes %d parameters, got %d", model_id,
                    cam::num_params(model_id), num_params);
    if (n == 0) return AMC_OK;
    if (!uv || !xy) return fail(AMC_E_INVALID, "amc_img_from_cam: NULL points");
    if (n > 0x7FFFFFFFull) return fail(AMC_E_INVALID, "amc_img_from_cam: too many points");
    if (cam::needs_libm(model_id)) {
        for (size_t i = 0; i < n; ++i) cam::img_from_cam(model_id, params, uv[2 * i], uv[2 * i + 1], xy[2 * i], xy[2 * i + 1]);
        return AMC_OK;
    }
    HIPCHK(hipSetDevice(c->device));
    CameraDev cd{};
    cd.model_id = model_id;
    for (int i = 0; i < num_params; ++i) cd.params[i] = params[i];
    DevBuf<double> buf;
    HIPCHK(buf.ensure(4 * n));
    hipStream_t st = c->stream;
    int rc = AMC_OK;
    auto chk = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == AMC_OK) rc = fail(AMC_E_HIP, "amc_img_from_cam: %s: %s", what, hipGetErrorString(e));
    };
    chk(hipMemcpyAsync(buf.p, uv, 2 * n * sizeof(double), hipMemcpyHostToDevice, st), "copy in");
    if (rc == AMC_OK) chk(launch_project(buf.p, (uint32_t)n, cd, buf.p + 2 * n, st), "launch");
    if (rc == AMC_OK) chk(hipMemcpyAsync(xy, buf.p + 2 * n, 2 * n * sizeof(double), hipMemcpyDeviceToHost, st), "copy out");
    chk(hipStreamSynchronize(st), "sync");
    buf.release();
    return rc;
}

static void fill_tvg_images(const amc_ctx* c, std::vector<TvgImage>& timgs) {
    timgs.resize(c->slots.size());
    for (size_t i = 0; i < timgs.size(); ++i) {
        const Slot& s = c->slots[i];
        timgs[i].kp = s.kp;
        timgs[i].kp64 = s.kp64;
        timgs[i].kpn = (s.kpn_valid && !cam::is_pinhole(s.cam.model_id)) ? s.kpn : nullptr;
        timgs[i].rows = s.kp_rows;
        timgs[i].pad = 0;
        timgs[i].cam = s.cam;
    }
}

namespace {

// RANSAC::ComputeNumTrials (colmap/optim/ransac.h) with the HOST libm, as COLMAP evaluates it
size_t compute_num_trials_host(size_t num_inliers, size_t num_samples, double confidence,
                               double multiplier, int kmin) {
    const double inlier_ratio = num_inliers / static_cast<double>(num_samples);
    const double nom = 1 - confidence;
    if (nom <= 0) return std::numeric_limits<size_t>::max();
    const double denom = 1 - std::pow(inlier_ratio, kmin);
    if (denom <= 0) return 1;
    if (denom == 1.0) return std::numeric_limits<size_t>::max();
    return static_cast<size_t>(std::ceil(std::log(nom) / std::log(denom) * multiplier));
}
size_t ransac_max_trials_host(const amc_ransac_opts& o, double min_inlier_ratio, int kmin) {
    const size_t kNumSamples = 100000;
    const size_t dyn = compute_num_trials_host(static_cast<size_t>(min_inlier_ratio * kNumSamples), kNumSamples,
                                               o.confidence, o.dyn_num_trials_multiplier, kmin);
    return std::min<size_t>(static_cast<size_t>(o.max_num_trials), dyn);
}

struct VerifyPriv {
    std::vector<amc_tvg> tvg;
    std::vector<uint8_t> mask;
    std::vector<amc_pose> pose;
    // single-geometry calls: plain storage, every element written from the device results (a vector would
    // zero tens of megabytes first)
    std::unique_ptr<amc_tvg[]> tvg_raw;
    std::unique_ptr<uint8_t[]> mask_raw;
    // verify_impl: pinned buffers leased from the context's pool (the D2H copies land in them; amc_verify_result_free
    // hands them back for the next call - no page faults on fresh heap memory, no copy out of a staging buffer)
    std::shared_ptr<PinnedPool> pool;
    PinBuf<uint32_t> tvg_pin, mask_pin;
    ~VerifyPriv() {
        if (pool) {
            pool->give_back(tvg_pin);
            pool->give_back(mask_pin);
        }
    }
};

void pose_default(amc_pose* q, int32_t config) {
    std::memset(q, 0, sizeof *q);
    q->config = config;
    q->qvec[0] = 1.0;
    q->R[0] = q->R[4] = q->R[8] = 1.0;
}

}  // namespace

// EstimateTwoViewGeometryPose for every listed pair (pose.hip); `inlier_matches` in CSR layout.
// kernel_ms (optional): the pose kernel's duration.
static int pose_impl(amc_ctx* c, const char* who, const uint32_t* slot1, const uint32_t* slot2, size_t npairs,
                     const uint64_t* match_offsets, const uint32_t* inlier_matches, const amc_tvg* geoms,
                     amc_pose* out, double* kernel_ms, const uint64_t* resident_mask_off = nullptr,
                     const uint32_t* resident_matches = nullptr, const uint64_t* resident_match_off = nullptr) {
    // resident_mask_off != nullptr (amc_verify_pairs): the matches of this call are still on the device - at
    // resident_matches, pair p's list at resident_match_off[p] (default: d_tmatches, the call's CSR offsets) - and
    // pair p's inlier bytes at d_mask_packed + resident_mask_off[p] (the packed masks: the call's CSR offsets); nothing
    // is uploaded again and the kernel takes the rows whose byte is set.  Their indices have been checked.
    const bool resident = resident_mask_off != nullptr;
    if (kernel_ms) *kernel_ms = 0.0;
    if (!c) return fail(AMC_E_INVALID, "%s: NULL ctx", who);
    if (npairs == 0) return AMC_OK;
    if (!slot1 || !slot2 || !match_offsets || !geoms || !out)
        return fail(AMC_E_INVALID, "%s: NULL pair arrays", who);
    const uint64_t total = match_offsets[npairs];
    if (total > 0 && !inlier_matches && !resident) return fail(AMC_E_INVALID, "%s: NULL matches", who);
    if (npairs > 0xFFFFFFFFull) return fail(AMC_E_INVALID, "%s: too many pairs", who);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));  // (an earlier call's upload of the staging buffer is over: every entry point blocks)
    HIPCHK(c->h_ppairs.ensure(npairs));
    PosePair* pp = c->h_ppairs.p;
    std::vector<uint8_t> need_lift(c->slots.size(), 0);
    for (size_t p = 0; p < npairs; ++p) {
        if (slot1[p] >= c->slots.size() || slot2[p] >= c->slots.size())
            return fail(AMC_E_INVALID, "%s: pair %zu references slot out of range", who, p);
        const Slot& a = c->slots[slot1[p]];
        const Slot& b = c->slots[slot2[p]];
        if (!a.has_kp || !b.has_kp || !a.has_cam || !b.has_cam)
            return fail(AMC_E_STATE, "%s: pair %zu: keypoints/camera not uploaded", who, p);
        if (match_offsets[p + 1] < match_offsets[p])
            return fail(AMC_E_INVALID, "%s: match_offsets not monotone at %zu", who, p);
        const uint64_t M = match_offsets[p + 1] - match_offsets[p];
        if (M > 0xFFFFFFFFull) return fail(AMC_E_INVALID, "%s: pair %zu has too many matches", who, p);
        const int32_t cfg = geoms[p].config;
        const bool has_geometry = cfg == AMC_TVG_CALIBRATED || cfg == AMC_TVG_UNCALIBRATED || cfg == AMC_TVG_PLANAR ||
                                  cfg == AMC_TVG_PANORAMIC || cfg == AMC_TVG_PLANAR_OR_PANORAMIC;
        if (has_geometry) need_lift[slot1[p]] = need_lift[slot2[p]] = 1;
        if (!resident)
            for (uint64_t k = match_offsets[p]; k < match_offsets[p + 1]; ++k)
                if (inlier_matches[2 * k] >= a.kp_rows || inlier_matches[2 * k + 1] >= b.kp_rows)
                    return fail(AMC_E_INVALID, "%s: pair %zu match %llu indexes past the keypoints", who, p,
                                (unsigned long long)(k - match_offsets[p]));
        pp[p].slot1 = slot1[p];
        pp[p].slot2 = slot2[p];
        pp[p].match_off = (resident && resident_match_off) ? resident_match_off[p] : match_offsets[p];
        pp[p].ws_off = match_offsets[p];
        pp[p].mask_off = resident ? resident_mask_off[p] : 0;
        pp[p].M = (uint32_t)M;
        pp[p].config = cfg;
        std::memcpy(pp[p].E, geoms[p].E, sizeof pp[p].E);
        std::memcpy(pp[p].H, geoms[p].H, sizeof pp[p].H);
    }
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    for (size_t i = 0; i < need_lift.size(); ++i)
        if (need_lift[i]) {
            const int rc = ensure_normalized(c, (uint32_t)i);
            if (rc != AMC_OK) return rc;
        }
    std::vector<TvgImage> timgs;
    fill_tvg_images(c, timgs);
    HIPCHK(c->d_timgs.ensure(timgs.size()));
    HIPCHK(c->d_ppairs.ensure(npairs));
    if (!resident) HIPCHK(c->d_pmatches.ensure(std::max<size_t>(2 * total, 2)));
    HIPCHK(c->d_pcos.ensure(std::max<size_t>(total, 1)));
    HIPCHK(c->d_pout.ensure(npairs));
    HIPCHK(hipMemcpyAsync(c->d_timgs.p, timgs.data(), timgs.size() * sizeof(TvgImage), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->d_ppairs.p, pp, npairs * sizeof(PosePair), hipMemcpyHostToDevice, st));
    if (total && !resident)
        HIPCHK(hipMemcpyAsync(c->d_pmatches.p, inlier_matches, 2 * total * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIPCHK(hipEventRecord(c->ev[4], st));
    HIPCHK(launch_pose(c->d_timgs.p, c->d_ppairs.p, (uint32_t)npairs,
                       resident ? (resident_matches ? resident_matches : c->d_tmatches.p) : c->d_pmatches.p,
                       resident ? c->d_mask_packed.p : nullptr, c->d_pcos.p, c->d_pout.p, st));
    HIPCHK(hipEventRecord(c->ev[5], st));
    HIPCHK(c->h_pout.ensure(npairs));
    const PoseOut* h = c->h_pout.p;
    HIPCHK(hipMemcpyAsync(c->h_pout.p, c->d_pout.p, npairs * sizeof(PoseOut), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (kernel_ms) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, c->ev[4], c->ev[5]);
        *kernel_ms = ms;
    }
    for (size_t p = 0; p < npairs; ++p) {
        amc_pose& q = out[p];
        pose_default(&q, geoms[p].config);
        if (!h[p].ok) continue;
        q.ok = 1;
        std::memcpy(q.R, h[p].R, sizeof q.R);
        std::memcpy(q.tvec, h[p].t, sizeof q.tvec);
        std::memcpy(q.qvec, h[p].q, sizeof q.qvec);
        q.num_points3D = h[p].num_points3D;
        // Median(CalculateTriangulationAngles(...)): libm acos of the selected cosine(s)
        q.tri_angle = amc::tvg::median_angle_host(h[p].num_points3D, h[p].cmed);
        if (q.config == AMC_TVG_PLANAR_OR_PANORAMIC) {
            if (h[p].t_is_zero) {
                q.config = AMC_TVG_PANORAMIC;
                q.tri_angle = 0.0;
            } else {
                q.config = AMC_TVG_PLANAR;
            }
        }
    }
    return AMC_OK;
}

// The sample stream: std::mt19937(seed)'s output words (operator() tempers them), `need` of them, kept across calls
// with the same seed.  Blocking (the ctx's stream is drained: the host vector goes out of scope).
static hipError_t ensure_sample_stream(amc_ctx* c, uint32_t seed, size_t need) {
    if (c->d_stream.p && c->stream_seed == seed && c->stream_len >= need) return hipSuccess;
    std::vector<uint32_t> words(need);
    std::mt19937 gen(seed);
    for (size_t i = 0; i < need; ++i) words[i] = (uint32_t)gen();
    hipError_t e = hipStreamSynchronize(c->stream);  // (a relaunch: nothing may still read the table that is freed below)
    if (e != hipSuccess) return e;
    e = c->d_stream.ensure(need);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(c->d_stream.p, words.data(), need * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(c->stream);  // `words` goes out of scope
    c->stream_seed = seed;
    c->stream_len = e == hipSuccess ? need : 0;
    return e;
}
// nothing of a verification run is left in flight (error paths; before buffers its kernels read are freed)
static void verify_streams_sync(amc_ctx* c) {
    if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
    for (auto vs : c->vstream)
        if (vs) (void)hipStreamSynchronize(vs);
    (void)hipStreamSynchronize(c->stream);
}

// ---- verification as a run of SLICES ---------------------------------------------------------------------------------
// A verification call used to be two kernel launches behind each other - tvg_e_kernel over every calibrated pair, then
// tvg_fh_kernel over every pair - and amc_match_verify_pairs ran them after the last match batch.  Both are persistent
// kernels whose tails (the last few long pairs on a few waves) leave most of the machine idle, and between them sat a
// kernel-level barrier; the host's preparation for 10^5 pairs (pair records, trial tables, class lists) ran with the
// device idle.  Round 6: the pairs of a call are cut into slices.  All essential-matrix launches go to one stream, all
// F/H launches to another, slice k's F/H waits for slice k's E by event: tvg_e_kernel(slice k + 1) runs beside
// tvg_fh_kernel(slice k), and a kernel's tail is filled by the other stream's waves.  amc_match_verify_pairs hands the
// pairs of match batch k over as a slice as soon as batch k's matches are in the resident table: the host prepares and
// launches it while the device scans batch k + 1, and what is left after the last batch is the last (small) batch's
// slice.  The kernels, the per-pair arithmetic and the results are unchanged: a pair's result does not depend on its
// slice (every pair re-seeds its generator and owns its output record).
//
// mode 0: EstimateTwoViewGeometry; 1 / 2 / 3: a single F / H / E LO-RANSAC per pair, reported
// through the same record (config = success, num_inliers, the model, its trial count, the mask)
namespace {

constexpr int kMaxVerifySlices = 12;

struct VerifyClassLaunch {  // one size class of one slice, as launched (kept for the rare relaunch after a stream overrun)
    int cls = 0;
    bool on_aux = false;
    bool one_stream = false;  // E and F/H behind each other on the E stream (a slice confined to a few CUs beside a scan)
    uint32_t n = 0, n_e = 0, mcap = 0, waves_e = 0, waves_fh = 0;
    int wpb = 4;
};
struct VerifySliceInfo {
    size_t begin = 0, end = 0;
    uint64_t mask_bytes = 0;
    std::vector<VerifyClassLaunch> launches;
};

struct VerifyRun {
    amc_ctx* c;
    int mode;
    const uint32_t* slot1;
    const uint32_t* slot2;
    size_t npairs;
    amc_tvg_opts o;
    uint32_t seed;
    TvgParams P{};
    TvgPair* tp = nullptr;  // npairs records in the ctx's pinned buffer (uploaded as they are by the packing step)
    std::vector<double> wm_cut;
    std::vector<VerifySliceInfo> slices;
    size_t submitted = 0;          // pairs [0, submitted) have been handed over
    const uint32_t* kernel_matches = nullptr;
    hipStream_t st_e = nullptr, st_fh = nullptr;  // all E launches / all F/H launches of the bulk classes
    bool started = false, aux_used = false;
    bool beside_match = false;     // the slices are submitted from the match loop (amc_match_verify_pairs)
    uint32_t launches = 0;
    uint32_t maxM = 0;
    int cus = 256;
    double t_tables = 0.0, t_lists = 0.0;

    bool uses_E(size_t p) const {
        if (mode == 3) return true;
        if (mode != 0 || o.force_H_use) return false;
        if (tp[p].M < (uint32_t)std::max(o.min_num_inliers, 0)) return false;
        return c->slots[slot1[p]].cam.has_prior != 0 && c->slots[slot2[p]].cam.has_prior != 0;
    }
    bool trivial(uint32_t M) const { return mode == 0 && M < (uint32_t)std::max(o.min_num_inliers, 0); }

    // everything that does not depend on the matches: option checks, the sample stream, the image table, the zeroed
    // records.  Issued on the ctx's stream; the verification streams wait for it (vev_setup).
    int begin(size_t total_hint);
    // pairs [begin, end): offs = the call's CSR (offs[p + 1] - offs[p] matches), dev_off = where pair p's rows start in
    // `matches_dev` (nullptr: at offs[p]); `ready` (may be null): an event after which the rows are in place
    int submit(size_t begin, size_t end, const uint64_t* offs, const uint64_t* dev_off, const uint32_t* matches_dev,
               const uint32_t* matches_host, hipEvent_t ready);
    // the two halves of submit(): pairs join the open slice (host only: checks, records, trial tables, size classes);
    // the slice is closed (class lists, uploads, launches).  amc_match_verify_pairs adds every match batch's pairs beside
    // the next batch's scan and closes ONE slice behind the last batch.
    int add_pairs(size_t begin, size_t end, const uint64_t* offs, const uint64_t* dev_off, const uint32_t* matches_dev,
                  const uint32_t* matches_host);
    int close_slice(hipEvent_t ready);
    struct OpenSlice {
        bool active = false;
        size_t begin = 0;
        uint32_t maxM = 0;
        uint64_t mask_bytes = 0;
        std::vector<uint32_t> tabs;
        std::vector<int64_t> tab_of_M;
        std::vector<size_t> cls[4];
    } open;
    int launch_slice(size_t si, hipEvent_t ready);
    int join();
    // amc_match_verify_pairs: how many CUs the next batch's scan shall leave to the verification of pairs [begin, end)
    int plan_cus(size_t begin, size_t end, const uint64_t* offs, double next_scan_ms) const;
    int slice_cus = 0;  // the slice being submitted runs beside a scan that left this many CUs free (0: the whole machine)
    bool defer_launch = false;  // submit() prepares and uploads; the launches follow when the caller says so (launch_deferred)
    int launch_deferred() {
        for (size_t si = 0; si < slices.size(); ++si)
            if (const int rc = launch_slice(si, nullptr)) return rc;
        return AMC_OK;
    }
    double est_ms_total = 0.0;
};

int VerifyRun::begin(size_t) {
    if (o.compute_relative_pose && mode != 0)
        return fail(AMC_E_INVALID, "amc_verify_pairs: internal: compute_relative_pose outside mode 0");
    if (o.multiple_models)
        return fail(AMC_E_INVALID, "amc_verify_pairs: internal: multiple_models reaches verify_impl");
    if (o.ransac.max_num_trials < 0 || o.ransac.min_num_trials < 0 || o.ransac.max_num_trials > (1 << 30))
        return fail(AMC_E_INVALID, "amc_verify_pairs: bad trial limits");
    std::vector<uint8_t> need_lift(c->slots.size(), 0);
    for (size_t p = 0; p < npairs; ++p) {
        if (slot1[p] >= c->slots.size() || slot2[p] >= c->slots.size())
            return fail(AMC_E_INVALID, "amc_verify_pairs: pair %zu references slot out of range", p);
        const Slot& a = c->slots[slot1[p]];
        const Slot& b = c->slots[slot2[p]];
        const bool need_cam = mode == 0 || mode == 3;
        if (!a.has_kp || !b.has_kp || (need_cam && (!a.has_cam || !b.has_cam)))
            return fail(AMC_E_STATE, "amc_verify_pairs: pair %zu: keypoints/camera not uploaded", p);
        const bool e = mode == 0 ? (!o.force_H_use && a.cam.has_prior && b.cam.has_prior) : mode == 3;
        if (e) need_lift[slot1[p]] = need_lift[slot2[p]] = 1;
    }
    P.min_num_inliers = o.min_num_inliers;
    P.detect_watermark = o.detect_watermark;
    P.force_H_use = o.force_H_use;
    P.min_num_trials = (int32_t)std::min<int64_t>(o.ransac.min_num_trials, 1 << 30);
    P.max_trials[0] = (int32_t)ransac_max_trials_host(o.ransac, o.ransac.min_inlier_ratio, 5);
    P.max_trials[1] = (int32_t)ransac_max_trials_host(o.ransac, o.ransac.min_inlier_ratio, 7);
    P.max_trials[2] = (int32_t)ransac_max_trials_host(o.ransac, o.ransac.min_inlier_ratio, 4);
    P.max_trials[3] = (int32_t)ransac_max_trials_host(o.ransac, o.watermark_min_inlier_ratio, 1);
    P.min_E_F_inlier_ratio = o.min_E_F_inlier_ratio;
    P.max_H_inlier_ratio = o.max_H_inlier_ratio;
    P.watermark_min_inlier_ratio = o.watermark_min_inlier_ratio;
    P.watermark_border_size = o.watermark_border_size;
    P.max_error = o.ransac.max_error;
    {
        const char* e = std::getenv("AMC_TVG_SLOW_SAMPLER");
        P.force_slow_sampler = (e && e[0] == '1') ? 1 : 0;
        const char* e2 = std::getenv("AMC_TVG_EXACT_COUNT");
        P.no_fast_count = (e2 && e2[0] == '1') ? 1 : 0;
        const char* e3 = std::getenv("AMC_TVG_NO_S32");
        P.no_fast32 = (e3 && e3[0] == '1') ? 1 : 0;
        P.mode = mode;
        P.bad_index_count = c->d_vscalars;
    }
    // inlier-ratio cut-offs of the watermark RANSAC's dynamic trial count (TvgParams::wm_cut)
    if (mode == 0 && o.detect_watermark) {
        auto dyn_of_ratio = [&](double r) -> size_t {  // ComputeNumTrials with inlier_ratio = r, kMinNumSamples = 1
            const double nom = 1 - o.ransac.confidence;
            if (nom <= 0) return std::numeric_limits<size_t>::max();
            const double denom = 1 - std::pow(r, 1);
            if (denom <= 0) return 1;
            if (denom == 1.0) return std::numeric_limits<size_t>::max();
            return static_cast<size_t>(std::ceil(std::log(nom) / std::log(denom) * o.ransac.dyn_num_trials_multiplier));
        };
        const int nT = std::max(P.max_trials[3], 0);
        // (the cut-offs depend on (confidence, multiplier, max_trials) only: kept across calls)
        if (c->wm_cut_cache.size() == (size_t)nT + 1 && c->wm_cut_conf == o.ransac.confidence &&
            c->wm_cut_mult == o.ransac.dyn_num_trials_multiplier) {
            wm_cut = c->wm_cut_cache;
        } else {
            wm_cut.assign((size_t)nT + 1, 2.0);
            for (int T = 0; T <= nT; ++T) {
                if (dyn_of_ratio(1.0) > (size_t)T) continue;  // not even r = 1 gets there: stays 2.0
                // doubles in [0, 1] order like their bit patterns: bisect the smallest r with dyn(r) <= T
                uint64_t lo = 0, hi = 0x3FF0000000000000ull;  // dyn(lo) > T (or lo is the answer at 0), dyn(hi) <= T
                if (dyn_of_ratio(0.0) <= (size_t)T) { wm_cut[T] = 0.0; continue; }
                while (hi - lo > 1) {
                    const uint64_t mid = lo + (hi - lo) / 2;
                    double r;
                    std::memcpy(&r, &mid, sizeof r);
                    if (dyn_of_ratio(r) <= (size_t)T) hi = mid; else lo = mid;
                }
                std::memcpy(&wm_cut[T], &hi, sizeof(double));
            }
            if (o.ransac.confidence == o.ransac.confidence && o.ransac.dyn_num_trials_multiplier == o.ransac.dyn_num_trials_multiplier) {
                c->wm_cut_cache = wm_cut;
                c->wm_cut_conf = o.ransac.confidence;
                c->wm_cut_mult = o.ransac.dyn_num_trials_multiplier;
            }
        }
    }
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
    HIPCHK(c->h_tp.ensure(std::max<size_t>(npairs, 1)));
    tp = c->h_tp.p;
    if (npairs == 0) return AMC_OK;
    // image table (cameras with distortion parameters: CamFromImg of their keypoints first)
    for (size_t i = 0; i < need_lift.size(); ++i)
        if (need_lift[i]) {
            const int rc = ensure_normalized(c, (uint32_t)i);
            if (rc != AMC_OK) return rc;
        }
    std::vector<TvgImage> timgs;
    fill_tvg_images(c, timgs);
    // The sample stream: std::mt19937(seed)'s output words (operator() tempers them).  Every pair re-seeds (D4), so
    // they all read the same table; its length covers every RANSAC of a pair running to its trial cap, plus the
    // words a chunk draws ahead and a margin for Lemire rejections.  Kept across calls with the same seed.
    size_t stream_need = (size_t)5 * P.max_trials[0] + (size_t)7 * P.max_trials[1] + (size_t)4 * P.max_trials[2] +
                         (size_t)P.max_trials[3] + 4 * 64 * 7 + 4096;
    // (test hook: a table a quarter as long, so that long RANSACs run off it and the relaunch path - every slice again on
    // a table twice as long - is exercised; production tables only ever overrun by a Lemire rejection streak)
    if (std::getenv("AMC_TVG_STREAM_SHORT")) stream_need = std::max<size_t>(8192, stream_need / 4);
    if (stream_need > kMaxStreamWords)
        return fail(AMC_E_INVALID, "amc_verify_pairs: ransac.max_num_trials / min_inlier_ratio allow %zu draws per pair: "
                    "more than the sample-stream table holds (%zu)", stream_need, kMaxStreamWords);
    HIPCHK(ensure_sample_stream(c, seed, stream_need));
    HIPCHK(c->d_timgs.ensure(timgs.size()));
    HIPCHK(c->d_estate.ensure(npairs));
    HIPCHK(c->d_tout.ensure(npairs));
    if (std::getenv("AMC_TVG_PROFILE")) HIPCHK(c->h_tout.ensure(npairs));
    // the image table: uploaded (from pinned memory) only when it differs from what the device holds
    if (c->timgs_on_device.size() != timgs.size() ||
        (!timgs.empty() && std::memcmp(c->timgs_on_device.data(), timgs.data(), timgs.size() * sizeof(TvgImage)) != 0)) {
        HIPCHK(hipStreamSynchronize(st));  // (h_timgs may still feed an earlier copy)
        HIPCHK(c->h_timgs.ensure(std::max<size_t>(timgs.size(), 1)));
        if (!timgs.empty()) std::memcpy(c->h_timgs.p, timgs.data(), timgs.size() * sizeof(TvgImage));
        c->timgs_on_device.clear();
        HIPCHK(hipMemcpyAsync(c->d_timgs.p, c->h_timgs.p, timgs.size() * sizeof(TvgImage), hipMemcpyHostToDevice, st));
        c->timgs_on_device = timgs;
    }
    P.wm_cut = nullptr;
    if (!wm_cut.empty()) {
        const bool same = c->wm_cut_on_device && c->wm_cut_cache.size() == wm_cut.size() && c->d_wmcut.cap >= wm_cut.size() &&
                          std::memcmp(c->wm_cut_cache.data(), wm_cut.data(), wm_cut.size() * sizeof(double)) == 0;
        if (!same) {
            HIPCHK(c->d_wmcut.ensure(wm_cut.size()));
            c->wm_cut_on_device = false;
            HIPCHK(hipMemcpy(c->d_wmcut.p, wm_cut.data(), wm_cut.size() * sizeof(double), hipMemcpyHostToDevice));  // (rare: options changed)
            c->wm_cut_on_device = c->wm_cut_cache.size() == wm_cut.size() &&
                                  std::memcmp(c->wm_cut_cache.data(), wm_cut.data(), wm_cut.size() * sizeof(double)) == 0;
        }
        P.wm_cut = c->d_wmcut.p;
    }
    // [0] pairs with a bad match index, [1] waves that ran off the stream table, [2 ..] the launches' queue heads; the
    // records' profile and work counters are accumulated by both kernels
    HIPCHK(memset_async(c->d_vscalars, 0, kVScalarWords * sizeof(uint32_t), st));
    HIPCHK(memset_async(c->d_tout.p, 0, npairs * sizeof(TvgOut), st));
    HIPCHK(hipEventRecord(c->vev_setup, st));
    P.stream = c->d_stream.p;
    P.stream_len = (uint32_t)std::min<size_t>(c->stream_len, 0xFFFFFFFFu);
    P.stream_err = c->d_vscalars + 1;
    started = true;
    return AMC_OK;
}

// add_pairs: pairs [begin, end) join the OPEN slice - checks, pair records, trial tables, size classes (host only).
int VerifyRun::add_pairs(size_t begin, size_t end, const uint64_t* offs, const uint64_t* dev_off, const uint32_t* matches_dev,
                         const uint32_t* matches_host) {
    if (begin != submitted || end < begin || end > npairs) return fail(AMC_E_INVALID, "amc_verify_pairs: internal: slices out of order");
    if (end == begin) return AMC_OK;
    kernel_matches = matches_dev;
    const auto t0 = std::chrono::steady_clock::now();
    if (!open.active) {
        open = OpenSlice{};
        open.active = true;
        open.begin = begin;
    }
    uint32_t add_maxM = 0;
    for (size_t p = begin; p < end; ++p) {
        if (offs[p + 1] < offs[p]) return fail(AMC_E_INVALID, "amc_verify_pairs: match_offsets not monotone at %zu", p);
        const uint64_t M = offs[p + 1] - offs[p];
        if (M > 65535) return fail(AMC_E_INVALID, "amc_verify_pairs: pair %zu has %llu matches (> 65535)", p, (unsigned long long)M);
        add_maxM = std::max<uint32_t>(add_maxM, (uint32_t)M);
        // Match indices are checked by the kernel where it gathers the points (bad_index_count); only the pairs no
        // kernel looks at - fewer matches than min_num_inliers - are checked here.
        if (trivial((uint32_t)M) && matches_host) {
            const Slot& a = c->slots[slot1[p]];
            const Slot& b = c->slots[slot2[p]];
            const uint32_t* mm = matches_host + 2 * offs[p];
            for (uint64_t k = 0; k < M; ++k)
                if (mm[2 * k] >= a.kp_rows || mm[2 * k + 1] >= b.kp_rows)
                    return fail(AMC_E_INVALID, "amc_verify_pairs: pair %zu match %llu indexes past the keypoints", p, (unsigned long long)k);
        }
    }
    maxM = std::max(maxM, add_maxM);
    open.maxM = std::max(open.maxM, add_maxM);
    if (open.tab_of_M.size() < (size_t)open.maxM + 1) open.tab_of_M.resize((size_t)open.maxM + 1, -1);
    const int kmins[3] = {5, 7, 4};
    auto make_table = [&](uint32_t M) {  // ComputeNumTrials for every inlier count 0 .. M and the three minimal sample sizes
        std::vector<uint32_t> t3;
        t3.reserve(3 * ((size_t)M + 1));
        for (int t = 0; t < 3; ++t)
            for (uint32_t i = 0; i <= M; ++i) {
                const size_t v = M ? compute_num_trials_host(i, M, o.ransac.confidence, o.ransac.dyn_num_trials_multiplier, kmins[t]) : 0;
                t3.push_back(v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v);
            }
        return t3;
    };
    // The tables these pairs need and the cache does not hold (a pow and two logs per entry: the first call of a run
    // sees a few hundred new match counts, ~40 ms on one core) are computed ahead on a few threads.
    const bool tabs_cacheable = o.ransac.confidence == o.ransac.confidence &&
                                o.ransac.dyn_num_trials_multiplier == o.ransac.dyn_num_trials_multiplier;
    std::vector<int32_t> fresh_of((size_t)add_maxM + 1, -1);
    std::vector<uint32_t> fresh_M;
    std::vector<std::vector<uint32_t>> fresh_tab;
    {
        size_t words = 0;
        for (size_t p = begin; p < end; ++p) {
            const uint32_t M = (uint32_t)(offs[p + 1] - offs[p]);
            if (trivial(M) || fresh_of[M] != -1 || open.tab_of_M[M] >= 0) continue;
            fresh_of[M] = -2;  // seen
            if (tabs_cacheable && c->trial_tabs.count(TrialTabKey{M, o.ransac.confidence, o.ransac.dyn_num_trials_multiplier})) continue;
            fresh_of[M] = (int32_t)fresh_M.size();
            fresh_M.push_back(M);
            words += 3 * ((size_t)M + 1);
        }
        fresh_tab.resize(fresh_M.size());
        const unsigned nth = words >= 65536 ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
        std::atomic<size_t> next{0};
        auto work = [&] {
            for (size_t k; (k = next.fetch_add(1)) < fresh_M.size();) fresh_tab[k] = make_table(fresh_M[k]);
        };
        std::vector<std::thread> th;
        for (unsigned k = 1; k < nth; ++k) th.emplace_back(work);
        work();
        for (auto& t : th) t.join();
    }
    std::vector<uint32_t>& tabs = open.tabs;
    for (size_t p = begin; p < end; ++p) {
        const uint32_t M = (uint32_t)(offs[p + 1] - offs[p]);
        TvgPair& q = tp[p];
        q.slot1 = slot1[p];
        q.slot2 = slot2[p];
        q.match_off = dev_off ? dev_off[p] : offs[p];
        q.M = M;
        q.orig = (uint32_t)p;
        q.mask_off = 0;
        q.tab_off[0] = q.tab_off[1] = q.tab_off[2] = 0;
        if (trivial(M)) continue;  // DEGENERATE without a kernel: pack_verify_kernel writes the record
        if (open.tab_of_M[M] < 0) {
            open.tab_of_M[M] = (int64_t)tabs.size();
            // the table of one match count depends on (M, confidence, multiplier) only: kept across calls
            // (a pow and two logs per entry; a pipeline sees the same few hundred counts again and again)
            const TrialTabKey key{M, o.ransac.confidence, o.ransac.dyn_num_trials_multiplier};
            // (NaN options would break the map's ordering: those tables are rebuilt every time)
            auto it = tabs_cacheable ? c->trial_tabs.find(key) : c->trial_tabs.end();
            if (it == c->trial_tabs.end()) {
                std::vector<uint32_t> t3 = (M <= add_maxM && fresh_of[M] >= 0) ? std::move(fresh_tab[(size_t)fresh_of[M]]) : make_table(M);
                if (!tabs_cacheable) {
                    tabs.insert(tabs.end(), t3.begin(), t3.end());
                } else {
                    if (c->trial_tab_words + t3.size() > kTrialTabCacheWords) {  // bounded: start over
                        c->trial_tabs.clear();
                        c->trial_tab_words = 0;
                    }
                    c->trial_tab_words += t3.size();
                    it = c->trial_tabs.emplace(key, std::move(t3)).first;
                }
            }
            if (it != c->trial_tabs.end()) tabs.insert(tabs.end(), it->second.begin(), it->second.end());
        }
        q.mask_off = open.mask_bytes;
        open.mask_bytes += ((uint64_t)M + 127) / 128 * 128;
        for (int t = 0; t < 3; ++t) q.tab_off[t] = (uint32_t)(open.tab_of_M[M] + (int64_t)t * (M + 1));
        // Size classes.  A wave's LDS share holds, besides a few KB of fixed state, two uint16 index arrays of mcap
        // entries (the sampler's permutation and the inlier list); everything else of a pair lives in the wave's global
        // workspace.  Pairs up to ~1,800 matches run at both kernels' full occupancy (E 2, F/H 3 waves per SIMD), 4 waves
        // per workgroup; larger ones in launches of their own with fewer resident waves; the largest (M <= ~38 k: covers
        // max_num_matches = 32768) one wave per workgroup with up to the whole 160 KB; beyond that (class 3, up to the
        // 65,535 matches the 16-bit indices name) the "big" builds of the kernels keep the two arrays in global memory.
        const uint32_t mc = std::max<uint32_t>(64, round_up(M, 64));
        const size_t lds = tvg_lds_bytes(mc, 1) + 64;
        const size_t lds_e = tvg_lds_bytes_e(mc, 1) + 64;  // (the E kernel's waves also carry the root finder's coefficients)
        if (lds <= 160 * 1024 / (4 * (size_t)kTvgFhWavesPerSimd) && lds_e <= 160 * 1024 / (4 * (size_t)kTvgEWavesPerSimd))
            open.cls[0].push_back(p);  // (full occupancy of BOTH kernels)
        else if (lds_e <= 160 * 1024 / 4) open.cls[1].push_back(p);  // 4-wave workgroups of either kernel fit a CU
        else if (lds_e <= 160 * 1024) open.cls[2].push_back(p);
        else open.cls[3].push_back(p);  // M <= 65535 was checked above
    }
    submitted = end;
    t_tables += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return AMC_OK;
}

// close_slice: the open slice's class lists (largest pairs first), its uploads and - unless deferred - its launches
int VerifyRun::close_slice(hipEvent_t ready) {
    if (!open.active) return AMC_OK;
    if (slices.size() >= (size_t)kMaxVerifySlices) return fail(AMC_E_INVALID, "amc_verify_pairs: internal: too many slices");
    const auto t1 = std::chrono::steady_clock::now();
    VerifySliceInfo sl;
    sl.begin = open.begin;
    sl.end = submitted;
    sl.mask_bytes = open.mask_bytes;
    const size_t si = slices.size();
    if (c->vslices.size() <= si) c->vslices.resize(si + 1);
    if (!c->vslices[si]) c->vslices[si].reset(new (std::nothrow) VerifySliceBufs());
    if (!c->vslices[si]) return fail(AMC_E_NOMEM, "amc_verify_pairs: out of host memory");
    VerifySliceBufs& B = *c->vslices[si];
    const std::vector<uint32_t>& tabs = open.tabs;
    const uint32_t slice_maxM = open.maxM;
    std::vector<size_t>(&cls)[4] = open.cls;
    HIPCHK(B.tabs.ensure(std::max<size_t>(tabs.size(), 1)));
    HIPCHK(B.outmask.ensure(std::max<size_t>(open.mask_bytes, 128)));
    HIPCHK(B.emask.ensure(std::max<size_t>(open.mask_bytes, 128)));
    // (pageable sources: these copies are done when the calls return - the vectors may go out of scope - and need no
    // stream synchronisation)
    if (!tabs.empty()) {
        HIPCHK(B.h_tabs.ensure(tabs.size()));
        std::memcpy(B.h_tabs.p, tabs.data(), tabs.size() * sizeof(uint32_t));
        HIPCHK(hipMemcpy(B.tabs.p, B.h_tabs.p, tabs.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    // The first non-empty class (the bulk of a slice) runs on the E / F/H streams; the others - few pairs, each several
    // milliseconds on one wave whatever the machine around it does - on the low-priority stream with their own lists
    // and workspaces, so that they fill the bulk class's tails instead of adding launches of pure latency behind it.
    int bulk = 0;
    while (bulk < 4 && cls[bulk].empty()) ++bulk;
    const bool serial_classes = std::getenv("AMC_TVG_SERIAL_CLASSES") != nullptr || !c->aux_stream;
    for (int k = 3; k >= 0; --k) {  // (the aux classes first: their few waves take their slots before the bulk class fills the machine)
        if (cls[k].empty()) continue;
        VerifyClassLaunch L;
        L.cls = k;
        L.on_aux = k != bulk && !serial_classes && slice_cus == 0;
        L.one_stream = slice_cus > 0;
        L.wpb = k >= 2 ? 1 : 4;
        const bool big = k == 3;  // index arrays in global memory (tvg_*_big.hip)
        // The waves pull pairs from a queue in this order.  A pair's cost grows with its match count (every
        // trial scores all matches), so the largest go first: what is left for the tail of the launch, when
        // most waves have run dry, are the cheap ones.  Results are stored by pair, the order is free.
        // (stable counting sort by match count, descending: M <= 65535)
        std::vector<size_t> idx(cls[k].size());
        {
            std::vector<uint32_t> start((size_t)slice_maxM + 2, 0);
            for (size_t p : cls[k]) ++start[slice_maxM - tp[p].M + 1];
            for (size_t b = 1; b <= (size_t)slice_maxM + 1; ++b) start[b] += start[b - 1];
            for (size_t p : cls[k]) idx[start[slice_maxM - tp[p].M]++] = p;
        }
        VerifyClassSlot& S = B.cls[k];
        HIPCHK(S.h_pairs.ensure(std::max<size_t>(idx.size(), 1)));
        HIPCHK(S.h_pairs_e.ensure(std::max<size_t>(idx.size(), 1)));
        TvgPair* const sub = S.h_pairs.p;      // the lists are written where the copies read them: pinned memory
        TvgPair* const sub_e = S.h_pairs_e.p;
        size_t n_e = 0;
        uint32_t cm = 0;
        for (size_t i = 0; i < idx.size(); ++i) {
            sub[i] = tp[idx[i]];
            cm = std::max(cm, sub[i].M);
            if (uses_E(idx[i])) sub_e[n_e++] = sub[i];
        }
        L.mcap = std::max<uint32_t>(64, round_up(cm, 64));
        auto waves_for = [&](size_t n, int waves_per_simd, size_t lds_block) {
            const uint32_t blocks_per_cu = (uint32_t)std::max<size_t>(
                1, std::min<size_t>(4 * (size_t)waves_per_simd / L.wpb, (160 * 1024) / std::max<size_t>(lds_block, 1)));
            const size_t use_cus = slice_cus > 0 ? (size_t)std::min(slice_cus, cus) : (size_t)cus;
            uint32_t nw = (uint32_t)std::min<size_t>(n, use_cus * blocks_per_cu * L.wpb);
            return std::max<uint32_t>(L.wpb, (nw + L.wpb - 1) / L.wpb * L.wpb);
        };
        const bool run_fh = mode != 3;
        L.n = (uint32_t)idx.size();
        L.n_e = (uint32_t)n_e;
        L.waves_e = n_e == 0 ? 0 : waves_for(n_e, kTvgEWavesPerSimd, big ? tvg_big_lds_bytes_e(L.wpb) : tvg_lds_bytes_e(L.mcap, L.wpb));
        L.waves_fh = run_fh ? waves_for(idx.size(), kTvgFhWavesPerSimd, big ? tvg_big_lds_bytes(L.wpb) : tvg_lds_bytes(L.mcap, L.wpb)) : 0;
        HIPCHK(S.pairs.ensure(idx.size()));
        HIPCHK(S.pairs_e.ensure(std::max<size_t>(n_e, 1)));
        const size_t idx_ws = big ? tvg_big_idx_doubles_host(L.mcap) : 0;  // per wave, behind the point workspaces
        // (E and F/H of one slice run behind each other, but slice k's F/H runs beside slice k + 1's E: own workspaces)
        HIPCHK(S.ws_e.ensure(std::max<size_t>((size_t)L.waves_e * (tvg_ws_doubles_e_host(L.mcap) + idx_ws), 1)));
        HIPCHK(S.ws.ensure(std::max<size_t>((size_t)L.waves_fh * (tvg_ws_doubles_host(L.mcap) + idx_ws), 1)));
        HIPCHK(S.maskws.ensure((size_t)std::max<uint32_t>(L.waves_fh, 1) * tvg_ws_mask_bytes_host(L.mcap)));
        HIPCHK(hipMemcpy(S.pairs.p, sub, idx.size() * sizeof(TvgPair), hipMemcpyHostToDevice));
        if (n_e) HIPCHK(hipMemcpy(S.pairs_e.p, sub_e, n_e * sizeof(TvgPair), hipMemcpyHostToDevice));
        sl.launches.push_back(L);
    }
    t_lists += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    slices.push_back(std::move(sl));
    open = OpenSlice{};
    if (defer_launch) return AMC_OK;
    return launch_slice(si, ready);
}

int VerifyRun::submit(size_t begin, size_t end, const uint64_t* offs, const uint64_t* dev_off, const uint32_t* matches_dev,
                      const uint32_t* matches_host, hipEvent_t ready) {
    if (const int rc = add_pairs(begin, end, offs, dev_off, matches_dev, matches_host)) return rc;
    return close_slice(ready);
}

// the launches of slice si: every class's E kernel(s), then - behind an event - its F/H kernel(s)
int VerifyRun::launch_slice(size_t si, hipEvent_t ready) {
    const VerifySliceInfo& sl = slices[si];
    VerifySliceBufs& B = *c->vslices[si];
    // (a slice confined to a few CUs beside a scan: one stream, E and F/H behind each other - two persistent grids side by
    // side would want twice the CUs the scan left)
    const bool one = !sl.launches.empty() && sl.launches[0].one_stream;
    hipStream_t st_fh = one ? st_e : this->st_fh;
    if (!B.ev[0]) {
        for (auto& e : B.ev)
            if (hipEventCreate(&e) != hipSuccess) return fail(AMC_E_HIP, "amc_verify_pairs: hipEventCreate failed");
        if (hipEventCreateWithFlags(&B.ev_e_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&B.ev_aux_done, hipEventDisableTiming) != hipSuccess)
            return fail(AMC_E_HIP, "amc_verify_pairs: hipEventCreate failed");
    }
    auto wait_inputs = [&](hipStream_t s) -> hipError_t {
        hipError_t e = s == c->stream ? hipSuccess : hipStreamWaitEvent(s, c->vev_setup, 0);
        if (e == hipSuccess && ready) e = hipStreamWaitEvent(s, ready, 0);
        return e;
    };
    bool any_aux = false, any_bulk_e = false;
    for (const VerifyClassLaunch& L : sl.launches) any_aux |= L.on_aux;
    if (any_aux) HIPCHK(wait_inputs(c->aux_stream));
    HIPCHK(wait_inputs(st_e));
    if (st_fh != st_e) HIPCHK(wait_inputs(st_fh));
    HIPCHK(hipEventRecord(B.ev[0], st_e));
    // E kernels: bulk classes on st_e, aux classes (E and F/H behind each other) on the aux stream
    for (const VerifyClassLaunch& L : sl.launches) {
        VerifyClassSlot& S = B.cls[L.cls];
        const bool big = L.cls == 3;
        uint32_t* const qhead = c->d_vscalars + 2 + 8 * si + 2 * L.cls;
        hipStream_t ks = L.on_aux ? c->aux_stream : st_e;
        if (L.n_e) {
            HIPCHK((big ? launch_tvg_e_big : launch_tvg_e)(c->d_timgs.p, S.pairs_e.p, L.n_e, kernel_matches, B.tabs.p, P, S.ws_e.p,
                                                            L.mcap, L.waves_e, L.wpb, qhead, c->d_estate.p, B.emask.p, c->d_tout.p,
                                                            B.outmask.p, ks));
            ++launches;
            any_bulk_e |= !L.on_aux;
        }
        if (L.on_aux && mode != 3) {
            HIPCHK((big ? launch_tvg_fh_big : launch_tvg_fh)(c->d_timgs.p, S.pairs.p, L.n, kernel_matches, B.tabs.p, P, S.ws.p,
                                                              S.maskws.p, L.mcap, L.waves_fh, L.wpb, qhead + 1, c->d_estate.p, B.emask.p,
                                                              c->d_tout.p, B.outmask.p, ks));
            ++launches;
        }
    }
    HIPCHK(hipEventRecord(B.ev[1], st_e));
    if (st_fh != st_e) {
        HIPCHK(hipEventRecord(B.ev_e_done, st_e));
        HIPCHK(hipStreamWaitEvent(st_fh, B.ev_e_done, 0));
    }
    (void)any_bulk_e;
    HIPCHK(hipEventRecord(B.ev[2], st_fh));
    if (mode != 3)
        for (const VerifyClassLaunch& L : sl.launches) {
            if (L.on_aux) continue;
            VerifyClassSlot& S = B.cls[L.cls];
            const bool big = L.cls == 3;
            uint32_t* const qhead = c->d_vscalars + 2 + 8 * si + 2 * L.cls;
            HIPCHK((big ? launch_tvg_fh_big : launch_tvg_fh)(c->d_timgs.p, S.pairs.p, L.n, kernel_matches, B.tabs.p, P, S.ws.p,
                                                              S.maskws.p, L.mcap, L.waves_fh, L.wpb, qhead + 1, c->d_estate.p, B.emask.p,
                                                              c->d_tout.p, B.outmask.p, st_fh));
            ++launches;
        }
    HIPCHK(hipEventRecord(B.ev[3], st_fh));
    if (any_aux) {
        HIPCHK(hipEventRecord(B.ev_aux_done, c->aux_stream));
        aux_used = true;
        B.aux_pending = true;
    }
    return AMC_OK;
}

// every launch of the run is done (the ctx's stream joins the others and is drained)
int VerifyRun::join() {
    hipStream_t st = c->stream;
    for (size_t si = 0; si < slices.size(); ++si) {
        VerifySliceBufs& B = *c->vslices[si];
        HIPCHK(hipStreamWaitEvent(st, B.ev[3], 0));
        HIPCHK(hipStreamWaitEvent(st, B.ev[1], 0));
        if (B.aux_pending) {
            HIPCHK(hipStreamWaitEvent(st, B.ev_aux_done, 0));
            B.aux_pending = false;
        }
    }
    HIPCHK(hipStreamSynchronize(st));
    return AMC_OK;
}

// The verification of pairs [begin, end) as machine-milliseconds (what the two kernels take with every CU: ~1.2 us per
// pair + ~6 ns per match, bench.py's verify and pipeline legs), against the scan that will run beside it: the share of
// the CUs that lets it finish within that scan, with 15 % to spare, between 8 and 96 CUs.  The scan is bound by the
// chip's power budget, not by its CU count - 48 of 256 CUs cost it 8 % of its rate (profiles/r06/scan_grid_v1.txt) - so
// the CUs it gives up are worth more to the verification than to the scan.
int VerifyRun::plan_cus(size_t begin, size_t end, const uint64_t* offs, double next_scan_ms) const {
    if (next_scan_ms <= 0.0) return 0;
    if (const char* e = std::getenv("AMC_VERIFY_CUS")) return std::max(0, std::min(128, std::atoi(e)));  // (A/B hook)
    double est = 0.0;
    for (size_t p = begin; p < end; ++p) {
        const uint64_t M = offs[p + 1] - offs[p];
        if (!trivial((uint32_t)M)) est += 1.2e-3 + 6.0e-6 * (double)M;
    }
    if (est < 0.25) return 0;  // (a launch's fixed costs are not worth hiding)
    // (confined to a few CUs the kernels do ~1.5x the work per CU they do with the whole machine - 35 ms x 256 CUs against
    // 123 ms x 48, profiles/r06/ab_cus_v1.txt: fewer waves share the fabric their workspaces stream through)
    const int v = (int)std::ceil((double)cus * 1.15 * est / (1.5 * next_scan_ms));
    return std::max(8, std::min(96, v));
}

}  // namespace

// dev_matches != nullptr (amc_match_verify_pairs without the streamed hand-over): the matches are already on this device -
// pair p's list starts at dev_matches + 2 * dev_off[p] and has match_offsets[p + 1] - match_offsets[p] rows; `matches` is not read.
// run_in: a VerifyRun whose slices were submitted beside the match batches (amc_match_verify_pairs): only what is
// left - the join, the packing, the download - happens here.
static int verify_finish(amc_ctx* c, VerifyRun& run, const uint64_t* match_offsets, const uint32_t* matches,
                         amc_verify_result* out, VerifyPriv* priv, const uint32_t* dev_matches, const uint64_t* dev_off,
                         double t_pre_ms);

static int verify_impl(amc_ctx* c, int mode, const uint32_t* slot1, const uint32_t* slot2, size_t npairs,
                       const uint64_t* match_offsets, const uint32_t* matches,
                       const amc_tvg_opts* opts_in, uint32_t seed, amc_verify_result* out,
                       const uint32_t* dev_matches = nullptr, const uint64_t* dev_off = nullptr) {
    if (!c || !out) return fail(AMC_E_INVALID, "amc_verify_pairs: NULL ctx/out");
    std::memset(out, 0, sizeof *out);
    c->vres = amc::VerifyResident{};
    if (npairs > 0 && (!slot1 || !slot2 || !match_offsets))
        return fail(AMC_E_INVALID, "amc_verify_pairs: NULL pair arrays");
    const auto wall0 = std::chrono::steady_clock::now();
    VerifyRun run{};
    run.c = c;
    run.mode = mode;
    run.slot1 = slot1;
    run.slot2 = slot2;
    run.npairs = npairs;
    if (opts_in) run.o = *opts_in; else amc_tvg_opts_default(&run.o);
    run.seed = seed;
    const uint64_t total = npairs ? match_offsets[npairs] : 0;
    if (total > 0 && !matches && !dev_matches) return fail(AMC_E_INVALID, "amc_verify_pairs: NULL matches");
    for (size_t p = 0; p < npairs; ++p)
        if (match_offsets[p + 1] < match_offsets[p])
            return fail(AMC_E_INVALID, "amc_verify_pairs: match_offsets not monotone at %zu", p);
    int rc = run.begin(total);
    if (rc != AMC_OK) return rc;
    VerifyPriv* priv = new (std::nothrow) VerifyPriv();
    if (!priv) return fail(AMC_E_NOMEM, "amc_verify_pairs: out of host memory");
    // every failure below (HIPCHK returns included) frees the result's storage and hands back a zeroed struct
    // - after nothing of the call is left in flight: launches on the other streams still run when an error returns, and
    // the next call would rewrite their lists and workspaces under them
    struct Guard {
        amc_ctx* c;
        VerifyPriv* p;
        amc_verify_result* o;
        ~Guard() {
            if (p) {
                verify_streams_sync(c);
                delete p;
                std::memset(o, 0, sizeof *o);
            }
        }
    } guard{c, priv, out};
    hipStream_t st = c->stream;
    if (npairs) {
        // the matches: uploaded once (host path), or where the matcher left them
        const uint32_t* km = dev_matches;
        if (!dev_matches) {
            HIPCHK(c->d_tmatches.ensure(std::max<size_t>(2 * total, 2)));
            if (total) HIPCHK(hipMemcpyAsync(c->d_tmatches.p, matches, 2 * total * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            HIPCHK(hipEventRecord(c->vev_matches, st));
            km = c->d_tmatches.p;
        }
        // Slices: E launches on the ctx's stream, F/H launches on the verification stream.  A slice wants enough pairs
        // to fill the machine several times over (its own tail is only hidden by the NEXT slice's E kernel).
        run.st_e = st;
        run.st_fh = c->vstream[0] ? c->vstream[0] : st;
        size_t nver = 0;
        for (size_t p = 0; p < npairs; ++p) nver += !run.trivial((uint32_t)(match_offsets[p + 1] - match_offsets[p]));
        // One slice by default since the kernels lost their scratch traffic (round 6, final code, 124,750 pairs, kernels:
        // 1 slice 338 ms, 2: 343, 3: 350, 4: 355 - profiles/r06/ab_slices_final_v1.txt): a SIMD that holds an E wave beside
        // F/H waves runs fewer of them, and the kernels' own tails are ~1 % of such a call.  Before that two slices were
        // +1.2 % (1 slice 433-439 ms, 2: 434-435, 4: 441, 8: 480 - ab_pipeline_v1.txt).  AMC_TVG_SLICES: the A/B hook and the
        // tests' way to the sliced path, which amc_match_verify_pairs' batches still take.
        int want = 1;
        if (const char* e = std::getenv("AMC_TVG_SLICES")) want = std::max(1, std::min(kMaxVerifySlices, std::atoi(e)));
        size_t min_per_slice = (size_t)run.cus * 12 * 2;  // two full F/H machine loads per slice
        if (const char* e = std::getenv("AMC_TVG_MIN_PER_SLICE")) min_per_slice = (size_t)std::max(1, std::atoi(e));  // (test hook)
        const int ns = (int)std::max<size_t>(1, std::min<size_t>((size_t)want, nver / std::max<size_t>(min_per_slice, 1)));
        if (ns <= 1) run.st_fh = st;  // one slice: E and F/H behind each other on the ctx's stream, as before
        // cut at equal shares of the verified pairs (the trivial ones cost nothing)
        size_t begin = 0, seen = 0;
        for (int k = 0; k < ns; ++k) {
            size_t end = begin;
            const size_t upto = k + 1 == ns ? nver : (nver * (size_t)(k + 1)) / (size_t)ns;
            if (k + 1 == ns) end = npairs;
            else
                while (end < npairs && seen < upto) {
                    seen += !run.trivial((uint32_t)(match_offsets[end + 1] - match_offsets[end]));
                    ++end;
                }
            rc = run.submit(begin, end, match_offsets, dev_off, km, dev_matches ? nullptr : matches,
                            dev_matches ? nullptr : c->vev_matches);
            if (rc != AMC_OK) return rc;
            begin = end;
        }
    }
    const double t_pre = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    rc = verify_finish(c, run, match_offsets, matches, out, priv, dev_matches, dev_off, t_pre);
    if (rc != AMC_OK) return rc;
    guard.p = nullptr;
    return AMC_OK;
}

static int verify_finish(amc_ctx* c, VerifyRun& run, const uint64_t* match_offsets, const uint32_t* matches,
                         amc_verify_result* out, VerifyPriv* priv, const uint32_t* dev_matches, const uint64_t* dev_off,
                         double t_pre_ms) {
    const size_t npairs = run.npairs;
    const uint64_t total = npairs ? match_offsets[npairs] : 0;
    const bool hprof = std::getenv("AMC_VERIFY_PROFILE") != nullptr;  // wall-clock of the call's host phases on stderr
    const auto wall0 = std::chrono::steady_clock::now();
    hipStream_t st = c->stream;
    priv->pool = c->verify_pool;
    priv->tvg_pin = c->verify_pool->acquire((std::max<size_t>(npairs, 1) * sizeof(amc_tvg) + 3) / 4);
    priv->mask_pin = c->verify_pool->acquire((size_t)(std::max<uint64_t>(total, 1) + 3) / 4);
    if (priv->tvg_pin.ensure((std::max<size_t>(npairs, 1) * sizeof(amc_tvg) + 3) / 4) != hipSuccess ||
        priv->mask_pin.ensure((size_t)(std::max<uint64_t>(total, 1) + 3) / 4) != hipSuccess)
        return fail(AMC_E_NOMEM, "amc_verify_pairs: out of pinned host memory");
    out->npairs = npairs;
    out->_priv = priv;
    out->tvg = reinterpret_cast<amc_tvg*>(priv->tvg_pin.p);
    out->inlier_mask = reinterpret_cast<uint8_t*>(priv->mask_pin.p);
    if (npairs == 0) return AMC_OK;
    if (run.submitted != npairs) return fail(AMC_E_INVALID, "amc_verify_pairs: internal: %zu of %zu pairs submitted", run.submitted, npairs);
    double kernel_ms = 0.0;
    for (int attempt = 0;; ++attempt) {
        {
            const int rc = run.join();
            if (rc != AMC_OK) return rc;
        }
        // kernel time: from the first slice's first launch to the last launch's end when the slices ran on the ctx's own
        // streams back to back (a verification call); the sum of the slices' E and F/H spans when they ran beside the
        // match batches (amc_match_verify_pairs: the span of the whole run would count the scans between them)
        kernel_ms = 0.0;
        if (!run.slices.empty()) {
            if (run.beside_match) {
                for (size_t si = 0; si < run.slices.size(); ++si) {
                    float a = 0.f, b = 0.f;
                    (void)hipEventElapsedTime(&a, c->vslices[si]->ev[0], c->vslices[si]->ev[1]);
                    (void)hipEventElapsedTime(&b, c->vslices[si]->ev[2], c->vslices[si]->ev[3]);
                    kernel_ms += a + b;
                }
            } else {
                HIPCHK(hipEventRecord(c->ev[3], st));
                HIPCHK(hipEventSynchronize(c->ev[3]));
                float kms = 0.f;
                (void)hipEventElapsedTime(&kms, c->vslices[0]->ev[0], c->ev[3]);
                kernel_ms = kms;
            }
        }
        uint32_t vs[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(vs, c->d_vscalars, sizeof vs, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (!vs[1]) break;
        // a Lemire rejection loop ran past the table (probability ~1e-6 per 4096 spare words): lay out more, redo -
        // every slice again (the lists are still on the device), behind each other
        if (attempt >= 4 || c->stream_len * 2 > kMaxStreamWords)
            return fail(AMC_E_HIP, "amc_verify_pairs: the sample stream table was exhausted %d times", attempt + 1);
        HIPCHK(ensure_sample_stream(c, run.seed, c->stream_len * 2));
        run.P.stream = c->d_stream.p;
        run.P.stream_len = (uint32_t)std::min<size_t>(c->stream_len, 0xFFFFFFFFu);
        HIPCHK(memset_async(c->d_vscalars, 0, kVScalarWords * sizeof(uint32_t), st));
        HIPCHK(memset_async(c->d_tout.p, 0, npairs * sizeof(TvgOut), st));
        HIPCHK(hipEventRecord(c->vev_setup, st));
        run.beside_match = false;
        for (size_t si = 0; si < run.slices.size(); ++si) {
            const int rc = run.launch_slice(si, nullptr);
            if (rc != AMC_OK) return rc;
        }
    }
    // The kernel stores a pair's record at the caller's pair index (TvgPair::orig) and its mask at a 128-byte
    // aligned offset of its slice's buffer; pack_verify_kernel lays both out as the caller reads them (records without
    // their counters, masks at the input's CSR offsets; a pair no kernel looked at - fewer matches than min_num_inliers -
    // becomes the DEGENERATE record EstimateTwoViewGeometry returns for it) and sums the work counters, so the copies
    // below land in the result itself.
    HIPCHK(c->d_tvg_packed.ensure(npairs));
    HIPCHK(c->d_mask_packed.ensure(std::max<uint64_t>(total, 1)));
    HIPCHK(c->d_moff.ensure(npairs + 1));
    HIPCHK(c->d_tp_all.ensure(npairs));
    HIPCHK(c->d_worksum.ensure(12));
    HIPCHK(c->h_moff.ensure(npairs + 1));
    std::memcpy(c->h_moff.p, match_offsets, (npairs + 1) * sizeof(uint64_t));
    HIPCHK(hipMemcpyAsync(c->d_moff.p, c->h_moff.p, (npairs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->d_tp_all.p, run.tp, npairs * sizeof(TvgPair), hipMemcpyHostToDevice, st));
    HIPCHK(memset_async(c->d_worksum.p, 0, 12 * sizeof(unsigned long long), st));
    const int32_t trivial_below = run.mode == 0 ? std::max(run.o.min_num_inliers, 0) : 0;
    for (size_t si = 0; si < run.slices.size(); ++si) {
        const VerifySliceInfo& sl = run.slices[si];
        HIPCHK(launch_pack_verify(c->d_tout.p + sl.begin, c->d_tp_all.p + sl.begin, (uint32_t)(sl.end - sl.begin),
                                  c->vslices[si]->outmask.p, c->d_moff.p + sl.begin, c->d_tvg_packed.p + sl.begin,
                                  c->d_mask_packed.p, c->d_worksum.p, trivial_below, st));
    }
    HIPCHK(hipMemcpyAsync(out->tvg, c->d_tvg_packed.p, npairs * sizeof(amc_tvg), hipMemcpyDeviceToHost, st));
    if (total) HIPCHK(hipMemcpyAsync(out->inlier_mask, c->d_mask_packed.p, total, hipMemcpyDeviceToHost, st));
    unsigned long long worksum[12];
    HIPCHK(hipMemcpyAsync(worksum, c->d_worksum.p, sizeof worksum, hipMemcpyDeviceToHost, st));
    const bool want_prof = std::getenv("AMC_TVG_PROFILE") != nullptr;
    if (want_prof)  // the per-pair cycle counters live in the full records
        HIPCHK(hipMemcpyAsync(c->h_tout.p, c->d_tout.p, npairs * sizeof(TvgOut), hipMemcpyDeviceToHost, st));
    uint32_t bad_pairs = 0;
    HIPCHK(hipMemcpyAsync(&bad_pairs, c->d_vscalars, sizeof bad_pairs, hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(c->ev[1], st));
    HIPCHK(hipEventSynchronize(c->ev[1]));
    if (bad_pairs) {  // the kernel met an index past an image's keypoints: find it for the message
        for (size_t p = 0; p < npairs && matches; ++p) {
            const Slot& a = c->slots[run.slot1[p]];
            const Slot& b = c->slots[run.slot2[p]];
            for (uint64_t k = match_offsets[p]; k < match_offsets[p + 1]; ++k)
                if (matches[2 * k] >= a.kp_rows || matches[2 * k + 1] >= b.kp_rows)
                    return fail(AMC_E_INVALID, "amc_verify_pairs: pair %zu match %llu indexes past the keypoints", p,
                                (unsigned long long)(k - match_offsets[p]));
        }
        return fail(AMC_E_INVALID, "amc_verify_pairs: %u pairs index past the keypoints", bad_pairs);
    }
    if (want_prof) {
        const TvgOut* h_out = c->h_tout.p;
        tvg_diag_report();
        tvg_diag_report_e();
        unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t p = 0; p < npairs; ++p)
            for (int i = 0; i < 8; ++i) acc[i] += h_out[p].prof[i];
        if (acc[4] == 0)
            std::fprintf(stderr, "[amc tvg profile] the kernels' cycle counters are compiled out of this build (-DAMC_TVG_PROF or the "
                         "-DAMC_TVG_LODIAG build: tools/variant_build_tvg.sh)\n");
        else
        std::fprintf(stderr, "[amc tvg profile] pairs=%zu cycles/pair: sampling=%.0f minimal=%.0f replay+score=%.0f "
                     "(of which LO=%.0f) total=%.0f\n", npairs, (double)acc[0] / npairs, (double)acc[1] / npairs,
                     (double)acc[2] / npairs, (double)acc[3] / npairs, (double)acc[4] / npairs);
        if (acc[4] != 0)
        std::fprintf(stderr, "[amc tvg profile] per pair: counting loop=%.0f local_estimate(E5)=%.0f local_estimate(F8)=%.0f\n",
                     (double)acc[5] / npairs, (double)acc[6] / npairs, (double)acc[7] / npairs);
    }
    for (int i = 0; i < 12; ++i) out->work[i] += worksum[i];
    float ms = 0.f;
    if (!run.slices.empty() && !run.beside_match) (void)hipEventElapsedTime(&ms, c->vslices[0]->ev[0], c->ev[1]);
    // (beside the match batches there is no span of its own: the kernels plus what followed the last batch)
    const double t_post = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    out->device_ms = run.beside_match ? kernel_ms : (double)ms;
    out->kernel_ms = kernel_ms;
    out->kernel_launches = run.launches;
    if (hprof)
        std::fprintf(stderr, "[amc verify profile] pairs=%zu slices=%zu%s: host before the join %.1f ms (tables %.1f, lists + uploads %.1f), "
                     "join + pack + download %.1f (kernels %.1f)\n", npairs, run.slices.size(), run.beside_match ? " beside the match batches" : "",
                     t_pre_ms, run.t_tables, run.t_lists, t_post, kernel_ms);
    if (run.o.compute_relative_pose) {
        // EstimateTwoViewGeometryPose on the selected inlier matches (mask order = match order): the matches
        // and the packed masks of this call are still on the device
        priv->pose.resize(npairs);
        double pose_ms = 0.0;
        const int rc = pose_impl(c, "amc_verify_pairs", run.slot1, run.slot2, npairs, match_offsets, matches, out->tvg,
                                 priv->pose.data(), &pose_ms, match_offsets, dev_matches, dev_off);
        if (rc != AMC_OK) return rc;
        for (size_t p = 0; p < npairs; ++p) out->tvg[p].config = priv->pose[p].config;
        out->pose = priv->pose.data();
        out->device_ms += pose_ms;
        out->kernel_ms += pose_ms;
        out->pose_kernel_ms = pose_ms;
        out->kernel_launches += 1;
    }
    if (run.mode == 0) {  // what the exchange step's verification half reads in place (amc_allgather_pair_records / _inlier_tables)
        c->vres.npairs = npairs;
        c->vres.total = total;
        // (EstimateTwoViewGeometryPose settles PLANAR_OR_PANORAMIC on the host copy of the records only)
        c->vres.tvg = run.o.compute_relative_pose ? nullptr : c->d_tvg_packed.p;
        c->vres.mask = c->d_mask_packed.p;
        c->vres.moff = c->d_moff.p;
        c->vres.tp = c->d_tp_all.p;
        c->vres.matches = run.kernel_matches;
    }
    return AMC_OK;
}

// EstimateMultipleTwoViewGeometries (TwoViewGeometryOptions.multiple_models): rounds of
// EstimateTwoViewGeometry over all still-active pairs at once, each on the matches its earlier
// rounds left over, until a pair's round comes back DEGENERATE.  All estimation runs in the kernel;
// the host only shrinks the match lists between rounds.
static int verify_multiple(amc_ctx* c, const uint32_t* slot1, const uint32_t* slot2, size_t npairs,
                           const uint64_t* match_offsets, const uint32_t* matches, const amc_tvg_opts& o,
                           uint32_t seed, amc_verify_result* out) {
    std::memset(out, 0, sizeof *out);
    if (npairs > 0 && (!slot1 || !slot2 || !match_offsets))
        return fail(AMC_E_INVALID, "amc_verify_pairs: NULL pair arrays");
    const uint64_t total = npairs ? match_offsets[npairs] : 0;
    if (total > 0 && !matches) return fail(AMC_E_INVALID, "amc_verify_pairs: NULL matches");
    for (size_t p = 0; p < npairs; ++p)
        if (match_offsets[p + 1] < match_offsets[p])
            return fail(AMC_E_INVALID, "amc_verify_pairs: match_offsets not monotone at %zu", p);
    amc_tvg_opts single = o;
    single.multiple_models = 0;
    VerifyPriv* priv = new (std::nothrow) VerifyPriv();
    if (!priv) return fail(AMC_E_NOMEM, "amc_verify_pairs: out of host memory");
    priv->tvg.resize(npairs);
    priv->mask.assign(total, 0);
    std::vector<std::vector<uint32_t>> remaining(npairs);  // indices into the pair's original matches
    std::vector<std::vector<amc_tvg>> kept(npairs);
    std::vector<amc_pose> first_pose(npairs);  // pose of a pair's first kept geometry
    for (size_t p = 0; p < npairs; ++p) pose_default(&first_pose[p], AMC_TVG_UNDEFINED);
    std::vector<size_t> active;
    for (size_t p = 0; p < npairs; ++p) {
        const size_t M = (size_t)(match_offsets[p + 1] - match_offsets[p]);
        remaining[p].resize(M);
        for (size_t i = 0; i < M; ++i) remaining[p][i] = (uint32_t)i;
        active.push_back(p);
    }
    double device_ms = 0.0, kernel_ms = 0.0, pose_ms = 0.0;
    uint64_t work[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t launches = 0;
    int rc = AMC_OK;
    for (int round = 0; round < 254 && !active.empty() && rc == AMC_OK; ++round) {
        std::vector<uint32_t> s1(active.size()), s2(active.size()), rm;
        std::vector<uint64_t> off(active.size() + 1, 0);
        for (size_t a = 0; a < active.size(); ++a) {
            const size_t p = active[a];
            s1[a] = slot1[p];
            s2[a] = slot2[p];
            const uint32_t* mm = matches + 2 * match_offsets[p];
            for (uint32_t i : remaining[p]) {
                rm.push_back(mm[2 * (size_t)i]);
                rm.push_back(mm[2 * (size_t)i + 1]);
            }
            off[a + 1] = off[a] + remaining[p].size();
        }
        amc_verify_result r;
        rc = verify_impl(c, 0, s1.data(), s2.data(), active.size(), off.data(), rm.data(), &single, seed, &r);
        if (rc != AMC_OK) break;
        device_ms += r.device_ms;
        kernel_ms += r.kernel_ms;
        for (int i = 0; i < 12; ++i) work[i] += r.work[i];
        pose_ms += r.pose_kernel_ms;
        launches += r.kernel_launches;
        std::vector<size_t> still;
        for (size_t a = 0; a < active.size(); ++a) {
            const size_t p = active[a];
            const amc_tvg& g = r.tvg[a];
            if (g.config == AMC_TVG_DEGENERATE) continue;  // this pair is finished
            const bool keep = !(o.multiple_ignore_watermark && g.config == AMC_TVG_WATERMARK);
            if (keep) {
                kept[p].push_back(g);
                if (kept[p].size() == 1 && r.pose) first_pose[p] = r.pose[a];
            }
            const uint8_t* mask = r.inlier_mask + off[a];
            std::vector<uint32_t> next;
            for (size_t k = 0; k < remaining[p].size(); ++k) {
                if (mask[k]) {
                    if (keep) priv->mask[match_offsets[p] + remaining[p][k]] = (uint8_t)kept[p].size();
                } else {
                    next.push_back(remaining[p][k]);
                }
            }
            if (next.size() == remaining[p].size()) continue;  // nothing left the pool: stop, do not spin
            remaining[p].swap(next);
            still.push_back(p);
        }
        amc_verify_result_free(&r);
        active.swap(still);
    }
    if (rc != AMC_OK) {
        delete priv;
        return rc;
    }
    if (o.compute_relative_pose) priv->pose.resize(npairs);
    for (size_t p = 0; p < npairs; ++p) {
        amc_tvg& t = priv->tvg[p];
        std::memset(&t, 0, sizeof t);
        if (o.compute_relative_pose) pose_default(&priv->pose[p], AMC_TVG_UNDEFINED);
        if (kept[p].empty()) {
            t.config = AMC_TVG_DEGENERATE;
            std::fill(priv->mask.begin() + match_offsets[p], priv->mask.begin() + match_offsets[p + 1], 0);
        } else if (kept[p].size() == 1) {
            t = kept[p][0];
            if (o.compute_relative_pose) priv->pose[p] = first_pose[p];
        } else {
            t.config = AMC_TVG_MULTIPLE;  // the models of a MULTIPLE geometry stay default (zero)
            for (const amc_tvg& g : kept[p]) t.num_inliers += g.num_inliers;
        }
        if (o.compute_relative_pose) priv->pose[p].config = t.config;
    }
    out->npairs = npairs;
    out->_priv = priv;
    out->tvg = priv->tvg.data();
    out->inlier_mask = priv->mask.data();
    out->pose = o.compute_relative_pose ? priv->pose.data() : nullptr;
    out->pose_kernel_ms = pose_ms;
    out->device_ms = device_ms;
    out->kernel_ms = kernel_ms;
    out->kernel_launches = launches;
    for (int i = 0; i < 12; ++i) out->work[i] = work[i];
    c->vres = amc::VerifyResident{};  // (the rounds' calls left the LAST round's shrunken lists: not this call's result)
    return AMC_OK;
}

int amc_verify_pairs(amc_ctx* c, const uint32_t* slot1, const uint32_t* slot2, size_t npairs,
                     const uint64_t* match_offsets, const uint32_t* matches,
                     const amc_tvg_opts* opts_in, uint32_t seed, amc_verify_result* out) {
    const uint64_t total = (c && out && npairs > 0 && match_offsets) ? match_offsets[npairs] : 0;
    if (total > 0 && !matches) {
        // the resident match table (amc_upload_matches, or the last match call's rows) instead of rows over PCIe
        if (opts_in && opts_in->multiple_models) {
            std::memset(out, 0, sizeof *out);
            return fail(AMC_E_INVALID, "amc_verify_pairs: multiple_models needs the match rows on the host (matches is NULL)");
        }
        if (c->resident_matches != total) {
            std::memset(out, 0, sizeof *out);
            return fail(AMC_E_STATE, "amc_verify_pairs: matches is NULL and the resident match table holds %llu rows, not the "
                        "%llu of match_offsets", (unsigned long long)c->resident_matches, (unsigned long long)total);
        }
        const uint64_t keep = c->resident_matches;  // (verify_impl drops a resident verification result, not the match table)
        const int rc = verify_impl(c, 0, slot1, slot2, npairs, match_offsets, nullptr, opts_in, seed, out, c->d_keep.p, match_offsets);
        c->resident_matches = keep;
        return rc;
    }
    if (c && out && opts_in && opts_in->multiple_models)
        return verify_multiple(c, slot1, slot2, npairs, match_offsets, matches, *opts_in, seed, out);
    return verify_impl(c, 0, slot1, slot2, npairs, match_offsets, matches, opts_in, seed, out);
}

int amc_match_verify_pairs(amc_ctx* c, const uint32_t* slot1, const uint32_t* slot2, size_t npairs,
                           const amc_match_opts* match_opts, const amc_tvg_opts* tvg_opts, uint32_t seed,
                           amc_match_result* match_out, amc_verify_result* verify_out) {
    if (!c || !match_out || !verify_out) return fail(AMC_E_INVALID, "amc_match_verify_pairs: NULL ctx/out");
    std::memset(verify_out, 0, sizeof *verify_out);
    if (tvg_opts && tvg_opts->multiple_models) {
        // EstimateMultipleTwoViewGeometries shrinks the match lists on the host between rounds: no resident path
        int rc = match_impl(c, slot1, slot2, npairs, match_opts, nullptr, 0.0, match_out);
        if (rc != AMC_OK) return rc;
        rc = amc_verify_pairs(c, slot1, slot2, npairs, match_out->offsets, match_out->matches, tvg_opts, seed, verify_out);
        if (rc != AMC_OK) amc_match_result_free(match_out);
        return rc;
    }
    if (std::getenv("AMC_PIPELINE_SERIAL")) {  // (A/B hook: the stages behind each other, as before round 6)
        std::vector<uint64_t> keep_off;
        int rc = match_impl(c, slot1, slot2, npairs, match_opts, nullptr, 0.0, match_out, &keep_off);
        if (rc != AMC_OK) return rc;
        rc = verify_impl(c, 0, slot1, slot2, npairs, match_out->offsets, match_out->matches, tvg_opts, seed, verify_out,
                         c->d_keep.p ? c->d_keep.p : reinterpret_cast<const uint32_t*>(c->d_scalars), keep_off.data());
        if (rc != AMC_OK) amc_match_result_free(match_out);
        return rc;
    }
    // The HOST sides of the two stages interleaved: the verification run is set up first (nothing of it depends on the
    // matches), and every match batch hands its pairs over while the next batch is scanned - their checks, pair records,
    // trial tables and size classes are done beside that scan.  ONE slice is closed and launched when the last batch
    // is done (three slices of ~3,000 verified pairs each have three tails: 38.9 ms of kernels against 35.3 for one,
    // profiles/r06/ab_final_v1.txt): on the device the stages stay behind each other, because the chip is bound by its
    // power budget - verification beside a scan takes from the scan what it gets (profiles/r06/ab_cus_v2.txt: the scan
    // leaving 24 .. 96 CUs to the verification of the batch before, 3 .. 8 batches, all within 1 % of the serial order).
    // AMC_PIPELINE_INTERLEAVE=1 keeps that variant reachable for the A/B.
    std::memset(match_out, 0, sizeof *match_out);
    if (npairs > 0 && (!slot1 || !slot2)) return fail(AMC_E_INVALID, "amc_match_verify_pairs: NULL pair arrays");
    c->vres = amc::VerifyResident{};
    const auto wall0 = std::chrono::steady_clock::now();
    VerifyRun run{};
    run.c = c;
    run.mode = 0;
    run.slot1 = slot1;
    run.slot2 = slot2;
    run.npairs = npairs;
    if (tvg_opts) run.o = *tvg_opts; else amc_tvg_opts_default(&run.o);
    run.seed = seed;
    run.beside_match = true;
    for (size_t p = 0; p < npairs; ++p)  // (the match call checks this too; the verification set-up reads the slots first)
        if (slot1[p] >= c->slots.size() || slot2[p] >= c->slots.size())
            return fail(AMC_E_INVALID, "amc_match_verify_pairs: pair %zu references slot out of range", p);
    int rc = run.begin(0);
    if (rc != AMC_OK) return rc;
    const double t_setup = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    run.st_e = c->vstream[0] ? c->vstream[0] : c->stream;
    run.st_fh = c->vstream[1] ? c->vstream[1] : run.st_e;
    VerifyPriv* priv = new (std::nothrow) VerifyPriv();
    if (!priv) return fail(AMC_E_NOMEM, "amc_match_verify_pairs: out of host memory");
    struct Guard {  // every failure: nothing left in flight, both results zeroed
        amc_ctx* c;
        VerifyPriv* p;
        amc_verify_result* o;
        amc_match_result* m;
        bool match_done = false;
        ~Guard() {
            if (p) {
                verify_streams_sync(c);
                delete p;
                std::memset(o, 0, sizeof *o);
                if (match_done) amc_match_result_free(m);
            }
        }
    } guard{c, priv, verify_out, match_out};
    std::vector<uint64_t> keep_off;
    const bool interleave = std::getenv("AMC_PIPELINE_INTERLEAVE") != nullptr;
    run.defer_launch = !interleave;
    BatchHook hook;
    if (interleave) hook.plan = [&](size_t begin, size_t end, const uint64_t* offsets, double next_scan_ms) -> int {
        if (run.slices.size() + 1 >= (size_t)kMaxVerifySlices) return 0;
        (void)begin;
        return run.plan_cus(run.submitted, end, offsets, next_scan_ms);
    };
    hook.submit = [&](size_t begin, size_t end, const uint64_t* offsets, const uint64_t* koff, hipEvent_t ready, int cus_free) -> int {
        // (the last slice slot is kept for whatever is left when the match call returns)
        (void)begin;
        if (run.slices.size() + 1 >= (size_t)kMaxVerifySlices) return AMC_OK;
        const uint32_t* km = c->d_keep.p ? c->d_keep.p : reinterpret_cast<const uint32_t*>(c->d_scalars);
        if (!interleave) return run.add_pairs(run.submitted, end, offsets, koff, km, nullptr);  // (host only; one slice, closed below)
        run.slice_cus = cus_free;
        return run.submit(run.submitted, end, offsets, koff, km, nullptr, ready);
    };
    rc = match_impl(c, slot1, slot2, npairs, match_opts, nullptr, 0.0, match_out, &keep_off, npairs ? &hook : nullptr);
    if (rc != AMC_OK) return rc;
    guard.match_done = true;
    const double t_match = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    {
        const uint32_t* km = c->d_keep.p ? c->d_keep.p : reinterpret_cast<const uint32_t*>(c->d_scalars);
        run.slice_cus = 0;
        if (run.submitted < npairs) {
            rc = run.add_pairs(run.submitted, npairs, match_out->offsets, keep_off.data(), km, nullptr);
            if (rc != AMC_OK) return rc;
        }
        run.kernel_matches = km;  // (the resident table may have moved while it grew)
        if (!interleave) {  // the one slice of this call: E and F/H behind each other on the ctx's stream
            run.beside_match = false;
            run.defer_launch = false;
            run.st_e = run.st_fh = c->stream;
        }
        rc = run.close_slice(nullptr);
        if (rc != AMC_OK) return rc;
    }
    const double t_pre = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    rc = verify_finish(c, run, match_out->offsets, match_out->matches, verify_out, priv,
                       c->d_keep.p ? c->d_keep.p : reinterpret_cast<const uint32_t*>(c->d_scalars), keep_off.data(), t_pre);
    if (rc != AMC_OK) return rc;
    guard.p = nullptr;
    {
        const double t_end = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
        const double tl[8] = {t_setup, t_match, t_pre, t_end, t_end, c->last_hook_ms, 0.0, 0.0};
        std::memcpy(c->timeline, tl, sizeof tl);
    }
    if (std::getenv("AMC_VERIFY_PROFILE"))  // the call's timeline on the host (ms since entry)
        std::fprintf(stderr, "[amc pipeline profile] pairs=%zu: verification set up at %.2f, match call back at %.2f, slice closed + launched at %.2f, "
                     "results on the host at %.2f\n", npairs, t_setup, t_match, t_pre,
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count());
    return AMC_OK;
}

int amc_ctx_last_timeline(amc_ctx* c, double out_ms[8]) {
    if (!c || !out_ms) return fail(AMC_E_INVALID, "amc_ctx_last_timeline: NULL argument");
    std::memcpy(out_ms, c->timeline, sizeof c->timeline);
    return AMC_OK;
}

int amc_pose_pairs(amc_ctx* c, const uint32_t* slot1, const uint32_t* slot2, size_t npairs,
                   const uint64_t* match_offsets, const uint32_t* inlier_matches, const amc_tvg* geoms,
                   amc_pose* out) {
    return pose_impl(c, "amc_pose_pairs", slot1, slot2, npairs, match_offsets, inlier_matches, geoms, out, nullptr);
}

namespace {
struct RansacPriv {
    std::vector<amc_ransac_report> reports;
    std::vector<uint8_t> mask;
};
}  // namespace

int amc_ransac_pairs(amc_ctx* c, int kind, const uint32_t* slot1, const uint32_t* slot2, size_t npairs,
                     const uint64_t* match_offsets, const uint32_t* matches,
                     const amc_ransac_opts* ropts, uint32_t seed, amc_ransac_result* out) {
    if (!c || !out) return fail(AMC_E_INVALID, "amc_ransac_pairs: NULL ctx/out");
    std::memset(out, 0, sizeof *out);
    if (kind != AMC_RANSAC_F && kind != AMC_RANSAC_H && kind != AMC_RANSAC_E)
        return fail(AMC_E_INVALID, "amc_ransac_pairs: unknown estimator kind %d", kind);
    amc_tvg_opts o;
    amc_tvg_opts_default(&o);
    if (ropts) o.ransac = *ropts;
    o.detect_watermark = 0;
    amc_verify_result v;
    const int mode = kind == AMC_RANSAC_F ? 1 : (kind == AMC_RANSAC_H ? 2 : 3);
    const int rc = verify_impl(c, mode, slot1, slot2, npairs, match_offsets, matches, &o, seed, &v);
    if (rc != AMC_OK) return rc;
    RansacPriv* priv = new (std::nothrow) RansacPriv();
    if (!priv) {
        amc_verify_result_free(&v);
        return fail(AMC_E_NOMEM, "amc_ransac_pairs: out of host memory");
    }
    const uint64_t total = npairs ? match_offsets[npairs] : 0;
    priv->reports.resize(npairs);
    priv->mask.assign(v.inlier_mask, v.inlier_mask + total);
    const int which = kind == AMC_RANSAC_F ? 1 : (kind == AMC_RANSAC_H ? 2 : 0);  // num_trials / inliers slot
    for (size_t p = 0; p < npairs; ++p) {
        const amc_tvg& g = v.tvg[p];
        amc_ransac_report& r = priv->reports[p];
        r.success = g.config;
        r.num_inliers = g.num_inliers;
        r.num_trials = g.num_trials[which];
        const double* m = kind == AMC_RANSAC_F ? g.F : (kind == AMC_RANSAC_H ? g.H : g.E);
        for (int i = 0; i < 9; ++i) r.model[i] = m[i];
    }
    out->npairs = npairs;
    out->reports = priv->reports.data();
    out->inlier_mask = priv->mask.data();
    out->device_ms = v.device_ms;
    out->_priv = priv;
    amc_verify_result_free(&v);
    return AMC_OK;
}

void amc_ransac_result_free(amc_ransac_result* r) {
    if (!r) return;
    delete static_cast<RansacPriv*>(r->_priv);
    std::memset(r, 0, sizeof *r);
}

int amc_squared_sampson_error(amc_ctx* c, const double* points1, const double* points2, size_t n,
                              const double E[9], double* out) {
    if (!c) return fail(AMC_E_INVALID, "amc_squared_sampson_error: ctx is NULL");
    if (n == 0) return AMC_OK;
    if (!points1 || !points2 || !E || !out) return fail(AMC_E_INVALID, "amc_squared_sampson_error: NULL argument");
    HIPCHK(hipSetDevice(c->device));
    DevBuf<double> buf;
    HIPCHK(buf.ensure(5 * n + 16));
    double* d1 = buf.p;
    double* d2 = d1 + 2 * n;
    double* dout = d2 + 2 * n;
    double* dE = dout + n;
    hipStream_t st = c->stream;
    int rc = AMC_OK;
    auto chk = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == AMC_OK) rc = fail(AMC_E_HIP, "amc_squared_sampson_error: %s: %s", what, hipGetErrorString(e));
    };
    chk(hipMemcpyAsync(d1, points1, 2 * n * sizeof(double), hipMemcpyHostToDevice, st), "copy points1");
    chk(hipMemcpyAsync(d2, points2, 2 * n * sizeof(double), hipMemcpyHostToDevice, st), "copy points2");
    chk(hipMemcpyAsync(dE, E, 9 * sizeof(double), hipMemcpyHostToDevice, st), "copy E");
    if (rc == AMC_OK) chk(launch_sampson(d1, d2, n, dE, dout, st), "launch");
    if (rc == AMC_OK) chk(hipMemcpyAsync(out, dout, n * sizeof(double), hipMemcpyDeviceToHost, st), "copy out");
    chk(hipStreamSynchronize(st), "sync");
    buf.release();
    return rc;
}

int amc_homography_decomposition(amc_ctx* c, const double H[9], const double K1[9], const double K2[9],
                                 const double* points1, const double* points2, size_t n, double R[9], double t[3],
                                 double normal[3], double* points3D, uint64_t* num_points3D) {
    if (!c) return fail(AMC_E_INVALID, "amc_homography_decomposition: ctx is NULL");
    if (!H || !K1 || !K2 || !R || !t || !normal || !num_points3D || (n > 0 && (!points1 || !points2 || !points3D)))
        return fail(AMC_E_INVALID, "amc_homography_decomposition: NULL argument");
    if (n > 0xFFFFFFFFull / 4) return fail(AMC_E_INVALID, "amc_homography_decomposition: too many points");
    HIPCHK(hipSetDevice(c->device));
    DevBuf<double> buf;
    HIPCHK(buf.ensure(7 * n + 27 + 16 + 8));
    double* d1 = buf.p;
    double* d2 = d1 + 2 * n;
    double* dX = d2 + 2 * n;
    double* din = dX + 3 * n;
    double* dout = din + 27;
    hipStream_t st = c->stream;
    int rc = AMC_OK;
    auto chk = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == AMC_OK) rc = fail(AMC_E_HIP, "amc_homography_decomposition: %s: %s", what, hipGetErrorString(e));
    };
    double in[27], o[16];
    std::memcpy(in, H, 9 * sizeof(double));
    std::memcpy(in + 9, K1, 9 * sizeof(double));
    std::memcpy(in + 18, K2, 9 * sizeof(double));
    if (n) {
        chk(hipMemcpyAsync(d1, points1, 2 * n * sizeof(double), hipMemcpyHostToDevice, st), "copy points1");
        chk(hipMemcpyAsync(d2, points2, 2 * n * sizeof(double), hipMemcpyHostToDevice, st), "copy points2");
    }
    chk(hipMemcpyAsync(din, in, sizeof in, hipMemcpyHostToDevice, st), "copy H, K1, K2");
    if (rc == AMC_OK) chk(launch_homography_decomposition(din, d1, d2, (uint32_t)n, dout, dX, st), "launch");
    if (rc == AMC_OK) chk(hipMemcpyAsync(o, dout, sizeof o, hipMemcpyDeviceToHost, st), "copy out");
    chk(hipStreamSynchronize(st), "sync");
    if (rc == AMC_OK) {
        std::memcpy(R, o, 9 * sizeof(double));
        std::memcpy(t, o + 9, 3 * sizeof(double));
        std::memcpy(normal, o + 12, 3 * sizeof(double));
        const uint64_t m = (uint64_t)o[15];
        *num_points3D = m;
        if (m) {
            chk(hipMemcpyAsync(points3D, dX, 3 * m * sizeof(double), hipMemcpyDeviceToHost, st), "copy points3D");
            chk(hipStreamSynchronize(st), "sync");
        }
    }
    buf.release();
    return rc;
}

void amc_verify_result_free(amc_verify_result* r) {
    if (!r) return;
    delete static_cast<VerifyPriv*>(r->_priv);
    std::memset(r, 0, sizeof *r);
}

}  // extern "C"
