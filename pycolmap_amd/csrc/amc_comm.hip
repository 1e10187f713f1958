// amc_comm.hip - the multi-GPU exchange step behind the C ABI (include/amc.h "multi-GPU exchange").
//
// Pairs shard over the GPUs of a node; every rank matches its share; one exchange gives every rank the whole match
// graph.  The reference's surface is SiftMatchingOptions.gpu_index (/root/reference/pycolmap/pipeline/match_features.h:76-81):
// COLMAP runs one matcher thread per listed GPU and joins their outputs on the host.  Here the tables never leave
// device memory on the way: they go from where the match kernels left them (the ctx's resident table) over xGMI to
// every other GPU.
//
// Why grouped ncclSend / ncclRecv and not ncclAllGather for the rows: the tables differ in size from rank to rank
// (ncclAllGather wants equal counts: padding to the largest rank would move up to world x the bytes), and xGMI is
// point to point - seven links per GPU, no switch - so the fastest all-gather is every rank writing its rows to
// every peer at once, each transfer on its own link; a ring would serialise the whole table through one link per hop.
// The (npairs, nmatches) sizes are one small ncclAllGather (equal counts by construction).
//
// RCCL is resolved with dlopen at the first call: the copy already in the process (PyTorch ships one) or the system's.
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdarg>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "amc_internal.h"

namespace amc {
namespace {

struct Rccl {
    void* handle = nullptr;
    std::string where;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional: a failed exchange aborts, so that the peers fail instead of waiting
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// nullptr + message when RCCL cannot be found
const Rccl* rccl(std::string* why) {
    static std::mutex mu;
    static Rccl r;
    static std::string err;
    std::lock_guard<std::mutex> lock(mu);
    if (r.handle) return &r;
    // ONE RCCL per process, and never in the global symbol scope:
    //  * a copy the process already holds is used as it is - found by walking the loaded objects (PyTorch maps its own
    //    as "librccl.so" from its lib directory: no soname or path this code could guess);
    //  * otherwise the loader's search path, then the ROCm install - with RTLD_LOCAL: a copy opened RTLD_GLOBAL would
    //    interpose on one a host loads LATER (import torch after the first exchange: the two copies' allocators then
    //    free each other's blocks at exit - "double free or corruption", seen in round 5's first GPU run).
    const char* forced = std::getenv("AMC_RCCL_LIBRARY");  // (a path: for hosts that keep RCCL somewhere else)
    void* h = nullptr;
    std::string got;
    if (forced && *forced) {
        h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        got = forced;
        if (!h) {  // an explicit choice that cannot be honoured is an error, not a reason to pick another library silently
            const char* de = dlerror();  // (read once: the call clears the message)
            err = std::string("AMC_RCCL_LIBRARY=") + forced + ": " + (de ? de : "dlopen failed");
            *why = err;
            return nullptr;
        }
    }
    if (!h) {
        std::string found;
        dl_iterate_phdr(
            [](struct dl_phdr_info* info, size_t, void* data) -> int {
                const char* name = info->dlpi_name;
                if (!name || !*name) return 0;
                const char* base = std::strrchr(name, '/');
                base = base ? base + 1 : name;
                if (std::strncmp(base, "librccl.so", 10) != 0) return 0;
                *static_cast<std::string*>(data) = name;
                return 1;
            },
            &found);
        if (!found.empty()) {
            h = dlopen(found.c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
            if (h) got = found + " (already loaded)";
        }
    }
    if (!h)
        for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (h) {
                got = n;
                break;
            }
        }
    if (!h) {
        const char* de = dlerror();  // (read once: the call clears the message)
        err = std::string("RCCL not found (librccl.so.1): ") + (de ? de : "dlopen failed");
        *why = err;
        return nullptr;
    }
    Rccl t;
    t.handle = h;
    t.where = got;
    bool ok = true;
    auto sym = [&](const char* name) {
        void* p = dlsym(h, name);
        if (!p) {
            ok = false;
            err = std::string("RCCL symbol missing: ") + name;
        }
        return p;
    };
    t.GetUniqueId = reinterpret_cast<decltype(t.GetUniqueId)>(sym("ncclGetUniqueId"));
    t.CommInitRank = reinterpret_cast<decltype(t.CommInitRank)>(sym("ncclCommInitRank"));
    t.CommDestroy = reinterpret_cast<decltype(t.CommDestroy)>(sym("ncclCommDestroy"));
    t.AllGather = reinterpret_cast<decltype(t.AllGather)>(sym("ncclAllGather"));
    t.Send = reinterpret_cast<decltype(t.Send)>(sym("ncclSend"));
    t.Recv = reinterpret_cast<decltype(t.Recv)>(sym("ncclRecv"));
    t.GroupStart = reinterpret_cast<decltype(t.GroupStart)>(sym("ncclGroupStart"));
    t.GroupEnd = reinterpret_cast<decltype(t.GroupEnd)>(sym("ncclGroupEnd"));
    t.GetErrorString = reinterpret_cast<decltype(t.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) {
        *why = err;
        return nullptr;
    }
    t.CommAbort = reinterpret_cast<decltype(t.CommAbort)>(dlsym(h, "ncclCommAbort"));
    r = t;
    return &r;
}

// grow-only device / pinned buffers of a communicator (freed with it)
struct DBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = std::max<size_t>(bytes, 256);
        const hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};
struct HBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = std::max<size_t>(bytes, 256);
        const hipError_t e = hipHostMalloc(&p, want, 0);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

// One wave per pair: the pair's rows from the rank-major receive buffer to their place in the global CSR.
__global__ __launch_bounds__(256) void gather_reorder_kernel(const uint64_t* __restrict__ src_off,
                                                             const uint64_t* __restrict__ dst_off,
                                                             const uint32_t* __restrict__ cnt, size_t npairs,
                                                             const uint2* __restrict__ src, uint2* __restrict__ dst) {
    const size_t p = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= npairs) return;
    const uint32_t n = cnt[p];
    const uint2* s = src + src_off[p];
    uint2* d = dst + dst_off[p];
    for (uint32_t i = threadIdx.x & 63; i < n; i += 64) d[i] = s[i];
}

// One wave per record: record k of the rank-major receive buffer (`words` uint64 each) to position pos[k] of the
// global array (amc_allgather_pair_records).
__global__ __launch_bounds__(256) void record_reorder_kernel(const uint64_t* __restrict__ pos, size_t nrec, uint32_t words,
                                                             const uint64_t* __restrict__ src, uint64_t* __restrict__ dst) {
    const size_t k = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= nrec) return;
    const uint64_t* s = src + k * words;
    uint64_t* d = dst + pos[k] * words;
    for (uint32_t i = threadIdx.x & 63; i < words; i += 64) d[i] = s[i];
}

// The inlier matches of a verified pair, one wave per pair (amc_allgather_inlier_tables).  mask: one byte per input
// match at the input's CSR offsets (amc_verify_result.inlier_mask as pack_verify_kernel left it on the device),
// 0 = outlier, g + 1 = inlier of the g-th geometry.  Pass 0 counts; pass 1 writes the rows ordered by (byte, position) -
// ExtractInlierMatches' order, the per-geometry lists of a MULTIPLE geometry one after the other.
__global__ __launch_bounds__(256) void inlier_count_kernel(const TvgPair* __restrict__ tp, const uint64_t* __restrict__ moff,
                                                           const uint8_t* __restrict__ mask, size_t npairs,
                                                           uint32_t* __restrict__ cnt) {
    const size_t p = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= npairs) return;
    const uint32_t lane = threadIdx.x & 63, M = tp[p].M;
    const uint8_t* m = mask + moff[p];
    uint32_t n = 0;
    for (uint32_t i = lane; i < M; i += 64) n += m[i] ? 1u : 0u;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) n += __shfl_xor(n, s);
    if (lane == 0) cnt[p] = n;
}
__global__ __launch_bounds__(256) void inlier_compact_kernel(const TvgPair* __restrict__ tp, const uint64_t* __restrict__ moff,
                                                             const uint8_t* __restrict__ mask, size_t npairs,
                                                             const uint2* __restrict__ matches,
                                                             const uint64_t* __restrict__ ioff, uint2* __restrict__ dst) {
    const size_t p = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= npairs) return;
    const uint32_t lane = threadIdx.x & 63, M = tp[p].M;
    const uint8_t* m = mask + moff[p];
    const uint2* rows = matches + tp[p].match_off;
    uint2* d = dst + ioff[p];
    uint32_t hi = 0;
    for (uint32_t i = lane; i < M; i += 64) hi = max(hi, (uint32_t)m[i]);
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) hi = max(hi, (uint32_t)__shfl_xor(hi, s));
    uint32_t run = 0;
    for (uint32_t g = 1; g <= hi; ++g)
        for (uint32_t base = 0; base < M; base += 64) {
            const uint32_t i = base + lane;
            const bool in = i < M && m[i] == g;
            const unsigned long long bal = __ballot(in);
            if (in) d[run + __popcll(bal & ((1ull << lane) - 1ull))] = rows[i];
            run += (uint32_t)__popcll(bal);
        }
}

struct GatherPriv {
    std::vector<uint64_t> offsets;
};

double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// test hook (read per call; never on a hot path): AMC_COMM_FAIL_ALLOC_RANK=r makes rank r report a failed allocation
// after the size exchange - the agreement step must fail the call on every rank
bool test_fail_alloc(int rank) {
    const char* e = std::getenv("AMC_COMM_FAIL_ALLOC_RANK");
    return e && *e && std::atoi(e) == rank;
}
// point-to-point transfers are issued in pieces of at most this many 8-byte words (1 GiB): no byte count beyond
// 2^31 ever reaches a transport call, whatever a rank's table weighs.  AMC_COMM_CHUNK_WORDS (tests): a smaller piece.
uint64_t chunk_words() {
    const char* e = std::getenv("AMC_COMM_CHUNK_WORDS");
    const long long v = e ? std::atoll(e) : 0;
    return v > 0 ? (uint64_t)v : ((uint64_t)1 << 27);
}

}  // namespace
}  // namespace amc

struct amc_comm {
    amc_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
    bool broken = false;  // a transfer failed half way: the communicator was aborted, every later call returns AMC_E_STATE
    amc::DBuf d_sizes, d_meta_send, d_meta_all, d_rows_send, d_rows_all, d_global, d_src_off, d_dst_off, d_cnt;
    amc::DBuf d_rec_send, d_rec_all, d_rec_global, d_inl_cnt, d_inl_off, d_inl_rows;
    amc::HBuf h_sizes, h_meta, h_rows, h_plan, h_rec, h_inl_cnt;
};

namespace amc {
namespace {

// One exchange call on one rank: the communicator, the stream, and how failures leave.
struct Xchg {
    amc_comm* c;
    const Rccl* R;
    hipStream_t st;
    const char* who;
    int W, me;

    int fail_hip(const char* what, hipError_t e) {
        (void)hipStreamSynchronize(st);
        return api_fail(AMC_E_HIP, "%s: %s -> %s", who, what, hipGetErrorString(e));
    }
    int fail_nccl(const char* what, ncclResult_t e) {
        (void)hipStreamSynchronize(st);
        return api_fail(AMC_E_HIP, "%s: %s -> %s", who, what, R->GetErrorString(e));
    }
    // A failure this rank alone has seen, with collectives of the call still ahead of the others: the communicator is
    // aborted (ncclCommAbort), so that the peers' pending and next calls fail instead of waiting for this rank for ever.
    void poison_comm() {
        if (c->comm && R->CommAbort) (void)R->CommAbort(c->comm);
        else if (c->comm) (void)R->CommDestroy(c->comm);
        c->comm = nullptr;
        c->broken = true;
    }

    // all-gather of n uint64 per rank, host to host (the sizes; the agreement words)
    int gather_words(const uint64_t* mine, size_t n, uint64_t* all) {
        const size_t bytes = sizeof(uint64_t) * n * (size_t)(W + 1);
        hipError_t e = c->h_sizes.ensure(bytes);
        if (e == hipSuccess) e = c->d_sizes.ensure(bytes);
        if (e != hipSuccess) {
            poison_comm();
            return fail_hip("size buffers", e);
        }
        uint64_t* hs = static_cast<uint64_t*>(c->h_sizes.p);
        uint64_t* ds = static_cast<uint64_t*>(c->d_sizes.p);
        std::memcpy(hs, mine, n * sizeof(uint64_t));
        e = memcpy_async(ds, hs, n * sizeof(uint64_t), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) {
            poison_comm();
            return fail_hip("H2D of the sizes", e);
        }
        const ncclResult_t ne = R->AllGather(ds, ds + n, n, ncclUint64, c->comm, st);
        if (ne != ncclSuccess) {
            poison_comm();
            return fail_nccl("ncclAllGather", ne);
        }
        e = memcpy_async(hs + n, ds + n, n * sizeof(uint64_t) * W, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            poison_comm();
            return fail_hip("D2H of the sizes", e);
        }
        std::memcpy(all, hs + n, n * sizeof(uint64_t) * W);
        return AMC_OK;
    }

    // Every rank says whether its part of the preparation went through (0) or not; all return the same verdict:
    // -1 = everybody is fine, else the first rank that is not.  An AMC_E_* code when the agreement itself failed.
    int agree(bool ok, int* first_bad) {
        std::vector<uint64_t> all((size_t)W, 0);
        const uint64_t mine = ok ? 0 : 1;
        *first_bad = -1;
        if (int rc = gather_words(&mine, 1, all.data())) return rc;
        for (int r = 0; r < W; ++r)
            if (all[r]) {
                *first_bad = r;
                break;
            }
        return AMC_OK;
    }

    // Each rank's `disp[r+1]-disp[r]` words (uint64) from `send` to slot disp[me] of every peer's `recv`: grouped
    // ncclSend / ncclRecv, device memory to device memory.  ncclGroupEnd is always reached; a transport error aborts
    // the communicator.
    int exchange(const uint64_t* send, uint64_t* recv, const std::vector<uint64_t>& disp, uint64_t* sent, uint64_t* received) {
        const uint64_t mine = disp[me + 1] - disp[me];
        if (mine) {
            const hipError_t e = memcpy_async(recv + disp[me], send, mine * sizeof(uint64_t), hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) {
                if (W > 1) poison_comm();
                return fail_hip("device copy of this rank's share", e);
            }
        }
        if (W == 1) return AMC_OK;
        const uint64_t piece = chunk_words();
        ncclResult_t ne = R->GroupStart();
        if (ne != ncclSuccess) {
            poison_comm();
            return fail_nccl("ncclGroupStart", ne);
        }
        const char* what = nullptr;
        for (int r = 0; r < W && ne == ncclSuccess; ++r) {
            if (r == me) continue;
            const uint64_t theirs = disp[r + 1] - disp[r];
            for (uint64_t o = 0; o < mine && ne == ncclSuccess; o += piece) {
                ne = R->Send(send + o, (size_t)std::min(piece, mine - o), ncclUint64, r, c->comm, st);
                what = "ncclSend";
            }
            if (ne == ncclSuccess && sent) *sent += mine;
            for (uint64_t o = 0; o < theirs && ne == ncclSuccess; o += piece) {
                ne = R->Recv(recv + disp[r] + o, (size_t)std::min(piece, theirs - o), ncclUint64, r, c->comm, st);
                what = "ncclRecv";
            }
            if (ne == ncclSuccess && received) *received += theirs;
        }
        const ncclResult_t ge = R->GroupEnd();  // (also after a failed Send / Recv: the thread's group must not stay open)
        if (ne == ncclSuccess && ge != ncclSuccess) {
            ne = ge;
            what = "ncclGroupEnd";
        }
        if (ne != ncclSuccess) {
            poison_comm();
            return fail_nccl(what, ne);
        }
        return AMC_OK;
    }
};

// The core of amc_allgather_match_tables / amc_allgather_inlier_tables: rows_dev = this rank's rows in device memory
// (nullptr with rows_host given: uploaded first).  `bad` = what is wrong with this rank's arguments (the rank still
// takes part in the size exchange, with a poison value: every rank returns the error together).
int gather_tables(const char* who, amc_ctx* ctx, amc_comm* c, const uint64_t* pair_index, size_t npairs_local,
                  const uint64_t* offsets, const uint64_t* rows_dev, const uint32_t* rows_host, const char* bad,
                  int download, amc_gathered_tables* out) {
    std::string why;
    const Rccl* R = rccl(&why);
    if (!R) return api_fail(AMC_E_HIP, "%s: %s", who, why.c_str());
    if (c->broken || !c->comm)
        return api_fail(AMC_E_STATE, "%s: the communicator was aborted by a failed exchange; create a new one", who);
    const CtxView v = ctx_view(ctx);
    Xchg x{c, R, v.stream, who, c->world, c->rank};
    const int W = c->world, me = c->rank;
    hipStream_t st = v.stream;
    {
        const hipError_t e = hipSetDevice(c->device);
        if (e != hipSuccess && !bad) bad = "hipSetDevice failed";
    }
    const uint64_t nm_local = bad ? 0 : offsets[npairs_local] - offsets[0];
    const auto t_all = std::chrono::steady_clock::now();
    auto t0 = t_all;

    // ---- 1. sizes: (npairs, nmatches) of every rank --------------------------------------------------------------
    constexpr uint64_t kPoison = ~0ull;
    std::vector<uint64_t> all((size_t)2 * W, 0);
    {
        const uint64_t mine[2] = {bad ? kPoison : (uint64_t)npairs_local, bad ? kPoison : nm_local};
        if (int rc = x.gather_words(mine, 2, all.data())) return rc;
    }
    for (int r = 0; r < W; ++r)
        if (all[2 * r] == kPoison) {
            if (r == me) return api_fail(AMC_E_INVALID, "%s: %s", who, bad);
            return api_fail(AMC_E_INVALID, "%s: rank %d reported invalid arguments", who, r);
        }
    std::vector<uint64_t> pair_disp((size_t)W + 1, 0), row_disp((size_t)W + 1, 0);
    for (int r = 0; r < W; ++r) {
        pair_disp[r + 1] = pair_disp[r] + all[2 * r];
        row_disp[r + 1] = row_disp[r] + all[2 * r + 1];
    }
    const uint64_t total_pairs = pair_disp[W], total_rows = row_disp[W];
    out->sizes_ms = ms_since(t0);
    t0 = std::chrono::steady_clock::now();

    // ---- 2. everything whose size the exchange just fixed is allocated NOW, and the ranks agree that it worked: an
    // allocation that fails on one rank (the one with less free HBM) must not leave the others inside ncclSend / ncclRecv
    GatherPriv* priv = nullptr;
    std::vector<uint8_t> seen;
    const char* alloc_what = nullptr;
    {
        const uint64_t np1 = std::max<uint64_t>(total_pairs, 1), nr1 = std::max<uint64_t>(total_rows, 1);
        hipError_t e = hipSuccess;
        auto need = [&](hipError_t r, const char* what) {
            if (e == hipSuccess && r != hipSuccess) {
                e = r;
                alloc_what = what;
            }
        };
        need(c->h_meta.ensure(sizeof(uint64_t) * np1), "pinned per-pair records");
        need(c->d_meta_send.ensure(sizeof(uint64_t) * std::max<size_t>(npairs_local, 1)), "per-pair records (send)");
        need(c->d_meta_all.ensure(sizeof(uint64_t) * np1), "per-pair records (all ranks)");
        if (!rows_dev) need(c->d_rows_send.ensure(sizeof(uint64_t) * std::max<uint64_t>(nm_local, 1)), "rows (send)");
        need(c->d_rows_all.ensure(sizeof(uint64_t) * nr1), "rows (all ranks)");
        need(c->d_global.ensure(sizeof(uint64_t) * nr1), "the global table");
        need(c->h_plan.ensure((size_t)np1 * (2 * sizeof(uint64_t) + sizeof(uint32_t))), "pinned reorder plan");
        need(c->d_src_off.ensure(sizeof(uint64_t) * np1), "reorder plan");
        need(c->d_dst_off.ensure(sizeof(uint64_t) * np1), "reorder plan");
        need(c->d_cnt.ensure(sizeof(uint32_t) * np1), "reorder plan");
        if (download && total_rows) need(c->h_rows.ensure(total_rows * sizeof(uint64_t)), "pinned rows");
        if (e != hipSuccess) (void)hipGetLastError();
        if (e == hipSuccess) {
            try {
                priv = new GatherPriv();
                priv->offsets.assign(total_pairs + 1, 0);
                seen.assign(total_pairs, 0);
            } catch (const std::bad_alloc&) {
                delete priv;
                priv = nullptr;
                alloc_what = "host memory for the global offsets";
            }
        }
        if (test_fail_alloc(me) && !alloc_what) alloc_what = "AMC_COMM_FAIL_ALLOC_RANK (test hook)";
    }
    struct PrivGuard {  // every failure below frees the offsets
        GatherPriv*& p;
        ~PrivGuard() { delete p; }
    } pg{priv};
    {
        int first_bad = -1;
        if (int rc = x.agree(alloc_what == nullptr, &first_bad)) return rc;
        if (first_bad >= 0) {
            if (first_bad == me || alloc_what)
                return api_fail(AMC_E_NOMEM, "%s: out of memory: %s (%llu pairs, %llu rows in all)", who,
                                alloc_what ? alloc_what : "?", (unsigned long long)total_pairs, (unsigned long long)total_rows);
            return api_fail(AMC_E_NOMEM, "%s: rank %d could not allocate its buffers for %llu pairs, %llu rows", who, first_bad,
                            (unsigned long long)total_pairs, (unsigned long long)total_rows);
        }
    }

    // ---- 3. per-pair records: (global position, count), 8 bytes per pair -------------------------------------------
    // positions travel as 32 bits: 2^32 pairs is 30 times BASELINE configs[4]
    bool bad_pos = total_pairs > 0xFFFFFFFFull;
    uint64_t* hm = static_cast<uint64_t*>(c->h_meta.p);
    for (size_t p = 0; p < npairs_local; ++p) {
        const uint64_t pos = pair_index ? pair_index[p] : pair_disp[me] + p;
        if (pos >= total_pairs) bad_pos = true;
        hm[p] = (pos & 0xFFFFFFFFull) | ((offsets[p + 1] - offsets[p]) << 32);
    }
    uint64_t* dm_send = static_cast<uint64_t*>(c->d_meta_send.p);
    uint64_t* dm_all = static_cast<uint64_t*>(c->d_meta_all.p);
    // from here on a failure of this rank alone would strand the others: it aborts the communicator (Xchg::poison_comm)
    auto local_fail = [&](const char* what, hipError_t e) {
        if (W > 1) x.poison_comm();
        return x.fail_hip(what, e);
    };
    if (npairs_local) {
        const hipError_t e = memcpy_async(dm_send, hm, npairs_local * sizeof(uint64_t), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return local_fail("H2D of the per-pair records", e);
    }
    if (int rc = x.exchange(dm_send, dm_all, pair_disp, nullptr, nullptr)) return rc;
    if (total_pairs) {
        const hipError_t e = memcpy_async(hm, dm_all, total_pairs * sizeof(uint64_t), hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return local_fail("D2H of the per-pair records", e);
    }

    // ---- 4. the rows: 8 bytes per match, from where the match kernels left them --------------------------------------
    const uint64_t* rows_send = rows_dev;
    if (!rows_dev) {
        if (nm_local) {
            const hipError_t e = hipMemcpyAsync(c->d_rows_send.p, rows_host, nm_local * sizeof(uint64_t), hipMemcpyHostToDevice, st);
            if (e != hipSuccess) return local_fail("H2D of the rows", e);
        }
        rows_send = static_cast<const uint64_t*>(c->d_rows_send.p);
    }
    {
        const hipError_t e = hipStreamSynchronize(st);  // the records are on the host
        if (e != hipSuccess) return local_fail("hipStreamSynchronize", e);
    }
    out->meta_ms = ms_since(t0);
    t0 = std::chrono::steady_clock::now();
    if (int rc = x.exchange(rows_send, static_cast<uint64_t*>(c->d_rows_all.p), row_disp, &out->rows_sent, &out->rows_received))
        return rc;
    // (no collective of this call is left: from here on a rank's failure is its own)

    // ---- 5. the global CSR (host, while the rows travel) and the reorder ---------------------------------------------
    uint64_t sum_cnt = 0;
    for (uint64_t k = 0; k < total_pairs && !bad_pos; ++k) {
        const uint64_t pos = hm[k] & 0xFFFFFFFFull, cnt = hm[k] >> 32;
        if (pos >= total_pairs || seen[pos]) {
            bad_pos = true;
            break;
        }
        seen[pos] = 1;
        priv->offsets[pos + 1] = cnt;
        sum_cnt += cnt;
    }
    if (bad_pos || sum_cnt != total_rows) {  // (every rank sees the same records: they all return here)
        (void)hipStreamSynchronize(st);
        return api_fail(AMC_E_INVALID, "%s: the ranks' pair positions are not a permutation of "
                        "0 .. %llu (or their counts disagree with their tables)", who, (unsigned long long)total_pairs);
    }
    for (uint64_t g = 0; g < total_pairs; ++g) priv->offsets[g + 1] += priv->offsets[g];
    // per received record (rank-major): where its rows start in the receive buffer, where they go, how many
    uint64_t* h_src = static_cast<uint64_t*>(c->h_plan.p);
    uint64_t* h_dst = h_src + std::max<uint64_t>(total_pairs, 1);
    uint32_t* h_cnt = reinterpret_cast<uint32_t*>(h_dst + std::max<uint64_t>(total_pairs, 1));
    uint64_t run = 0;
    for (uint64_t k = 0; k < total_pairs; ++k) {
        const uint64_t pos = hm[k] & 0xFFFFFFFFull, cnt = hm[k] >> 32;
        h_src[k] = run;
        h_dst[k] = priv->offsets[pos];
        h_cnt[k] = (uint32_t)cnt;
        run += cnt;
    }
#define CHK_HIP(expr)                                                     \
    do {                                                                  \
        const hipError_t e_ = (expr);                                     \
        if (e_ != hipSuccess) return x.fail_hip(#expr, e_);               \
    } while (0)
    if (total_pairs) {
        CHK_HIP(memcpy_async(c->d_src_off.p, h_src, total_pairs * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        CHK_HIP(memcpy_async(c->d_dst_off.p, h_dst, total_pairs * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        CHK_HIP(memcpy_async(c->d_cnt.p, h_cnt, total_pairs * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    }
    CHK_HIP(hipStreamSynchronize(st));  // rows received
    out->rows_ms = ms_since(t0);
    t0 = std::chrono::steady_clock::now();
    if (total_pairs && total_rows) {
        hipLaunchKernelGGL(gather_reorder_kernel, dim3((unsigned)((total_pairs + 3) / 4)), dim3(256), 0, st,
                           static_cast<const uint64_t*>(c->d_src_off.p), static_cast<const uint64_t*>(c->d_dst_off.p),
                           static_cast<const uint32_t*>(c->d_cnt.p), (size_t)total_pairs,
                           static_cast<const uint2*>(c->d_rows_all.p), static_cast<uint2*>(c->d_global.p));
        CHK_HIP(hipGetLastError());
    }
    CHK_HIP(hipStreamSynchronize(st));
    out->reorder_ms = ms_since(t0);
    t0 = std::chrono::steady_clock::now();
    if (download && total_rows) {
        CHK_HIP(hipMemcpyAsync(c->h_rows.p, c->d_global.p, total_rows * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        CHK_HIP(hipStreamSynchronize(st));
        out->matches = static_cast<const uint32_t*>(c->h_rows.p);
    }
#undef CHK_HIP
    out->download_ms = ms_since(t0);
    out->total_ms = ms_since(t_all);
    out->npairs = (size_t)total_pairs;
    out->offsets = priv->offsets.data();
    out->matches_device = total_rows ? static_cast<const uint32_t*>(c->d_global.p) : nullptr;
    out->num_matches = total_rows;
    out->world_size = W;
    out->rank = me;
    out->_priv = priv;
    priv = nullptr;  // (the guard's reference: nothing left to free)
    return AMC_OK;
}

// amc_allgather_pair_records' core.  rec_dev: this rank's records in device memory, or nullptr with rec_host.
int gather_records(const char* who, amc_ctx* ctx, amc_comm* c, const uint64_t* pair_index, size_t npairs_local,
                   const void* rec_dev, const void* rec_host, size_t record_bytes, const char* bad, int download,
                   amc_gathered_records* out) {
    std::string why;
    const Rccl* R = rccl(&why);
    if (!R) return api_fail(AMC_E_HIP, "%s: %s", who, why.c_str());
    if (c->broken || !c->comm)
        return api_fail(AMC_E_STATE, "%s: the communicator was aborted by a failed exchange; create a new one", who);
    const CtxView v = ctx_view(ctx);
    Xchg x{c, R, v.stream, who, c->world, c->rank};
    const int W = c->world, me = c->rank;
    hipStream_t st = v.stream;
    {
        const hipError_t e = hipSetDevice(c->device);
        if (e != hipSuccess && !bad) bad = "hipSetDevice failed";
    }
    const auto t_all = std::chrono::steady_clock::now();
    const uint64_t words = record_bytes / 8;

    // ---- 1. (pairs, record size) of every rank: the sizes must agree ------------------------------------------------
    constexpr uint64_t kPoison = ~0ull;
    std::vector<uint64_t> all((size_t)2 * W, 0);
    {
        const uint64_t mine[2] = {bad ? kPoison : (uint64_t)npairs_local, bad ? kPoison : (uint64_t)record_bytes};
        if (int rc = x.gather_words(mine, 2, all.data())) return rc;
    }
    for (int r = 0; r < W; ++r)
        if (all[2 * r] == kPoison) {
            if (r == me) return api_fail(AMC_E_INVALID, "%s: %s", who, bad);
            return api_fail(AMC_E_INVALID, "%s: rank %d reported invalid arguments", who, r);
        }
    for (int r = 0; r < W; ++r)
        if (all[2 * r + 1] != all[1])
            return api_fail(AMC_E_INVALID, "%s: rank %d passes records of %llu bytes, rank 0 of %llu", who, r,
                            (unsigned long long)all[2 * r + 1], (unsigned long long)all[1]);
    std::vector<uint64_t> pair_disp((size_t)W + 1, 0), word_disp((size_t)W + 1, 0);
    for (int r = 0; r < W; ++r) {
        pair_disp[r + 1] = pair_disp[r] + all[2 * r];
        word_disp[r + 1] = pair_disp[r + 1] * words;
    }
    const uint64_t total_pairs = pair_disp[W];

    // ---- 2. allocate, agree -------------------------------------------------------------------------------------------
    std::vector<uint8_t> seen;
    const char* alloc_what = nullptr;
    {
        const uint64_t np1 = std::max<uint64_t>(total_pairs, 1);
        hipError_t e = hipSuccess;
        auto need = [&](hipError_t r, const char* what) {
            if (e == hipSuccess && r != hipSuccess) {
                e = r;
                alloc_what = what;
            }
        };
        need(c->h_meta.ensure(sizeof(uint64_t) * np1), "pinned positions");
        need(c->d_meta_send.ensure(sizeof(uint64_t) * std::max<size_t>(npairs_local, 1)), "positions (send)");
        need(c->d_meta_all.ensure(sizeof(uint64_t) * np1), "positions (all ranks)");
        if (!rec_dev) need(c->d_rec_send.ensure(record_bytes * std::max<size_t>(npairs_local, 1)), "records (send)");
        need(c->d_rec_all.ensure(record_bytes * np1), "records (all ranks)");
        need(c->d_rec_global.ensure(record_bytes * np1), "the global records");
        if (download && total_pairs) need(c->h_rec.ensure(record_bytes * np1), "pinned records");
        if (e != hipSuccess) (void)hipGetLastError();
        if (e == hipSuccess) {
            try {
                seen.assign(total_pairs, 0);
            } catch (const std::bad_alloc&) {
                alloc_what = "host memory";
            }
        }
        if (test_fail_alloc(me) && !alloc_what) alloc_what = "AMC_COMM_FAIL_ALLOC_RANK (test hook)";
    }
    {
        int first_bad = -1;
        if (int rc = x.agree(alloc_what == nullptr, &first_bad)) return rc;
        if (first_bad >= 0) {
            if (alloc_what)
                return api_fail(AMC_E_NOMEM, "%s: out of memory: %s (%llu records of %zu bytes in all)", who, alloc_what,
                                (unsigned long long)total_pairs, record_bytes);
            return api_fail(AMC_E_NOMEM, "%s: rank %d could not allocate its buffers for %llu records of %zu bytes", who,
                            first_bad, (unsigned long long)total_pairs, record_bytes);
        }
    }
    auto local_fail = [&](const char* what, hipError_t e) {
        if (W > 1) x.poison_comm();
        return x.fail_hip(what, e);
    };

    // ---- 3. positions, then the records ---------------------------------------------------------------------------------
    uint64_t* hm = static_cast<uint64_t*>(c->h_meta.p);
    for (size_t p = 0; p < npairs_local; ++p) hm[p] = pair_index ? pair_index[p] : pair_disp[me] + p;
    uint64_t* dm_send = static_cast<uint64_t*>(c->d_meta_send.p);
    uint64_t* dm_all = static_cast<uint64_t*>(c->d_meta_all.p);
    if (npairs_local) {
        const hipError_t e = memcpy_async(dm_send, hm, npairs_local * sizeof(uint64_t), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return local_fail("H2D of the positions", e);
    }
    if (int rc = x.exchange(dm_send, dm_all, pair_disp, nullptr, nullptr)) return rc;
    if (total_pairs) {
        const hipError_t e = memcpy_async(hm, dm_all, total_pairs * sizeof(uint64_t), hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return local_fail("D2H of the positions", e);
    }
    const uint64_t* rec_send = static_cast<const uint64_t*>(rec_dev);
    if (!rec_dev) {
        if (npairs_local) {
            const hipError_t e = hipMemcpyAsync(c->d_rec_send.p, rec_host, npairs_local * record_bytes, hipMemcpyHostToDevice, st);
            if (e != hipSuccess) return local_fail("H2D of the records", e);
        }
        rec_send = static_cast<const uint64_t*>(c->d_rec_send.p);
    }
    uint64_t sent = 0, received = 0;
    if (int rc = x.exchange(rec_send, static_cast<uint64_t*>(c->d_rec_all.p), word_disp, &sent, &received)) return rc;
    {
        const hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) return x.fail_hip("hipStreamSynchronize", e);
    }
    // ---- 4. the positions are a permutation (every rank sees the same list), reorder, download ---------------------------
    for (uint64_t k = 0; k < total_pairs; ++k) {
        if (hm[k] >= total_pairs || seen[hm[k]])
            return api_fail(AMC_E_INVALID, "%s: the ranks' pair positions are not a permutation of 0 .. %llu", who,
                            (unsigned long long)total_pairs);
        seen[hm[k]] = 1;
    }
#define CHK_HIP(expr)                                                     \
    do {                                                                  \
        const hipError_t e_ = (expr);                                     \
        if (e_ != hipSuccess) return x.fail_hip(#expr, e_);               \
    } while (0)
    if (total_pairs) {
        hipLaunchKernelGGL(record_reorder_kernel, dim3((unsigned)((total_pairs + 3) / 4)), dim3(256), 0, st,
                           static_cast<const uint64_t*>(c->d_meta_all.p), (size_t)total_pairs, (uint32_t)words,
                           static_cast<const uint64_t*>(c->d_rec_all.p), static_cast<uint64_t*>(c->d_rec_global.p));
        CHK_HIP(hipGetLastError());
        if (download)
            CHK_HIP(hipMemcpyAsync(c->h_rec.p, c->d_rec_global.p, total_pairs * record_bytes, hipMemcpyDeviceToHost, st));
        CHK_HIP(hipStreamSynchronize(st));
    }
#undef CHK_HIP
    out->npairs = (size_t)total_pairs;
    out->record_bytes = record_bytes;
    out->records = (download && total_pairs) ? c->h_rec.p : nullptr;
    out->records_device = total_pairs ? c->d_rec_global.p : nullptr;
    out->bytes_sent = sent * 8;
    out->bytes_received = received * 8;
    out->world_size = W;
    out->rank = me;
    out->total_ms = ms_since(t_all);
    return AMC_OK;
}

// extern "C" bodies run under this: no exception crosses the C boundary.  One thrown between two collectives of a call
// (std::bad_alloc of a small vector) would strand the peers: the communicator is aborted as for any local failure.
template <class F>
int guarded(const char* who, amc_comm* c, F&& body) {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        std::string why;
        const Rccl* R = rccl(&why);
        if (c && R && c->comm && c->world > 1) {
            if (R->CommAbort) (void)R->CommAbort(c->comm);
            c->comm = nullptr;
            c->broken = true;
        }
        return api_fail(AMC_E_NOMEM, "%s: out of host memory", who);
    } catch (const std::exception& e) {
        return api_fail(AMC_E_HIP, "%s: %s", who, e.what());
    }
}

}  // namespace
}  // namespace amc

extern "C" {

int amc_comm_unique_id(void* id) {
    using namespace amc;
    if (!id) return api_fail(AMC_E_INVALID, "amc_comm_unique_id: id is NULL");
    return guarded("amc_comm_unique_id", nullptr, [&]() -> int {
        std::string why;
        const Rccl* R = rccl(&why);
        if (!R) return api_fail(AMC_E_HIP, "amc_comm_unique_id: %s", why.c_str());
        static_assert(sizeof(ncclUniqueId) == AMC_COMM_ID_BYTES, "ncclUniqueId is AMC_COMM_ID_BYTES");
        ncclUniqueId u;
        const ncclResult_t e = R->GetUniqueId(&u);
        if (e != ncclSuccess) return api_fail(AMC_E_HIP, "ncclGetUniqueId: %s", R->GetErrorString(e));
        std::memcpy(id, &u, sizeof u);
        return AMC_OK;
    });
}

int amc_comm_create(amc_ctx* ctx, int world_size, int rank, const void* id, amc_comm** out) {
    using namespace amc;
    if (!ctx || !id || !out) return api_fail(AMC_E_INVALID, "amc_comm_create: NULL argument");
    *out = nullptr;
    if (world_size < 1 || rank < 0 || rank >= world_size)
        return api_fail(AMC_E_INVALID, "amc_comm_create: rank %d of %d", rank, world_size);
    return guarded("amc_comm_create", nullptr, [&]() -> int {
        std::string why;
        const Rccl* R = rccl(&why);
        if (!R) return api_fail(AMC_E_HIP, "amc_comm_create: %s", why.c_str());
        const CtxView v = ctx_view(ctx);
        hipError_t he = hipSetDevice(v.device);
        if (he != hipSuccess) return api_fail(AMC_E_HIP, "amc_comm_create: hipSetDevice: %s", hipGetErrorString(he));
        amc_comm* c = new (std::nothrow) amc_comm();
        if (!c) return api_fail(AMC_E_NOMEM, "amc_comm_create: out of host memory");
        c->ctx = ctx;
        c->world = world_size;
        c->rank = rank;
        c->device = v.device;
        ncclUniqueId u;
        std::memcpy(&u, id, sizeof u);
        const ncclResult_t e = R->CommInitRank(&c->comm, world_size, u, rank);
        if (e != ncclSuccess) {
            delete c;
            return api_fail(AMC_E_HIP, "ncclCommInitRank(rank %d of %d, %s): %s", rank, world_size, R->where.c_str(),
                            R->GetErrorString(e));
        }
        *out = c;
        return AMC_OK;
    });
}

void amc_comm_destroy(amc_comm* c) {
    using namespace amc;
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    std::string why;
    const Rccl* R = nullptr;
    try {
        R = rccl(&why);
    } catch (...) {
    }
    if (R && c->comm) (void)R->CommDestroy(c->comm);
    for (DBuf* b : {&c->d_sizes, &c->d_meta_send, &c->d_meta_all, &c->d_rows_send, &c->d_rows_all, &c->d_global,
                    &c->d_src_off, &c->d_dst_off, &c->d_cnt, &c->d_rec_send, &c->d_rec_all, &c->d_rec_global,
                    &c->d_inl_cnt, &c->d_inl_off, &c->d_inl_rows})
        b->release();
    for (HBuf* b : {&c->h_sizes, &c->h_meta, &c->h_rows, &c->h_plan, &c->h_rec, &c->h_inl_cnt}) b->release();
    delete c;
}

void amc_gathered_tables_free(amc_gathered_tables* t) {
    if (!t) return;
    delete static_cast<amc::GatherPriv*>(t->_priv);
    std::memset(t, 0, sizeof *t);
}

void amc_gathered_records_free(amc_gathered_records* t) {
    if (t) std::memset(t, 0, sizeof *t);  // (the buffers are the communicator's)
}

int amc_allgather_match_tables(amc_ctx* ctx, amc_comm* c, const uint64_t* pair_index, size_t npairs_local,
                               const uint64_t* offsets, const uint32_t* matches, int download,
                               amc_gathered_tables* out) {
    using namespace amc;
    const char* who = "amc_allgather_match_tables";
    if (!ctx || !c || !out) return api_fail(AMC_E_INVALID, "%s: NULL ctx / comm / out", who);
    std::memset(out, 0, sizeof *out);
    if (c->ctx != ctx) return api_fail(AMC_E_INVALID, "%s: the comm belongs to another ctx", who);
    return guarded(who, c, [&]() -> int {
        const CtxView v = ctx_view(ctx);
        // A rank whose arguments are bad must not leave the others waiting inside a collective: it takes part in the
        // size exchange with a poison value, and every rank returns the error together.
        const char* bad = nullptr;
        if (!offsets) bad = "offsets is NULL";
        else if (offsets[0] != 0) bad = "offsets[0] != 0";
        for (size_t p = 0; !bad && p < npairs_local; ++p)
            if (offsets[p + 1] < offsets[p] || offsets[p + 1] - offsets[p] > 0xFFFFFFFFull)
                bad = "offsets not monotone (or a pair with 2^32 matches)";
        const uint64_t nm_local = bad ? 0 : offsets[npairs_local];
        if (!bad && !matches && nm_local != v.resident_rows) bad = "offsets[npairs] differs from the ctx's resident match table";
        if (!bad && !matches && nm_local && !v.resident) bad = "no resident match table";
        return gather_tables(who, ctx, c, pair_index, npairs_local, offsets,
                             matches ? nullptr : reinterpret_cast<const uint64_t*>(v.resident), matches, bad, download, out);
    });
}

int amc_allgather_pair_records(amc_ctx* ctx, amc_comm* c, const uint64_t* pair_index, size_t npairs_local,
                               const void* records, size_t record_bytes, int download, amc_gathered_records* out) {
    using namespace amc;
    const char* who = "amc_allgather_pair_records";
    if (!ctx || !c || !out) return api_fail(AMC_E_INVALID, "%s: NULL ctx / comm / out", who);
    std::memset(out, 0, sizeof *out);
    if (c->ctx != ctx) return api_fail(AMC_E_INVALID, "%s: the comm belongs to another ctx", who);
    return guarded(who, c, [&]() -> int {
        const VerifyResident vr = verify_resident(ctx);
        const char* bad = nullptr;
        if (record_bytes == 0 || record_bytes % 8 != 0 || record_bytes > ((size_t)1 << 20)) bad = "record_bytes must be a multiple of 8 (at most 1 MiB)";
        const void* rec_dev = nullptr;
        if (!bad && !records) {  // the two-view geometries of the last verification call, where pack_verify_kernel left them
            if (record_bytes != sizeof(amc_tvg)) bad = "records is NULL (= the resident verification records) but record_bytes != sizeof(amc_tvg)";
            else if (!vr.tvg && npairs_local) bad = "records is NULL and the ctx holds no resident verification records";
            else if (vr.npairs != npairs_local) bad = "records is NULL and the last verification call had another number of pairs";
            rec_dev = vr.tvg;
        }
        return gather_records(who, ctx, c, pair_index, npairs_local, rec_dev, records, record_bytes, bad, download, out);
    });
}

int amc_allgather_inlier_tables(amc_ctx* ctx, amc_comm* c, const uint64_t* pair_index, size_t npairs_local, int download,
                                amc_gathered_tables* out) {
    using namespace amc;
    const char* who = "amc_allgather_inlier_tables";
    if (!ctx || !c || !out) return api_fail(AMC_E_INVALID, "%s: NULL ctx / comm / out", who);
    std::memset(out, 0, sizeof *out);
    if (c->ctx != ctx) return api_fail(AMC_E_INVALID, "%s: the comm belongs to another ctx", who);
    return guarded(who, c, [&]() -> int {
        const CtxView v = ctx_view(ctx);
        const VerifyResident vr = verify_resident(ctx);
        hipStream_t st = v.stream;
        const char* bad = nullptr;
        if (vr.npairs != npairs_local) bad = "npairs_local differs from the last verification call's number of pairs";
        else if (npairs_local && (!vr.tp || !vr.moff)) bad = "the ctx holds no resident verification result";
        std::vector<uint64_t> ioff(npairs_local + 1, 0);
        // this rank's inlier lists, compacted on the device: counts (4 bytes per pair to the host), offsets, rows
        if (!bad && npairs_local) {
            hipError_t e = hipSetDevice(c->device);
            if (e == hipSuccess) e = c->d_inl_cnt.ensure(npairs_local * sizeof(uint32_t));
            if (e == hipSuccess) e = c->h_inl_cnt.ensure(npairs_local * sizeof(uint32_t));
            if (e == hipSuccess) e = c->d_inl_off.ensure((npairs_local + 1) * sizeof(uint64_t));
            if (e == hipSuccess) {
                hipLaunchKernelGGL(inlier_count_kernel, dim3((unsigned)((npairs_local + 3) / 4)), dim3(256), 0, st, vr.tp, vr.moff,
                                   vr.mask, npairs_local, static_cast<uint32_t*>(c->d_inl_cnt.p));
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(c->h_inl_cnt.p, c->d_inl_cnt.p, npairs_local * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e == hipSuccess) {
                const uint32_t* hc = static_cast<const uint32_t*>(c->h_inl_cnt.p);
                for (size_t p = 0; p < npairs_local; ++p) ioff[p + 1] = ioff[p] + hc[p];
                e = c->d_inl_rows.ensure(std::max<uint64_t>(ioff[npairs_local], 1) * sizeof(uint64_t));
            }
            if (e == hipSuccess && ioff[npairs_local]) {
                e = hipMemcpyAsync(c->d_inl_off.p, ioff.data(), (npairs_local + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st);
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(inlier_compact_kernel, dim3((unsigned)((npairs_local + 3) / 4)), dim3(256), 0, st, vr.tp,
                                       vr.moff, vr.mask, npairs_local, reinterpret_cast<const uint2*>(vr.matches),
                                       static_cast<const uint64_t*>(c->d_inl_off.p), static_cast<uint2*>(c->d_inl_rows.p));
                    e = hipGetLastError();
                }
                if (e == hipSuccess) e = hipStreamSynchronize(st);  // (ioff, a pageable source, is read by the copy above)
            }
            if (e != hipSuccess) {
                (void)hipGetLastError();
                bad = "compacting the inlier matches on the device failed";
            }
        }
        return gather_tables(who, ctx, c, pair_index, npairs_local, ioff.data(), static_cast<const uint64_t*>(c->d_inl_rows.p),
                             nullptr, bad, download, out);
    });
}

}  // extern "C"
