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
        err = std::string("RCCL not found (librccl.so.1): ") + (dlerror() ? dlerror() : "dlopen failed");
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

struct GatherPriv {
    std::vector<uint64_t> offsets;
};

double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace
}  // namespace amc

struct amc_comm {
    amc_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
    amc::DBuf d_sizes, d_meta_send, d_meta_all, d_rows_send, d_rows_all, d_global, d_src_off, d_dst_off, d_cnt;
    amc::HBuf h_sizes, h_meta, h_rows, h_plan;
};

extern "C" {

int amc_comm_unique_id(void* id) {
    using namespace amc;
    if (!id) return api_fail(AMC_E_INVALID, "amc_comm_unique_id: id is NULL");
    std::string why;
    const Rccl* R = rccl(&why);
    if (!R) return api_fail(AMC_E_HIP, "amc_comm_unique_id: %s", why.c_str());
    static_assert(sizeof(ncclUniqueId) == AMC_COMM_ID_BYTES, "ncclUniqueId is AMC_COMM_ID_BYTES");
    ncclUniqueId u;
    const ncclResult_t e = R->GetUniqueId(&u);
    if (e != ncclSuccess) return api_fail(AMC_E_HIP, "ncclGetUniqueId: %s", R->GetErrorString(e));
    std::memcpy(id, &u, sizeof u);
    return AMC_OK;
}

int amc_comm_create(amc_ctx* ctx, int world_size, int rank, const void* id, amc_comm** out) {
    using namespace amc;
    if (!ctx || !id || !out) return api_fail(AMC_E_INVALID, "amc_comm_create: NULL argument");
    *out = nullptr;
    if (world_size < 1 || rank < 0 || rank >= world_size)
        return api_fail(AMC_E_INVALID, "amc_comm_create: rank %d of %d", rank, world_size);
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
}

void amc_comm_destroy(amc_comm* c) {
    using namespace amc;
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    std::string why;
    const Rccl* R = rccl(&why);
    if (R && c->comm) (void)R->CommDestroy(c->comm);
    for (DBuf* b : {&c->d_sizes, &c->d_meta_send, &c->d_meta_all, &c->d_rows_send, &c->d_rows_all, &c->d_global,
                    &c->d_src_off, &c->d_dst_off, &c->d_cnt})
        b->release();
    for (HBuf* b : {&c->h_sizes, &c->h_meta, &c->h_rows, &c->h_plan}) b->release();
    delete c;
}

void amc_gathered_tables_free(amc_gathered_tables* t) {
    if (!t) return;
    delete static_cast<amc::GatherPriv*>(t->_priv);
    std::memset(t, 0, sizeof *t);
}

int amc_allgather_match_tables(amc_ctx* ctx, amc_comm* c, const uint64_t* pair_index, size_t npairs_local,
                               const uint64_t* offsets, const uint32_t* matches, int download,
                               amc_gathered_tables* out) {
    using namespace amc;
    if (!ctx || !c || !out) return api_fail(AMC_E_INVALID, "amc_allgather_match_tables: NULL ctx / comm / out");
    std::memset(out, 0, sizeof *out);
    if (c->ctx != ctx) return api_fail(AMC_E_INVALID, "amc_allgather_match_tables: the comm belongs to another ctx");
    if (!offsets) return api_fail(AMC_E_INVALID, "amc_allgather_match_tables: offsets is NULL");
    std::string why;
    const Rccl* R = rccl(&why);
    if (!R) return api_fail(AMC_E_HIP, "amc_allgather_match_tables: %s", why.c_str());
    const CtxView v = ctx_view(ctx);
    const int W = c->world, me = c->rank;
    hipStream_t st = v.stream;
    const uint64_t nm_local = offsets[npairs_local] - offsets[0];
    // A rank whose arguments are bad must not leave the others waiting inside a collective: it takes part in the
    // size exchange with a poison value, and every rank returns the error together.
    const char* bad = nullptr;
    if (offsets[0] != 0) bad = "offsets[0] != 0";
    for (size_t p = 0; !bad && p < npairs_local; ++p)
        if (offsets[p + 1] < offsets[p] || offsets[p + 1] - offsets[p] > 0xFFFFFFFFull) bad = "offsets not monotone (or a pair with 2^32 matches)";
    if (!bad && !matches && nm_local != v.resident_rows) bad = "offsets[npairs] differs from the ctx's resident match table";
    if (!bad && !matches && nm_local && !v.resident) bad = "no resident match table";

#define CHK_HIP(expr)                                                                                          \
    do {                                                                                                       \
        const hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) {                                                                                \
            (void)hipStreamSynchronize(st);                                                                    \
            return api_fail(AMC_E_HIP, "amc_allgather_match_tables: %s -> %s", #expr, hipGetErrorString(e_));  \
        }                                                                                                      \
    } while (0)
#define CHK_NCCL(expr)                                                                                         \
    do {                                                                                                       \
        const ncclResult_t e_ = (expr);                                                                        \
        if (e_ != ncclSuccess) {                                                                               \
            (void)hipStreamSynchronize(st);                                                                    \
            return api_fail(AMC_E_HIP, "amc_allgather_match_tables: %s -> %s", #expr, R->GetErrorString(e_));  \
        }                                                                                                      \
    } while (0)

    CHK_HIP(hipSetDevice(c->device));
    const auto t_all = std::chrono::steady_clock::now();
    auto t0 = t_all;

    // ---- 1. sizes: (npairs, nmatches) of every rank --------------------------------------------------------------
    constexpr uint64_t kPoison = ~0ull;
    CHK_HIP(c->h_sizes.ensure(sizeof(uint64_t) * 2 * (size_t)(W + 1)));
    CHK_HIP(c->d_sizes.ensure(sizeof(uint64_t) * 2 * (size_t)(W + 1)));
    uint64_t* hs = static_cast<uint64_t*>(c->h_sizes.p);
    uint64_t* ds = static_cast<uint64_t*>(c->d_sizes.p);
    hs[0] = bad ? kPoison : npairs_local;
    hs[1] = bad ? kPoison : nm_local;
    CHK_HIP(memcpy_async(ds, hs, 2 * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    CHK_NCCL(R->AllGather(ds, ds + 2, 2, ncclUint64, c->comm, st));
    CHK_HIP(memcpy_async(hs + 2, ds + 2, 2 * sizeof(uint64_t) * W, hipMemcpyDeviceToHost, st));
    CHK_HIP(hipStreamSynchronize(st));
    const uint64_t* all = hs + 2;
    for (int r = 0; r < W; ++r)
        if (all[2 * r] == kPoison) {
            if (r == me) return api_fail(AMC_E_INVALID, "amc_allgather_match_tables: %s", bad);
            return api_fail(AMC_E_INVALID, "amc_allgather_match_tables: rank %d reported invalid arguments", r);
        }
    std::vector<uint64_t> pair_disp(W + 1, 0), row_disp(W + 1, 0);
    for (int r = 0; r < W; ++r) {
        pair_disp[r + 1] = pair_disp[r] + all[2 * r];
        row_disp[r + 1] = row_disp[r] + all[2 * r + 1];
    }
    const uint64_t total_pairs = pair_disp[W], total_rows = row_disp[W];
    out->sizes_ms = ms_since(t0);
    t0 = std::chrono::steady_clock::now();

    // ---- 2. per-pair records: (global position, count), 8 bytes per pair -------------------------------------------
    // positions travel as 32 bits: 2^32 pairs is 30 times BASELINE configs[4]
    bool bad_pos = total_pairs > 0xFFFFFFFFull;
    CHK_HIP(c->h_meta.ensure(sizeof(uint64_t) * std::max<uint64_t>(total_pairs, 1)));
    CHK_HIP(c->d_meta_send.ensure(sizeof(uint64_t) * std::max<size_t>(npairs_local, 1)));
    CHK_HIP(c->d_meta_all.ensure(sizeof(uint64_t) * std::max<uint64_t>(total_pairs, 1)));
    uint64_t* hm = static_cast<uint64_t*>(c->h_meta.p);
    for (size_t p = 0; p < npairs_local; ++p) {
        const uint64_t pos = pair_index ? pair_index[p] : pair_disp[me] + p;
        if (pos >= total_pairs) bad_pos = true;
        hm[p] = (pos & 0xFFFFFFFFull) | ((offsets[p + 1] - offsets[p]) << 32);
    }
    uint64_t* dm_send = static_cast<uint64_t*>(c->d_meta_send.p);
    uint64_t* dm_all = static_cast<uint64_t*>(c->d_meta_all.p);
    if (npairs_local) CHK_HIP(memcpy_async(dm_send, hm, npairs_local * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    auto exchange = [&](const uint64_t* send, uint64_t* recv, const std::vector<uint64_t>& disp, uint64_t* sent,
                        uint64_t* received) -> int {
        const uint64_t mine = disp[me + 1] - disp[me];
        if (mine) CHK_HIP(memcpy_async(recv + disp[me], send, mine * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
        if (W == 1) return AMC_OK;
        CHK_NCCL(R->GroupStart());
        for (int r = 0; r < W; ++r) {
            if (r == me) continue;
            const uint64_t theirs = disp[r + 1] - disp[r];
            if (mine) {
                CHK_NCCL(R->Send(send, mine, ncclUint64, r, c->comm, st));
                if (sent) *sent += mine;
            }
            if (theirs) {
                CHK_NCCL(R->Recv(recv + disp[r], theirs, ncclUint64, r, c->comm, st));
                if (received) *received += theirs;
            }
        }
        CHK_NCCL(R->GroupEnd());
        return AMC_OK;
    };
    if (int rc = exchange(dm_send, dm_all, pair_disp, nullptr, nullptr)) return rc;
    if (total_pairs) CHK_HIP(memcpy_async(hm, dm_all, total_pairs * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    out->meta_ms = 0.0;  // (ends below, with the rows in flight behind it: the stream is drained once for both)

    // ---- 3. the rows: 8 bytes per match, from where the match kernels left them --------------------------------------
    const uint64_t* rows_send = reinterpret_cast<const uint64_t*>(v.resident);
    if (matches) {
        CHK_HIP(c->d_rows_send.ensure(sizeof(uint64_t) * std::max<uint64_t>(nm_local, 1)));
        if (nm_local) CHK_HIP(hipMemcpyAsync(c->d_rows_send.p, matches, nm_local * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        rows_send = static_cast<const uint64_t*>(c->d_rows_send.p);
    }
    CHK_HIP(c->d_rows_all.ensure(sizeof(uint64_t) * std::max<uint64_t>(total_rows, 1)));
    CHK_HIP(c->d_global.ensure(sizeof(uint64_t) * std::max<uint64_t>(total_rows, 1)));
    CHK_HIP(hipStreamSynchronize(st));  // the records are on the host
    out->meta_ms = ms_since(t0);
    t0 = std::chrono::steady_clock::now();
    if (int rc = exchange(rows_send, static_cast<uint64_t*>(c->d_rows_all.p), row_disp, &out->rows_sent, &out->rows_received))
        return rc;

    // ---- 4. the global CSR (host, while the rows travel) and the reorder ---------------------------------------------
    GatherPriv* priv = new (std::nothrow) GatherPriv();
    if (!priv) {
        (void)hipStreamSynchronize(st);
        return api_fail(AMC_E_NOMEM, "amc_allgather_match_tables: out of host memory");
    }
    priv->offsets.assign(total_pairs + 1, 0);
    std::vector<uint8_t> seen(total_pairs, 0);
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
        delete priv;
        return api_fail(AMC_E_INVALID, "amc_allgather_match_tables: the ranks' pair positions are not a permutation of "
                        "0 .. %llu (or their counts disagree with their tables)", (unsigned long long)total_pairs);
    }
    for (uint64_t g = 0; g < total_pairs; ++g) priv->offsets[g + 1] += priv->offsets[g];
    // per received record (rank-major): where its rows start in the receive buffer, where they go, how many
    const size_t plan_bytes = (size_t)std::max<uint64_t>(total_pairs, 1) * (2 * sizeof(uint64_t) + sizeof(uint32_t));
    hipError_t pe = c->h_plan.ensure(plan_bytes);
    if (pe == hipSuccess) pe = c->d_src_off.ensure(sizeof(uint64_t) * std::max<uint64_t>(total_pairs, 1));
    if (pe == hipSuccess) pe = c->d_dst_off.ensure(sizeof(uint64_t) * std::max<uint64_t>(total_pairs, 1));
    if (pe == hipSuccess) pe = c->d_cnt.ensure(sizeof(uint32_t) * std::max<uint64_t>(total_pairs, 1));
    if (pe != hipSuccess) {
        (void)hipStreamSynchronize(st);
        delete priv;
        return api_fail(AMC_E_HIP, "amc_allgather_match_tables: plan buffers: %s", hipGetErrorString(pe));
    }
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
    struct PrivGuard {  // every failure below frees the offsets
        GatherPriv* p;
        ~PrivGuard() { delete p; }
    } pg{priv};
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
        CHK_HIP(c->h_rows.ensure(total_rows * sizeof(uint64_t)));
        CHK_HIP(hipMemcpyAsync(c->h_rows.p, c->d_global.p, total_rows * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        CHK_HIP(hipStreamSynchronize(st));
        out->matches = static_cast<const uint32_t*>(c->h_rows.p);
    }
    out->download_ms = ms_since(t0);
    out->total_ms = ms_since(t_all);
    out->npairs = (size_t)total_pairs;
    out->offsets = priv->offsets.data();
    out->matches_device = total_rows ? static_cast<const uint32_t*>(c->d_global.p) : nullptr;
    out->num_matches = total_rows;
    out->world_size = W;
    out->rank = me;
    out->_priv = priv;
    pg.p = nullptr;
    return AMC_OK;
#undef CHK_HIP
#undef CHK_NCCL
}

}  // extern "C"
