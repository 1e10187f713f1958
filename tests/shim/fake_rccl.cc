// Test-only stand-in for librccl.so.1 between THREADS of one process that share ONE device (tests/test_multigpu_gpu.py,
// tests/comm_threads.py).  The round-end GPU box has a single MI355X, and RCCL refuses two ranks on one device - so the
// N > 1 logic of amc_allgather_match_tables (displacements, exact per-rank counts, the reorder into the global CSR,
// appended lists, the poisoned size exchange) would otherwise meet two ranks for the first time on the day an 8-GPU
// node appears.  libamc.so loads this file instead of RCCL when AMC_RCCL_LIBRARY names it (amc_comm.hip: the library
// resolves RCCL with dlopen); every rank is a thread with its own amc_ctx on device 0, and "the wire" is a
// device-to-device copy after a rendezvous.  Only what amc_comm.hip calls is here; semantics follow the NCCL API:
// collectives are matched by call order, point-to-point calls inside ncclGroupStart / ncclGroupEnd complete at
// ncclGroupEnd, everything is ordered after the work already on the caller's stream.
#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef struct FakeComm* ncclComm_t;
}

namespace {

struct Post {  // one rank's side of a collective / group
    const void* send = nullptr;          // all-gather: the send buffer
    struct P2p { int peer; const void* ptr; size_t bytes; };
    std::vector<P2p> sends;              // group: what this rank sends, in call order
};

struct World {
    int n = 0, joined = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<Post> posts;
    bool aborted = false;  // ncclCommAbort by any rank: every pending and later rendezvous fails on every rank
    bool barrier() {  // all n ranks; false = the world was aborted
        std::unique_lock<std::mutex> lock(mu);
        if (aborted) return false;
        const uint64_t g = generation;
        if (++arrived == n) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lock, [&] { return generation != g || aborted; });
        }
        return !aborted;
    }
    void abort() {
        std::unique_lock<std::mutex> lock(mu);
        aborted = true;
        cv.notify_all();
    }
};

std::mutex g_mu;
std::map<std::string, World*> g_worlds;
uint64_t g_next_id = 1;

size_t dtype_size(ncclDataType_t t) {
    static const size_t sz[] = {1, 1, 4, 4, 8, 8, 2, 4, 8, 2};
    return (t >= 0 && t < 10) ? sz[t] : 0;
}

struct GroupState {
    int depth = 0;
    struct Op { bool is_send; int peer; void* ptr; size_t bytes; struct FakeComm* comm; hipStream_t stream; };
    std::vector<Op> ops;
};
thread_local GroupState t_group;
thread_local struct FakeComm* t_comm = nullptr;   // the calling thread's communicator (one per thread in the tests)
thread_local hipStream_t t_stream = nullptr;      // ... and the stream of its last call

}  // namespace

struct FakeComm {
    World* world;
    int rank;
};

namespace {

// (every rank of the world passes through here for every group, also with nothing to send or receive: the rendezvous
// below counts all of them)
ncclResult_t run_group(std::vector<GroupState::Op>& ops) {
    FakeComm* c = ops.empty() ? t_comm : ops[0].comm;
    if (!c) return 3;
    World* w = c->world;
    // everything already on the caller's stream (the buffers' contents) first
    if (hipStreamSynchronize(ops.empty() ? t_stream : ops[0].stream) != hipSuccess) return 1;
    Post& mine = w->posts[c->rank];
    mine.sends.clear();
    for (const auto& op : ops)
        if (op.is_send) mine.sends.push_back(Post::P2p{op.peer, op.ptr, op.bytes});
    if (!w->barrier()) return 6;  // every rank's sends are posted (6: the world was aborted)
    ncclResult_t rc = 0;
    std::vector<size_t> next_from(w->n, 0);  // k-th recv from a peer matches that peer's k-th send to this rank
    for (const auto& op : ops) {
        if (op.is_send) continue;
        const Post& theirs = w->posts[op.peer];
        size_t seen = 0;
        const Post::P2p* match = nullptr;
        for (const auto& s : theirs.sends)
            if (s.peer == c->rank && seen++ == next_from[op.peer]) {
                match = &s;
                break;
            }
        ++next_from[op.peer];
        if (!match || match->bytes != op.bytes) {
            rc = 2;  // a send / recv pair that does not match: a protocol bug in the caller
            continue;
        }
        // on the receiver's own stream, like RCCL's recv, and drained before the rendezvous below: a device-to-device
        // hipMemcpy on the null stream may return before the copy is done, and the caller's stream (non-blocking) would
        // then read the buffer ahead of it (seen once as a stale record on a two-rank run)
        if (op.bytes && hipMemcpyAsync(op.ptr, match->ptr, op.bytes, hipMemcpyDeviceToDevice, op.stream) != hipSuccess) rc = 1;
    }
    for (const auto& op : ops)
        if (!op.is_send && hipStreamSynchronize(op.stream) != hipSuccess) rc = 1;
    if (!w->barrier()) return 6;  // nobody reuses a send buffer before its readers are done
    return rc;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::lock_guard<std::mutex> lock(g_mu);
    std::memset(id, 0, sizeof *id);
    std::snprintf(id->internal, sizeof id->internal, "fake-rccl-%llu", (unsigned long long)g_next_id++);
    return 0;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    World* w;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        World*& slot = g_worlds[std::string(id.internal)];
        if (!slot) {
            slot = new World();
            slot->n = nranks;
            slot->posts.resize(nranks);
        }
        w = slot;
        if (w->n != nranks || rank < 0 || rank >= nranks) return 3;
    }
    *comm = new FakeComm{w, rank};
    t_comm = *comm;
    w->barrier();  // collective, like ncclCommInitRank
    return 0;
}

ncclResult_t ncclCommAbort(ncclComm_t comm) {  // the peers' pending and later calls fail instead of waiting for this rank
    comm->world->abort();
    delete comm;
    return 0;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete comm;  // (worlds are leaked: a test process)
    return 0;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t c, hipStream_t s) {
    World* w = c->world;
    t_stream = s;
    const size_t bytes = count * dtype_size(dt);
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    w->posts[c->rank].send = send;
    if (!w->barrier()) return 6;
    ncclResult_t rc = 0;
    for (int r = 0; r < w->n; ++r) {
        void* dst = static_cast<char*>(recv) + (size_t)r * bytes;
        if (dst == w->posts[r].send) continue;  // in place
        if (bytes && hipMemcpyAsync(dst, w->posts[r].send, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) rc = 1;
    }
    if (hipStreamSynchronize(s) != hipSuccess) rc = 1;  // (see run_group: the copies are done before anybody moves on)
    if (!w->barrier()) return 6;
    return rc;
}

ncclResult_t ncclGroupStart() {
    ++t_group.depth;
    return 0;
}

ncclResult_t ncclGroupEnd() {
    if (t_group.depth <= 0) return 3;
    if (--t_group.depth > 0) return 0;
    std::vector<GroupState::Op> ops;
    ops.swap(t_group.ops);
    return run_group(ops);
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t s) {
    t_group.ops.push_back(GroupState::Op{true, peer, const_cast<void*>(buf), count * dtype_size(dt), c, s});
    if (t_group.depth == 0) {
        std::vector<GroupState::Op> ops;
        ops.swap(t_group.ops);
        return run_group(ops);
    }
    return 0;
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t s) {
    t_group.ops.push_back(GroupState::Op{false, peer, buf, count * dtype_size(dt), c, s});
    if (t_group.depth == 0) {
        std::vector<GroupState::Op> ops;
        ops.swap(t_group.ops);
        return run_group(ops);
    }
    return 0;
}

const char* ncclGetErrorString(ncclResult_t e) {
    switch (e) {
        case 0: return "no error";
        case 1: return "fake rccl: HIP failure";
        case 2: return "fake rccl: a send and its recv do not match";
        case 6: return "fake rccl: the communicator was aborted (ncclCommAbort on some rank)";
        default: return "fake rccl: invalid usage";
    }
}

}  // extern "C"
