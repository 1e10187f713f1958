// slot_arena.h - the sub-allocator behind the image slots' device memory (amc_api.hip instantiates it over hipMalloc /
// hipFree; tests/shim/slot_arena_fuzz.cc over malloc, to fuzz the block bookkeeping on the CPU).
#pragma once

#include <algorithm>
#include <cstddef>
#include <map>
#include <unordered_map>
#include <vector>

namespace amc {

// Device memory of the image slots (descriptors, keypoints, grids): sub-allocated from a few large slabs.  A
// 500-image database is 1,500 buffers; through hipMalloc / hipFree each costs 50-100 us at upload time and again at
// teardown (measured: 100 ms of a 700 ms match_exhaustive call in amc_ctx_destroy alone).  Blocks are 256-byte
// aligned.  Free blocks are kept by address and by size: an allocation takes the best fit and leaves the rest of the
// block free, a freed block merges with its free neighbours (a context that re-uploads its slots with growing or
// varied sizes no longer strands memory: ADVICE r4), slabs nothing lives in go back to the driver at amc_ctx_trim,
// all of them when the context is destroyed or its slots are released (amc_ctx_reserve_slots).  Calls on one context
// are serialised by its owner (include/amc.h), so no lock.
template <class Raw>
struct SlotArenaT {
    static constexpr size_t kSlabBytes = (size_t)256 << 20;
    struct Slab {
        char* p;
        size_t cap;
    };
    std::vector<Slab> slabs;
    std::map<char*, size_t> idle;               // free blocks by address (neighbours of one slab merge)
    std::multimap<size_t, char*> idle_by_size;  // the same blocks by size (best fit)
    std::unordered_map<void*, size_t> live;     // blocks handed out
    void idle_insert(char* p, size_t n) {
        idle.emplace(p, n);
        idle_by_size.emplace(n, p);
    }
    void idle_erase(std::map<char*, size_t>::iterator it) {
        auto range = idle_by_size.equal_range(it->second);
        for (auto q = range.first; q != range.second; ++q)
            if (q->second == it->first) {
                idle_by_size.erase(q);
                break;
            }
        idle.erase(it);
    }
    const Slab* slab_of(const char* p) const {
        for (const Slab& b : slabs)
            if (p >= b.p && p < b.p + b.cap) return &b;
        return nullptr;
    }
    int alloc(void** out, size_t bytes) {  // 0, or the raw allocator's error code
        bytes = std::max<size_t>((bytes + 255) / 256 * 256, 256);
        auto fit = idle_by_size.lower_bound(bytes);  // best fit; what is left of the block stays free
        if (fit == idle_by_size.end()) {
            Slab nb{nullptr, std::max(bytes, kSlabBytes)};
            int e = Raw::alloc(reinterpret_cast<void**>(&nb.p), nb.cap);
            if (e != 0 && nb.cap > bytes) {  // no room for a whole slab: exactly what is asked for
                Raw::clear_error();  // (the failed attempt is not this call's status: the runtime keeps the last error)
                nb.cap = bytes;
                e = Raw::alloc(reinterpret_cast<void**>(&nb.p), nb.cap);
            }
            if (e != 0) return e;
            slabs.push_back(nb);
            idle_insert(nb.p, nb.cap);
            fit = idle_by_size.lower_bound(bytes);
        }
        char* p = fit->second;
        const size_t have = fit->first;
        idle_erase(idle.find(p));
        if (have > bytes) idle_insert(p + bytes, have - bytes);
        live[p] = bytes;
        *out = p;
        return 0;
    }
    template <class T>
    int alloc(T** out, size_t bytes) {
        void* p = nullptr;
        const int e = alloc(&p, bytes);
        *out = static_cast<T*>(p);
        return e;
    }
    // The caller has synchronised with whatever read the block - the stream of the call AND, for blocks verification or
    // a match call's copies touched, the aux and copy streams (hipFree's implicit device-wide wait is gone with it).
    void free(void* vp) {
        if (!vp) return;
        auto it = live.find(vp);
        if (it == live.end()) return;
        char* p = static_cast<char*>(vp);
        size_t n = it->second;
        live.erase(it);
        const Slab* sb = slab_of(p);
        // merge with the free neighbours inside the same slab
        auto next = idle.lower_bound(p);
        if (next != idle.end() && next->first == p + n && sb && next->first < sb->p + sb->cap) {
            n += next->second;
            idle_erase(next);
        }
        auto prev = idle.lower_bound(p);
        if (prev != idle.begin()) {
            --prev;
            if (prev->first + prev->second == p && sb && prev->first >= sb->p) {
                p = prev->first;
                n += prev->second;
                idle_erase(prev);
            }
        }
        idle_insert(p, n);
    }
    // slabs nothing lives in go back to the driver (amc_ctx_trim)
    void release_idle_slabs() {
        for (size_t k = 0; k < slabs.size();) {
            auto it = idle.find(slabs[k].p);
            if (it != idle.end() && it->second == slabs[k].cap) {
                idle_erase(it);
                Raw::free(slabs[k].p);
                slabs.erase(slabs.begin() + k);
            } else {
                ++k;
            }
        }
    }
    void release_all() {
        for (Slab& b : slabs) Raw::free(b.p);
        slabs.clear();
        idle.clear();
        idle_by_size.clear();
        live.clear();
    }
};

}  // namespace amc
