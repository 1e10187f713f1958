// CPU fuzz of pycolmap_amd/csrc/slot_arena.h (the allocator behind the image slots' device memory) over a fake raw
// allocator: random alloc / free / trim sequences; after every step the live blocks must be disjoint, 256-byte
// aligned, inside a slab; the free blocks must tile the rest of every slab exactly, with no two adjacent ones left
// unmerged; a fully idle slab must go at trim.  tests/test_host_layer_cpu.py builds and runs it (also under ASan).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>

#include "../../pycolmap_amd/csrc/slot_arena.h"

static size_t g_raw_live = 0, g_fail_next = 0;
struct FakeRaw {
    static int alloc(void** p, size_t bytes) {
        if (g_fail_next && bytes == ((size_t)256 << 20)) {  // "no room for a whole slab": the exact-size retry must follow
            --g_fail_next;
            return 2;
        }
        *p = std::aligned_alloc(256, bytes);
        if (!*p) return 2;
        ++g_raw_live;
        return 0;
    }
    static void free(void* p) {
        std::free(p);
        --g_raw_live;
    }
    static void clear_error() {}
};
using Arena = amc::SlotArenaT<FakeRaw>;

static int check(const Arena& a, const char* when) {
    size_t cap = 0, live = 0, idle = 0;
    for (const auto& s : a.slabs) cap += s.cap;
    for (const auto& kv : a.live) {
        const char* p = static_cast<const char*>(kv.first);
        if (((uintptr_t)p & 255) || !a.slab_of(p) || !a.slab_of(p + kv.second - 1)) return std::printf("%s: bad live block\n", when), 1;
        live += kv.second;
    }
    const char* prev_end = nullptr;
    for (const auto& kv : a.idle) {
        if (!a.slab_of(kv.first) || a.slab_of(kv.first) != a.slab_of(kv.first + kv.second - 1)) return std::printf("%s: free block leaves its slab\n", when), 1;
        if (prev_end == kv.first && a.slab_of(prev_end - 1) == a.slab_of(kv.first)) return std::printf("%s: adjacent free blocks not merged\n", when), 1;
        for (const auto& lv : a.live) {
            const char* p = static_cast<const char*>(lv.first);
            if (p < kv.first + kv.second && kv.first < p + lv.second) return std::printf("%s: free block overlaps a live one\n", when), 1;
        }
        prev_end = kv.first + kv.second;
        idle += kv.second;
    }
    if (a.idle.size() != a.idle_by_size.size()) return std::printf("%s: the two free indexes disagree\n", when), 1;
    if (live + idle != cap) return std::printf("%s: %zu live + %zu free != %zu in slabs\n", when, live, idle, cap), 1;
    if (a.slabs.size() != g_raw_live) return std::printf("%s: slab count %zu vs raw blocks %zu\n", when, a.slabs.size(), g_raw_live), 1;
    return 0;
}

int main(int argc, char** argv) {
    const unsigned seed = argc > 1 ? (unsigned)std::atoi(argv[1]) : 1u;
    std::mt19937 rng(seed);
    Arena a;
    std::vector<std::pair<char*, size_t>> held;
    const size_t sizes[] = {1, 255, 256, 4096, 100000, 532480, 1064960, 9000000, (size_t)300 << 20};
    for (int step = 0; step < 20000; ++step) {
        const unsigned r = rng() % 100;
        if (r < 55 || held.empty()) {
            size_t n = sizes[rng() % 9];
            if (rng() % 3 == 0) n = 1 + rng() % 2000000;
            if (n > ((size_t)64 << 20) && rng() % 4) n = 1 + rng() % 4096;  // (few giant blocks)
            if (rng() % 50 == 0) g_fail_next = 1;
            char* p = nullptr;
            if (a.alloc(&p, n) != 0) return std::printf("alloc of %zu failed\n", n), 1;
            g_fail_next = 0;
            for (const auto& h : held)
                if (p < h.first + h.second && h.first < p + n) return std::printf("step %d: overlapping allocation\n", step), 1;
            held.emplace_back(p, n);
        } else if (r < 97) {
            const size_t k = rng() % held.size();
            a.free(held[k].first);
            held[k] = held.back();
            held.pop_back();
        } else {
            a.release_idle_slabs();
            for (const auto& s : a.slabs) {
                auto it = a.idle.find(s.p);
                if (it != a.idle.end() && it->second == s.cap) return std::printf("step %d: an idle slab survived the trim\n", step), 1;
            }
        }
        if (step % 64 == 0 && check(a, "sweep")) return 1;
    }
    for (auto& h : held) a.free(h.first);
    if (check(a, "all freed")) return 1;
    if (a.idle.size() != a.slabs.size()) return std::printf("everything freed: %zu free blocks in %zu slabs\n", a.idle.size(), a.slabs.size()), 1;
    a.release_idle_slabs();
    if (!a.slabs.empty() || g_raw_live) return std::printf("slabs left after the final trim\n"), 1;
    a.release_all();
    std::printf("slot arena fuzz ok (seed %u)\n", seed);
    return 0;
}
