/*
 * oracle/match_vnni.c — the same matcher as match_oracle.c (COLMAP 3.9.1 FindBestMatchesBruteForce),
 * written the way a CPU would want it: AVX-512 VNNI (vpdpbusd) dot products, 64 byte-MACs per instruction, with the
 * top-2 scan vectorised beside them.  It is the "optimised" CPU baseline bench.py reports next to the literal port
 * (SURVEY.md section 8d(b)): the literal triple loop says what COLMAP's semantics cost when written down naively, this
 * file what the host's cores can actually do.
 *
 * TEST INFRASTRUCTURE ONLY, like everything under oracle/: only tests/ and bench.py's cpu_baseline leg call it, and
 * tests/test_oracle_match.py checks it against match_oracle.c result for result.
 *
 * vpdpbusd multiplies UNSIGNED bytes by SIGNED bytes.  Descriptors are unsigned, so one side is shifted by the zero
 * point 128:  sum_k a_k b_k = sum_k (a_k - 128) b_k + 128 sum_k b_k  - exact in int32 (|sum| <= 128 * 128 * 255).
 * (The GPU kernel uses the same identity for its int8 MFMA, pycolmap_amd/csrc/match_mfma.hip.)
 * Layout: the column image is repacked so that one 64-byte vector holds bytes 4s..4s+3 of SIXTEEN columns; a row's
 * four bytes 4s..4s+3 are broadcast against it, and after 32 steps the sixteen int32 lanes hold sixteen dot products.
 * Top-2: lane c keeps (best, index, second) of the columns c, c+16, c+32, ... in scan order - `if (d > best) ...
 * else if (d > second) ...` is two masked moves and a max - and the sixteen partial states are merged per row exactly
 * as SURVEY.md A.2's "order-independent reformulation" says (winner: largest best, lowest index among ties; second:
 * the largest of the winner's second and everybody else's best).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#define AMC_DIM 128
#define VNNI_TARGET __attribute__((target("avx512f,avx512bw,avx512vl,avx512vnni")))

int oracle_vnni_available(void) {
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vnni");
}

/* one direction: best match of every row of A (na x 128) among the columns B (nb x 128) -> matches[na] (-1: none) */
VNNI_TARGET static int one_way_vnni(const uint8_t* A, int na, const uint8_t* B, int nb, float max_ratio, float max_distance,
                                    int32_t* matches) {
    const int nt = (nb + 15) / 16;                     /* column tiles */
    const int ntp = (nt + 1) & ~1;                     /* processed two at a time */
    uint32_t* Bt = (uint32_t*)aligned_alloc(64, (size_t)ntp * 32 * 16 * sizeof(uint32_t));
    int32_t* cs128 = (int32_t*)aligned_alloc(64, (size_t)ntp * 16 * sizeof(int32_t));
    int8_t* As = (int8_t*)aligned_alloc(64, ((size_t)na * AMC_DIM + 63) / 64 * 64);
    if (!Bt || !cs128 || !As) {
        free(Bt); free(cs128); free(As);
        return -1;
    }
    memset(Bt, 0, (size_t)ntp * 32 * 16 * sizeof(uint32_t));
    memset(cs128, 0, (size_t)ntp * 16 * sizeof(int32_t));
    for (int c = 0; c < nb; ++c) {
        const uint8_t* b = B + (size_t)c * AMC_DIM;
        int32_t s = 0;
        for (int k = 0; k < AMC_DIM; ++k) s += b[k];
        cs128[c] = 128 * s;
        for (int st = 0; st < 32; ++st) memcpy(&Bt[((size_t)(c / 16) * 32 + st) * 16 + (c % 16)], b + 4 * st, 4);
    }
    for (size_t i = 0; i < (size_t)na * AMC_DIM; ++i) As[i] = (int8_t)((int)A[i] - 128);
    const float kDistNorm = 1.0f / (512.0f * 512.0f);
    const __m512i iota = _mm512_set_epi32(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    for (int r0 = 0; r0 < na; r0 += 4) {
        const int nr = na - r0 < 4 ? na - r0 : 4;
        __m512i best[4], second[4], idx[4];
        for (int r = 0; r < 4; ++r) {
            best[r] = _mm512_setzero_si512();
            second[r] = _mm512_setzero_si512();
            idx[r] = _mm512_set1_epi32(-1);
        }
        const int8_t* arow[4];
        for (int r = 0; r < 4; ++r) arow[r] = As + (size_t)(r0 + (r < nr ? r : 0)) * AMC_DIM;
        for (int t = 0; t < ntp; t += 2) {
            __m512i acc[4][2];
            const __m512i c0 = _mm512_load_si512(cs128 + (size_t)t * 16), c1 = _mm512_load_si512(cs128 + (size_t)(t + 1) * 16);
            for (int r = 0; r < 4; ++r) { acc[r][0] = c0; acc[r][1] = c1; }
            const uint32_t* b0p = Bt + (size_t)t * 32 * 16;
            const uint32_t* b1p = Bt + (size_t)(t + 1) * 32 * 16;
            for (int st = 0; st < 32; ++st) {
                const __m512i b0 = _mm512_load_si512(b0p + st * 16), b1 = _mm512_load_si512(b1p + st * 16);
                for (int r = 0; r < 4; ++r) {
                    int32_t w;
                    memcpy(&w, arow[r] + 4 * st, 4);
                    const __m512i wv = _mm512_set1_epi32(w);
                    acc[r][0] = _mm512_dpbusd_epi32(acc[r][0], b0, wv);
                    acc[r][1] = _mm512_dpbusd_epi32(acc[r][1], b1, wv);
                }
            }
            for (int h = 0; h < 2; ++h) {
                const __m512i col = _mm512_add_epi32(iota, _mm512_set1_epi32((t + h) * 16));
                for (int r = 0; r < 4; ++r) {
                    const __m512i d = acc[r][h];
                    const __mmask16 gt = _mm512_cmpgt_epi32_mask(d, best[r]);
                    second[r] = _mm512_mask_mov_epi32(_mm512_max_epi32(second[r], d), gt, best[r]);
                    best[r] = _mm512_mask_mov_epi32(best[r], gt, d);
                    idx[r] = _mm512_mask_mov_epi32(idx[r], gt, col);
                }
            }
        }
        for (int r = 0; r < nr; ++r) {
            int32_t bv[16], iv[16], sv[16];
            _mm512_storeu_si512(bv, best[r]);
            _mm512_storeu_si512(iv, idx[r]);
            _mm512_storeu_si512(sv, second[r]);
            int w = -1;
            for (int l = 0; l < 16; ++l) {
                if (iv[l] < 0) continue;                 /* the lane saw nothing > 0 */
                if (w < 0 || bv[l] > bv[w] || (bv[l] == bv[w] && iv[l] < iv[w])) w = l;
            }
            const int i1 = r0 + r;
            matches[i1] = -1;
            if (w < 0) continue;
            int32_t sec = sv[w];
            for (int l = 0; l < 16; ++l) {
                if (l != w && bv[l] > sec) sec = bv[l];
                if (sv[l] > sec) sec = sv[l];
            }
            const int32_t bst = bv[w];
            const float best_dist_normed = acosf(fminf(kDistNorm * (float)bst, 1.0f));
            if (best_dist_normed > max_distance) continue;
            const float second_best_dist_normed = acosf(fminf(kDistNorm * (float)sec, 1.0f));
            if (best_dist_normed >= max_ratio * second_best_dist_normed) continue;
            matches[i1] = iv[w];
        }
    }
    free(Bt); free(cs128); free(As);
    return 0;
}

int oracle_match_vnni(const uint8_t* d1, int n1, const uint8_t* d2, int n2, double max_ratio, double max_distance,
                      int cross_check, uint32_t* out_matches) {
    if (n1 <= 0 || n2 <= 0) return 0;
    int32_t* m12 = (int32_t*)malloc((size_t)n1 * sizeof(int32_t));
    int32_t* m21 = (int32_t*)malloc((size_t)n2 * sizeof(int32_t));
    if (!m12 || !m21) {
        free(m12); free(m21);
        return -1;
    }
    const float r = (float)max_ratio, t = (float)max_distance;
    int rc = one_way_vnni(d1, n1, d2, n2, r, t, m12);
    if (rc == 0 && cross_check) rc = one_way_vnni(d2, n2, d1, n1, r, t, m21);
    int num = rc == 0 ? 0 : -1;
    for (int i1 = 0; rc == 0 && i1 < n1; ++i1) {
        if (m12[i1] == -1) continue;
        if (cross_check && m21[m12[i1]] != i1) continue;
        out_matches[2 * num] = (uint32_t)i1;
        out_matches[2 * num + 1] = (uint32_t)m12[i1];
        ++num;
    }
    free(m12); free(m21);
    return num;
}
#else
int oracle_vnni_available(void) { return 0; }
int oracle_match_vnni(const uint8_t* d1, int n1, const uint8_t* d2, int n2, double max_ratio, double max_distance,
                      int cross_check, uint32_t* out_matches) {
    (void)d1; (void)n1; (void)d2; (void)n2; (void)max_ratio; (void)max_distance; (void)cross_check; (void)out_matches;
    return -1;
}
#endif

/* oracle_match_pairs (match_oracle.c) on the vectorised matcher: same arguments, same results */
int oracle_match_pairs_vnni(const uint8_t* arena, const uint64_t* row_offset, const uint32_t* rows, const uint32_t* slot1,
                            const uint32_t* slot2, size_t npairs, double max_ratio, double max_distance, int cross_check,
                            const uint64_t* out_offsets, uint32_t* counts, uint32_t* out_matches, int num_threads) {
    int failed = 0;
    (void)num_threads;
    if (!oracle_vnni_available()) return -2;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads > 0 ? num_threads : 1)
#endif
    for (long p = 0; p < (long)npairs; ++p) {
        const uint32_t s1 = slot1[p], s2 = slot2[p];
        const int c = oracle_match_vnni(arena + row_offset[s1] * 128, (int)rows[s1], arena + row_offset[s2] * 128,
                                        (int)rows[s2], max_ratio, max_distance, cross_check,
                                        out_matches + 2 * out_offsets[p]);
        if (c < 0) {
            failed = 1;
            counts[p] = 0;
        } else {
            counts[p] = (uint32_t)c;
        }
    }
    return failed ? -1 : 0;
}
