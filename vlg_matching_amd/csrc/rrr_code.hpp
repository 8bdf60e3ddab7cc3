// Block code of the rrr-63 variant of an index (bv_kind == 1).
//
// A block of 63 bits is stored as rrr_vector<63> stores it (include/sdsl/rrr_vector.hpp:145-237): its class k = number of ones in
// 6 bits, and an offset of ceil(log2 C(63,k)) bits that numbers the block among the C(63,k) blocks of its class -- so the headers
// and the offset stream have exactly the sizes they have there.  What differs is the ORDER in which a class is numbered.  The
// reference numbers it bit by bit (rrr_helper.hpp:304-320), which makes a decode up to 62 dependent steps
// (rrr_helper.hpp:411-460); on a GPU that is 62 dependent LDS reads per lane and a wave pays the longest of its 64 lanes.  Here a
// class is numbered by halves instead:
//
//     63 = 31 + 32,  31 = 15 + 16,  32 = 16 + 16,  15 = 7 + 8,  16 = 8 + 8        (first part = low bits)
//     offset(x; n, k) = cum_n[k][j] + offset(first; n1, j) * C(n2, k - j) + offset(second; n2, k - j),      j = ones in the first part
//     cum_n[k][j] = sum over j' < j of C(n1, j') * C(n2, k - j')                 (blocks of the class with fewer ones in front)
//     offset(x; 8 or 7, k) = how many smaller values have k ones
//
// (C(n,k) = sum_j C(n1,j) C(n2,k-j), so the offsets of a class are exactly 0 .. C(n,k)-1.)  A rank inside a block follows ONE path
// down the halves: three small table searches, three divisions by a tabulated binomial, one byte of pattern at the end -- the same
// work for every lane.  (K6, the stand-alone rrr bit-vector of kernels.hip, keeps the reference's bit-by-bit order: it is the
// parity check of rank_support_rrr itself.)
#pragma once
#include <cstdint>
#include <cstring>

#ifdef __HIPCC__
#define VLG_HD __host__ __device__ __forceinline__
#else
#define VLG_HD inline                    // the CPU unit test of the code (tests/test_rrr_code.py) compiles this header with g++
#endif

namespace vlg {

struct RrrTables {
    uint64_t cum63[64][32];        // [k][j], j = ones in bits 0..30
    uint32_t cum2[2][33][17];      // [0]: 31 = 15 + 16, [1]: 32 = 16 + 16; entries past the last possible j are ~0
    uint16_t cum3[2][17][9];       // [0]: 15 = 7 + 8,   [1]: 16 = 8 + 8
    uint32_t c32[33];              // C(32, i)
    uint32_t c16[17];
    uint32_t c8[9];
    double inv32[33];              // 1 / C(32, i): quotients are taken by multiplication and corrected by one
    float inv16[17];
    float inv8[9];
    uint8_t dec8[9][70];           // [k][i] = the i-th smallest byte with k ones (the 7-bit values come first)
    uint8_t enc8[256];             // inverse of dec8
    uint8_t space[64];             // bits of the offset of class k
};
static_assert(sizeof(RrrTables) <= 64 * 64 * 8, "the tables live in the blob region of the former binomial table");

inline void build_rrr_tables(RrrTables& t)
{
    static uint64_t C[65][65];
    memset(C, 0, sizeof C);
    for (int n = 0; n <= 64; ++n) {
        C[n][0] = 1;
        for (int k = 1; k <= n; ++k) C[n][k] = C[n - 1][k - 1] + (k <= n - 1 ? C[n - 1][k] : 0);
    }
    auto cc = [&](int n, int k) -> uint64_t { return (k < 0 || k > n) ? 0 : C[n][k]; };
    memset(&t, 0, sizeof t);
    for (int k = 0; k < 64; ++k) {
        uint64_t acc = 0;
        for (int j = 0; j < 32; ++j) { t.cum63[k][j] = acc; acc += cc(31, j) * cc(32, k - j); }
        const uint64_t c = cc(63, k);
        uint8_t bits = 0;
        while (bits < 64 && (bits == 0 ? 1ull : (1ull << bits)) < c) ++bits;     // ceil(log2 c); 0 for c == 1
        t.space[k] = c == 1 ? 0 : bits;
    }
    for (int w = 0; w < 2; ++w) {
        const int n1 = 15 + w;
        for (int k = 0; k <= 32; ++k) {
            uint64_t acc = 0;
            for (int j = 0; j < 17; ++j) {
                t.cum2[w][k][j] = j <= n1 ? (uint32_t)acc : 0xFFFFFFFFu;
                acc += cc(n1, j) * cc(16, k - j);
            }
        }
        const int m1 = 7 + w;
        for (int k = 0; k <= 16; ++k) {
            uint64_t acc = 0;
            for (int j = 0; j < 9; ++j) {
                t.cum3[w][k][j] = j <= m1 ? (uint16_t)acc : 0xFFFFu;
                acc += cc(m1, j) * cc(8, k - j);
            }
        }
    }
    for (int i = 0; i <= 32; ++i) { t.c32[i] = (uint32_t)C[32][i]; t.inv32[i] = 1.0 / (double)C[32][i]; }
    for (int i = 0; i <= 16; ++i) { t.c16[i] = (uint32_t)C[16][i]; t.inv16[i] = 1.0f / (float)C[16][i]; }
    for (int i = 0; i <= 8; ++i) { t.c8[i] = (uint32_t)C[8][i]; t.inv8[i] = 1.0f / (float)C[8][i]; }
    uint32_t fill[9] = {0};
    for (int v = 0; v < 256; ++v) {
        const int k = __builtin_popcount(v);
        t.enc8[v] = (uint8_t)fill[k];
        t.dec8[k][fill[k]++] = (uint8_t)v;
    }
}

// ---- encode: pattern -> (class, offset) -----------------------------------------------------------------------------------------
// w = 1: 16 bits as 8 + 8;  w = 0: 15 bits as 7 + 8
VLG_HD uint32_t rrr_enc16(const RrrTables& t, uint32_t v, uint32_t w, uint32_t& k)
{
    const uint32_t m1 = 7 + w, lo = v & ((1u << m1) - 1u), hi = v >> m1;
    const uint32_t k1 = (uint32_t)__builtin_popcount(lo), k2 = (uint32_t)__builtin_popcount(hi);
    k = k1 + k2;
    return (uint32_t)t.cum3[w][k][k1] + (uint32_t)t.enc8[lo] * t.c8[k2] + t.enc8[hi];
}
// w = 1: 32 bits as 16 + 16;  w = 0: 31 bits as 15 + 16
VLG_HD uint32_t rrr_enc32(const RrrTables& t, uint32_t v, uint32_t w, uint32_t& k)
{
    const uint32_t n1 = 15 + w, lo = v & ((1u << n1) - 1u), hi = v >> n1;
    uint32_t k1, k2;
    const uint32_t r1 = rrr_enc16(t, lo, w, k1), r2 = rrr_enc16(t, hi, 1, k2);
    k = k1 + k2;
    return t.cum2[w][k][k1] + r1 * t.c16[k2] + r2;
}
VLG_HD uint64_t rrr_enc63(const RrrTables& t, uint64_t bin, uint32_t& k)
{
    uint32_t k1, k2;
    const uint32_t r1 = rrr_enc32(t, (uint32_t)(bin & 0x7FFFFFFFull), 0, k1), r2 = rrr_enc32(t, (uint32_t)(bin >> 31), 1, k2);
    k = k1 + k2;
    return t.cum63[k][k1] + (uint64_t)r1 * t.c32[k2] + r2;
}

// ---- decode: ones among the first `off` bits (off < 63) of the block (k, o), and bit number `off` ------------------------------
VLG_HD uint32_t rrr_dec63(const RrrTables& t, uint32_t k, uint64_t o, uint32_t off, uint32_t& bit)
{
    // 63 = 31 + 32
    const uint64_t* row = t.cum63[k];
    uint32_t j = 0;
#pragma unroll
    for (uint32_t st = 16; st; st >>= 1) if (row[j + st] <= o) j += st;
    const uint64_t rem = o - row[j];
    const uint32_t k2 = k - j, d = t.c32[k2];
    uint64_t q = (uint64_t)((double)rem * t.inv32[k2]);
    int64_t r = (int64_t)(rem - q * d);
    if (r < 0) { --q; r += d; } else if (r >= (int64_t)d) { ++q; r -= d; }
    const bool s1 = off >= 31;
    uint32_t ones = s1 ? j : 0;
    const uint32_t ka = s1 ? k2 : j, ia = s1 ? (uint32_t)r : (uint32_t)q, wa = s1 ? 1u : 0u, oa = s1 ? off - 31 : off;
    // 31 = 15 + 16 or 32 = 16 + 16
    const uint32_t* row2 = t.cum2[wa][ka];
    uint32_t j2 = 0;
#pragma unroll
    for (uint32_t st = 8; st; st >>= 1) if (row2[j2 + st] <= ia) j2 += st;
    if (row2[16] <= ia) j2 = 16;
    const uint32_t rem2 = ia - row2[j2], kb2 = ka - j2, d2 = t.c16[kb2];
    uint32_t q2 = (uint32_t)((float)rem2 * t.inv16[kb2]);
    int32_t r2 = (int32_t)(rem2 - q2 * d2);
    if (r2 < 0) { --q2; r2 += d2; } else if (r2 >= (int32_t)d2) { ++q2; r2 -= d2; }
    const uint32_t n1 = 15 + wa;
    const bool s2 = oa >= n1;
    ones += s2 ? j2 : 0;
    const uint32_t kb = s2 ? kb2 : j2, ib = s2 ? (uint32_t)r2 : q2, wb = s2 ? 1u : wa, ob = s2 ? oa - n1 : oa;
    // 15 = 7 + 8 or 16 = 8 + 8
    const uint16_t* row3 = t.cum3[wb][kb];
    uint32_t j3 = 0;
#pragma unroll
    for (uint32_t st = 4; st; st >>= 1) if (row3[j3 + st] <= ib) j3 += st;
    if (row3[8] <= ib) j3 = 8;
    const uint32_t rem3 = ib - row3[j3], kc3 = kb - j3, d3 = t.c8[kc3];
    uint32_t q3 = (uint32_t)((float)rem3 * t.inv8[kc3]);
    int32_t r3 = (int32_t)(rem3 - q3 * d3);
    if (r3 < 0) { --q3; r3 += d3; } else if (r3 >= (int32_t)d3) { ++q3; r3 -= d3; }
    const uint32_t m1 = 7 + wb;
    const bool s3 = ob >= m1;
    ones += s3 ? j3 : 0;
    const uint32_t kc = s3 ? kc3 : j3, ic = s3 ? (uint32_t)r3 : q3, oc = s3 ? ob - m1 : ob;
    const uint32_t p = t.dec8[kc][ic];
    bit = (p >> oc) & 1u;
    return ones + (uint32_t)__builtin_popcount(p & ((1u << oc) - 1u));
}

}  // namespace vlg
