/* Seeded synthetic workloads: the Pizza&Chili corpora are unreachable offline, so
 * every text is a splitmix64 stream with a stated seed.  Integer arithmetic only, so the same seed
 * gives the same bytes on every machine.  Host-side, no GPU involved. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t splitmix64(uint64_t* s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* iid symbols: symbol j is drawn when the 32 high bits of the stream fall below cum[j] (cum[nsym-1] = 2^32). */
void vlgw_gen_iid(uint8_t* out, uint64_t n, uint64_t seed, const uint8_t* alphabet, const uint64_t* cum, uint32_t nsym)
{
    /* table on the top 16 bits of the draw: the symbol when the whole bucket maps to one symbol, 0xFF otherwise */
    static uint8_t table[65536];
    for (uint32_t b = 0; b < 65536; ++b) {
        uint64_t lo = (uint64_t)b << 16, hi = lo + 0xFFFF;
        uint32_t jl = 0, jh = 0;
        while (jl + 1 < nsym && lo >= cum[jl]) ++jl;
        while (jh + 1 < nsym && hi >= cum[jh]) ++jh;
        table[b] = (jl == jh) ? (uint8_t)jl : 0xFF;
    }
    uint64_t s = seed;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t u = splitmix64(&s) >> 32;
        uint32_t j = table[u >> 16];
        if (j == 0xFF) { j = 0; while (j + 1 < nsym && u >= cum[j]) ++j; }
        out[i] = alphabet[j];
    }
}

/* English-like: words drawn Zipf(1.0) from a synthetic vocabulary (lengths min_len..max_len over a-z),
 * separated by single spaces. */
int vlgw_gen_zipf_words(uint8_t* out, uint64_t n, uint64_t seed, uint32_t vocab, uint32_t min_len, uint32_t max_len)
{
    uint64_t s = seed;
    uint32_t span = max_len - min_len + 1;
    uint8_t* words = (uint8_t*)malloc((size_t)vocab * max_len);
    uint8_t* lens = (uint8_t*)malloc(vocab);
    uint64_t* cum = (uint64_t*)malloc(8 * (size_t)vocab);
    if (!words || !lens || !cum) { free(words); free(lens); free(cum); return -1; }
    for (uint32_t r = 0; r < vocab; ++r) {
        lens[r] = (uint8_t)(min_len + splitmix64(&s) % span);
        for (uint32_t j = 0; j < lens[r]; ++j) words[(size_t)r * max_len + j] = (uint8_t)('a' + splitmix64(&s) % 26);
    }
    uint64_t tot = 0;
    for (uint32_t r = 0; r < vocab; ++r) { tot += (1ULL << 40) / (r + 1); cum[r] = tot; }   /* weight of rank r+1 */
    uint64_t i = 0;
    while (i < n) {
        uint64_t u = splitmix64(&s) % tot;
        uint32_t lo = 0, hi = vocab - 1;
        while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (cum[mid] > u) hi = mid; else lo = mid + 1; }
        const uint8_t* w = words + (size_t)lo * max_len;
        for (uint32_t j = 0; j < lens[lo] && i < n; ++j) out[i++] = w[j];
        if (i < n) out[i++] = ' ';
    }
    free(words); free(lens); free(cum);
    return 0;
}

/* Query sub-pattern start positions: uniform in [0, n_text - m]. */
void vlgw_gen_positions(uint64_t* out, uint64_t count, uint64_t seed, uint64_t n_text, uint64_t m)
{
    uint64_t s = seed;
    uint64_t range = n_text >= m ? n_text - m + 1 : 1;
    for (uint64_t i = 0; i < count; ++i) out[i] = splitmix64(&s) % range;
}
