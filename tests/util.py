"""Shared helpers for the tests: seeded synthetic texts and brute-force checkers."""
import numpy as np


def dna_text(n, seed, probs=(0.25, 0.25, 0.25, 0.25)):
    rng = np.random.default_rng(seed)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.choice(4, size=n, p=probs)].copy()


def skewed_text(n, seed, sigma=40):
    """Zipf-ish byte text over `sigma` printable symbols (deep Huffman tree)."""
    rng = np.random.default_rng(seed)
    p = 1.0 / np.arange(1, sigma + 1) ** 1.3
    p /= p.sum()
    return (33 + rng.choice(sigma, size=n, p=p)).astype(np.uint8)


def naive_sa(text_with_sentinel):
    t = bytes(text_with_sentinel)
    return np.array(sorted(range(len(t)), key=lambda i: t[i:]), dtype=np.uint64)


def naive_occurrences(text, pat):
    t, p = bytes(text), bytes(pat)
    out, i = [], t.find(p)
    while i >= 0:
        out.append(i)
        i = t.find(p, i + 1)
    return out


def bwt_from_sa(text_with_sentinel, sa):
    t = np.asarray(text_with_sentinel, dtype=np.uint8)
    n = len(t)
    return t[(np.asarray(sa, dtype=np.int64) - 1) % n]


def reference_semantics_join(lists, lo, hi, end_len):
    """Pure-Python restatement of SURVEY Appendix C (small cases only)."""
    k = len(lists)
    p = [0] * k
    out = []
    if any(len(l) == 0 for l in lists):
        return out
    while p[0] < len(lists[0]):
        prev = lists[0][p[0]]
        stop = again = False
        for i in range(1, k):
            while p[i] < len(lists[i]) and lists[i][p[i]] < prev + lo[i - 1]:
                p[i] += 1
            if p[i] == len(lists[i]):
                stop = True
                break
            if lists[i][p[i]] > prev + hi[i - 1]:
                p[i - 1] += 1
                again = True
                break
            prev = lists[i][p[i]]
        if stop:
            break
        if again:
            continue
        out.append([int(lists[i][p[i]]) for i in range(k)])
        end = lists[k - 1][p[k - 1]] + end_len
        while p[0] < len(lists[0]) and lists[0][p[0]] < end:
            p[0] += 1
    return out
