// Host check of the searches of rhfit::OctView (fit_shared.h): lower_bound against std::lower_bound and select
// against the list of set bits, on random sorted codes / enabled words of ragged sizes.  Built and run by
// tests/test_abi.py::test_octview_searches_match_std (g++, no GPU).
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
#include <random>
#include "ransac_hip.h"
#include "fit_shared.h"
int main() {
    std::mt19937_64 g(5);
    long bad = 0, checks = 0;
    for (int trial = 0; trial < 400; trial++) {
        int64_t n = 1 + g() % (trial < 200 ? 70 : 20000);
        std::vector<uint64_t> code(n);
        uint64_t range = (trial % 3 == 0) ? 16 : (trial % 3 == 1 ? 100000 : ~0ULL);
        for (auto &c : code) c = g() % range;
        std::sort(code.begin(), code.end());
        int64_t nwords = (n + 63) / 64;
        std::vector<uint64_t> men(nwords);
        for (auto &w : men) { w = g(); if (g() % 3 == 0) w = 0; if (g() % 5 == 0) w &= g(); }
        if (n % 64) men[nwords - 1] &= (~0ULL) >> (64 - n % 64);
        std::vector<int32_t> prefix(nwords + 1);
        int32_t acc = 0;
        for (int64_t w = 0; w < nwords; w++) { prefix[w] = acc; acc += __builtin_popcountll(men[w]); }
        prefix[nwords] = acc;
        rhfit::OctView oc;
        oc.code = code.data(); oc.perm = nullptr; oc.pos = nullptr; oc.men = men.data(); oc.prefix = prefix.data();
        oc.n = n; oc.nwords = nwords; oc.depth = 5;
        for (int q = 0; q < 300; q++) {
            uint64_t key = (q % 4 == 0) ? code[g() % n] : (q % 4 == 1 ? code[g() % n] + 1 : g() % (range ? range : 1));
            if (q == 7) key = 0;
            if (q == 8) key = ~0ULL;
            int64_t exp = std::lower_bound(code.begin(), code.end(), key) - code.begin();
            int64_t got = oc.lower_bound(key);
            checks++;
            if (exp != got) { bad++; if (bad < 10) printf("lower_bound n=%ld key=%lu exp=%ld got=%ld\n", (long)n, (unsigned long)key, (long)exp, (long)got); }
        }
        // select: every rank
        std::vector<int64_t> posv;
        for (int64_t i = 0; i < n; i++) if ((men[i >> 6] >> (i & 63)) & 1) posv.push_back(i);
        for (size_t r = 1; r <= posv.size(); r += 1 + posv.size() / 500) {
            int64_t got = oc.select((int64_t)r);
            checks++;
            if (got != posv[r - 1]) { bad++; if (bad < 10) printf("select n=%ld r=%zu exp=%ld got=%ld\n", (long)n, r, (long)posv[r - 1], (long)got); }
        }
        if (!posv.empty()) { checks++; if (oc.select((int64_t)posv.size()) != posv.back()) { bad++; printf("select last\n"); } }
    }
    printf("%ld checks, %ld bad\n", checks, bad);
    return bad != 0;
}
