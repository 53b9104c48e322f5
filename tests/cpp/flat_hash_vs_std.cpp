// FlatHash (alvaar_amd/csrc/slam/flat_hash.hpp) against the real libstdc++ containers: the same random operation sequences on both,
// iteration order compared after every step.  Exit code 0 = identical everywhere.
#include "../../alvaar_amd/csrc/slam/flat_hash.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <unordered_map>
#include <unordered_set>

using namespace alva_slam;

static long g_checks = 0;

template <class Std, class Flat>
static bool same_order(const Std &a, const Flat &b, const char *what, long step) {
    g_checks++;
    if (a.size() != b.size() || a.bucket_count() != b.bucket_count()) {
        std::fprintf(stderr, "%s step %ld: size %zu/%zu buckets %zu/%zu\n", what, step, a.size(), b.size(), a.bucket_count(), b.bucket_count());
        return false;
    }
    int s = b.first();
    for (const auto &e: a) {
        if (s == Flat::END) return std::fprintf(stderr, "%s step %ld: flat list ends early\n", what, step), false;
        if (e.first != b.key(s) || e.second != b.val(s)) return std::fprintf(stderr, "%s step %ld: order differs\n", what, step), false;
        s = b.next(s);
    }
    return s == Flat::END;
}

static bool same_order_set(const std::unordered_set<int> &a, const FlatSet &b, const char *what, long step) {
    g_checks++;
    if (a.size() != b.size() || a.bucket_count() != b.bucket_count()) return std::fprintf(stderr, "%s step %ld: size / buckets\n", what, step), false;
    int s = b.first();
    for (int k: a) {
        if (s == FlatSet::END || k != b.key(s)) return std::fprintf(stderr, "%s step %ld: order differs\n", what, step), false;
        s = b.next(s);
    }
    return s == FlatSet::END;
}

int main(int argc, char **argv) {
    const int seeds = argc > 1 ? std::atoi(argv[1]) : 24;
    for (int seed = 0; seed < seeds; seed++) {
        std::mt19937 rng((unsigned) seed);
        // ---- map: keypoint-table life cycle (ids grow, erases, id changes = erase + insert, copies, clears)
        std::unordered_map<int, long> sm;
        FlatHash<long> fm;
        int next_id = seed % 3 == 0 ? -5 : 0;   // negative keys too: hash<int> sign-extends
        const int range = 50 + (int) (rng() % 4000);
        for (long step = 0; step < 30000; step++) {
            const unsigned op = rng() % 100;
            if (op < 45) {
                const int k = next_id++;
                const long v = (long) rng();
                sm.emplace(k, v);
                fm.insert_slot(k, v);
            } else if (op < 55) {  // re-insert an existing or random key
                const int k = next_id > 0 ? (int) (rng() % (unsigned) (next_id + 3)) - 1 : 0;
                const long v = (long) rng();
                sm.emplace(k, v);
                fm.insert_slot(k, v);
            } else if (op < 85) {
                const int k = next_id > 0 ? (int) (rng() % (unsigned) (next_id + 3)) - 1 : 0;
                sm.erase(k);
                fm.erase(k);
            } else if (op < 88 && !sm.empty()) {  // erase through an iterator
                auto it = sm.begin();
                std::advance(it, (long) (rng() % sm.size()));
                const int k = it->first;
                sm.erase(it);
                fm.erase_slot(fm.find_slot(k));
            } else if (op < 93) {  // copy (keyframe = copy of the frame) and continue on the copy
                std::unordered_map<int, long> c(sm);
                FlatHash<long> fc(fm);
                if (!same_order(c, fc, "map copy", step)) return 1;
                if (rng() & 1) {
                    sm = c;
                    fm = fc;
                }
            } else if (op < 94) {
                sm.clear();
                fm.clear();
            } else if ((int) sm.size() > range) {  // shrink phase: erase most
                for (int k = next_id - 1; k >= 0 && sm.size() > (size_t) range / 4; k -= 1 + (int) (rng() % 3)) {
                    sm.erase(k);
                    fm.erase(k);
                }
            }
            if (!same_order(sm, fm, "map", step)) return 1;
            // look-ups agree
            const int q = (int) (rng() % (unsigned) (next_id + 10)) - 5;
            auto it = sm.find(q);
            const int fs = fm.find_slot(q);
            if ((it == sm.end()) != (fs == FlatHash<long>::END) || (fs != FlatHash<long>::END && fm.val(fs) != it->second)) return std::fprintf(stderr, "find differs\n"), 1;
        }
        // ---- set: local-map life cycle (range inserts from other sets, swaps, clears)
        std::unordered_set<int> sa, sb;
        FlatSet fa, fb;
        for (long step = 0; step < 4000; step++) {
            const unsigned op = rng() % 100;
            if (op < 50) {
                const int k = (int) (rng() % 6000);
                sa.insert(k);
                fa.insert(k);
            } else if (op < 65) {
                const int k = (int) (rng() % 6000);
                sb.insert(k);
                fb.insert(k);
            } else if (op < 72) {
                const int k = (int) (rng() % 6000);
                sa.erase(k);
                fa.erase(k);
            } else if (op < 80) {  // a.insert(b.begin(), b.end())
                sa.insert(sb.begin(), sb.end());
                fa.insert(fb.begin(), fb.end());
            } else if (op < 86) {
                sa.swap(sb);
                fa.swap(fb);
            } else if (op < 88) {
                sb.clear();
                fb.clear();
            } else if (op < 92) {  // copy-construct
                std::unordered_set<int> c(sa);
                FlatSet fc(fa);
                sb = c;
                fb = fc;
            } else {
                for (int i = 0; i < 300; i++) {
                    const int k = (int) (rng() % 20000);
                    sb.insert(k);
                    fb.insert(k);
                }
            }
            if (!same_order_set(sa, fa, "set a", step) || !same_order_set(sb, fb, "set b", step)) return 1;
        }
        // ---- the inline small set (a map point's descriptor keys: keyframe ids of a sliding window, at most CAP of them): inserts of a
        // moving key range, erases, clears (bucket count and policy state survive a clear), struct copies, a fresh reset, and inserts
        // beyond the capacity, which must be refused without a trace
        {
            typedef SmallFlatSet<48, 59> Small;
            std::unordered_set<int> ss;
            Small fs;
            int base = seed % 4 == 0 ? -20 : 0;
            for (long step = 0; step < 40000; step++) {
                const unsigned op = rng() % 100;
                const int k = base + (int) (rng() % 70);
                if (op < 50) {
                    const bool fits = ss.count(k) || ss.size() < 48;
                    const int r = fs.insert(k);
                    if (fits) {
                        const bool ins = ss.insert(k).second;
                        if (r != (ins ? 1 : 0)) return std::fprintf(stderr, "small set step %ld: insert says %d\n", step, r), 1;
                    } else if (r != -1) {   // (a refusal leaves no trace: 59 buckets hold 59 keys, the growth policy had nothing to decide)
                        return std::fprintf(stderr, "small set step %ld: an insert beyond the capacity was not refused\n", step), 1;
                    }
                } else if (op < 85) {
                    if (ss.erase(k) != (fs.erase(k) ? 1u : 0u)) return std::fprintf(stderr, "small set step %ld: erase\n", step), 1;
                } else if (op < 88) {
                    ss.clear();
                    fs.clear();
                } else if (op < 92) {   // a copy continues in place of the original (mergeMapPoints walks a copy)
                    Small c = fs;
                    fs = c;
                } else if (op < 94) {   // a new map point takes the slot
                    ss = std::unordered_set<int>();
                    fs.reset();
                } else {
                    base += (int) (rng() % 3);   // the window slides
                }
                g_checks++;
                if (ss.size() != fs.size() || ss.bucket_count() != fs.bucket_count())
                    return std::fprintf(stderr, "small set step %ld: size %zu/%zu buckets %zu/%zu\n", step, ss.size(), fs.size(), ss.bucket_count(), fs.bucket_count()), 1;
                int s2 = fs.first();
                for (int key: ss) {
                    if (s2 == Small::END || key != fs.key(s2)) return std::fprintf(stderr, "small set step %ld: order differs\n", step), 1;
                    s2 = fs.next(s2);
                }
                if (s2 != Small::END) return std::fprintf(stderr, "small set step %ld: list too long\n", step), 1;
                if (fs.count(k) != ss.count(k)) return std::fprintf(stderr, "small set step %ld: count\n", step), 1;
            }
        }
    }
    std::printf("ok %ld comparisons\n", g_checks);
    return 0;
}
