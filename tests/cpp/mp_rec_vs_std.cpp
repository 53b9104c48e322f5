// The map-point record (alvaar_amd/csrc/slam/mp_rec.hpp) against the containers it stands for, through random operation sequences: the
// observing keyframes as a std::set<int> (MapPoint::observedKeyframeIds_), "keyframe kf holds the keypoint" + its positions and "has a
// descriptor for kf" + its 32 bytes as std::map<int, ...>.  After EVERY operation: the record's entries are sorted, carry exactly the union
// of the three key sets with the right flags and payloads, n_obs / n_ent are right, the descriptor bytes of the side arena sit beside
// their entries (they are shifted with them), rec_observers() is the set's ascending walk, rec_find / rec_in_kf agree.  Overflow beyond
// MP_ENT_CAP is refused and flagged.   usage: mp_rec_vs_std [seeds]
#include "../../alvaar_amd/csrc/slam/mp_rec.hpp"
#include <array>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <set>
using namespace alva_slam;

struct Px {
    float px[2], unpx[2];
};
static long checks = 0;
static void fail(const char *what, int seed, int step) {
    std::printf("MISMATCH %s (seed %d, step %d)\n", what, seed, step);
    std::exit(1);
}

int main(int argc, char **argv) {
    const int seeds = argc > 1 ? std::atoi(argv[1]) : 8;
    for (int seed = 1; seed <= seeds; seed++) {
        std::mt19937 rng((unsigned) seed);
        MpRec r;
        DescBytes side[MP_ENT_CAP];
        rec_init(r, 7, 0, 3);
        std::set<int> obs;
        std::map<int, Px> inkf;
        std::map<int, std::array<uint8_t, 32>> desc;
        const int key_span = seed % 3 == 0 ? 30 : 60;   // (span 60 > capacity 40: the overflow path is exercised)
        for (int step = 0; step < 20000; step++) {
            const int kf = (int) (rng() % (unsigned) key_span), op = (int) (rng() % 7);
            std::set<int> keys(obs);
            for (auto &e: inkf) keys.insert(e.first);
            for (auto &e: desc) keys.insert(e.first);
            const bool full = (int) keys.size() >= MP_ENT_CAP && !keys.count(kf);
            if (op == 0) {   // addObservedKeyframeId
                ObsEnt *e = rec_slot(r, kf, side);
                if (full) {
                    if (e || !r.overflow) fail("overflow not refused", seed, step);
                    r.overflow = 0;
                } else {
                    if (!e) fail("slot refused below capacity", seed, step);
                    if (!(e->flags & MPF_OBS)) { e->flags |= MPF_OBS; r.n_obs++; }
                    obs.insert(kf);
                }
            } else if (op == 1) {   // the observation goes
                const int i = rec_find(r, kf);
                if (i >= 0 && (r.ent[i].flags & MPF_OBS)) rec_clear_flag(r, i, MPF_OBS, side);
                obs.erase(kf);
            } else if (op == 2) {   // the keyframe holds the keypoint (positions)
                ObsEnt *e = rec_slot(r, kf, side);
                if (full) { r.overflow = 0; continue; }
                Px p{{(float) (rng() % 640), (float) (rng() % 480)}, {(float) (rng() % 640) + 0.5f, (float) (rng() % 480) + 0.25f}};
                e->flags |= MPF_INKF;
                e->px[0] = p.px[0]; e->px[1] = p.px[1]; e->unpx[0] = p.unpx[0]; e->unpx[1] = p.unpx[1];
                inkf[kf] = p;
            } else if (op == 3) {
                const int i = rec_find(r, kf);
                if (i >= 0 && (r.ent[i].flags & MPF_INKF)) rec_clear_flag(r, i, MPF_INKF, side);
                inkf.erase(kf);
            } else if (op == 4) {   // a descriptor for the keyframe (first one wins, like unordered_map::emplace)
                if (desc.count(kf)) continue;
                ObsEnt *e = rec_slot(r, kf, side);
                if (full) { r.overflow = 0; continue; }
                std::array<uint8_t, 32> d;
                for (auto &b: d) b = (uint8_t) rng();
                e->flags |= MPF_DESC;
                std::memcpy(side[e - r.ent], d.data(), 32);
                desc[kf] = d;
            } else if (op == 5) {
                const int i = rec_find(r, kf);
                if (i >= 0 && (r.ent[i].flags & MPF_DESC)) rec_clear_flag(r, i, MPF_DESC, side);
                desc.erase(kf);
            } else if (rng() % 50 == 0) {   // the last observation goes: every descriptor is dropped (MapPoint::removeObservedKeyframeId)
                for (int i = r.n_ent; i-- > 0;)
                    if (r.ent[i].flags & MPF_DESC) rec_clear_flag(r, i, MPF_DESC, side);
                desc.clear();
            }
            // ---- compare
            keys = obs;
            for (auto &e: inkf) keys.insert(e.first);
            for (auto &e: desc) keys.insert(e.first);
            if ((int) keys.size() != r.n_ent || (int) obs.size() != r.n_obs) fail("counts", seed, step);
            int i = 0;
            for (int k: keys) {
                const ObsEnt &e = r.ent[i];
                if (e.kf != k) fail("entry order / keys", seed, step);
                const uint8_t want = (uint8_t) ((obs.count(k) ? MPF_OBS : 0) | (inkf.count(k) ? MPF_INKF : 0) | (desc.count(k) ? MPF_DESC : 0));
                if (e.flags != want) fail("flags", seed, step);
                if (inkf.count(k) && std::memcmp(e.px, inkf[k].px, 8)) fail("px", seed, step);
                if (inkf.count(k) && std::memcmp(e.unpx, inkf[k].unpx, 8)) fail("unpx", seed, step);
                if (desc.count(k) && std::memcmp(side[i], desc[k].data(), 32)) fail("descriptor bytes beside their entry", seed, step);
                if (rec_find(r, k) != i) fail("rec_find", seed, step);
                if ((rec_in_kf(r, k) != nullptr) != (inkf.count(k) != 0)) fail("rec_in_kf", seed, step);
                i++;
                checks++;
            }
            const ObsList l = rec_observers(r);
            if (l.size() != obs.size()) fail("observer snapshot size", seed, step);
            i = 0;
            for (int k: obs)
                if (l.kf[i++] != k) fail("observer snapshot order", seed, step);
            if (rec_find(r, key_span + 5) != -1) fail("rec_find of an absent key", seed, step);
        }
    }
    std::printf("ok %ld\n", checks);
    return 0;
}
