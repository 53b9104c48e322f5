// MapPoint's descriptor table -- mapKeyframeDescriptors_ + mapDescriptorsDist_ + desc_ (map_point.hpp:80-83) -- as ONE fixed-size record
// per map point that a GPU thread can own, and the two routines that edit it: MapPoint::addDesc (map_point.cpp:131-181) and the
// descriptor half of MapPoint::removeObservedKeyframeId (:93-128).
//
// The reference keeps the descriptors in std::unordered_map<int, cv::Mat> and picks the representative ("medoid") descriptor while
// ITERATING that map: the first strict minimum wins, so libstdc++'s iteration order decides ties (identical descriptors do occur: a
// stream that revisits a view).  The order is a small state machine over one singly linked list + bucket heads (flat_hash.hpp, which
// states and tests it against the real containers); the same machine runs here on index arrays inside the record.  Growth is NOT decided
// here: the map layer keeps the key sets of these tables in FlatHash (it needs them for its own control flow) and passes the bucket
// count of every rehash along with the insert that caused it, so the library's own rehash policy stays the only one.
//
// A table entry holds the 32 descriptor bytes (the reference's cv::Mat keeps them alive as long as the entry exists, whatever happens
// to the keyframe they came from); the distance sums are floats like the reference's (sums of popcounts: exact).  Every logged
// operation carries its descriptor, so the log is all the device needs.  Compiled for the device (medoid.hip:
// one wavefront per map point replays that point's operations in order) and for the host (the default Stages, used by the GPU-less
// harness under oracle/, and tests/cpp/medoid_table_vs_std.cpp, which drives it against std::unordered_map-based MapPoint logic).
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define ALVA_MED_HD __host__ __device__
#else
#define ALVA_MED_HD
#endif

namespace alva_medoid {

constexpr int CAP = 48;      // descriptors per map point: one per keyframe that holds it; the mapper's window is 30 keyframes
constexpr int NBKT = 59;     // libstdc++'s bucket counts on the way: 1, 13, 29, 59 (the 60th element would rehash to 127)
constexpr int END = -1, EMPTY = -1, BEFORE_BEGIN = -2;
constexpr int FREE_KEY = (int) 0x80000000;   // key of a slot on the free list (a reader that scans slots 0 .. used - 1 instead of walking the list)

enum Op : int { OP_ADD = 0, OP_REMOVE = 1, OP_CLEAR = 2, OP_RESET = 3 };
// one logged operation; `next` chains the operations of ONE map point in program order (-1 ends the chain)
struct MedoidOp {
    int op;
    int kf;          // keyframe id (the hash key)
    int rehash_to;   // OP_ADD: bucket count of the rehash this insert triggers in the reference's container, 0 = none
    int next;
    uint8_t desc[32];   // OP_ADD: the descriptor
    int pad[4];
};
static_assert(sizeof(MedoidOp) == 64, "MedoidOp layout");

struct Slot {
    int key, next;
    float dist;      // mapDescriptorsDist_[key]
    int pad;
    uint8_t desc[32];   // mapKeyframeDescriptors_[key]
};
struct Table {
    int head, free_, count, nbkt, used;
    int medoid_valid;    // !desc_.empty()
    int overflow;        // sticky: more than CAP descriptors or NBKT buckets were asked for (the export reports it)
    int medoid_kf;       // the key desc_ was last taken from (diagnostics; desc_ itself is the bytes below, which outlive their entry)
    uint8_t medoid[32];  // desc_
    Slot slot[CAP];
    int bkt[NBKT];
    int pad[1];
};
static_assert(sizeof(Table) % 16 == 0, "Table stride");

ALVA_MED_HD inline int bucket_of(int k, int n) { return (int) ((unsigned long long) (long long) k % (unsigned long long) n); }   // hash<int> = identity, sign-extended

ALVA_MED_HD inline void reset(Table &t) {   // a freshly constructed MapPoint
    t.head = END; t.free_ = END; t.count = 0; t.nbkt = 1; t.used = 0;
    t.medoid_valid = 0; t.overflow = 0; t.medoid_kf = -1;
    t.bkt[0] = EMPTY;
}
ALVA_MED_HD inline void clear_keep(Table &t) {   // unordered_map::clear (bucket count stays) + desc_.release()
    t.head = END; t.free_ = END; t.count = 0; t.used = 0;
    for (int b = 0; b < t.nbkt; b++) t.bkt[b] = EMPTY;
    t.medoid_valid = 0;
    t.medoid_kf = -1;
}
ALVA_MED_HD inline int find_slot(const Table &t, int k) {
    const int b = bucket_of(k, t.nbkt);
    const int p = t.bkt[b];
    if (p == EMPTY) return END;
    for (int s = p == BEFORE_BEGIN ? t.head : t.slot[p].next; s != END && bucket_of(t.slot[s].key, t.nbkt) == b; s = t.slot[s].next)
        if (t.slot[s].key == k) return s;
    return END;
}
ALVA_MED_HD inline void rehash(Table &t, int n) {   // _M_rehash_aux(n, unique keys)
    int nb[NBKT];
    for (int b = 0; b < n; b++) nb[b] = EMPTY;
    int p = t.head;
    t.head = END;
    int bbegin_bkt = 0;
    while (p != END) {
        const int nx = t.slot[p].next;
        const int b = bucket_of(t.slot[p].key, n);
        if (nb[b] == EMPTY) {
            t.slot[p].next = t.head;
            t.head = p;
            nb[b] = BEFORE_BEGIN;
            if (t.slot[p].next != END) nb[bbegin_bkt] = p;
            bbegin_bkt = b;
        } else {
            const int q = nb[b];
            if (q == BEFORE_BEGIN) {
                t.slot[p].next = t.head;
                t.head = p;
            } else {
                t.slot[p].next = t.slot[q].next;
                t.slot[q].next = p;
            }
        }
        p = nx;
    }
    for (int b = 0; b < n; b++) t.bkt[b] = nb[b];
    t.nbkt = n;
}
// unordered_map::emplace of a new key (the caller has checked that it is absent); returns the slot or END on overflow
ALVA_MED_HD inline int insert(Table &t, int k, const uint8_t *d, int rehash_to) {
    if (rehash_to > NBKT || (t.free_ == END && t.used >= CAP)) {
        t.overflow = 1;
        return END;
    }
    if (rehash_to > 0) rehash(t, rehash_to);
    int s = t.free_;
    if (s != END) t.free_ = t.slot[s].next;
    else s = t.used++;
    t.slot[s].key = k;
    t.slot[s].next = END;
    t.slot[s].dist = 0.f;
    memcpy(t.slot[s].desc, d, 32);
    const int b = bucket_of(k, t.nbkt);   // _M_insert_bucket_begin
    if (t.bkt[b] != EMPTY) {
        const int p = t.bkt[b];
        if (p == BEFORE_BEGIN) {
            t.slot[s].next = t.head;
            t.head = s;
        } else {
            t.slot[s].next = t.slot[p].next;
            t.slot[p].next = s;
        }
    } else {
        t.slot[s].next = t.head;
        t.head = s;
        const int nx = t.slot[s].next;
        if (nx != END) t.bkt[bucket_of(t.slot[nx].key, t.nbkt)] = s;
        t.bkt[b] = BEFORE_BEGIN;
    }
    t.count++;
    return s;
}
ALVA_MED_HD inline void erase_slot(Table &t, int s) {   // unordered_map::erase(iterator): _M_erase(bkt, prev, n)
    const int b = bucket_of(t.slot[s].key, t.nbkt);
    int prev = t.bkt[b];
    int c = prev == BEFORE_BEGIN ? t.head : t.slot[prev].next;
    while (c != s) {
        prev = c;
        c = t.slot[c].next;
    }
    const int nx = t.slot[s].next;
    if (prev == t.bkt[b]) {
        const int nb = nx != END ? bucket_of(t.slot[nx].key, t.nbkt) : 0;
        if (nx == END || nb != b) {
            if (nx != END) t.bkt[nb] = t.bkt[b];
            if (t.bkt[b] == BEFORE_BEGIN) t.head = nx;
            t.bkt[b] = EMPTY;
        }
    } else if (nx != END) {
        const int nb = bucket_of(t.slot[nx].key, t.nbkt);
        if (nb != b) t.bkt[nb] = prev;
    }
    if (prev == BEFORE_BEGIN) t.head = nx;
    else t.slot[prev].next = nx;
    t.slot[s].next = t.free_;
    t.slot[s].key = FREE_KEY;
    t.free_ = s;
    t.count--;
}

ALVA_MED_HD inline int popcount256(const uint8_t *a, const uint8_t *b) {
    int c = 0;
    for (int w = 0; w < 4; w++) {
        unsigned long long x, y;
        memcpy(&x, a + 8 * w, 8);
        memcpy(&y, b + 8 * w, 8);
#if defined(__HIP_DEVICE_COMPILE__)
        c += __popcll(x ^ y);
#else
        c += __builtin_popcountll(x ^ y);
#endif
    }
    return c;
}

// ---- the two routines, one thread (the host form; medoid.hip has the same steps spread over a wavefront) -------------------------------
// MapPoint::addDesc(kf, d)
ALVA_MED_HD inline void add_desc(Table &t, int kf, const uint8_t *d, int rehash_to) {
    if (find_slot(t, kf) != END) return;                       // :133-137
    const int sn = insert(t, kf, d, rehash_to);                // :140-143 (distance 0)
    if (sn == END) return;
    if (t.count == 1) {                                        // :145-149
        memcpy(t.medoid, d, 32);
        t.medoid_valid = 1;
        t.medoid_kf = kf;
        return;
    }
    float min_dist = t.medoid_valid ? 256.f : 0.f;             // desc_.cols * 8. (:152)
    int min_id = -1, min_slot = END;
    float nd = 0.f;
    for (int s = t.head; s != END; s = t.slot[s].next) {       // :156-172, the new entry included (distance 0 to itself)
        const float dist = (float) popcount256(d, t.slot[s].desc);
        if (s != sn) t.slot[s].dist += dist;
        if (dist < min_dist) {
            min_dist = dist;
            min_id = t.slot[s].key;
            min_slot = s;
        }
        nd += dist;
    }
    t.slot[sn].dist = nd;   // `newDescriptorDist` IS the new entry's sum (a reference into the map, :143)
    if (nd < min_dist) {                                       // :175-178
        min_id = kf;
        min_slot = sn;
    }
    if (min_slot != END) {                                     // :180 (.at(minId) throws in the reference when nothing was chosen)
        memcpy(t.medoid, t.slot[min_slot].desc, 32);
        t.medoid_valid = 1;
        t.medoid_kf = min_id;
    }
}
// the descriptor part of MapPoint::removeObservedKeyframeId(kf) (:93-128; the caller handles the observation set and the "last
// observation gone" branch, which is clear_keep)
ALVA_MED_HD inline void remove_desc(Table &t, int kf) {
    const int sd = find_slot(t, kf);
    if (sd == END) return;
    float min_dist = t.medoid_valid ? 256.f : 0.f;
    int min_id = -1, min_slot = END;
    for (int s = t.head; s != END; s = t.slot[s].next) {
        if (s == sd) continue;
        const float dist = (float) popcount256(t.slot[sd].desc, t.slot[s].desc);
        t.slot[s].dist -= dist;
        if (t.slot[s].dist < min_dist) {
            min_dist = t.slot[s].dist;
            min_id = t.slot[s].key;
            min_slot = s;
        }
    }
    erase_slot(t, sd);
    if (min_id > 0) {   // sic: keyframe 0 is never chosen (:123)
        memcpy(t.medoid, t.slot[min_slot].desc, 32);
        t.medoid_valid = 1;
        t.medoid_kf = min_id;
    }
}
ALVA_MED_HD inline void apply(Table &t, const MedoidOp &o) {
    switch (o.op) {
        case OP_ADD: add_desc(t, o.kf, o.desc, o.rehash_to); break;
        case OP_REMOVE: remove_desc(t, o.kf); break;
        case OP_CLEAR: clear_keep(t); break;
        default: reset(t); break;
    }
}

}  // namespace alva_medoid
