// std::unordered_map<int, V> / std::unordered_set<int> of libstdc++ (GCC 11 headers, the ones the reference is built with) as FLAT arrays
// with the SAME ITERATION ORDER after the same sequence of operations.
//
// Why: the reference iterates its hash containers and that order reaches the numerics (slam.hpp), so the map layer must walk its
// keypoints / local map / descriptor tables in libstdc++'s order.  Real node-based containers cost one allocation per element and a cache
// miss per step; a keyframe is a COPY of ~2600 nodes, a local map a few thousand.  libstdc++'s order is fully determined by a small state
// machine (bits/hashtable.h): ONE singly linked list of all nodes + per bucket a pointer to the node BEFORE the bucket's first node;
//   insert   : at the FRONT of its bucket's run if the bucket is non-empty, else at the front of the whole list (_M_insert_bucket_begin)
//   erase    : unlink, repair the bucket heads (_M_erase / _M_remove_bucket_begin)
//   rehash   : walk the list, re-insert every node with the same rule into the new bucket array (_M_rehash_aux, unique keys)
//   growth   : std::__detail::_Prime_rehash_policy (the library's own object is used here, so the prime table and the load-factor rule
//              are the library's, not a transcription)
//   copy     : same bucket count, same policy state, nodes in the source's order (_M_assign); clear keeps the bucket count
// hash<int> is the identity, the bucket is (size_t) key % bucket_count.  The same machine on index arrays: `next` links slots, `bkt_` holds
// per bucket the slot before its first node (BEFORE_BEGIN for the list head).  A slot is {key, next, tag, live, value} in ONE array (two
// allocations per container: slots + buckets); a copy is two vector copies, an insert is an append.
// tests/cpp/flat_hash_vs_std.cpp drives this class and the real containers with the same random operation sequences (inserts, erases,
// clears, copies, swaps, range inserts across many growth steps) and compares the iteration order after every step.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <unordered_map>   // std::__detail::_Prime_rehash_policy
#include <utility>
#include <vector>

namespace alva_slam {

struct FlatNoValue {};

// _Prime_rehash_policy::_M_need_rehash(buckets, elements, 1) -- the library's own function decides every growth step -- behind the test
// it starts with (hashtable_c++0x.cc: "if (n_elt + n_ins > _M_next_resize) ... else return {false, 0}"): below the threshold the answer
// is "no" and nothing changes, and a call into libstdc++.so per insert (17 000 per keyframe) is most of what an insert costs.
// tests/cpp/flat_hash_vs_std.cpp compares the bucket counts with the real containers after every operation.
inline std::pair<bool, std::size_t> need_rehash(std::__detail::_Prime_rehash_policy &pol, std::size_t buckets, std::size_t elements) {
    if (elements + 1 <= pol._M_state()) return {false, 0};
    return pol._M_need_rehash(buckets, elements, 1);
}

template <class V>
class FlatHash {
public:
    static constexpr int END = -1;

    FlatHash() : bkt_(1, EMPTY) {}
    FlatHash(const FlatHash &) = default;
    FlatHash &operator=(const FlatHash &) = default;
    // moves leave the source a valid EMPTY container (a moved-from vector would leave bucket_of() dividing by a bucket count of zero)
    FlatHash(FlatHash &&o) : FlatHash() { swap(o); }
    FlatHash &operator=(FlatHash &&o) {
        if (this != &o) {
            reset();
            swap(o);
        }
        return *this;
    }

    size_t size() const { return count_; }
    bool empty() const { return count_ == 0; }
    size_t bucket_count() const { return bkt_.size(); }

    // iteration in the container's order: for (int s = c.first(); s != FlatHash::END; s = c.next(s)) use c.key(s), c.val(s), c.tag(s)
    int first() const { return head_; }
    int next(int slot) const { return slot_[(size_t) slot].next; }
    int key(int slot) const { return slot_[(size_t) slot].key; }
    V &val(int slot) { return slot_[(size_t) slot].v; }
    const V &val(int slot) const { return slot_[(size_t) slot].v; }
    // one user byte per element, kept beside the key (hot walks of a set-like table read 12-byte slots)
    uint8_t tag(int slot) const { return slot_[(size_t) slot].tag; }
    void set_tag(int slot, uint8_t t) { slot_[(size_t) slot].tag = t; }
    // walks in MEMORY order, for loops whose result does not depend on the order: slot indices 0 .. slots() - 1, erased ones are not live
    size_t slots() const { return slot_.size(); }
    const void *slot_storage() const { return slot_.data(); }   // for software prefetch
    bool slot_live(size_t slot) const { return slot_[slot].live != 0; }

    int find_slot(int k) const {  // slot or END
        const size_t b = bucket_of(k);
        const int p = bkt_[b];
        if (p == EMPTY) return END;
        for (int s = p == BEFORE_BEGIN ? head_ : slot_[(size_t) p].next; s != END && bucket_of(slot_[(size_t) s].key) == b; s = slot_[(size_t) s].next)
            if (slot_[(size_t) s].key == k) return s;
        return END;
    }
    size_t count(int k) const { return find_slot(k) != END ? 1 : 0; }

    // unordered_map::emplace / unordered_set::insert: returns (slot, inserted)
    std::pair<int, bool> insert_slot(int k, const V &v = V(), uint8_t tag = 0) {
        const int f = find_slot(k);
        if (f != END) return {f, false};
        const std::pair<bool, std::size_t> grow = need_rehash(pol_, bkt_.size(), count_);   // _M_insert_unique_node
        if (grow.first) rehash(grow.second);
        const int s = alloc(k, v, tag);
        link_front_of_bucket(s);
        count_++;
        return {s, true};
    }

    // the same for a key the caller KNOWS to be absent (it filters repeats itself, e.g. with a mark table): no look-up
    int insert_new_slot(int k, const V &v = V(), uint8_t tag = 0) {
        const std::pair<bool, std::size_t> grow = need_rehash(pol_, bkt_.size(), count_);
        if (grow.first) rehash(grow.second);
        const int s = alloc(k, v, tag);
        link_front_of_bucket(s);
        count_++;
        return s;
    }

    bool erase(int k) {
        const size_t b = bucket_of(k);
        int prev = bkt_[b];
        if (prev == EMPTY) return false;
        int s = prev == BEFORE_BEGIN ? head_ : slot_[(size_t) prev].next;
        while (s != END && bucket_of(slot_[(size_t) s].key) == b && slot_[(size_t) s].key != k) {
            prev = s;
            s = slot_[(size_t) s].next;
        }
        if (s == END || bucket_of(slot_[(size_t) s].key) != b) return false;
        erase_linked(b, prev, s);
        return true;
    }
    void erase_slot(int s) {  // unordered_map::erase(iterator)
        const size_t b = bucket_of(slot_[(size_t) s].key);
        int prev = bkt_[b];
        int c = prev == BEFORE_BEGIN ? head_ : slot_[(size_t) prev].next;
        while (c != s) {
            prev = c;
            c = slot_[(size_t) c].next;
        }
        erase_linked(b, prev, s);
    }

    void clear() {  // keeps the bucket count and the policy state, like _Hashtable::clear
        slot_.clear();
        for (int &b: bkt_) b = EMPTY;
        head_ = END;
        free_ = END;
        count_ = 0;
    }

    // a freshly constructed container (one bucket, policy reset) that keeps the arrays' capacity: for function-local containers of the
    // reference that are rebuilt on every call
    void reset() {
        slot_.clear();
        bkt_.assign(1, EMPTY);
        head_ = END;
        free_ = END;
        count_ = 0;
        pol_ = std::__detail::_Prime_rehash_policy();
    }

    // ---- the std::unordered_map spelling of the same operations, for code that reads like the reference's
    struct Ref {
        int first;
        V &second;
    };
    struct CRef {
        int first;
        const V &second;
    };
    template <class H, class R>
    struct Iter {
        H *h;
        int s;
        R operator*() const { return R{h->key(s), h->val(s)}; }
        struct Arrow {
            R r;
            R *operator->() { return &r; }
        };
        Arrow operator->() const { return Arrow{**this}; }
        Iter &operator++() {
            s = h->next(s);
            return *this;
        }
        bool operator!=(const Iter &o) const { return s != o.s; }
        bool operator==(const Iter &o) const { return s == o.s; }
    };
    typedef Iter<FlatHash, Ref> iterator;
    typedef Iter<const FlatHash, CRef> const_iterator;
    iterator begin() { return iterator{this, head_}; }
    iterator end() { return iterator{this, END}; }
    const_iterator begin() const { return const_iterator{this, head_}; }
    const_iterator end() const { return const_iterator{this, END}; }
    iterator find(int k) { return iterator{this, find_slot(k)}; }
    const_iterator find(int k) const { return const_iterator{this, find_slot(k)}; }
    std::pair<iterator, bool> emplace(int k, const V &v) {
        const std::pair<int, bool> r = insert_slot(k, v);
        return {iterator{this, r.first}, r.second};
    }
    void erase(iterator it) { erase_slot(it.s); }
    V &at(int k) {
        const int s = find_slot(k);
        if (s == END) throw std::out_of_range("FlatHash::at");
        return val(s);
    }
    const V &at(int k) const { return const_cast<FlatHash *>(this)->at(k); }

    void swap(FlatHash &o) {
        slot_.swap(o.slot_); bkt_.swap(o.bkt_);
        std::swap(head_, o.head_); std::swap(free_, o.free_); std::swap(count_, o.count_); std::swap(pol_, o.pol_);
    }
    // copy construction / assignment: the defaults (two vector copies) ARE _M_assign: same buckets, same policy, same order

private:
    static constexpr int EMPTY = -1, BEFORE_BEGIN = -2;
    struct Slot {
        int key, next;
        uint8_t tag, live;
        V v;   // a set's slot (V = FlatNoValue) is 12 bytes
    };
    std::vector<Slot> slot_;
    std::vector<int> bkt_;   // per bucket: the slot BEFORE its first node, BEFORE_BEGIN, or EMPTY
    int head_ = END, free_ = END;
    size_t count_ = 0;
    std::__detail::_Prime_rehash_policy pol_;

    size_t bucket_of(int k) const { return (size_t) k % bkt_.size(); }   // std::hash<int> = identity (sign-extended), _Mod_range_hashing
    static size_t bucket_of(int k, size_t n) { return (size_t) k % n; }

    int alloc(int k, const V &v, uint8_t tag) {
        int s = free_;
        if (s != END) {
            free_ = slot_[(size_t) s].next;
            slot_[(size_t) s] = Slot{k, END, tag, 1, v};
        } else {
            s = (int) slot_.size();
            slot_.push_back(Slot{k, END, tag, 1, v});
        }
        return s;
    }

    void link_front_of_bucket(int s) {  // _M_insert_bucket_begin
        const size_t b = bucket_of(slot_[(size_t) s].key);
        if (bkt_[b] != EMPTY) {
            const int p = bkt_[b];
            int &pn = p == BEFORE_BEGIN ? head_ : slot_[(size_t) p].next;
            slot_[(size_t) s].next = pn;
            pn = s;
        } else {
            slot_[(size_t) s].next = head_;
            head_ = s;
            const int nx = slot_[(size_t) s].next;
            if (nx != END) bkt_[bucket_of(slot_[(size_t) nx].key)] = s;
            bkt_[b] = BEFORE_BEGIN;
        }
    }

    void rehash(size_t n) {  // _M_rehash_aux(n, unique keys)
        std::vector<int> nb(n, EMPTY);
        int p = head_;
        head_ = END;
        size_t bbegin_bkt = 0;
        while (p != END) {
            const int nx = slot_[(size_t) p].next;
            const size_t b = bucket_of(slot_[(size_t) p].key, n);
            if (nb[b] == EMPTY) {
                slot_[(size_t) p].next = head_;
                head_ = p;
                nb[b] = BEFORE_BEGIN;
                if (slot_[(size_t) p].next != END) nb[bbegin_bkt] = p;
                bbegin_bkt = b;
            } else {
                const int q = nb[b];
                int &qn = q == BEFORE_BEGIN ? head_ : slot_[(size_t) q].next;
                slot_[(size_t) p].next = qn;
                qn = p;
            }
            p = nx;
        }
        bkt_.swap(nb);
    }

    void erase_linked(size_t b, int prev, int s) {  // _M_erase(bkt, prev_n, n)
        const int nx = slot_[(size_t) s].next;
        if (prev == bkt_[b]) {
            // _M_remove_bucket_begin(bkt, next, next_bkt)
            const size_t nb = nx != END ? bucket_of(slot_[(size_t) nx].key) : 0;
            if (nx == END || nb != b) {
                if (nx != END) bkt_[nb] = bkt_[b];
                if (bkt_[b] == BEFORE_BEGIN) head_ = nx;
                bkt_[b] = EMPTY;
            }
        } else if (nx != END) {
            const size_t nb = bucket_of(slot_[(size_t) nx].key);
            if (nb != b) bkt_[nb] = prev;
        }
        if (prev == BEFORE_BEGIN) head_ = nx;
        else slot_[(size_t) prev].next = nx;
        slot_[(size_t) s].live = 0;
        slot_[(size_t) s].next = free_;
        free_ = s;
        count_--;
    }
};

// std::unordered_set<int> of at most CAP keys on at most NBKT buckets, everything INLINE (no heap): the same state machine as FlatHash on
// byte-sized index arrays.  For the thousands of tiny tables the map layer keeps one of per map point (the keys of a map point's
// mapKeyframeDescriptors_: one per keyframe of the window): two heap vectors per table cost an allocation pair per new map point, a
// copy pair per merge and two cache misses per edit; here a table is 300 bytes in an arena indexed by the map point's slot.  insert()
// refuses (returns -1, nothing changed but the growth policy's state) what would not fit; the caller fails the frame.
template <int CAP, int NBKT>
class SmallFlatSet {
    static_assert(CAP < 127 && NBKT < 32767, "index widths");

public:
    static constexpr int END = -1;
    SmallFlatSet() { reset(); }
    void reset() {   // a freshly constructed container: one bucket, the policy reset
        head_ = END; free_ = END; used_ = 0; count_ = 0; nbkt_ = 1;
        bkt_[0] = EMPTY;
        pol_ = std::__detail::_Prime_rehash_policy();
    }
    size_t size() const { return (size_t) count_; }
    bool empty() const { return count_ == 0; }
    size_t bucket_count() const { return (size_t) nbkt_; }
    int first() const { return head_; }
    int next(int slot) const { return next_[slot]; }
    int key(int slot) const { return key_[slot]; }
    int find_slot(int k) const {
        const int b = bucket_of(k);
        const int p = bkt_[b];
        if (p == EMPTY) return END;
        for (int s = p == BEFORE_BEGIN ? head_ : next_[p]; s != END && bucket_of(key_[s]) == b; s = next_[s])
            if (key_[s] == k) return s;
        return END;
    }
    size_t count(int k) const { return find_slot(k) != END ? 1 : 0; }
    int next_slot() const { return free_ != END ? free_ : used_; }   // the slot the next new key will take (for a caller that prefetches its payload)
    // unordered_set::insert: 1 = inserted, 0 = already there, -1 = does not fit (CAP keys / NBKT buckets); *slot = the key's slot when it
    // is (now) in the set -- a key keeps its slot for as long as it stays (a caller may keep per-key payload in a parallel array)
    int insert(int k, int *slot = nullptr) {
        const int have = find_slot(k);
        if (have != END) {
            if (slot) *slot = have;
            return 0;
        }
        const std::pair<bool, std::size_t> grow = need_rehash(pol_, (size_t) nbkt_, (size_t) count_);   // _M_insert_unique_node
        if ((grow.first && grow.second > (size_t) NBKT) || (free_ == END && used_ >= CAP)) return -1;
        if (grow.first) rehash((int) grow.second);
        int s = free_;
        if (s != END) free_ = next_[s];
        else s = used_++;
        key_[s] = k;
        const int b = bucket_of(k);   // _M_insert_bucket_begin
        if (bkt_[b] != EMPTY) {
            const int p = bkt_[b];
            if (p == BEFORE_BEGIN) {
                next_[s] = (int8_t) head_;
                head_ = (int16_t) s;
            } else {
                next_[s] = next_[p];
                next_[p] = (int8_t) s;
            }
        } else {
            next_[s] = (int8_t) head_;
            head_ = (int16_t) s;
            const int nx = next_[s];
            if (nx != END) bkt_[bucket_of(key_[nx])] = (int8_t) s;
            bkt_[b] = BEFORE_BEGIN;
        }
        count_++;
        if (slot) *slot = s;
        return 1;
    }
    bool erase(int k) {
        const int b = bucket_of(k);
        int prev = bkt_[b];
        if (prev == EMPTY) return false;
        int s = prev == BEFORE_BEGIN ? head_ : next_[prev];
        while (s != END && bucket_of(key_[s]) == b && key_[s] != k) {
            prev = s;
            s = next_[s];
        }
        if (s == END || bucket_of(key_[s]) != b) return false;
        const int nx = next_[s];   // _M_erase(bkt, prev_n, n)
        if (prev == bkt_[b]) {
            const int nb = nx != END ? bucket_of(key_[nx]) : 0;   // _M_remove_bucket_begin
            if (nx == END || nb != b) {
                if (nx != END) bkt_[nb] = bkt_[b];
                if (bkt_[b] == BEFORE_BEGIN) head_ = (int16_t) nx;
                bkt_[b] = EMPTY;
            }
        } else if (nx != END) {
            const int nb = bucket_of(key_[nx]);
            if (nb != b) bkt_[nb] = (int8_t) prev;
        }
        if (prev == BEFORE_BEGIN) head_ = (int16_t) nx;
        else next_[prev] = (int8_t) nx;
        next_[s] = (int8_t) free_;
        free_ = (int16_t) s;
        count_--;
        return true;
    }
    void clear() {   // keeps the bucket count and the policy state, like _Hashtable::clear
        for (int b = 0; b < nbkt_; b++) bkt_[b] = EMPTY;
        head_ = END; free_ = END; used_ = 0; count_ = 0;
    }

private:
    static constexpr int EMPTY = -1, BEFORE_BEGIN = -2;
    int bucket_of(int k) const { return (int) ((size_t) k % (size_t) nbkt_); }   // std::hash<int> = identity (sign-extended)
    void rehash(int n) {   // _M_rehash_aux(n, unique keys)
        int8_t nb[NBKT];
        for (int b = 0; b < n; b++) nb[b] = EMPTY;
        int p = head_;
        head_ = END;
        int bbegin_bkt = 0;
        while (p != END) {
            const int nx = next_[p];
            const int b = (int) ((size_t) key_[p] % (size_t) n);
            if (nb[b] == EMPTY) {
                next_[p] = (int8_t) head_;
                head_ = (int16_t) p;
                nb[b] = BEFORE_BEGIN;
                if (next_[p] != END) nb[bbegin_bkt] = (int8_t) p;
                bbegin_bkt = b;
            } else {
                const int q = nb[b];
                if (q == BEFORE_BEGIN) {
                    next_[p] = (int8_t) head_;
                    head_ = (int16_t) p;
                } else {
                    next_[p] = next_[q];
                    next_[q] = (int8_t) p;
                }
            }
            p = nx;
        }
        for (int b = 0; b < n; b++) bkt_[b] = nb[b];
        nbkt_ = (int16_t) n;
    }
    int16_t head_, free_, used_, count_, nbkt_;
    std::__detail::_Prime_rehash_policy pol_;
    int key_[CAP];
    int8_t next_[CAP];
    int8_t bkt_[NBKT];
};

// std::unordered_set<int>
class FlatSet : public FlatHash<FlatNoValue> {
public:
    struct KeyIter {
        const FlatSet *h;
        int s;
        int operator*() const { return h->key(s); }
        KeyIter &operator++() {
            s = h->next(s);
            return *this;
        }
        bool operator!=(const KeyIter &o) const { return s != o.s; }
    };
    KeyIter begin() const { return KeyIter{this, first()}; }
    KeyIter end() const { return KeyIter{this, END}; }
    bool insert(int k) { return insert_slot(k).second; }
    void insert_new(int k) { insert_new_slot(k); }   // k is known to be absent
    template <class It>
    void insert(It a, It b) {  // _M_insert_range, unique keys: one insert per element
        for (; a != b; ++a) insert_slot(*a);
    }
};

}  // namespace alva_slam
