// ambi_sort.hpp -- operation-for-operation restatement of libstdc++'s std::sort (GCC 11 bits/stl_algo.h:
// introsort with median-of-3 pivot, 16-element threshold, heapsort fallback at depth 2*floor(log2 n), final
// (unguarded) insertion sort).
//
// Why: LocalGenomicMap::constructDAG sorts node2loop with `compareLoops` (LGM.cpp:3267-3274, :3303), a
// comparator that is NOT a strict weak order (an empty slot compares equivalent to everything).  The resulting
// permutation therefore depends on the exact sequence of comparisons/swaps the library performs, and the DAG
// node numbering -- hence the order in which candidate BFB paths are tried -- depends on that permutation.
// To stay bit-exact on the GPU the engine replays the library's algorithm itself.  tests/test_sort_replica.py
// checks this file against the real std::sort with the same comparator on random inputs (n = 0..64).
//
// Records are 3 x int32 (a, b, cn); a == 0 marks an empty slot.  The comparator looks at (empty?, |a-b|) only, so the
// replay runs on one 32-bit word per record: key = { rank : 24 | original position : 8 } with rank = 0 for an empty
// slot and |a-b| + 1 otherwise; compareLoops(x, y) == (rank(y) != 0 && rank(x) > rank(y)).  The caller permutes the
// records by the position bytes afterwards.
//
// The algorithm is sequential (K <= 64), so on the GPU its cost is the latency of its dependent array accesses.  The
// array type is a template parameter: MemWords = plain memory (host, tests); LaneWords = ONE vector register, element
// i in lane i, read and written with v_readlane / v_writelane -- every lane of the wavefront runs the same scalar
// program (indices and keys live in scalar registers), and an access costs a few cycles instead of an LDS round trip.
#pragma once
#include "ambi_common.hpp"

namespace ambi {

struct Rec3 { int32_t v[3]; };

// compareLoops (LGM.cpp:3267-3274): |a0-a1| > |b0-b1| when both are non-empty, else false.
AMBI_HD bool compare_loops(const Rec3& x, const Rec3& y) {
    int d1 = 0, d2 = 0;
    if (x.v[0] != 0 && y.v[0] != 0) { d1 = iabs(x.v[0] - x.v[1]); d2 = iabs(y.v[0] - y.v[1]); }
    return d1 > d2;
}
AMBI_HD uint32_t loop_sort_key(int a, int b, int pos) { return ((a != 0 ? (uint32_t)iabs(a - b) + 1u : 0u) << 8) | (uint32_t)pos; }
AMBI_HD bool compare_keys(uint32_t x, uint32_t y) { return (y >> 8) != 0u && (x >> 8) > (y >> 8); }

// Besides get / set the array types answer the three linear scans of the library's inner loops and do its block move
// -- element by element in memory, with one ballot / one lane shift in the register form; the permutation after each
// library operation is the same either way:
//   first_not_before(from, n, pk)  smallest i in [from, n) with !comp(a[i], pk), else n   (__unguarded_partition, left scan)
//   last_not_after(from, x)        largest  i in [0, from] with !comp(x, a[i]), else -1   (right scan; __unguarded_linear_insert)
//   shift_up(lo, hi)               a[k + 1] = a[k] for k = hi-1 .. lo                     (move_backward / the insert's moves)
struct MemWords {
    static constexpr int kCap = 255;   // elements the replay may be asked for (the record's position rides in the key's low byte)
    uint32_t* a;
    AMBI_HD uint32_t get(int i) const { return a[i]; }
    AMBI_HD void set(int i, uint32_t x) { a[i] = x; }
    AMBI_HD int first_not_before(int from, int n, uint32_t pk) const { int i = from; while (i < n && compare_keys(a[i], pk)) i++; return i; }
    AMBI_HD int last_not_after(int from, uint32_t x) const { int i = from; while (i >= 0 && compare_keys(x, a[i])) i--; return i; }
    AMBI_HD void shift_up(int lo, int hi) { for (int k = hi; k > lo; --k) a[k] = a[k - 1]; }
};
struct LaneWords {   // all 64 lanes call get/set together with the same (uniform) arguments; device code only
    static constexpr int kCap = 64;
    uint32_t v;
    int lane;   // this thread's lane number
#if defined(__HIP_DEVICE_COMPILE__)
    __device__ inline uint32_t get(int i) const { return (uint32_t)__builtin_amdgcn_readlane((int)v, i); }
    __device__ inline void set(int i, uint32_t x) { v = (lane == i) ? x : v; }   // v_cmp + v_cndmask on scalar operands
    __device__ inline int first_not_before(int from, int n, uint32_t pk) const {
        const unsigned long long m = __ballot(lane >= from && lane < n && !compare_keys(v, pk));
        return m ? (int)__builtin_ctzll(m) : n;
    }
    __device__ inline int last_not_after(int from, uint32_t x) const {
        const unsigned long long m = __ballot(lane <= from && !compare_keys(x, v));
        return m ? 63 - (int)__builtin_clzll(m) : -1;
    }
    __device__ inline void shift_up(int lo, int hi) {
        const uint32_t below = (uint32_t)__shfl_up((int)v, 1, 64);   // lane k-1's element
        v = (lane > lo && lane <= hi) ? below : v;
    }
#else
    AMBI_HD uint32_t get(int) const { return v; }
    AMBI_HD void set(int, uint32_t x) { v = x; }
    AMBI_HD int first_not_before(int, int n, uint32_t) const { return n; }
    AMBI_HD int last_not_after(int, uint32_t) const { return -1; }
    AMBI_HD void shift_up(int, int) {}
#endif
};

template <class ARR>
struct SortCtx {
    ARR& a;       // the keys
    int n;        // number of records (guards)
    bool ub;      // an "unguarded" loop left [0,n): the library would have read out of bounds
};

namespace sortdetail {

template <class C> AMBI_HD void swp(C& c, int i, int j) { const uint32_t t = c.a.get(i); c.a.set(i, c.a.get(j)); c.a.set(j, t); }

template <class C> AMBI_HD void unguarded_linear_insert(C& c, int last) {
    // the library moves a[next] up while comp(val, a[next]) and stops at the first element (from the right) for which
    // the comparison fails -- or runs off the front (the "unguarded" read out of bounds)
    const uint32_t val = c.a.get(last);
    const int stop = c.a.last_not_after(last - 1, val);
    if (stop < 0) c.ub = true;
    c.a.shift_up(stop + 1, last);
    c.a.set(stop + 1, val);
}

template <class C> AMBI_HD void insertion_sort(C& c, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (compare_keys(c.a.get(i), c.a.get(first))) {
            const uint32_t val = c.a.get(i);
            c.a.shift_up(first, i);                                       // move_backward(first, i, i+1)
            c.a.set(first, val);
        } else {
            unguarded_linear_insert(c, i);
        }
    }
}

template <class C> AMBI_HD void push_heap(C& c, int first, int hole, int top, uint32_t value) {
    int parent = (hole - 1) / 2;
    while (hole > top && compare_keys(c.a.get(first + parent), value)) {
        c.a.set(first + hole, c.a.get(first + parent));
        hole = parent;
        parent = (hole - 1) / 2;
    }
    c.a.set(first + hole, value);
}

template <class C> AMBI_HD void adjust_heap(C& c, int first, int hole, int len, uint32_t value) {
    const int top = hole;
    int second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (compare_keys(c.a.get(first + second), c.a.get(first + (second - 1)))) second--;
        c.a.set(first + hole, c.a.get(first + second));
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        c.a.set(first + hole, c.a.get(first + (second - 1)));
        hole = second - 1;
    }
    push_heap(c, first, hole, top, value);
}

template <class C> AMBI_HD void pop_heap(C& c, int first, int last, int result) {
    const uint32_t value = c.a.get(result);
    c.a.set(result, c.a.get(first));
    adjust_heap(c, first, 0, last - first, value);
}

template <class C> AMBI_HD void heap_sort_range(C& c, int first, int last) {   // __partial_sort(first, last, last)
    int len = last - first;
    if (len >= 2) {   // __make_heap
        int parent = (len - 2) / 2;
        while (true) {
            const uint32_t value = c.a.get(first + parent);
            adjust_heap(c, first, parent, len, value);
            if (parent == 0) break;
            parent--;
        }
    }
    // __heap_select's scan over [middle,last) is empty (middle == last); then __sort_heap
    int l = last;
    while (l - first > 1) { --l; pop_heap(c, first, l, l); }
}

template <class C> AMBI_HD void move_median_to_first(C& c, int result, int a, int b, int cc) {
    const uint32_t ka = c.a.get(a), kb = c.a.get(b), kc = c.a.get(cc);
    if (compare_keys(ka, kb)) {
        if (compare_keys(kb, kc)) swp(c, result, b);
        else if (compare_keys(ka, kc)) swp(c, result, cc);
        else swp(c, result, a);
    } else if (compare_keys(ka, kc)) swp(c, result, a);
    else if (compare_keys(kb, kc)) swp(c, result, cc);
    else swp(c, result, b);
}

template <class C> AMBI_HD int unguarded_partition(C& c, int first, int last, int pivot) {
    const uint32_t pk = c.a.get(pivot);      // the pivot sits in front of the range and is not moved by the swaps below
    while (true) {
        first = c.a.first_not_before(first, c.n, pk);
        if (first >= c.n) { c.ub = true; return first; }
        last = c.a.last_not_after(last - 1, pk);
        if (last < 0) { c.ub = true; return first; }
        if (!(first < last)) return first;
        swp(c, first, last);
        ++first;
    }
}

AMBI_HD int floor_lg(int n) { int k = 0; while (n > 1) { n >>= 1; ++k; } return k; }

}  // namespace sortdetail

// std::sort(a, a+n, compareLoops) on the keys, n <= ARR::kCap (64 in lane registers, 255 in memory).  `stack`: kSortStack words for the parked left parts, one
// packed word { first : 10, last : 10, depth : 12 } each.  The parked parts have strictly increasing depth budgets from
// the top of the stack down, so 2*floor(log2 n) + 1 <= 15 words suffice.
constexpr int kSortStack = 24;
template <class ARR, class STK>
AMBI_HD void libstdcxx_sort_keys(ARR& a, int n, bool* ub, STK& stack) {
    using namespace sortdetail;
    SortCtx<ARR> c{a, n, false};
    if (n > ARR::kCap) { if (ub) *ub = true; return; }
    if (n > 0) {
        // __introsort_loop.  The library recurses on the RIGHT part [cut,last) first and afterwards loops on the
        // left part [first,cut) with the same (already decremented) depth; the explicit stack below keeps exactly
        // that processing order: the left part is parked, the right part is handled at once.
        int sp = 0;
        stack.set(sp++, 0u | ((uint32_t)n << 10) | ((uint32_t)(floor_lg(n) * 2) << 20));
        while (sp > 0 && !c.ub) {
            const uint32_t f = stack.get(--sp);
            int first = (int)(f & 1023u), last = (int)((f >> 10) & 1023u), depth = (int)(f >> 20);
            while (last - first > 16) {
                if (depth == 0) { heap_sort_range(c, first, last); break; }
                --depth;
                int mid = first + (last - first) / 2;
                move_median_to_first(c, first, first + 1, mid, last - 1);
                int cut = unguarded_partition(c, first + 1, last, first);
                if (c.ub) break;
                if (sp < kSortStack) stack.set(sp++, (uint32_t)first | ((uint32_t)cut << 10) | ((uint32_t)depth << 20));   // left part, resumed after the right part
                first = cut;
            }
        }
        if (!c.ub) {
            // __final_insertion_sort
            if (n > 16) {
                insertion_sort(c, 0, 16);
                for (int i = 16; i != n; ++i) unguarded_linear_insert(c, i);
            } else {
                insertion_sort(c, 0, n);
            }
        }
    }
    if (ub) *ub = c.ub;
}

// Record form (tests/hostsim/sort_probe.cpp checks it against the real std::sort with compareLoops): keys from the
// records, replay, records permuted by the position bytes.  n <= 255; `stack`: kSortStack + 256 words.
AMBI_HD void libstdcxx_sort_loops(Rec3* a, int n, bool* ub, uint32_t* stack) {
    if (n > 255) { if (ub) *ub = true; return; }
    MemWords keys{stack + kSortStack}, stk{stack};
    for (int i = 0; i < n; i++) keys.a[i] = loop_sort_key(a[i].v[0], a[i].v[1], i);
    bool u = false;
    libstdcxx_sort_keys(keys, n, &u, stk);
    if (ub) *ub = u;
    if (u) return;
    Rec3 tmp[256];
    for (int i = 0; i < n; i++) tmp[i] = a[i];
    for (int i = 0; i < n; i++) a[i] = tmp[keys.a[i] & 255u];
}

}  // namespace ambi
