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
// Records are 3 x int32 (a, b, cn); a == 0 marks an empty slot.  Single-threaded by design (K <= 64).
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

struct SortCtx {
    Rec3* a;      // array base
    int n;        // number of records (guards)
    bool ub;      // an "unguarded" loop left [0,n): the library would have read out of bounds
};

namespace sortdetail {

AMBI_HD void swp(SortCtx& c, int i, int j) { Rec3 t = c.a[i]; c.a[i] = c.a[j]; c.a[j] = t; }

AMBI_HD void unguarded_linear_insert(SortCtx& c, int last) {
    Rec3 val = c.a[last];
    int next = last - 1;
    while (true) {
        if (next < 0) { c.ub = true; break; }
        if (!compare_loops(val, c.a[next])) break;
        c.a[last] = c.a[next];
        last = next;
        --next;
    }
    c.a[last] = val;
}

AMBI_HD void insertion_sort(SortCtx& c, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (compare_loops(c.a[i], c.a[first])) {
            Rec3 val = c.a[i];
            for (int k = i; k > first; --k) c.a[k] = c.a[k - 1];   // move_backward(first, i, i+1)
            c.a[first] = val;
        } else {
            unguarded_linear_insert(c, i);
        }
    }
}

AMBI_HD void push_heap(SortCtx& c, int first, int hole, int top, Rec3 value) {
    int parent = (hole - 1) / 2;
    while (hole > top && compare_loops(c.a[first + parent], value)) {
        c.a[first + hole] = c.a[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    c.a[first + hole] = value;
}

AMBI_HD void adjust_heap(SortCtx& c, int first, int hole, int len, Rec3 value) {
    const int top = hole;
    int second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (compare_loops(c.a[first + second], c.a[first + (second - 1)])) second--;
        c.a[first + hole] = c.a[first + second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        c.a[first + hole] = c.a[first + (second - 1)];
        hole = second - 1;
    }
    push_heap(c, first, hole, top, value);
}

AMBI_HD void pop_heap(SortCtx& c, int first, int last, int result) {
    Rec3 value = c.a[result];
    c.a[result] = c.a[first];
    adjust_heap(c, first, 0, last - first, value);
}

AMBI_HD void heap_sort_range(SortCtx& c, int first, int last) {   // __partial_sort(first, last, last)
    int len = last - first;
    if (len >= 2) {   // __make_heap
        int parent = (len - 2) / 2;
        while (true) {
            Rec3 value = c.a[first + parent];
            adjust_heap(c, first, parent, len, value);
            if (parent == 0) break;
            parent--;
        }
    }
    // __heap_select's scan over [middle,last) is empty (middle == last); then __sort_heap
    int l = last;
    while (l - first > 1) { --l; pop_heap(c, first, l, l); }
}

AMBI_HD void move_median_to_first(SortCtx& c, int result, int a, int b, int cc) {
    if (compare_loops(c.a[a], c.a[b])) {
        if (compare_loops(c.a[b], c.a[cc])) swp(c, result, b);
        else if (compare_loops(c.a[a], c.a[cc])) swp(c, result, cc);
        else swp(c, result, a);
    } else if (compare_loops(c.a[a], c.a[cc])) swp(c, result, a);
    else if (compare_loops(c.a[b], c.a[cc])) swp(c, result, cc);
    else swp(c, result, b);
}

AMBI_HD int unguarded_partition(SortCtx& c, int first, int last, int pivot) {
    while (true) {
        while (true) {
            if (first >= c.n) { c.ub = true; return first; }
            if (!compare_loops(c.a[first], c.a[pivot])) break;
            ++first;
        }
        --last;
        while (true) {
            if (last < 0) { c.ub = true; return first; }
            if (!compare_loops(c.a[pivot], c.a[last])) break;
            --last;
        }
        if (!(first < last)) return first;
        swp(c, first, last);
        ++first;
    }
}

AMBI_HD int floor_lg(int n) { int k = 0; while (n > 1) { n >>= 1; ++k; } return k; }

}  // namespace sortdetail

// std::sort(a, a+n, compareLoops), n <= 1023.  `stack`: kSortStack words of caller memory (group memory on the GPU: a
// private array indexed at run time would live in scratch, i.e. in HBM) for the parked left parts: one packed word
// { first : 10, last : 10, depth : 12 } each.  The parked parts have strictly increasing depth budgets from the top of
// the stack down, so 2*floor(log2 n) + 1 <= 19 words suffice.
constexpr int kSortStack = 24;
AMBI_HD void libstdcxx_sort_loops(Rec3* a, int n, bool* ub, uint32_t* stack) {
    using namespace sortdetail;
    SortCtx c{a, n, false};
    if (n > 1023) { if (ub) *ub = true; return; }
    if (n > 0) {
        // __introsort_loop.  The library recurses on the RIGHT part [cut,last) first and afterwards loops on the
        // left part [first,cut) with the same (already decremented) depth; the explicit stack below keeps exactly
        // that processing order: the left part is parked, the right part is handled at once.
        int sp = 0;
        stack[sp++] = 0u | ((uint32_t)n << 10) | ((uint32_t)(floor_lg(n) * 2) << 20);
        while (sp > 0 && !c.ub) {
            const uint32_t f = stack[--sp];
            int first = (int)(f & 1023u), last = (int)((f >> 10) & 1023u), depth = (int)(f >> 20);
            while (last - first > 16) {
                if (depth == 0) { heap_sort_range(c, first, last); break; }
                --depth;
                int mid = first + (last - first) / 2;
                move_median_to_first(c, first, first + 1, mid, last - 1);
                int cut = unguarded_partition(c, first + 1, last, first);
                if (c.ub) break;
                if (sp < kSortStack) stack[sp++] = (uint32_t)first | ((uint32_t)cut << 10) | ((uint32_t)depth << 20);   // left part, resumed after the right part
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

}  // namespace ambi
