// ambi_group.hpp -- thread-group policies for the SPMD algorithm headers.
//
// Every cooperative routine takes a group `g` and is written so that
//   * all threads of the group execute it with identical (group-uniform) control flow,
//   * per-element work is strided:  for (i = g.tid(); i < n; i += g.size()),
//   * data handed between threads through the group's memory is separated by g.sync(),
//   * decisions come from group reductions (g.min_i32 / g.max_i32 / g.any).
// Policies:
//   HostGroup   1 thread, reductions are identities        -> CPU host simulation (tests, sanitizers)
//   WaveGroup   one 64-lane CDNA wavefront, DPP/shuffle reductions, no LDS scratch needed
//   BlockGroup  one workgroup of up to 1024 threads (wave reduce + LDS exchange)
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif
#include "ambi_common.hpp"

namespace ambi {

// atomic add on group or device memory, usable from both builds (the host simulation is single-threaded)
AMBI_HD int atomic_add_i32(int* p, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(p, v);
#else
    int old = *p; *p = old + v; return old;
#endif
}

// element `i` (uniform over the wavefront) of a value held one element per lane; device code of wavefront groups only
AMBI_HD uint32_t lane_get_u32(uint32_t v, int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_readlane((int)v, i);
#else
    (void)i; return v;
#endif
}

// compare-and-swap on group memory, usable from both builds; returns the old value
AMBI_HD int atomic_cas_i32(int* p, int expected, int desired) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicCAS(p, expected, desired);
#else
    int old = *p; if (old == expected) *p = desired; return old;
#endif
}

struct HostGroup {
    static constexpr bool kIsBlock = false;
    static constexpr bool kLaneArrays = false;   // no cross-lane register arrays (ambi_sort.hpp: LaneWords)
    AMBI_HD int tid() const { return 0; }
    AMBI_HD int size() const { return 1; }
    AMBI_HD void sync() const {}
    AMBI_HD int min_i32(int v) const { return v; }
    AMBI_HD int max_i32(int v) const { return v; }
    AMBI_HD int sum_i32(int v) const { return v; }
    AMBI_HD bool any(bool p) const { return p; }
    AMBI_HD int bcast_i32(int v, int /*src*/) const { return v; }
    AMBI_HD uint64_t bcast_u64(uint64_t v, int /*src*/) const { return v; }   // src must be uniform over the group
    AMBI_HD int bcast_i32_u(int v, int /*src*/) const { return v; }
    AMBI_HD uint64_t ballot_u64(bool q) const { return q ? 1ull : 0ull; }
    AMBI_HD int first_flag(bool q) const { return q ? 0 : -1; }   // lowest thread with the flag set, -1 if none
    // exclusive prefix sum over the group in thread order; total returned through *total
    AMBI_HD int exscan_i32(int v, int* total) const { *total = v; return 0; }
    // sub-groups: runs of up to 64 consecutive threads (a wavefront on the GPU) that can rank flags without a barrier
    AMBI_HD int sub_size() const { return 1; }
    AMBI_HD int sub_id() const { return 0; }
    AMBI_HD int n_subs() const { return 1; }
    AMBI_HD int sub_lane() const { return 0; }
    // number of set flags among the lower threads of my sub-group; *count = set flags in the whole sub-group
    AMBI_HD int flag_rank(bool q, int* count) const { *count = q ? 1 : 0; return 0; }
    // the same over the WHOLE group: number of set flags among the lower threads; *count = set flags in the group
    AMBI_HD int flag_exscan(bool q, int* count) const { *count = q ? 1 : 0; return 0; }
};

#if defined(__HIPCC__)

// One wavefront (64 lanes on gfx950). Lanes run in lockstep; sync() only has to order LDS traffic.
struct WaveGroup {
    static constexpr bool kIsBlock = false;
    static constexpr bool kLaneArrays = true;
    __device__ inline int tid() const { return (int)(threadIdx.x & 63u); }
    __device__ inline int size() const { return 64; }
    __device__ inline void sync() const {
        // Workgroup-scope fence: waits for this wave's outstanding LDS/global operations (all waves of a
        // workgroup share one CU and its L1, so no cache maintenance is involved) and pins the compiler's ordering.
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    // min / max over the 64 lanes: the DPP sequence of the prefix sum with the operator swapped (lanes without a
    // source take the identity), result in lane 63, broadcast through a scalar register
    __device__ inline int min_i32(int v) const {
        const int I = 0x7fffffff;
        int x = v, t;
        t = __builtin_amdgcn_update_dpp(I, x, 0x111, 0xf, 0xf, false); x = t < x ? t : x;
        t = __builtin_amdgcn_update_dpp(I, x, 0x112, 0xf, 0xf, false); x = t < x ? t : x;
        t = __builtin_amdgcn_update_dpp(I, x, 0x114, 0xf, 0xf, false); x = t < x ? t : x;
        t = __builtin_amdgcn_update_dpp(I, x, 0x118, 0xf, 0xf, false); x = t < x ? t : x;
        t = __builtin_amdgcn_update_dpp(I, x, 0x142, 0xa, 0xf, false); x = t < x ? t : x;
        t = __builtin_amdgcn_update_dpp(I, x, 0x143, 0xc, 0xf, false); x = t < x ? t : x;
        return __builtin_amdgcn_readlane(x, 63);
    }
    __device__ inline int max_i32(int v) const {
        const int I = (int)0x80000000;
        int x = v, t;
        t = __builtin_amdgcn_update_dpp(I, x, 0x111, 0xf, 0xf, false); x = t > x ? t : x;
        t = __builtin_amdgcn_update_dpp(I, x, 0x112, 0xf, 0xf, false); x = t > x ? t : x;
        t = __builtin_amdgcn_update_dpp(I, x, 0x114, 0xf, 0xf, false); x = t > x ? t : x;
        t = __builtin_amdgcn_update_dpp(I, x, 0x118, 0xf, 0xf, false); x = t > x ? t : x;
        t = __builtin_amdgcn_update_dpp(I, x, 0x142, 0xa, 0xf, false); x = t > x ? t : x;
        t = __builtin_amdgcn_update_dpp(I, x, 0x143, 0xc, 0xf, false); x = t > x ? t : x;
        return __builtin_amdgcn_readlane(x, 63);
    }
    // Inclusive prefix sum over the 64 lanes with DPP moves (VALU speed; the shuffle form costs six LDS-crossbar
    // round trips): Hillis-Steele inside every row of 16 lanes (row_shr 1,2,4,8; lanes without a source add 0), then
    // the last lane of row 0/2 into row 1/3 (row_bcast:15) and lane 31 into the upper half (row_bcast:31).
    __device__ inline int incl_scan_i32(int v) const {
        int x = v;
        x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
        x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
        x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
        x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8
        x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1 and 3
        x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2 and 3
        return x;
    }
    __device__ inline int sum_i32(int v) const { return __builtin_amdgcn_readlane(incl_scan_i32(v), 63); }
    __device__ inline bool any(bool p) const { return __ballot(p) != 0ull; }
    __device__ inline int bcast_i32(int v, int src) const { return __shfl(v, src, 64); }
    __device__ inline int bcast_i32_u(int v, int src) const { return __builtin_amdgcn_readlane(v, src); }   // src uniform over the wave
    __device__ inline uint64_t ballot_u64(bool q) const { return __ballot(q); }
    __device__ inline int first_flag(bool q) const { const unsigned long long b = __ballot(q); return b ? (int)__builtin_ctzll(b) : -1; }
    __device__ inline uint64_t bcast_u64(uint64_t v, int src) const {   // uniform src: two v_readlane, no LDS crossbar
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
        return ((uint64_t)hi << 32) | lo;
    }
    __device__ inline int exscan_i32(int v, int* total) const {
        const int x = incl_scan_i32(v);
        *total = __builtin_amdgcn_readlane(x, 63);
        return x - v;
    }
    __device__ inline int sub_size() const { return 64; }
    __device__ inline int sub_id() const { return 0; }
    __device__ inline int n_subs() const { return 1; }
    __device__ inline int sub_lane() const { return tid(); }
    __device__ inline int flag_rank(bool q, int* count) const {
        const unsigned long long b = __ballot(q);
        *count = __popcll(b);
        return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
    }
    __device__ inline int flag_exscan(bool q, int* count) const { return flag_rank(q, count); }
};

// One workgroup. `scratch` points at >= 40 ints of LDS reserved for the reductions.
struct BlockGroup {
    static constexpr bool kIsBlock = true;       // several wavefronts: group operations cost workgroup barriers
    static constexpr bool kLaneArrays = false;
    int* scratch;
    __device__ inline explicit BlockGroup(int* s) : scratch(s) {}
    __device__ inline int tid() const { return (int)threadIdx.x; }
    __device__ inline int size() const { return (int)blockDim.x; }
    __device__ inline void sync() const { __syncthreads(); }
    __device__ inline int nwaves() const { return ((int)blockDim.x + 63) >> 6; }
    // wave-level result (already uniform inside every wave) combined across the waves through LDS
    template <class Op> __device__ inline int combine(int wave_value, Op op, int identity) const {
        __syncthreads();   // protect scratch against the previous reduction's readers
        if ((threadIdx.x & 63u) == 0) scratch[threadIdx.x >> 6] = wave_value;
        __syncthreads();
        int r = identity;
        const int nw = nwaves();
        for (int i = 0; i < nw; i++) r = op(r, scratch[i]);
        return r;
    }
    __device__ inline int min_i32(int v) const { WaveGroup w; return combine(w.min_i32(v), [](int a, int b) { return a < b ? a : b; }, 0x7fffffff); }
    __device__ inline int max_i32(int v) const { WaveGroup w; return combine(w.max_i32(v), [](int a, int b) { return a > b ? a : b; }, (int)0x80000000); }
    __device__ inline int sum_i32(int v) const { WaveGroup w; return combine(w.sum_i32(v), [](int a, int b) { return a + b; }, 0); }
    __device__ inline bool any(bool p) const { return __syncthreads_or(p ? 1 : 0) != 0; }
    __device__ inline int bcast_i32(int v, int src) const {
        __syncthreads();
        if ((int)threadIdx.x == src) scratch[32] = v;
        __syncthreads();
        return scratch[32];
    }
    __device__ inline uint64_t bcast_u64(uint64_t v, int src) const {
        const uint32_t lo = (uint32_t)bcast_i32((int)(uint32_t)v, src), hi = (uint32_t)bcast_i32((int)(uint32_t)(v >> 32), src);
        return ((uint64_t)hi << 32) | lo;
    }
    __device__ inline int exscan_i32(int v, int* total) const {
        WaveGroup w;
        int wt;
        int x = w.exscan_i32(v, &wt);
        __syncthreads();
        if ((threadIdx.x & 63u) == 63u) scratch[threadIdx.x >> 6] = wt;
        __syncthreads();
        int base = 0, tot = 0, nw = nwaves(), me = (int)(threadIdx.x >> 6);
        for (int i = 0; i < nw; i++) { int s = scratch[i]; if (i < me) base += s; tot += s; }
        *total = tot;
        return base + x;
    }
    __device__ inline int sub_size() const { return 64; }
    __device__ inline int sub_id() const { return (int)(threadIdx.x >> 6); }
    __device__ inline int n_subs() const { return nwaves(); }
    __device__ inline int sub_lane() const { return (int)(threadIdx.x & 63u); }
    __device__ inline int flag_rank(bool q, int* count) const { WaveGroup w; return w.flag_rank(q, count); }
    __device__ inline int flag_exscan(bool q, int* count) const { return exscan_i32(q ? 1 : 0, count); }
    __device__ inline int first_flag(bool q) const { const int m = min_i32(q ? (int)threadIdx.x : 0x7fffffff); return m == 0x7fffffff ? -1 : m; }
};

#endif  // __HIPCC__

}  // namespace ambi
