// ambi_engine.hip -- the HIP (gfx950 / MI355X) backend: kernels + stream orchestration.
//
// Kernels (all integer / index work, no MFMA; bound by HBM writes of the order table and by LDS latency / issue):
//   ambi_prepare_kernel           1 wave  / unit   junction ends staged into LDS; getJuncCN, bias, getIndelBias,
//                                                  targetCN, constructDAG; order-ideal lattice (level-synchronous
//                                                  search in LDS), completion counts, frozen automaton; R
//   ambi_plan_kernel              1 block / slice  64-bit scans: order-table offsets, enumerate work blocks
//   ambi_blocks_build_kernel      1 block / unit   block-emission image (block directory + suffix rows) -> HBM
//   ambi_enumerate_blocks_kernel  1 block / work block: image -> LDS, rows streamed block by block with fully
//                                                  coalesced 16-byte stores                       <- HBM-bound
//   ambi_enumerate_kernel         general path (per-lane unrank + lexicographic successor) for units whose image
//                                                  does not fit the LDS budget
//   ambi_first_kernel             1 wave  / unit   getBFB scan for the first valid order, bkp in LDS
//   ambi_search/resolve_kernel    1 wave  / chunk of orders (only for units whose scan budget ran out)
//   ambi_finish_kernel            1 block / unit   bkp -> path (LDS int16), indelBFB, output junctions
//   ambi_pack_*                   optional end-of-batch packing of the paths for an RCCL gather
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "ambi_backend.hpp"
#include "ambi_ilp_rows.hpp"
#include "ambi_stages.hpp"

namespace ambi {

#define HIP_CK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            fprintf(stderr, "ambigram_hip: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -31;                                                                             \
        }                                                                                           \
    } while (0)

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
extern __shared__ __align__(16) uint8_t ambi_lds[];

// Register budgets of the kernels that share the CUs during a step.  A SIMD holds 512 vector registers per lane and 800 scalar
// registers: five wavefronts of the order-table kernel at 96 vector registers (what the compiler takes when left alone) fill 480
// of them, and a finish wavefront (96) then only fits where an order-table workgroup has been pushed off the CU.  Asking for
// AMBI_*_WAVES wavefronts per SIMD caps the kernel's vector registers at 512 / waves (in granules of 8).
// Measured on the bench batch, builds interleaved on one box (profiles/r04_notes.md): left alone (96 / 95 / 112 registers)
// 0.865 ms per step; order table 80 alone 0.91-0.92 (six of its workgroups then fit a CU and squeeze the finish kernels out);
// order table 80 + lean finish 80 + direct full finish 96: 0.824-0.830.
#ifndef AMBI_ENUM_WAVES
#define AMBI_ENUM_WAVES 6
#endif
#ifndef AMBI_LEAN_WAVES
#define AMBI_LEAN_WAVES 6
#endif
#ifndef AMBI_EXT_WAVES
#define AMBI_EXT_WAVES 5
#endif
#ifndef AMBI_FIRST_WAVES
#define AMBI_FIRST_WAVES 0
#endif
#ifndef AMBI_PREP_WAVES
#define AMBI_PREP_WAVES 0
#endif
#define AMBI_WAVES_ATTR_(n) __attribute__((amdgpu_waves_per_eu(n)))
#define AMBI_WAVES_ATTR(n) AMBI_WAVES_ATTR_(n)
#if AMBI_ENUM_WAVES > 0
#define AMBI_ENUM_ATTR AMBI_WAVES_ATTR(AMBI_ENUM_WAVES)
#else
#define AMBI_ENUM_ATTR
#endif
#if AMBI_LEAN_WAVES > 0
#define AMBI_LEAN_ATTR AMBI_WAVES_ATTR(AMBI_LEAN_WAVES)
#else
#define AMBI_LEAN_ATTR
#endif
#ifndef AMBI_EDIT_WAVES
#define AMBI_EDIT_WAVES 6
#endif
#if AMBI_EDIT_WAVES > 0
#define AMBI_EDIT_ATTR AMBI_WAVES_ATTR(AMBI_EDIT_WAVES)
#else
#define AMBI_EDIT_ATTR
#endif
#if AMBI_EXT_WAVES > 0
#define AMBI_EXT_ATTR AMBI_WAVES_ATTR(AMBI_EXT_WAVES)
#else
#define AMBI_EXT_ATTR
#endif
#if AMBI_FIRST_WAVES > 0
#define AMBI_FIRST_ATTR AMBI_WAVES_ATTR(AMBI_FIRST_WAVES)
#else
#define AMBI_FIRST_ATTR
#endif
#if AMBI_PREP_WAVES > 0
#define AMBI_PREP_ATTR AMBI_WAVES_ATTR(AMBI_PREP_WAVES)
#else
#define AMBI_PREP_ATTR
#endif

constexpr uint32_t kGuardWord = 0xA5B1C3D7u;
// results of one unit as the express kernel mirrors them into the pinned mailbox (MailLayout); whole workgroup
__device__ inline void mail_unit(const BatchArgs& A, int u, bool path_there = false) {   // path_there: the finish stage wrote the path into the slot already
    const UnitIn& U = A.units[u];
    const UnitOut* h = unit_out(A.results, u);
    const UnitLayout L = unit_layout(U.n_seg, U.bkp_cap, U.path_cap, U.out_cap);
    const MailLayout M = mail_layout(U.path_cap, U.out_cap);
    uint8_t* slot = A.mail + A.mail_off[u];
    const uint8_t* res = A.results + U.res_off;
    auto copy8 = [&](uint8_t* dst, const uint8_t* src, int64_t bytes) {   // (every part is padded to 8 bytes on both sides)
        for (int64_t i = threadIdx.x; i < (bytes + 7) / 8; i += blockDim.x) reinterpret_cast<uint64_t*>(dst)[i] = reinterpret_cast<const uint64_t*>(src)[i];
    };
    copy8(slot, reinterpret_cast<const uint8_t*>(h), (int64_t)sizeof(UnitOut));
    if (!path_there) copy8(slot + M.path, res + L.path, (int64_t)sizeof(rcell_t) * h->path_len);
    if (h->path_ind_stored) copy8(slot + M.path_ind, res + L.path_ind, (int64_t)sizeof(rcell_t) * h->path_indel_len);
    copy8(slot + M.out_junc, res + L.out_junc, (int64_t)sizeof(OutJunc) * h->n_out_junc);
    __threadfence_system();
}
// A small batch's input image goes from pinned host memory to HBM and its zero-filled region is cleared by ONE kernel (the
// threads read host memory over the link): a copy command + a fill command cost two engine hand-overs in front of the
// first kernel, several times what moving 15 KB takes.
__global__ __launch_bounds__(256) void ambi_ingest_kernel(const uint4* src, uint4* dst, int64_t n16, uint4* zero, int64_t z16) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = i; k < n16; k += stride) dst[k] = src[k];
    const uint4 z = {0u, 0u, 0u, 0u};
    for (int64_t k = i; k < z16; k += stride) zero[k] = z;
}
// guard words on both sides of every path area of the direct full-finish launch (AMBI_DEBUG: does that kernel leave its slot?)
constexpr int kCellGuardBytes = 64;
__global__ void ambi_guard_fill_kernel(uint8_t* cells, int64_t stride, int slots) {
    const int s = blockIdx.x, t = threadIdx.x;   // 32 threads: 16 words in front of the area, 16 behind
    if (s >= slots) return;
    uint32_t* w = reinterpret_cast<uint32_t*>(cells + (int64_t)s * stride + (t < 16 ? 0 : stride - kCellGuardBytes));
    w[t & 15] = kGuardWord;
}
__global__ void ambi_guard_check_kernel(const uint8_t* cells, int64_t stride, int slots, int32_t* bad) {
    const int s = blockIdx.x, t = threadIdx.x;
    if (s >= slots) return;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(cells + (int64_t)s * stride + (t < 16 ? 0 : stride - kCellGuardBytes));
    if (w[t & 15] != kGuardWord) atomicAdd(bad, 1);
}


__global__ __launch_bounds__(64) AMBI_PREP_ATTR void ambi_prepare_kernel(BatchArgs A) {
    WaveGroup g;
    if (A.zero_pending && blockIdx.x == 0 && threadIdx.x == 0) { *A.n_pending = 0; A.refin_count[0] = 0; A.refin_count[1] = 0; }   // nothing counts pending / handed-over units before the scan
    stage_prepare(g, A, A.unit_base + (int)blockIdx.x, ambi_lds);
}

// Small batches: the whole reconstruction of a unit whose first order assembles, one workgroup per unit (ambi_stages.hpp:
// stage_express) -- junction side and DAG side on two wavefronts at once, then imperfectFBI, then the finish stage on the
// workgroup.  The lattice / order table follow in the kernels behind (ambi_lattice_kernel, plan, enumerate).
__global__ __launch_bounds__(1024) void ambi_express_kernel(BatchArgs A) {
    __shared__ int scratch[40];
    BlockGroup gb(scratch);
    WaveGroup gw;
    if (A.zero_pending && blockIdx.x == 0 && threadIdx.x == 0) *A.n_pending = 0;
    const int wave = threadIdx.x >> 6, u = A.unit_base + (int)blockIdx.x;
    const bool path_there = stage_express(gw, gb, wave < 2 ? wave : 2, A, u, ambi_lds);
    __syncthreads();
    if (A.mail && unit_out(A.results, u)->reserved) { mail_unit(A, u, path_there); __syncthreads(); }   // header, final path(s), output junctions -> pinned host memory
    AMBI_MARK(A, gb, u, 25);
    if (threadIdx.x == 0 && A.express_left) {
        if (!unit_out(A.results, u)->reserved) *A.express_left = 1;
        __threadfence_system();
        if (atomicAdd(A.blocks_done, 1) == (int)gridDim.x - 1) {   // last workgroup: tell the host (it spins on this word)
            *A.blocks_done = 0;
            __threadfence_system();
            *(volatile int32_t*)A.express_seq = A.run_seq;
        }
    }
}
__global__ __launch_bounds__(64) void ambi_lattice_kernel(BatchArgs A) {
    WaveGroup g;
    if (A.zero_pending && blockIdx.x == 0 && threadIdx.x == 0) { A.refin_count[0] = 0; A.refin_count[1] = 0; }   // the lean finish kernel behind fills the list ([1]: the edit kernel's hand-over list)
    stage_lattice(g, A, A.unit_base + (int)blockIdx.x, ambi_lds);
}

__device__ inline int64_t wave_incl_scan_i64(int64_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int64_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// block-wide exclusive scan of one int64 per thread (blockDim.x <= 1024); *total = sum
__device__ inline int64_t block_exscan_i64(int64_t v, int64_t* total, int64_t* sh /*[17]*/) {
    int64_t inc = wave_incl_scan_i64(v);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    int64_t base = 0, tot = 0;
    for (int i = 0; i < nw; i++) { int64_t s = sh[i]; if (i < w) base += s; tot += s; }
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(64) void ambi_lattice_own_kernel(BatchArgs A) {
    WaveGroup g;
    if (A.zero_pending && blockIdx.x == 0 && threadIdx.x == 0) { A.refin_count[0] = 0; A.refin_count[1] = 0; }   // (as ambi_lattice_kernel)
    const int u = A.unit_base + (int)blockIdx.x;
    stage_lattice_own(g, A, u, ambi_lds);
    // the verdict the host waits for on a fresh batch: no lattice failed and the tables of ALL units fit the arena together
    // (then every prefix fits, whatever subset the plan kernel ends up placing)
    if (threadIdx.x == 0 && A.lat_seq) {
        const bool ok = A.lat_status[u] == ST_OK || A.lat_status[u] == ST_ERR_NO_ELEMENTS;   // (no elements: the express stage reports that itself)
        const int64_t bytes = A.lat_status[u] == ST_OK ? order_bytes((int64_t)A.lat_R[u], A.units[u].n_elem, A.order_align) : 0;
        if (!ok) *A.lat_unsure = 1;
        __threadfence_system();
        const unsigned long long total = atomicAdd(reinterpret_cast<unsigned long long*>(A.lat_sum), (unsigned long long)bytes) + (unsigned long long)bytes;
        (void)total;
        __threadfence();
        if (atomicAdd(reinterpret_cast<unsigned long long*>(A.lat_sum + 1), 1ull) == (unsigned long long)gridDim.x - 1) {   // last wave
            const int64_t all = (int64_t)atomicAdd(reinterpret_cast<unsigned long long*>(A.lat_sum), 0ull);
            if (all > A.order_arena_bytes) *A.lat_unsure = 1;
            A.lat_sum[0] = 0; A.lat_sum[1] = 0;
            __threadfence_system();
            *(volatile int32_t*)A.lat_seq = A.run_seq;
        }
    }
}

// before the plan kernel is run a SECOND time over the same headers (tables written on demand, or after the arena grew): every
// unit that has or was refused rows wants rows again
__global__ void ambi_plan_reset_kernel(BatchArgs A) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= A.n_units) return;
    UnitOut* out = unit_out(A.results, A.unit_base + i);
    if (out->order_off >= 0 || out->order_off == kOrderOffNoRoom) out->order_off = kOrderOffWanted;
}

// Parallel form of plan_serial (ambi_stages.hpp): same prefix-sum semantics.
__global__ __launch_bounds__(1024) void ambi_plan_kernel(BatchArgs A) {
    __shared__ int64_t sh[17];
    if (A.lat_R) {   // the lattice ran beside the express kernel: its outcome goes into the headers first
        for (int i = threadIdx.x; i < A.n_units; i += blockDim.x) plan_merge_lattice(A, A.unit_base + i);
        __threadfence();
        __syncthreads();
    }
    // Every thread takes kPlanPer CONSECUTIVE units of a chunk of kPlanPer * blockDim.x: their headers are read together (one round
    // trip to memory per chunk and pass instead of one per unit), the prefix sums inside a thread are serial, and a chunk costs two
    // block scans (offsets; work blocks, which depend on what fits) -- round 3's form read and scanned blockDim.x units at a time: nine
    // scans and five dependent round trips for the bench batch's 4096 units, 27 us with the chip otherwise idle.
    constexpr int kPlanPer = 4;
    const int chunk = kPlanPer * (int)blockDim.x;
    const bool one_chunk = A.n_units <= chunk;
    int64_t R_[kPlanPer]; int K_[kPlanPer]; bool wanted_[kPlanPer];
    auto load_chunk = [&](int base) {
#pragma unroll
        for (int k = 0; k < kPlanPer; k++) {
            const int i = base + (int)threadIdx.x * kPlanPer + k;
            R_[k] = 0; K_[k] = 0; wanted_[k] = false;
            if (i < A.n_units) {
                const UnitOut* o = unit_out(A.results, A.unit_base + i);
                wanted_[k] = o->order_off == kOrderOffWanted;   // (not the status: the scan for the first valid order may be rewriting it)
                R_[k] = o->num_orders; K_[k] = o->K;
            }
        }
    };
    // pass 1: rows of the whole batch -> rows per lane of the enumerate kernel
    int64_t my_rows = 0;
    for (int base = 0; base < A.n_units; base += chunk) {
        load_chunk(base);
#pragma unroll
        for (int k = 0; k < kPlanPer; k++) if (wanted_[k] && R_[k] < (int64_t)kCountSat) my_rows += R_[k];
    }
    int64_t total_rows;
    (void)block_exscan_i64(my_rows, &total_rows, sh);
    const int TL = rows_per_lane_for(total_rows, A.target_lanes);
    int64_t off_carry = 0, blk_carry = 0;
    for (int base = 0; base < A.n_units; base += chunk) {
        if (!one_chunk) load_chunk(base);
        const int T = TL;
        int64_t bytes[kPlanPer], blocks[kPlanPer], sum_b = 0;
        bool live[kPlanPer], toobig[kPlanPer];
#pragma unroll
        for (int k = 0; k < kPlanPer; k++) {
            bytes[k] = 0; blocks[k] = 0; live[k] = false; toobig[k] = false;
            if (wanted_[k]) {
                if (R_[k] >= (int64_t)kCountSat) toobig[k] = true;
                else {
                    live[k] = true;
                    bytes[k] = order_bytes(R_[k], K_[k], A.order_align);
                    blocks[k] = (R_[k] + 256ll * T - 1) / (256ll * T);   // one work block = one workgroup = 4 waves x 64*T rows
                }
            }
            sum_b += bytes[k];
        }
        int64_t tot_b, tot_k;
        int64_t off = off_carry + block_exscan_i64(sum_b, &tot_b, sh);
        // a unit that does not fit contributes no work blocks (and still advances the offset: orders_needed is the true total)
        bool fits[kPlanPer];
        int64_t offs_[kPlanPer], sum_k = 0;
#pragma unroll
        for (int k = 0; k < kPlanPer; k++) {
            offs_[k] = off;
            fits[k] = live[k] && (off + bytes[k] <= A.order_arena_bytes);
            off += bytes[k];
            sum_k += fits[k] ? blocks[k] : 0;
        }
        int64_t blk = blk_carry + block_exscan_i64(sum_k, &tot_k, sh);
#pragma unroll
        for (int k = 0; k < kPlanPer; k++) {
            const int i = base + (int)threadIdx.x * kPlanPer + k;
            if (i < A.n_units) {
                const int u = A.unit_base + i;
                UnitOut* out = unit_out(A.results, u);
                A.blk_off[i] = blk;
                A.rows_per_lane[u] = T;
                if (toobig[k]) out->order_off = kOrderOffNoRoom;
                else if (live[k]) out->order_off = fits[k] ? A.arena_base + offs_[k] : kOrderOffNoRoom;   // (no room: the finish stage turns the status into ORDERS_CAPACITY)
                if ((toobig[k] || (live[k] && !fits[k])) && A.late_flag) { *A.late_flag = 1; __threadfence_system(); }   // a result the express stage published is void
            }
            blk += fits[k] ? blocks[k] : 0;
        }
        off_carry += tot_b;
        blk_carry += tot_k;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        A.blk_off[A.n_units] = blk_carry;
        *A.orders_needed = off_carry;
        if (A.host_needed) *A.host_needed = off_carry;
        if (A.plan_seq) { __threadfence_system(); *(volatile int32_t*)A.plan_seq = A.run_seq; }   // orders_needed and late_flag of this run are final
    }
}

// One wavefront per work block of 64*T consecutive ranks of one unit.  Lane l unranks rank base + l*T from the
// automaton and walks T lexicographic successors; rows leave the registers as 16-byte stores (4 rows per group), so
// every byte of the table is written exactly once and nothing is staged through an LDS tile.
// LDS per wave: [enum_stack_lds] per-lane DFS stacks (depth-major) | [enum_auto_lds] compact copy of the unit's
// automaton (avail masks, child bases, child links) when it fits, else the automaton is read through L2.
template <int CLS>
__global__ __launch_bounds__(256) void ambi_enumerate_kernel(BatchArgs A) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const int wave_bytes = A.enum_stack_lds + A.enum_auto_lds;
    uint8_t* stacks = ambi_lds + (size_t)wave * wave_bytes;
    uint8_t* amem = stacks + A.enum_stack_lds;
    WaveGroup g;
    const int64_t total = A.blk_off[A.n_units];
    int staged_unit = -1;
    bool in_lds = false;
    (void)wpb;
    // general path: only units whose block tables did not fit (flagged by ambi_enumerate_blocks_kernel)
    for (int64_t b = (int64_t)blockIdx.x; b < total; b += (int64_t)gridDim.x) {
        int lo = 0, hi = A.n_units;
        if (b < A.n_units && A.blk_off[b] == b && A.blk_off[b + 1] == b + 1) lo = (int)b;   // one work block per unit so far: two independent reads
        else while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (A.blk_off[mid] <= b) lo = mid; else hi = mid; }
        const int u = A.unit_base + lo;
        const UnitOut* out = unit_out(A.results, u);
        const int K = out->K, T = A.rows_per_lane[u];
        if (enum_class_of(K) != CLS) continue;   // rows of this width belong to another instantiation
        if (!A.unit_fallback[u]) continue;
        const int64_t R = out->num_orders;
        const int64_t base_rank = (b - A.blk_off[lo]) * 256ll * T + (int64_t)wave * 64 * T;
        const IdealTable tbl = unit_ideal_table(A, u);
        const AutoView V = auto_view(tbl);
        const int nI = V.nI, nC = tbl.counter[1];
        const bool wide = K > 32;
        const int msz = wide ? 8 : 4;
        uint8_t* av_l = amem;
        uint32_t* rec_l = reinterpret_cast<uint32_t*>(amem + (size_t)nI * msz);
        uint16_t* cb_l = reinterpret_cast<uint16_t*>(rec_l + nI);
        uint16_t* ch_l = cb_l + nI;
        if (u != staged_unit) {
            g.sync();
            const int64_t need = (int64_t)nI * (msz + 6) + 2ll * nC + 16;
            in_lds = need <= A.enum_auto_lds && nC < 65536;
            if (in_lds) {
                for (int i = lane; i < nI; i += 64) {
                    const uint64_t av = V.avail[i];
                    if (wide) reinterpret_cast<uint64_t*>(av_l)[i] = av;
                    else reinterpret_cast<uint32_t*>(av_l)[i] = (uint32_t)av;
                    cb_l[i] = (uint16_t)V.cbase[i];
                    rec_l[i] = av ? make_rec(av, V.child[V.cbase[i]]) : 0u;
                }
                for (int i = lane; i < nC; i += 64) ch_l[i] = V.child[i];
            }
            staged_unit = u;
            g.sync();
        }
        uint8_t* rows = A.order_arena + out->order_off;   // 16-byte aligned; lane ranges start at multiples of 4 rows
        const int64_t first = base_rank + (int64_t)lane * T;
        if (in_lds) {
            LdsAuto<uint32_t> a32{reinterpret_cast<const uint32_t*>(av_l), rec_l, cb_l, ch_l};
            LdsAuto<uint64_t> a64{reinterpret_cast<const uint64_t*>(av_l), rec_l, cb_l, ch_l};
            enumerate_lane_dispatch<CLS>(a32, a64, V, K, R, first, T, stacks, lane, 64, rows);
        } else {
            GlobalAuto ga{V};
            enumerate_lane_dispatch<CLS>(ga, ga, V, K, R, first, T, stacks, lane, 64, rows);
        }
    }
}

// Builds the block-emission image of every unit whose table is written by MORE THAN ONE workgroup and parks it in HBM
// (a unit with a single work block gets its image built in LDS by the enumerate workgroup itself, below).
__global__ __launch_bounds__(256) void ambi_blocks_build_kernel(BatchArgs A) {
    __shared__ int scratch[40];
    BlockGroup g(scratch);
    const int u = A.unit_base + (int)blockIdx.x;
    const UnitOut* out = unit_out(A.results, u);
    const int64_t nblocks = A.blk_off[blockIdx.x + 1] - A.blk_off[blockIdx.x];
    // (order_off, not status: the scan for the first valid order runs beside this kernel and rewrites the status; the
    // plan kernel gave order_off >= 0 to exactly the units that have rows to write)
    if (out->order_off < 0 || nblocks <= 0 || (nblocks == 1 && A.build_in_emit)) {   // no rows to write / image built by the enumerate kernel
        if (threadIdx.x == 0) A.unit_fallback[u] = 0;
        return;
    }
    const IdealTable tbl = unit_ideal_table(A, u);
    const int K = out->K;
    BlockImageHeader H;
    // the automaton copy sits in LDS, the image is assembled word by word straight in its HBM slot
    uint8_t* slot = A.block_img + (int64_t)u * A.block_lds;
    bool fits = build_block_image(g, tbl, K, row_stride(K) / 4, out->num_orders, A.block_max, ambi_lds, A.block_scratch_lds, slot, A.block_lds, H,
                                  A.stage_clk ? A.stage_clk + (int64_t)u * kStageSlots : nullptr);
    if (!fits && A.block_dfs) {
        // one directory entry per block does not fit (many rows): the directory-free image -- [build tables][suffix rows],
        // walked block by block at emission (emit_blocks_dfs_wave) -- with the largest block size whose suffix rows fit
        BuildTables dummy;
        const int64_t scr = carve_build_tables(ambi_lds, tbl.counter[0], tbl.counter[1], dummy);
        const int64_t budget = (int64_t)A.block_lds - kDfsStateBytes - scr;
        for (int bm = A.block_max; bm >= 8 && !fits && budget > 0 && scr <= A.block_scratch_lds; bm >>= 1) {
            __syncthreads();
            fits = build_block_image(g, tbl, K, row_stride(K) / 4, out->num_orders, bm, ambi_lds, A.block_scratch_lds, slot + scr, budget, H, nullptr, false);
        }
        if (fits) {   // the tables the walk reads travel with the suffix rows
            const uint32_t* src = reinterpret_cast<const uint32_t*>(ambi_lds);
            uint32_t* dst = reinterpret_cast<uint32_t*>(slot);
            for (int64_t i = threadIdx.x; i < scr / 4; i += blockDim.x) dst[i] = src[i];
            H.pad = (int32_t)scr;
            H.image_bytes += (int32_t)scr;
        }
    }
    if (threadIdx.x == 0) {
        *reinterpret_cast<BlockImageHeader*>(A.block_hdr + 8 * (int64_t)u) = H;
        A.unit_fallback[u] = fits ? 0 : 1;
    }
}

// Fast path: block emission (ambi_enum_blocks.hpp).  One workgroup per work block of 256*T rows; the workgroup copies
// the unit's image (block directory + suffix rows) from HBM into LDS, then every wave streams its 64*T rows block by
// block: four LDS reads, four ORs and one fully coalesced 16-byte store per lane and step.
// A unit with ONE work block has no image in HBM: its workgroup builds the image right here in LDS (automaton copy in
// front, image behind it) -- the build is a latency-bound chain that hides under the store stream of the other
// workgroups of the CU -- and the separate build kernel only serves the units that several workgroups share.
// LDS: [block_lds] image (+ automaton copy while building).
template <int CLS>
__global__ __launch_bounds__(1024) AMBI_ENUM_ATTR void ambi_enumerate_blocks_kernel(BatchArgs A) {
    __shared__ int scratch[40];
    BlockGroup g(scratch);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint8_t* tmem = ambi_lds;
    const uint32_t* image = reinterpret_cast<const uint32_t*>(tmem);
    const int64_t total = A.blk_off[A.n_units];
    int staged_unit = -1;
    bool fits = false;
    int nB = 0;
    // directory-free images (header fits == 2): tables at the front of the image, suffix rows behind; per-wave walk state
    // in the last kDfsStateBytes of the workgroup's group memory
    bool dfs = false;
    int dfs_block_max = 0;
    BuildTables Bt;
    const uint32_t* dfs_suf = nullptr;
    uint8_t* wave_state = tmem + A.block_lds - kDfsStateBytes + wave * kDfsWaveStride;
    for (int64_t b = (int64_t)blockIdx.x; b < total; b += (int64_t)gridDim.x) {
        int lo = 0, hi = A.n_units;
        if (b < A.n_units && A.blk_off[b] == b && A.blk_off[b + 1] == b + 1) lo = (int)b;   // one work block per unit so far: two independent reads
        else while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (A.blk_off[mid] <= b) lo = mid; else hi = mid; }
        const int u = A.unit_base + lo;
        const UnitOut* out = unit_out(A.results, u);
        const int K = out->K, T = A.rows_per_lane[u];
        if (enum_class_of(K) != CLS) continue;
        if (A.first_rows && out->num_orders <= A.first_budget) {   // the table is among the rows unranked for the scan
            copy_first_rows(g, A.first_rows + (int64_t)u * A.first_budget * kFirstRowStride, K, out->num_orders, A.order_arena + out->order_off);
            continue;
        }
        if (u != staged_unit) {
            g.sync();
            if (A.build_in_emit && A.blk_off[lo + 1] - A.blk_off[lo] == 1) {
                const IdealTable tbl = unit_ideal_table(A, u);
                BuildTables dummy;
                const int64_t scr = carve_build_tables(tmem, tbl.counter[0], tbl.counter[1], dummy);
                BlockImageHeader H;
                fits = scr < A.block_lds &&
                       build_block_image(g, tbl, K, row_stride(K) / 4, out->num_orders, A.block_max, tmem, scr, tmem + scr, A.block_lds - scr, H);
                dfs = false;
                if (!fits && A.block_dfs) {   // directory too large: tables + suffix rows, walked at emission (as the build kernel does)
                    const int64_t budget = (int64_t)A.block_lds - kDfsStateBytes - scr;
                    for (int bm = A.block_max; bm >= 8 && !fits && budget > 0; bm >>= 1) {
                        g.sync();
                        fits = build_block_image(g, tbl, K, row_stride(K) / 4, out->num_orders, bm, tmem, scr, tmem + scr, budget, H, nullptr, false);
                    }
                    if (fits) {
                        dfs = true;
                        (void)carve_build_tables(tmem, tbl.counter[0], tbl.counter[1], Bt);
                        dfs_suf = reinterpret_cast<const uint32_t*>(tmem + scr);
                        dfs_block_max = H.block_max;
                    }
                }
                nB = H.nB;
                image = reinterpret_cast<const uint32_t*>(tmem + scr);
                if (!fits && threadIdx.x == 0) A.unit_fallback[u] = 1;   // the general enumerate kernel takes the unit
                if (threadIdx.x == 0) {   // what this unit's workgroup really needs of its group memory (the host sizes later launches by it)
                    int32_t* hdr = A.block_hdr + 8 * (int64_t)u;
                    hdr[0] = fits ? (dfs ? 2 : 1) : 0;
                    hdr[3] = (int32_t)(scr + (fits ? H.image_bytes : 0) + (dfs ? kDfsStateBytes : 0));
                }
            } else {
                const BlockImageHeader* hdr = reinterpret_cast<const BlockImageHeader*>(A.block_hdr + 8 * (int64_t)u);
                fits = hdr->fits != 0;
                dfs = hdr->fits == 2;
                nB = hdr->nB;
                image = reinterpret_cast<const uint32_t*>(tmem);
                if (dfs) {
                    (void)carve_build_tables(tmem, hdr->nI, hdr->nC, Bt);
                    dfs_suf = reinterpret_cast<const uint32_t*>(tmem + hdr->pad);
                    dfs_block_max = hdr->block_max;
                }
                if (fits) {   // coalesced copy of the unit's image into LDS
                    const int64_t nvec = ((int64_t)hdr->image_bytes + 15) >> 4;
                    const uint4* src = reinterpret_cast<const uint4*>(A.block_img + (int64_t)u * A.block_lds);
                    uint4* dst = reinterpret_cast<uint4*>(tmem);
                    for (int64_t i = threadIdx.x; i < nvec; i += blockDim.x) dst[i] = src[i];
                }
            }
            staged_unit = u;
            g.sync();
        }
        if (!fits) continue;
        const int64_t R = out->num_orders;   // < 2^32 for every unit that has an image
        // the rows of this work block, split evenly over the workgroup's waves (any row boundary will do: emission
        // handles unaligned heads and tails)
        const int64_t blo = (b - A.blk_off[lo]) * 256ll * T;
        int64_t bhi = blo + 256ll * T;
        if (bhi > R) bhi = R;
        const int nwave = (int)(blockDim.x >> 6);
        const int64_t per = (bhi - blo + nwave - 1) / nwave;
        const int64_t wlo = blo + (int64_t)wave * per;
        int64_t whi = wlo + per;
        if (whi > bhi) whi = bhi;
        if (!dfs && A.emit_interleave) {   // the blocks of the whole work block dealt round-robin to the waves
            if (blo < bhi)
                emit_blocks_dispatch<CLS>(image, nB, K, (uint32_t)blo, (uint32_t)bhi, A.order_arena + out->order_off, lane, lane + 1, wave, nwave);
        } else if (wlo < whi) {
            if (dfs)
                emit_blocks_dfs_dispatch<CLS>(Bt, dfs_suf, K, dfs_block_max, (uint32_t)wlo, (uint32_t)whi, A.order_arena + out->order_off,
                                              reinterpret_cast<uint16_t*>(wave_state), reinterpret_cast<uint32_t*>(wave_state + 128), lane, lane + 1);
            else
                emit_blocks_dispatch<CLS>(image, nB, K, (uint32_t)wlo, (uint32_t)whi,
                                          A.order_arena + out->order_off, lane, lane + 1);
        }
    }
}

// Order tables of the wide units (64..127 nodes, row class 3): 64 workgroups per wide unit, one thread per row -- the row is
// unranked from the unit's completion counts (unrank_wide) and written as 128 or 256 bytes (nodes, then 0xFF).  Rare units, at most
// kWideMaxOrders rows each: written for correctness, not for the roofline.
__global__ __launch_bounds__(256) void ambi_enumerate_wide_kernel(BatchArgs A, const int32_t* wide_units, int n_wide) {
    const int w = blockIdx.y;
    if (w >= n_wide) return;
    const int u = wide_units[w];
    if (u < A.unit_base || u >= A.unit_base + A.n_units) return;   // another slice's unit (its WideUnit may be being rebuilt right now)
    const UnitOut* out = unit_out(A.results, u);
    if (out->order_off < 0 || out->num_orders <= 0) return;
    const WideUnit& X = A.wide[A.wide_index[u]];
    const int K = out->K, stride = row_stride(K);
    uint8_t* rows = A.order_arena + out->order_off;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < out->num_orders; r += (int64_t)gridDim.x * blockDim.x) {
        alignas(16) uint8_t row[kWideNodeCap];
        unrank_wide(X, (uint64_t)r, row);
        for (int d = K; d < stride; d++) row[d] = 0xFF;
        uint4* dst = reinterpret_cast<uint4*>(rows + r * stride);
        const uint4* src = reinterpret_cast<const uint4*>(row);
        for (int q = 0; q < stride / 16; q++) dst[q] = src[q];
    }
}

__global__ __launch_bounds__(64) AMBI_FIRST_ATTR void ambi_first_kernel(BatchArgs A) {
    WaveGroup g;
    stage_first(g, A, A.unit_base + (int)blockIdx.x, ambi_lds);
}

// Slow path: units whose first-valid scan ran out of budget.  One wave per chunk of `chunk` consecutive orders
// (ambi_stages.hpp: stage_search_chunk / stage_resolve).
struct SearchArgs {
    const int32_t* pending;      // [np] unit indices
    const int64_t* chunk_off;    // [np+1] prefix of chunk counts
    SearchSlot* slots;           // [np] per pass: least valid index / least undefined index
    int32_t np, chunk, forward, wave_lds;
};
__global__ __launch_bounds__(256) void ambi_search_kernel(BatchArgs A, SearchArgs S) {
    const int wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    uint8_t* work = ambi_lds + (size_t)wave * S.wave_lds;
    WaveGroup g;
    const int64_t total = S.chunk_off[S.np];
    for (int64_t c = (int64_t)blockIdx.x * wpb + wave; c < total; c += (int64_t)gridDim.x * wpb) {
        int lo = 0, hi = S.np;
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (S.chunk_off[mid] <= c) lo = mid; else hi = mid; }
        const int p = lo, u = S.pending[p];
        const UnitIn& U = A.units[u];
        if (unit_out(A.results, u)->status != ST_PENDING) continue;   // resolved by the previous pass
        const int64_t first = (c - S.chunk_off[p]) * S.chunk;
        if (first >= search_limit(load_now_i64(&S.slots[p].found), load_now_i64(&S.slots[p].err_key))) continue;   // an earlier hit is already known
        FirstWork W = carve_first(work, U.n_seg, U.bkp_cap, U.n_elem > kMaxNodes);
        load_first_work(g, A, u, W);
        stage_search_chunk(g, A, u, W, first, S.chunk, S.forward != 0, &S.slots[p]);
    }
}
__global__ void ambi_search_init_kernel(SearchSlot* slots, int np) {
    const int p = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (p < np) { slots[p].found = kSearchNone; slots[p].err_key = kSearchNone; }
}
// after a search pass: re-evaluate the winning order to materialise its bkp and fill the header
__global__ __launch_bounds__(64) void ambi_resolve_kernel(BatchArgs A, SearchArgs S, int pass) {
    WaveGroup g;
    const int p = blockIdx.x;
    stage_resolve(g, A, S.pending[p], ambi_lds, &S.slots[p], S.forward != 0, pass);
}

// --all (LGM.cpp:3672-3695), fused enumerate + evaluate (ambi_stages.hpp: stage_all_chunk): one wave per 64 consecutive
// orders of a unit, all units of the batch in ONE launch per pass; the orders are unranked from the automaton by the
// lanes, never read from the table.  Output: one 64-bit word of the unit's validity bitmap per wave-chunk.
// Work item c of the launch = word c of the concatenated pass-0 maps (all_off is also the chunk prefix, up to the
// factor 2 for the two passes).  LDS per wave: first-work area + 64 unranked rows.
// One thread per order (stage_all_chunk_lanes) for units whose breakpoint path has at most lane_cap cells -- the rule on
// real inputs (a path of ~100 cells at 256 segments); ambi_all_kernel below keeps the longer ones.
// LDS per wave: DAG + fold-back map | 64 x 64 bytes of transposed orders | lane_cap x 64 cells.
__global__ __launch_bounds__(256) void ambi_all_lanes_kernel(BatchArgs A, int pass, int wave_lds, int64_t total_chunks, int lane_cap, int head_bytes,
                                                            int lane_cells, int auto_bytes, int rows_bytes) {
    const int wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    uint8_t* work = ambi_lds + (size_t)wave * wave_lds;
    WaveGroup g;
    int loaded = -1;
    bool staged_ok = false;
    AutoView SV{};
    for (int64_t c = (int64_t)blockIdx.x * wpb + wave; c < total_chunks; c += (int64_t)gridDim.x * wpb) {
        int lo = 0, hi = A.n_units;
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (A.all_off[mid] <= 2 * c) lo = mid; else hi = mid; }
        const int u = lo;
        const UnitIn& U = A.units[u];
        if (U.bkp_cap > lane_cap || U.n_elem > kMaxNodes) continue;   // (long breakpoint paths and wide units: ambi_all_kernel)
        const int64_t R = unit_out(A.results, u)->num_orders;
        if (!all_chunk_is_mine(A, c, c - A.all_off[u] / 2, R)) continue;   // another rank's chunk
        if (pass == 1 && all_pass0_last_valid(A, u, R)) continue;
        FirstWork W = carve_first(work, U.n_seg, 8);   // (the wavefront form's breakpoint area is not used here)
        uint8_t* rows_t = work + head_bytes;
        cell_t* cells = reinterpret_cast<cell_t*>(rows_t + rows_bytes);
        // the unit's automaton behind the cells: every lane unranks its own order (K dependent steps), from group memory
        // instead of 64 scattered walks through L2
        uint8_t* amem = reinterpret_cast<uint8_t*>(cells) + (size_t)lane_cells * 64 * sizeof(cell_t);
        if (u != loaded) {
            g.sync();
            load_first_work(g, A, u, W);
            const IdealTable T = unit_ideal_table(A, u);
            const int nI = T.counter[0], nC = T.counter[1];
            const int64_t need = 16ll * nI + 4ll * (nI + 1) + 2ll * nC + 16;
            staged_ok = need <= auto_bytes;
            if (staged_ok) {
                uint64_t* av = reinterpret_cast<uint64_t*>(amem);
                uint64_t* cn = av + nI;
                int32_t* cb = reinterpret_cast<int32_t*>(cn + nI);
                uint16_t* ch = reinterpret_cast<uint16_t*>(cb + nI + 1);
                for (int i = threadIdx.x & 63; i < nI; i += 64) { av[i] = T.a_avail[i]; cn[i] = T.a_cnt[i]; }
                for (int i = threadIdx.x & 63; i <= nI; i += 64) cb[i] = T.a_cbase[i];
                for (int i = threadIdx.x & 63; i < nC; i += 64) ch[i] = T.a_child[i];
                SV = AutoView{av, cn, cb, ch, nI};
            }
            loaded = u;
            g.sync();
        }
        stage_all_chunk_lanes(g, A, u, W, rows_t, cells, c - A.all_off[u] / 2, pass, staged_ok ? &SV : nullptr);
        g.sync();
    }
}
__global__ __launch_bounds__(256) void ambi_all_kernel(BatchArgs A, int pass, int wave_lds, int64_t total_chunks, int lane_cap) {
    const int wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    uint8_t* work = ambi_lds + (size_t)wave * wave_lds;
    WaveGroup g;
    for (int64_t c = (int64_t)blockIdx.x * wpb + wave; c < total_chunks; c += (int64_t)gridDim.x * wpb) {
        // all_off[u] = 2 * (chunks of the units before u)
        int lo = 0, hi = A.n_units;
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (A.all_off[mid] <= 2 * c) lo = mid; else hi = mid; }
        const int u = lo;
        const int64_t R = unit_out(A.results, u)->num_orders;
        if (pass == 1 && all_pass0_last_valid(A, u, R)) continue;   // no orientation flip for this unit (LGM.cpp:3691-3695)
        const UnitIn& U = A.units[u];
        const bool wide = U.n_elem > kMaxNodes;
        if (U.bkp_cap <= lane_cap && !wide) continue;                // taken by ambi_all_lanes_kernel
        if (!all_chunk_is_mine(A, c, c - A.all_off[u] / 2, R)) continue;   // another rank's chunk
        FirstWork W = carve_first(work, U.n_seg, U.bkp_cap, wide);
        uint8_t* rows = work + first_work_bytes(U.n_seg, U.bkp_cap, wide);
        g.sync();
        load_first_work(g, A, u, W);
        stage_all_chunk(g, A, u, W, rows, c - A.all_off[u] / 2, pass);
    }
}
__global__ __launch_bounds__(256) void ambi_all_finalize_kernel(BatchArgs A) {
    const int u = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (u < A.n_units) all_finalize_unit(A, u);
}
// Paths of a list of orders of one unit (the valid ones, in print order): one wave per order, breakpoints in LDS,
// the expanded path (absolute signed ids) straight to cells[j * stride ...], its length to lengths[j].
__global__ __launch_bounds__(64) void ambi_order_paths_kernel(BatchArgs A, int u, int forward, const int64_t* order_idx, int32_t* lengths,
                                                              int32_t* cells, int64_t stride) {
    WaveGroup g;
    const UnitIn& U = A.units[u];
    const int j = blockIdx.x;
    const bool wide = U.n_elem > kMaxNodes;
    FirstWork W = carve_first(ambi_lds, U.n_seg, U.bkp_cap, wide);
    int32_t* offs = reinterpret_cast<int32_t*>(ambi_lds + first_work_bytes(U.n_seg, U.bkp_cap, wide));
    load_first_work(g, A, u, W);
    int L = 0;
    const int v = eval_indexed(g, A, u, W, order_idx[j], forward != 0, &L);
    int P = -1;
    if (v == 1) P = expand_bkp(g, W.bkp, L, (cell_t*)nullptr, (int)(stride < U.path_cap ? stride : U.path_cap), offs, cells + (int64_t)j * stride, U.seg_base);
    if (g.tid() == 0) lengths[j] = P;
}

// Full finish stage.  unit_list == nullptr: every unit of the slice (one workgroup each).  With a list: the listed units;
// list_count == nullptr: one workgroup per entry (host-built list of the slow path), else the list is the one the lean
// kernel in front of this launch filled on the device (refin_list / refin_count) and the workgroups share it in strides
// -- no host round trip between the two kernels.
__global__ __launch_bounds__(1024) void ambi_finish_kernel(BatchArgs A, const int32_t* unit_list, const int32_t* list_count, int fixed_count = -1) {
    __shared__ int scratch[40];
    BlockGroup g(scratch);
    if (unit_list && (list_count || fixed_count >= 0)) {
        const int n = list_count ? *list_count : fixed_count;
        for (int i = (int)blockIdx.x; i < n; i += (int)gridDim.x) {
            if (!plan_refused(g, unit_out(A.results, unit_list[i])) && !unit_out(A.results, unit_list[i])->reserved) stage_finish(g, A, unit_list[i], ambi_lds);   // (reserved: done by the express kernel; refusal first, as in the lean stage)
            __syncthreads();
        }
        return;
    }
    const int u = unit_list ? unit_list[blockIdx.x] : A.unit_base + (int)blockIdx.x;
    // the scan kernel is complete (stream order): its count of units left for the parallel search goes to the host
    if (A.host_pending && !unit_list && blockIdx.x == 0 && threadIdx.x == 0) *A.host_pending = *A.n_pending;
    stage_finish(g, A, u, ambi_lds);
}
// The direct full-stage launch with the path cells in device memory (stage_finish<true>): a workgroup works in its own
// slot of `cells` (stride bytes apart; gridDim.x slots) for every unit it takes.
__global__ __launch_bounds__(1024) AMBI_EXT_ATTR void ambi_finish_ext_kernel(BatchArgs A, const int32_t* unit_list, int count, uint8_t* cells, int64_t stride,
                                                                             const int32_t* count_ptr = nullptr) {
    __shared__ int scratch[40];
    BlockGroup g(scratch);
    if (count_ptr) count = *count_ptr;   // a list the kernel in front of this one filled on the device (ambi_finish_edit_kernel)
    for (int i = (int)blockIdx.x; i < count; i += (int)gridDim.x) {
        if (!plan_refused(g, unit_out(A.results, unit_list[i])) && !unit_out(A.results, unit_list[i])->reserved)
            stage_finish<true>(g, A, unit_list[i], ambi_lds, reinterpret_cast<cell_t*>(cells + (int64_t)blockIdx.x * stride + kCellGuardBytes));
        __syncthreads();
    }
}

// The direct launch for units whose SVs may edit the path, on the runs of the path (stage_finish_edit); what it hands on goes
// to `hand_list` for an ambi_finish_ext_kernel launch behind it.
__global__ __launch_bounds__(256) AMBI_EDIT_ATTR void ambi_finish_edit_kernel(BatchArgs A, const int32_t* unit_list, int count, int32_t* hand_list, int32_t* hand_count) {
    __shared__ int scratch[40];
    BlockGroup g(scratch);
    for (int i = (int)blockIdx.x; i < count; i += (int)gridDim.x) {
        stage_finish_edit(g, A, unit_list[i], ambi_lds, hand_list, hand_count);
        __syncthreads();
    }
}

// Lean finish (ambi_stages.hpp: stage_finish_lean): every unit of the slice; units it cannot take are counted in
// n_pending with status ST_REFINISH.  The last workgroup to finish reports n_pending to the host.
__global__ __launch_bounds__(256) AMBI_LEAN_ATTR void ambi_finish_lean_kernel(BatchArgs A, const int32_t* unit_list = nullptr, int list_count = 0) {
    __shared__ int scratch[40];
    BlockGroup g(scratch);
    if (unit_list) {   // the listed units only (the units the parallel search has just resolved)
        for (int i = (int)blockIdx.x; i < list_count; i += (int)gridDim.x) {
            stage_finish_lean(g, A, unit_list[i], ambi_lds);
            __syncthreads();
        }
        return;
    }
    // a workgroup takes every gridDim.x-th unit: the grid is sized to the number of workgroups that should be resident
    // at a time (one per CU fits beside the enumerate workgroups), not to the batch
    for (int i = (int)blockIdx.x; i < A.n_units; i += (int)gridDim.x) {
        stage_finish_lean(g, A, A.unit_base + i, ambi_lds);
        __syncthreads();
    }
    if (A.host_pending) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            if (atomicAdd(A.blocks_done, 1) == (int)gridDim.x - 1) {
                *A.host_pending = atomicAdd(A.n_pending, 0);
                *A.blocks_done = 0;
            }
        }
    }
}

// The lean stage with ONE WAVEFRONT per unit: the stage is a chain of short dependent phases (offsets, SV collection, look-ups,
// output junctions) in which the 256 threads of the workgroup form mostly wait for each other at barriers -- a CU gets through one
// unit per ~10 us whether one or five such workgroups are resident (profiles/r04_notes.md).  A lone wavefront has no barriers, a
// quarter of the registers and the same group memory per unit, so several units per CU are in flight beside the order-table kernel.
__global__ __launch_bounds__(64) AMBI_LEAN_ATTR void ambi_finish_lean_wave_kernel(BatchArgs A) {
    WaveGroup g;
    for (int i = (int)blockIdx.x; i < A.n_units; i += (int)gridDim.x) {
        stage_finish_lean(g, A, A.unit_base + i, ambi_lds);
        g.sync();
    }
    if (A.host_pending && threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(A.blocks_done, 1) == (int)gridDim.x - 1) {
            *A.host_pending = atomicAdd(A.n_pending, 0);
            *A.blocks_done = 0;
        }
    }
}

__global__ __launch_bounds__(1024) void ambi_pack_scan_kernel(BatchArgs A, int which, int32_t* lengths, int64_t* pack_off, int64_t* total) {
    __shared__ int64_t sh[17];
    int64_t carry = 0;
    for (int base = 0; base < A.n_units; base += blockDim.x) {
        const int u = base + (int)threadIdx.x;
        int64_t len = 0;
        if (u < A.n_units) {
            const UnitOut* h = unit_out(A.results, u);
            len = which ? h->path_indel_len : h->path_len;
            lengths[u] = (int32_t)len;
        }
        int64_t tot;
        int64_t ex = block_exscan_i64(len, &tot, sh);
        if (u < A.n_units) pack_off[u] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) { pack_off[A.n_units] = carry; if (total) *total = carry; }
}
__global__ __launch_bounds__(256) void ambi_pack_copy_kernel(BatchArgs A, int which, const int64_t* pack_off, int32_t* cells, int64_t cap) {
    const int u = blockIdx.x;
    const UnitIn& U = A.units[u];
    const UnitLayout L = unit_layout(U.n_seg, U.bkp_cap, U.path_cap, U.out_cap);
    const bool stored = which && unit_out(A.results, u)->path_ind_stored;   // else the edited path equals `path`
    const rcell_t* src = reinterpret_cast<const rcell_t*>(A.results + U.res_off + (stored ? L.path_ind : L.path));
    const int64_t off = pack_off[u], len = pack_off[u + 1] - off;
    for (int64_t i = threadIdx.x; i < len; i += blockDim.x)
        if (off + i < cap) cells[off + i] = abs_cell(src[i], U.seg_base);
}

// ---- run-length form of the final paths (payload of the end-of-batch exchange) ----
// A run starts where a cell is not its predecessor + 1.  One workgroup per unit, two passes over the path in the result
// blob: count the runs (all units) -> offsets (one scan) -> write {start value, length}.
__device__ inline const rcell_t* unit_final_path(const BatchArgs& A, int u, int which, int* len) {
    const UnitIn& U = A.units[u];
    const UnitOut* h = unit_out(A.results, u);
    const UnitLayout L = unit_layout(U.n_seg, U.bkp_cap, U.path_cap, U.out_cap);
    const bool stored = which && h->path_ind_stored;   // else the edited path equals `path`
    *len = which ? h->path_indel_len : h->path_len;
    return reinterpret_cast<const rcell_t*>(A.results + U.res_off + (stored ? L.path_ind : L.path));
}
__global__ __launch_bounds__(256) void ambi_pack_runs_count_kernel(BatchArgs A, int which, int32_t* lengths, int32_t* run_counts) {
    __shared__ int scratch[40];
    BlockGroup g(scratch);
    const int u = blockIdx.x;
    int P;
    const rcell_t* src = unit_final_path(A, u, which, &P);
    int mine = 0;
    for (int i = threadIdx.x; i < P; i += blockDim.x) mine += (i == 0 || src[i] != src[i - 1] + 1) ? 1 : 0;
    const int total = g.sum_i32(mine);
    if (threadIdx.x == 0) { run_counts[u] = total; lengths[u] = P; }
}
__global__ __launch_bounds__(1024) void ambi_pack_runs_scan_kernel(BatchArgs A, const int32_t* lengths, const int32_t* run_counts, int64_t* run_off,
                                                                   int64_t* totals) {
    __shared__ int64_t sh[17];
    int64_t carry = 0, cells = 0;
    for (int base = 0; base < A.n_units; base += blockDim.x) {
        const int u = base + (int)threadIdx.x;
        const int64_t c = u < A.n_units ? run_counts[u] : 0, l = u < A.n_units ? lengths[u] : 0;
        int64_t tot, totl;
        const int64_t ex = block_exscan_i64(c, &tot, sh);
        (void)block_exscan_i64(l, &totl, sh);
        if (u < A.n_units) run_off[u] = carry + ex;
        carry += tot; cells += totl;
    }
    if (threadIdx.x == 0) { run_off[A.n_units] = carry; if (totals) { totals[0] = carry; totals[1] = cells; } }
}
__global__ __launch_bounds__(256) void ambi_pack_runs_write_kernel(BatchArgs A, int which, const int64_t* run_off, int32_t* run_start,
                                                                   int32_t* run_len, int64_t cap) {
    __shared__ int scratch[40];
    BlockGroup g(scratch);
    const int u = blockIdx.x;
    int P;
    const rcell_t* src = unit_final_path(A, u, which, &P);
    const int64_t off = run_off[u];
    const int n = (int)(run_off[u + 1] - off);
    if (off + n > cap) return;   // the caller's buffers are too small: nothing is written for this unit (totals tell)
    int done = 0;
    for (int base = 0; base < P; base += blockDim.x) {      // run starts in path order: value and, for now, position
        const int i = base + (int)threadIdx.x;
        const int flag = (i < P && (i == 0 || src[i] != src[i - 1] + 1)) ? 1 : 0;
        int tot;
        const int ex = g.exscan_i32(flag, &tot);
        if (flag) { run_start[off + done + ex] = abs_cell(src[i], A.units[u].seg_base); run_len[off + done + ex] = i; }
        done += tot;
    }
    __syncthreads();
    // positions -> lengths (the next run's position is read before anyone overwrites it: two phases)
    for (int base = 0; base < n; base += blockDim.x) {
        const int k = base + (int)threadIdx.x;
        int len = 0;
        if (k < n) len = (k + 1 < n ? run_len[off + k + 1] : P) - run_len[off + k];
        __syncthreads();
        if (k < n) run_len[off + k] = len;
        __syncthreads();
    }
}
// one wavefront per run
__global__ __launch_bounds__(256) void ambi_expand_runs_kernel(const int32_t* run_start, const int32_t* run_len, const int64_t* cell_off, int64_t n_runs,
                                                               int32_t* cells, int64_t cap) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r = wave; r < n_runs; r += nwaves) {
        const int32_t s = run_start[r], len = run_len[r];
        const int64_t o = cell_off[r];
        for (int k = lane; k < len; k += 64) if (o + k < cap) cells[o + k] = s + k;
    }
}

// ---- do two streams dispatch side by side? ----
// A kernel with a backlog of workgroups on stream a, one tiny workgroup on stream b: if b's workgroup only starts when a's backlog
// has been handed out, the two streams feed the same dispatch pipe of the command processor (queues of one pipe hand out their
// workgroups one kernel after the other), and a finish kernel on b would run BEHIND an order-table kernel on a instead of beside
// it (measured: 1.28 instead of 0.86 ms per step, profiles/r04_notes.md).  The engine asks this question about the caller's
// stream and its own side streams instead of assuming a mapping of streams to hardware queues.
__global__ __launch_bounds__(256) void ambi_spin_kernel(int64_t ticks, int32_t* sink) {
    const int64_t t0 = (int64_t)wall_clock64();
    while ((int64_t)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (sink && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *sink = 1;
}
__global__ void ambi_touch_kernel(int32_t* sink) { if (sink && blockIdx.x == 0x7fffffff) *sink = 1; }
// microseconds from the start of the backlog kernel on a to the end of the tiny kernel on b (both streams idle before and after)
static int stream_probe_us(hipStream_t a, hipStream_t b, float* us) {
    static hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    static int64_t ticks_per_us = 0;
    if (!e0) { HIP_CK(hipEventCreate(&e0)); HIP_CK(hipEventCreate(&e1)); HIP_CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming)); }
    if (!ticks_per_us) { int dev = 0, khz = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000; ticks_per_us = khz / 1000 > 0 ? khz / 1000 : 100; }
    HIP_CK(hipStreamSynchronize(a)); HIP_CK(hipStreamSynchronize(b));
    // 16 rounds of workgroups that hold a CU slot for ~6 us each: ~100 us of backlog
    int ncu = 256; { hipDeviceProp_t pr; int dev = 0; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ncu = pr.multiProcessorCount; }
    HIP_CK(hipEventRecord(e0, a));
    hipLaunchKernelGGL(ambi_spin_kernel, dim3(ncu * 8 * 16), dim3(256), 0, a, (int64_t)(6 * ticks_per_us), (int32_t*)nullptr);
    HIP_CK(hipEventRecord(e2, a));   // (b must not start before a's kernel has been queued: it waits for nothing of a's, only the host order matters)
    hipLaunchKernelGGL(ambi_touch_kernel, dim3(1), dim3(64), 0, b, (int32_t*)nullptr);
    HIP_CK(hipEventRecord(e1, b));
    HIP_CK(hipStreamSynchronize(a)); HIP_CK(hipStreamSynchronize(b));
    float ms = 0;
    HIP_CK(hipEventElapsedTime(&ms, e0, e1));
    *us = ms * 1e3f;
    return 0;
}
int backend_stream_probe(void* a, void* b, float* us) { return stream_probe_us((hipStream_t)a, (hipStream_t)b, us); }

// The dynamic-LDS ceiling of a kernel is a per-function, process-wide attribute: every batch asks for the device limit,
// so that a second batch with smaller units cannot lower it under a first batch that is still launching (the launches
// themselves request only what their units need).
constexpr int kLdsMaxDynamic = 160 * 1024 - 1024;

// ------------------------------------------------------------------------------------------------
// Per-device resources with PROCESS lifetime.
//
// Round 2 created and destroyed 3-4 streams (one of them with a dispatch priority), 6+ events and three pinned allocations
// with every batch, and a long create / run / destroy loop showed rare stray writes into host memory of the process
// (DESIGN.md 8b).  Nothing the HIP runtime hands out is destroyed per batch any more: a batch LEASES a context -- side
// streams, events, the pinned words the kernels report through, a pinned staging block, a pinned result mailbox and
// grow-only device blocks -- from a per-device free list and gives it back when it is destroyed (after synchronising every
// stream it used).  Leases are never destroyed; at process exit they are left to the runtime.
// ------------------------------------------------------------------------------------------------
struct PinnedWords {                    // what kernels write into host memory (BatchArgs::host_pending, host_needed, express_*, plan_seq, late_flag)
    uint32_t guard_lo[16];
    int32_t npending, express_left, express_seq, plan_seq, late_flag, lat_seq, lat_unsure, pad_[9];
    int64_t needed[16];
    uint32_t guard_hi[16];
};
struct TimingEvents { const char* name; hipEvent_t a, b; };
struct Lease {
    int device = 0;
    hipStream_t side[3][3] = {};        // [back, full, first][default, lowest, highest priority], created on first use
    hipEvent_t ev_fork = nullptr, ev_prep = nullptr, ev_back = nullptr, ev_first = nullptr, ev_full = nullptr, ev_plan = nullptr, ev_express = nullptr, ev_lat = nullptr, ev_tail = nullptr;
    PinnedWords* h_words = nullptr; PinnedWords* dh_words = nullptr;
    uint8_t* d_block = nullptr; int64_t d_block_bytes = 0;      // inputs + working set + result blob of the batch
    uint8_t* d_arena = nullptr; int64_t d_arena_bytes = 0;      // order tables
    uint8_t* d_cells = nullptr; int64_t d_cells_bytes = 0;      // path areas of the direct full-finish launch
    uint8_t* h_stage = nullptr; uint8_t* dh_stage = nullptr; int64_t h_stage_bytes = 0;   // pinned image of the inputs (one H2D copy, or read by the ingest kernel)
    uint8_t* h_mail = nullptr; uint8_t* dh_mail = nullptr; int64_t h_mail_bytes = 0;   // pinned result mailbox (express path)
    // final paths in run-length form on their way to the host (runs_to_host): two slots, each a device block and its pinned mirror,
    // a copy stream of the lease's own and two events per slot (packed / arrived)
    uint8_t* d_runs[2] = {nullptr, nullptr}; int64_t d_runs_bytes[2] = {0, 0};
    uint8_t* h_runs[2] = {nullptr, nullptr}; int64_t h_runs_bytes[2] = {0, 0};
    hipStream_t run_stream = nullptr;   // own_stream(): for callers without a stream (ambi_batch_run_sharded's shares)
    hipStream_t copy_stream = nullptr; hipEvent_t ev_runs_packed[2] = {nullptr, nullptr}, ev_runs_done[2] = {nullptr, nullptr};
    std::vector<TimingEvents> evs;
    std::vector<hipStream_t> slice_streams; std::vector<hipEvent_t> slice_events;     // AMBI_SLICES experiments
    long uses = 0;
    int32_t seq = 0;   // run sequence numbers (the kernels report completion by storing the run's number into a pinned word)
};
// what stays cached in a lease between batches (larger blocks go back to the device when the batch is destroyed)
constexpr int64_t kKeepBlock = 64ll << 20, kKeepArena = 256ll << 20, kKeepCells = 64ll << 20, kKeepStage = 16ll << 20, kKeepMail = 8ll << 20, kKeepRuns = 16ll << 20;

class DevicePool {
    std::mutex mu_;
    std::vector<Lease*> free_;
  public:
    static DevicePool& get() { static DevicePool* p = new DevicePool(); return *p; }   // (never destroyed: no HIP call from a static destructor)
    int acquire(Lease** out) {
        int dev = 0;
        HIP_CK(hipGetDevice(&dev));
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (size_t i = 0; i < free_.size(); i++)
                if (free_[i]->device == dev) { *out = free_[i]; free_.erase(free_.begin() + (long)i); (*out)->uses++; return 0; }
        }
        {   // the engine's streams want hardware queues of their own: say so when the runtime was started without them
            static std::once_flag warned;
            std::call_once(warned, [] {
                const char* q = getenv("GPU_MAX_HW_QUEUES");
                if (!q || atoi(q) < 8)
                    fprintf(stderr, "ambigram_hip: warning: GPU_MAX_HW_QUEUES is %s when the engine creates its streams; with fewer than 8 hardware queues two of "
                                    "them can share one and the finish kernels then run behind the order-table kernel instead of beside it (measured: 1.55 "
                                    "instead of 1.10 ms per 4096-sample step).  Set GPU_MAX_HW_QUEUES=8 in the environment before the HIP runtime starts.\n", q ? q : "unset");
            });
        }
        Lease* L = new Lease();
        L->device = dev;
        hipEvent_t* evs[] = {&L->ev_fork, &L->ev_prep, &L->ev_back, &L->ev_first, &L->ev_full, &L->ev_plan, &L->ev_express, &L->ev_lat, &L->ev_tail};
        for (hipEvent_t* e : evs) HIP_CK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        HIP_CK(hipHostMalloc((void**)&L->h_words, sizeof(PinnedWords)));
        memset(L->h_words, 0, sizeof(PinnedWords));
        for (int i = 0; i < 16; i++) { L->h_words->guard_lo[i] = kGuardWord; L->h_words->guard_hi[i] = kGuardWord; }
        HIP_CK(hipHostGetDevicePointer((void**)&L->dh_words, L->h_words, 0));
        L->uses = 1;
        *out = L;
        return 0;
    }
    void release(Lease* L) {
        if (!L) return;
        std::lock_guard<std::mutex> lk(mu_);
        free_.push_back(L);
    }
};
// a side stream of the lease: kind 0 lean finish / scan, 1 direct full finish, 2 scan ahead; prio 0 default, 1 lowest, 2 highest
static int lease_stream(Lease* L, int kind, int prio, hipStream_t* out) {
    static const bool shared = [] { const char* e = ambi_env("AMBI_SHARE_STREAMS"); return e && atoi(e) != 0; }();
    if (shared && L->device >= 0 && L->device < 16) {
        // experiment: one set of side streams per DEVICE, used by every lease (several resident batches then need no more hardware queues than one)
        static std::recursive_mutex mu; static Lease* holder[16] = {};
        std::lock_guard<std::recursive_mutex> lk(mu);
        if (!holder[L->device]) holder[L->device] = L;
        if (holder[L->device] != L) { const int rc = lease_stream(holder[L->device], kind, prio, out); if (!rc) L->side[kind][prio] = *out; return rc; }
    }
    if (!L->side[kind][prio]) {
        int least = 0, greatest = 0;
        if (prio != 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) {
            HIP_CK(hipStreamCreateWithPriority(&L->side[kind][prio], hipStreamNonBlocking, prio == 1 ? least : greatest));
        } else {
            if (prio != 0) { (void)hipGetLastError(); return lease_stream(L, kind, 0, out); }   // no priorities on this device
            HIP_CK(hipStreamCreateWithFlags(&L->side[kind][0], hipStreamNonBlocking));
        }
    }
    *out = L->side[kind][prio];
    return 0;
}
// ---- side streams that really run beside the caller's stream ----
// The command processor hands out workgroups through four dispatch pipes; the streams of a process are spread over them in
// creation order, and two streams on one pipe hand out their kernels' workgroups one kernel after the other.  Which pipe a stream
// got cannot be asked, only observed (stream_probe_us), so the pool keeps eight candidate streams per device, sorted into
// classes of streams that do NOT run side by side, learns the class of every caller's stream when it first sees it, and gives a
// batch side streams from the other classes.  (Round 3 relied on the creation order: correct for the legacy default stream in a
// process that creates no other streams first, 1.28 instead of 0.86 ms per step on a stream of torch's pool.)
struct StreamClasses {
    std::vector<hipStream_t> cand; std::vector<int> cls;    // candidates and their classes
    std::vector<hipStream_t> rep;                            // one stream per class (class index = position)
    std::vector<std::pair<hipStream_t, uint32_t>> callers;   // callers' streams seen so far, with the SET of classes each collides with
    bool built = false;
};
constexpr float kProbeSharedUs = 55.f;   // measured: 10-17 us side by side, 105-125 us behind the backlog
static std::mutex g_classes_mu;
static StreamClasses g_classes[16];
static int stream_class_of(StreamClasses& C, hipStream_t s, int* out) {   // -1: runs beside every known class
    for (size_t c = 0; c < C.rep.size(); c++) {
        float us = 0;
        if (int rc = stream_probe_us(C.rep[c], s, &us)) return rc;
        if (us > kProbeSharedUs) { *out = (int)c; return 0; }
    }
    *out = -1;
    return 0;
}
static int build_stream_classes(StreamClasses& C) {
    if (C.built) return 0;
    { float us; hipStream_t t0, t1;   // first use of the probe kernels (code loading) outside the measurements
      HIP_CK(hipStreamCreateWithFlags(&t0, hipStreamNonBlocking)); HIP_CK(hipStreamCreateWithFlags(&t1, hipStreamNonBlocking));
      C.cand.push_back(t0); C.cand.push_back(t1);
      if (int rc = stream_probe_us(t0, t1, &us)) return rc; }
    // eight candidates; up to sixteen while fewer than four classes have shown up (other libraries' streams -- RCCL's -- sit between
    // the engine's in creation order)
    for (size_t i = 0; i < 16; i++) {
        if (i >= 8 && C.rep.size() >= 4) break;
        if (i >= C.cand.size()) { hipStream_t t; HIP_CK(hipStreamCreateWithFlags(&t, hipStreamNonBlocking)); C.cand.push_back(t); }
        int c = -1;
        if (int rc = stream_class_of(C, C.cand[i], &c)) return rc;
        if (c < 0) { c = (int)C.rep.size(); C.rep.push_back(C.cand[i]); }
        C.cls.push_back(c);
    }
    C.built = true;
    if (ambi_env("AMBI_DEBUG")) { fprintf(stderr, "ambigram_hip: %zu dispatch classes among %zu candidate streams:", C.rep.size(), C.cand.size()); for (int c : C.cls) fprintf(stderr, " %d", c); fprintf(stderr, "\n"); }
    return 0;
}
// side streams for a batch whose kernels start on `caller`: out[k], k = 0 lean finish, 1 direct full finish, 2 scan / lattice -- from
// classes other than the caller's and, while there are enough classes, from different ones; `set`: which of the candidates of a class
static int classified_side_streams(int device, hipStream_t caller, int set, hipStream_t out[3]) {
    if (device < 0 || device >= 16) return -31;
    std::lock_guard<std::mutex> lk(g_classes_mu);
    StreamClasses& C = g_classes[device];
    if (int rc = build_stream_classes(C)) return rc;
    // Every class the caller's stream collides with, in either direction: a stream can sit on the dispatch pipe of one class AND share a
    // hardware queue with a candidate of another (the legacy default stream of a process that creates it after the candidates: the lattice
    // kernel of a single sample then ran BEHIND the express kernel on one queue, 142 instead of 89 us end to end from a C caller).
    const int nc = (int)C.rep.size();
    uint32_t mask = 0; bool known = false;
    for (auto& pr : C.callers) if (pr.first == caller) { mask = pr.second; known = true; }
    if (!known) {
        for (size_t i = 0; i < C.cand.size(); i++) if (C.cand[i] == caller) { mask = 1u << C.cls[i]; known = true; }
        if (!known) {
            for (int c = 0; c < nc; c++) {
                float ab = 0, ba = 0;
                if (int rc = stream_probe_us(C.rep[c], caller, &ab)) return rc;
                if (ab <= kProbeSharedUs) { if (int rc = stream_probe_us(caller, C.rep[c], &ba)) return rc; }
                if (ab > kProbeSharedUs || ba > kProbeSharedUs) mask |= 1u << c;
            }
        }
        C.callers.push_back({caller, mask});
        if (ambi_env("AMBI_DEBUG")) fprintf(stderr, "ambigram_hip: caller's stream %p collides with the dispatch classes 0x%x (of %d)\n", (void*)caller, mask, nc);
    }
    int cc = -1;
    for (int c = 0; c < nc; c++) if ((mask >> c) & 1u) { cc = c; break; }
    int k = 0;
    for (int step = 1; step <= nc && k < 3; step++) {
        const int c = ((cc < 0 ? 0 : cc) + step) % nc;
        if ((mask >> c) & 1u) continue;
        // the set-th candidate of class c (wrapping)
        std::vector<hipStream_t> of;
        for (size_t i = 0; i < C.cand.size(); i++) if (C.cls[i] == c) of.push_back(C.cand[i]);
        if (of.empty()) continue;
        out[k++] = of[(size_t)set % of.size()];
    }
    for (int j = k; j < 3; j++) out[j] = k > 0 ? out[j % k] : nullptr;   // fewer classes than streams: some share
    return k > 0 ? 0 : 1;   // 1: every stream behaves like the caller's (a profiler that serialises the kernels): nothing to choose
}

// grow-only block of the lease (device memory, or pinned host memory with its device address)
static int lease_device_block(uint8_t** p, int64_t* have, int64_t want) {
    if (*have >= want && *p) return 0;
    if (*p) { (void)hipFree(*p); *p = nullptr; *have = 0; }
    want = (want + 4095) & ~int64_t(4095);
    HIP_CK(hipMalloc((void**)p, (size_t)want));
    *have = want;
    return 0;
}
static int lease_pinned_block(uint8_t** p, uint8_t** dp, int64_t* have, int64_t want) {
    if (*have >= want && *p) return 0;
    if (*p) { (void)hipHostFree(*p); *p = nullptr; *have = 0; }
    want = (want + 4095) & ~int64_t(4095);
    HIP_CK(hipHostMalloc((void**)p, (size_t)want));
    if (dp) HIP_CK(hipHostGetDevicePointer((void**)dp, *p, 0));
    *have = want;
    return 0;
}

// scoped device allocation / event pair for the helpers that allocate per call (freed on every return path)
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};
// Every backend entry that touches the GPU runs on the device of the batch's lease, whatever device the calling thread has
// current (a batch dealt over several devices answers its per-unit getters from the device that holds the unit; a host
// program working on another device finds its current device unchanged afterwards).  dev < 0: nothing to select.
struct DeviceGuard {
    int prev = -1; bool switched = false;
    explicit DeviceGuard(int dev) {
        if (dev < 0 || hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); return; }
        if (prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete; DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// ------------------------------------------------------------------------------------------------
// backend
// ------------------------------------------------------------------------------------------------
class HipBackend : public Backend {
    const HostBatch* hbp_ = nullptr;   // the batch's packed inputs (owned by the ambi_batch that owns this backend)
    const HostBatch& hb() const { return *hbp_; }
    EngineConfig cfg_;
    bool uploaded_ = false, timing_ = false, arena_checked_ = false, ran_ = false;
    bool inflight_ = false;            // a run has been queued and not yet waited for
    bool upload_pending_ = false;      // the input image sits in pinned staging: the first run queues its copy (and the zero fill) ahead of the kernels
    bool late_refusal_ = false;        // a unit of this batch is refused by the lattice / plan stage (known after one complete run: inputs are immutable)
    bool tuned_ = false;               // the launch parameters that depend on the first run's outcome have been chosen
    uint32_t timing_mask_ = ~0u;   // kernels that get events (bit = kernel index)
    hipStream_t stream_ = nullptr;
    Lease* lease_ = nullptr;
    int device_ = -1;                  // the device of the lease (set at upload): every entry below selects it for its duration
    // device buffers: all carved from ONE block of the lease (layout()); inputs first, in the order of the staging image
    UnitIn* d_units_ = nullptr; double* d_seg_cn_ = nullptr; double* d_junc_cn_ = nullptr; JuncEnds* d_junc_ends_ = nullptr; Element* d_elems_ = nullptr;
    Dag* d_dags_ = nullptr; uint8_t* d_results_ = nullptr;
    uint64_t* d_ikeys_ = nullptr; uint64_t* d_icnt_ = nullptr; uint32_t* d_ilink_ = nullptr; int32_t* d_ilvl_off_ = nullptr; int32_t* d_icounter_ = nullptr;
    int32_t* d_ipos_ = nullptr; uint64_t* d_aavail_ = nullptr; uint64_t* d_acnt_ = nullptr; int32_t* d_acbase_ = nullptr; uint16_t* d_achild_ = nullptr;
    int enum_stack_lds_ = 0, enum_auto_lds_ = 4096, enum_classes_ = 0, block_lds_ = 49152, lds_blocks_ = 0, block_scratch_lds_ = 32768, lds_build_ = 0, block_max_ = 256;
    int64_t* d_stage_clk_ = nullptr;
    int32_t* d_fallback_ = nullptr; uint8_t* d_blk_img_ = nullptr; int32_t* d_blk_hdr_ = nullptr;
    uint8_t* d_arena_ = nullptr; int64_t arena_bytes_ = 0;
    int64_t* d_blk_off_ = nullptr; int32_t* d_rows_ = nullptr; int32_t* d_npending_ = nullptr; int64_t* d_needed_ = nullptr;
    int32_t* d_scratch_ = nullptr; int64_t* d_scratch_off_ = nullptr; int64_t* d_pack_off_ = nullptr;
    int64_t* d_mail_off_ = nullptr; int32_t* d_guard_bad_ = nullptr;
    int64_t* d_run_slot_ = nullptr; int32_t* d_run_blk_[2] = {nullptr, nullptr}; int64_t run_total_ = 0; int runs_parity_ = -1;   // finish-stage runs: [cnt U][cells U][start total][len total] per run parity
    WideUnit* d_wide_ = nullptr; int32_t* d_wide_index_ = nullptr; int32_t* d_wide_units_ = nullptr; int n_wide_ = 0;   // units with 64..127 nodes (ambi_wide.hpp)
    int32_t* h_npending_ = nullptr; int64_t* h_needed_ = nullptr;   // pinned (words of the lease)
    int32_t* dh_npending_ = nullptr; int64_t* dh_needed_ = nullptr; // the same two, as the device addresses them
    int64_t in_bytes_ = 0, zero_off_ = 0, zero_bytes_ = 0;          // input image / zero-filled region of the block
    std::vector<uint8_t> stage_big_;                                // input image of a batch too large for the pinned staging block
    std::vector<int64_t> mail_off_; int64_t mail_bytes_ = 0; bool mail_on_ = false, mail_valid_ = false;
    BatchArgs A_{};
    int lds_prepare_ = 0, lds_first_ = 0, lds_finish_ = 0, lds_finish_lean_ = 0, lds_enum_ = 0;
    bool lean_finish_ = true;   // env AMBI_LEAN_FINISH=0: every unit through the full finish stage
    int lean_threads_ = 256;   // env AMBI_LEAN_THREADS (128 / 256): threads per workgroup of the lean finish kernel
    int lean_wave_ = 0, lean_wave_grid_ = 0;   // env AMBI_LEAN_WAVE=1: the lean stage on one wavefront per unit; AMBI_LEAN_WAVE_GRID: its wavefronts
    int finish_grid_ = 0;       // workgroups of the lean finish kernel; 0 = sized per run (env AMBI_FINISH_GRID overrides)
    int32_t* d_blocks_done_ = nullptr; int32_t* d_refin_list_ = nullptr; int32_t* d_refin_count_ = nullptr;
    uint32_t* d_anblk_ = nullptr; uint8_t* d_adepth_ = nullptr;
    int finish_path_cells_ = 0;
    std::vector<KernelTime> times_;
    typedef TimingEvents Ev;
    std::vector<Ev>& evs() { return lease_->evs; }
    int64_t last_needed_ = 0;
    int general_path_ = -1;   // units that take the general enumerate path: -1 unknown (first run), else the count (inputs are immutable)
    int shared_units_ = -1;   // units whose table is written by several workgroups: -1 unknown, else the count
    long timed_runs_ = 0;
    // slices: contiguous unit ranges whose kernel chains run on different streams (see BatchArgs)
    static constexpr int kMaxSlices = 16;
    int n_slices_ = 1;
    std::vector<int> slice_lo_;                        // [n_slices_+1]
    std::vector<int64_t> slice_base_, slice_bytes_;    // arena regions
    std::vector<hipStream_t> side_;                    // n_slices_-1 internal streams (slice 0 runs on the caller's); owned by the lease
    hipEvent_t ev_fork_ = nullptr;
    std::vector<hipEvent_t> ev_join_, ev_stage_;
    bool stagger_ = true;
    // overlap of [first valid order, finish] with the enumerate kernel (one slice, arena sized): own stream + two events (the lease's)
    bool overlap_back_ = false, want_overlap_ = true;
    hipStream_t back_stream_ = nullptr;
    hipEvent_t ev_prep_ = nullptr, ev_back_ = nullptr, ev_first_ = nullptr, ev_full_ = nullptr;
    hipStream_t full_stream_ = nullptr;
    bool first_launched_ = false; hipEvent_t ev_plan_ = nullptr;
    uint8_t* d_direct_cells_ = nullptr; int64_t direct_stride_ = 0; int direct_slots_ = 0; int lds_finish_ext_ = 0; bool direct_ext_ = true;   // env AMBI_DIRECT_EXT: the direct launch keeps its path cells in device memory
    int direct_cells_ = 0; bool direct_retry_ = false;   // path area of the direct full-finish launch (0: the batch's capacity bound); env AMBI_DIRECT_CELLS
    int full_threads_ = 1024;  // env AMBI_FULL_THREADS: threads per workgroup of the direct full-finish launch (256 / 512 / 1024)
    bool classed_ = true;              // env AMBI_STREAM_CLASSES=0: side streams by creation order (rounds 1-3) instead of by observed dispatch class
    hipStream_t classed_for_ = (hipStream_t)-1; hipStream_t classed_streams_[3] = {nullptr, nullptr, nullptr};
    bool want_back_ = false, want_full_ = false, want_first_ = false, want_lattice_ = false;
    hipStream_t first_stream_ = nullptr; int first_ahead_ = 3;   // env AMBI_FIRST_AHEAD: 1 the enumerate kernel waits for the scan, 2 the scan on a highest-priority stream beside it
    int32_t* d_direct_list_ = nullptr; int direct_n_ = 0, direct_grid_ = 1024;
    int32_t* d_edit_list_ = nullptr; bool direct_edit_ = true; int lds_finish_edit_ = 0, edit_grid_ = 1024;   // env AMBI_DIRECT_EDIT: these units through stage_finish_edit first (ambi_finish_edit_kernel), the ext launch behind it for what that hands on
    // express path (small batches): one kernel reconstructs every unit whose first order assembles; results are complete at ev_express_
    int express_threads_ = 512;   // env AMBI_EXPRESS_THREADS (256 / 512 / 1024): the serial stages use three wavefronts, the finish stage and the mailbox copy all of them
    int express_units_ = 32, lds_express_ = 0, lds_lattice_ = 0, lds_lattice_own_ = 0;
    bool lazy_ = false, tables_written_ = false;   // FLAG_LAZY_ORDERS: the run leaves the order tables out; they are written on demand
    bool side_lattice_ = false, flushed_ = false; hipStream_t lattice_stream_ = nullptr;   // env AMBI_SIDE_LATTICE=0: the lattice kernel behind the express kernel (round 2)
    uint64_t* d_lat_R_ = nullptr; int32_t* d_lat_status_ = nullptr; int64_t* d_lat_sum_ = nullptr;
    bool express_ = false;
    hipEvent_t ev_express_ = nullptr;
    int32_t* h_express_left_ = nullptr; int32_t* dh_express_left_ = nullptr;   // [0] left flag, [1] sequence word (words of the lease)
    int32_t run_seq_ = 0;
    uint8_t* d_first_rows_ = nullptr;
    int build_in_emit_ = 1;   // env AMBI_BUILD_IN_EMIT=0: every image through the build kernel and HBM
    int order_align_ = 4096;  // env AMBI_ORDER_ALIGN: every unit's table starts on a 4 KB boundary of the arena (a store stream whose 1 KB pieces are
                              // line-aligned: 5.4 -> 5.7 TB/s on a pure store stream of this shape, profiles/tools/hbm_write_pat5.hip)
    int emit_interleave_ = 1; // env AMBI_EMIT_INTERLEAVE=0: every wave a contiguous quarter of the work block instead of every fourth block
    int block_dfs_ = 1;       // env AMBI_BLOCK_DFS=0: no directory-free images (units whose directory does not fit take the general path)
    std::vector<std::vector<int64_t>> all_idx_[2];   // --all: valid order indices per pass and unit (filled when asked for)
    std::vector<std::vector<uint64_t>> all_cache_;   // --all: a unit's bitmap words once fetched
    std::vector<int64_t> all_off_, all_R_;
    std::vector<int32_t> all_counts_;
    uint64_t* d_all_bits_ = nullptr; int64_t* d_all_off_ = nullptr; int32_t* d_all_count_ = nullptr; int32_t* d_all_flags_ = nullptr;
    int8_t* d_inject_ = nullptr; int64_t* d_inject_off_ = nullptr;
    float all_kernel_ms_ = -1.f;
    int64_t all_bits_cap_ = 0, all_pool_bytes_ = 0;
    int all_rank_ = 0, all_world_ = 1;
    bool all_done_ = false;
    int enum_grid_ = 2048;
    bool emit_lds_auto_ = true;
    bool emit_lds_tight_ = true;   // env AMBI_EMIT_LDS_TIGHT=0: the whole 160 KB / k share
    double avg_path_ = 0;
    int enum_threads_ = 256;  // threads per workgroup of the block-emission kernel (env AMBI_ENUM_THREADS: 256 / 512 / 1024)
    std::chrono::steady_clock::time_point t_run_; double t_launched_ = 0;   // env AMBI_DEBUG_LATENCY
    bool debug_ = false;      // env AMBI_DEBUG: budgets and grids chosen; guard words of the direct path areas checked at wait()

    // Everything this batch queued is complete when this returns: the caller's stream (only if a run is still in flight -- the
    // stream must outlive its work) and every side stream of the lease, each synchronised EXPLICITLY (round 2 relied on
    // hipFree's implicit device synchronisation and destroyed streams it had never synchronised).
    void sync_all() {
        if (!lease_) return;
        if (inflight_) { (void)hipStreamSynchronize(stream_); inflight_ = false; }
        for (auto& kind : lease_->side) for (hipStream_t s : kind) if (s) (void)hipStreamSynchronize(s);
        for (hipStream_t s : lease_->slice_streams) (void)hipStreamSynchronize(s);
        if (lease_->copy_stream) (void)hipStreamSynchronize(lease_->copy_stream);
        if (lease_->run_stream) (void)hipStreamSynchronize(lease_->run_stream);
        for (hipStream_t s : classed_streams_) if (s) (void)hipStreamSynchronize(s);
        (void)hipGetLastError();
    }
    // guard words around the pinned words the kernels write through (always) and around the path areas of the direct
    // full-finish launch (AMBI_DEBUG); called with every stream idle
    int check_guards(const char* when) {
        if (!lease_) return 0;
        int bad = 0;
        for (int i = 0; i < 16; i++) bad += (lease_->h_words->guard_lo[i] != kGuardWord) + (lease_->h_words->guard_hi[i] != kGuardWord);
        if (bad) fprintf(stderr, "ambigram_hip: GUARD: %d guard words around the pinned status words overwritten (%s)\n", bad, when);
        if (debug_ && d_direct_cells_ && d_guard_bad_ && direct_slots_ > 0) {
            int32_t n = 0;
            hipLaunchKernelGGL(ambi_guard_check_kernel, dim3(direct_slots_), dim3(32), 0, nullptr, (const uint8_t*)d_direct_cells_, direct_stride_, direct_slots_, d_guard_bad_);
            if (hipMemcpy(&n, d_guard_bad_, sizeof(n), hipMemcpyDeviceToHost) == hipSuccess && n) {
                fprintf(stderr, "ambigram_hip: GUARD: %d guard words around the path areas of the direct full-finish launch overwritten (%s)\n", n, when);
                bad += n;
                (void)hipMemset(d_guard_bad_, 0, sizeof(int32_t));
            }
        }
        return bad;
    }
    void free_all() {
        DeviceGuard dg_(device_);
        sync_all();
        (void)check_guards("release");
        // per-batch allocations outside the lease's blocks (--all bitmaps, stage profile)
        void* ptrs[] = {d_all_bits_, d_all_off_, d_all_count_, d_stage_clk_};
        for (void* p : ptrs) if (p) (void)hipFree(p);
        d_all_bits_ = nullptr; d_all_off_ = nullptr; d_all_count_ = nullptr; d_all_flags_ = nullptr; d_stage_clk_ = nullptr; all_bits_cap_ = 0;
        d_units_ = nullptr; d_results_ = nullptr; d_arena_ = nullptr; d_direct_cells_ = nullptr; direct_n_ = 0;
        if (lease_) {
            Lease* L = lease_;
            // what a large batch grew goes back to the device; small blocks stay with the lease for the next batch
            if (L->d_block_bytes > kKeepBlock) { (void)hipFree(L->d_block); L->d_block = nullptr; L->d_block_bytes = 0; }
            if (L->d_arena_bytes > kKeepArena) { (void)hipFree(L->d_arena); L->d_arena = nullptr; L->d_arena_bytes = 0; }
            if (L->d_cells_bytes > kKeepCells) { (void)hipFree(L->d_cells); L->d_cells = nullptr; L->d_cells_bytes = 0; }
            if (L->h_stage_bytes > kKeepStage) { (void)hipHostFree(L->h_stage); L->h_stage = nullptr; L->dh_stage = nullptr; L->h_stage_bytes = 0; }
            if (L->h_mail_bytes > kKeepMail) { (void)hipHostFree(L->h_mail); L->h_mail = nullptr; L->dh_mail = nullptr; L->h_mail_bytes = 0; }
            for (int k = 0; k < 2; k++) {
                if (L->d_runs_bytes[k] > kKeepRuns) { (void)hipFree(L->d_runs[k]); L->d_runs[k] = nullptr; L->d_runs_bytes[k] = 0; }
                if (L->h_runs_bytes[k] > kKeepRuns) { (void)hipHostFree(L->h_runs[k]); L->h_runs[k] = nullptr; L->h_runs_bytes[k] = 0; }
            }
            lease_ = nullptr;
            DevicePool::get().release(L);
        }
        stage_big_.clear(); stage_big_.shrink_to_fit();
        mail_valid_ = false;
        device_ = -1;
    }

    // Layout of the device block: [inputs, in the order of the staging image] [zero-filled: result blob, image headers, flags,
    // counters] [working set].  pass 0 measures, pass 1 assigns the pointers.
    struct Carver {
        uint8_t* base; int64_t off = 0;
        template <class T> void take(T** p, size_t count) {
            if (base) *p = reinterpret_cast<T*>(base + off);
            off = (off + (int64_t)((count ? count : 1) * sizeof(T)) + 255) & ~int64_t(255);
        }
    };
    int64_t layout(uint8_t* base, const std::vector<int32_t>& direct_list) {
        const HostBatch& H = hb();
        const size_t U = H.units.size();
        Carver c{base};
        c.take(&d_units_, U); c.take(&d_seg_cn_, H.seg_cn.size()); c.take(&d_junc_cn_, H.junc_cn.size()); c.take(&d_junc_ends_, H.junc_ends.size());
        c.take(&d_elems_, H.elems.size()); c.take(&d_scratch_off_, U); c.take(&d_direct_list_, direct_list.size()); c.take(&d_mail_off_, U);
        c.take(&d_wide_index_, U); c.take(&d_wide_units_, (size_t)H.n_wide);
        c.take(&d_inject_, H.inject.size()); c.take(&d_inject_off_, H.inject.empty() ? 0 : 2 * U);
        c.take(&d_run_slot_, U + 1);
        in_bytes_ = c.off;
        zero_off_ = c.off;
        c.take(&d_results_, (size_t)H.result_bytes); c.take(&d_blk_hdr_, U * 8); c.take(&d_fallback_, U);
        c.take(&d_refin_count_, 2); c.take(&d_blocks_done_, 1); c.take(&d_npending_, 1); c.take(&d_guard_bad_, 1); c.take(&d_lat_sum_, 2);
        zero_bytes_ = c.off - zero_off_;
        c.take(&d_dags_, U);
        c.take(&d_ikeys_, (size_t)H.ideal_slots); c.take(&d_icnt_, (size_t)H.ideal_slots); c.take(&d_ilink_, (size_t)H.ideal_slots * 4 + 8);
        c.take(&d_ilvl_off_, U * (kMaxNodes + 3)); c.take(&d_icounter_, 2 * U); c.take(&d_ipos_, (size_t)H.ideal_slots);
        c.take(&d_aavail_, (size_t)H.ideal_slots / 2 + 1); c.take(&d_acnt_, (size_t)H.ideal_slots / 2 + 1); c.take(&d_acbase_, (size_t)H.ideal_slots / 2 + U + 1);
        c.take(&d_achild_, (size_t)H.ideal_slots * 4 + 8); c.take(&d_anblk_, (size_t)H.ideal_slots / 2 + 1); c.take(&d_adepth_, (size_t)H.ideal_slots / 2 + 8);
        c.take(&d_edit_list_, direct_list.size()); c.take(&d_blk_off_, U + kMaxSlices + 1); c.take(&d_rows_, U); c.take(&d_needed_, kMaxSlices); c.take(&d_lat_R_, U); c.take(&d_lat_status_, U);
        c.take(&d_scratch_, (size_t)H.scratch_ints + 8); c.take(&d_pack_off_, U + 1); c.take(&d_refin_list_, U);
        c.take(&d_first_rows_, U * (size_t)(cfg_.first_budget > 0 ? cfg_.first_budget : 1) * kFirstRowStride);
        const size_t img_stride = (size_t)std::max(block_lds_, ((160 * 1024) / 3) & ~15);   // (the budget per workgroup may be re-chosen after the first run)
        c.take(&d_blk_img_, U * img_stride);
        c.take(&d_wide_, (size_t)H.n_wide);
        run_total_ = H.run_slot.empty() ? 0 : H.run_slot.back();
        for (int k = 0; k < 2; k++) c.take(&d_run_blk_[k], 2 * U + 2 * (size_t)run_total_);
        return c.off;
    }

  public:
    ~HipBackend() override { free_all(); }
    const char* name() const override { return "hip"; }
    int device_count(int* n) override {
        int c = 0;
        hipError_t e = hipGetDeviceCount(&c);
        if (e != hipSuccess) c = 0;
        if (n) *n = c;
        return 0;
    }
    int set_device(int d) override { HIP_CK(hipSetDevice(d)); return 0; }

    template <class T> int dalloc(T** p, size_t count) {
        HIP_CK(hipMalloc((void**)p, (count ? count : 1) * sizeof(T)));
        return 0;
    }

    int upload(const HostBatch& hb_in, const EngineConfig& cfg) override {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return -30;   // AMBI_ERR_NO_DEVICE: no CPU fallback
        if (uploaded_) { free_all(); uploaded_ = false; }   // a second upload replaces the first (every stream idle first)
        ran_ = false; general_path_ = -1; timed_runs_ = 0; tuned_ = false; late_refusal_ = false; last_needed_ = 0;
        express_ = false;
        shared_units_ = -1;
        hbp_ = &hb_in; cfg_ = cfg;
        const HostBatch& H = hb();
        const size_t U = H.units.size();
        int rc;
        debug_ = ambi_env("AMBI_DEBUG") != nullptr;
        if ((rc = DevicePool::get().acquire(&lease_))) return rc;
        Lease* L = lease_;
        if (device_ != L->device) classed_for_ = (hipStream_t)-1;   // (another device: its own stream classes)
        device_ = L->device;
        h_npending_ = &L->h_words->npending; h_needed_ = L->h_words->needed;
        dh_npending_ = &L->dh_words->npending; dh_needed_ = L->dh_words->needed;
        h_express_left_ = &L->h_words->express_left; dh_express_left_ = &L->dh_words->express_left;
        L->h_words->npending = 0; L->h_words->express_left = 1; L->h_words->late_flag = 0;
        ev_fork_ = L->ev_fork; ev_prep_ = L->ev_prep; ev_back_ = L->ev_back; ev_first_ = L->ev_first; ev_full_ = L->ev_full; ev_plan_ = L->ev_plan; ev_express_ = L->ev_express;
        if (ambi_env("AMBI_STAGE_PROFILE")) { if ((rc = dalloc(&d_stage_clk_, U * kStageSlots))) return rc; HIP_CK(hipMemset(d_stage_clk_, 0, U * kStageSlots * sizeof(int64_t))); }
        { const char* e9 = ambi_env("AMBI_ORDER_ALIGN"); order_align_ = e9 ? atoi(e9) : 4096; if (order_align_ < 16 || (order_align_ & (order_align_ - 1))) order_align_ = 4096; }
        // LDS budgets (dynamic shared memory), sized for the largest unit of the batch
        lds_prepare_ = (int)prepare_work_bytes(H.max_n, H.max_m, H.max_k);
        n_wide_ = H.n_wide;
        lds_first_ = (int)first_work_bytes(H.max_n, H.max_bkp, n_wide_ > 0);
        // the full finish stage keeps the path cells in LDS: as many as fit beside its other arrays (longer paths are
        // served by the lean stage alone)
        finish_path_cells_ = H.max_path < kPathLdsCells ? H.max_path : kPathLdsCells;
        while (finish_path_cells_ > 4096 && finish_work_bytes(H.max_n, H.max_m, H.max_bkp, finish_path_cells_, H.max_out) > 160 * 1024 - 1024)
            finish_path_cells_ -= 2048;
        lds_finish_ = (int)finish_work_bytes(H.max_n, H.max_m, H.max_bkp, finish_path_cells_, H.max_out);
        lds_finish_lean_ = (int)finish_lean_work_bytes(H.max_n, H.max_m, H.max_bkp);
        { const char* e = ambi_env("AMBI_LEAN_FINISH"); lean_finish_ = e ? atoi(e) != 0 : true; }
        { const char* e = ambi_env("AMBI_FINISH_GRID"); finish_grid_ = e ? atoi(e) : 0; if (finish_grid_ < 0) finish_grid_ = 0; }
        { const char* e = ambi_env("AMBI_LEAN_WAVE"); lean_wave_ = e ? atoi(e) : 0; }
        // (measured, lean grid re-tuned for each: 128 threads 0.94-1.00, 256 threads 0.90 ms per step on one box)
        { const char* e = ambi_env("AMBI_LEAN_THREADS"); lean_threads_ = e ? atoi(e) : 256; if (lean_threads_ != 128) lean_threads_ = 256; }
        { const char* e = ambi_env("AMBI_LEAN_WAVE_GRID"); lean_wave_grid_ = e ? atoi(e) : 0; if (lean_wave_grid_ < 0) lean_wave_grid_ = 0; }
        {   // the general enumerate kernel serves ordinary units only (wide ones have their own table kernel): its per-lane stacks are
            // sized by the largest ORDINARY unit -- with the wide units' node count they outgrew a CU's group memory at 255 nodes
            int mk = 1;
            for (const UnitIn& un : H.units) if (un.n_elem <= kMaxNodes && un.n_elem > mk) mk = un.n_elem;
            enum_stack_lds_ = (int)enum_stack_bytes(mk);
        }
        lds_enum_ = 4 * (enum_stack_lds_ + enum_auto_lds_);
        { const char* env = ambi_env("AMBI_BLOCK_LDS"); block_lds_ = env ? atoi(env) : cfg.block_lds; if (block_lds_ < 64) block_lds_ = 64; block_lds_ = (block_lds_ + 15) & ~15; }
        { const char* env = ambi_env("AMBI_BLOCK_MAX"); block_max_ = env ? atoi(env) : cfg.block_max; if (block_max_ < 1) block_max_ = 1; if (block_max_ > kBlockMaxLimit) block_max_ = kBlockMaxLimit; }
        lds_blocks_ = block_lds_;
        const int kLdsLimit = 160 * 1024 - 1024;
        emit_lds_auto_ = !ambi_env("AMBI_BLOCK_LDS");
        { const char* e = ambi_env("AMBI_EMIT_LDS_TIGHT"); emit_lds_tight_ = e ? atoi(e) != 0 : true; }
        { const char* env = ambi_env("AMBI_BLOCK_SCRATCH_LDS"); block_scratch_lds_ = ((env ? atoi(env) : cfg.block_scratch_lds) + 15) & ~15; }
        if (block_scratch_lds_ > kLdsLimit) block_scratch_lds_ = kLdsLimit & ~15;
        lds_build_ = block_scratch_lds_;   // the image itself is assembled in HBM
        if (lds_prepare_ > kLdsLimit || lds_first_ > kLdsLimit || lds_finish_ > kLdsLimit || lds_enum_ > kLdsLimit || lds_blocks_ > kLdsLimit) {
            fprintf(stderr, "ambigram_hip: a unit needs more LDS than one CU has (prepare %d, first %d, finish %d, general enumerate %d, block emission %d bytes)\n",
                    lds_prepare_, lds_first_, lds_finish_, lds_enum_, lds_blocks_);
            return ST_ERR_BAD_INPUT;
        }
        {   // the dynamic-LDS ceiling of every kernel: a per-function, process-wide attribute, set once
            static std::mutex mu; static uint64_t done = 0; hipError_t err = hipSuccess;   // (per device: one bit each)
            std::lock_guard<std::mutex> lk(mu);
            if (!((done >> (L->device & 63)) & 1ull)) {
                const void* fns[] = {(const void*)ambi_blocks_build_kernel, (const void*)ambi_prepare_kernel, (const void*)ambi_first_kernel, (const void*)ambi_resolve_kernel,
                                     (const void*)ambi_finish_kernel, (const void*)ambi_finish_ext_kernel, (const void*)ambi_finish_edit_kernel, (const void*)ambi_finish_lean_kernel, (const void*)ambi_finish_lean_wave_kernel,
                                     (const void*)ambi_enumerate_kernel<0>, (const void*)ambi_enumerate_kernel<1>, (const void*)ambi_enumerate_kernel<2>,
                                     (const void*)ambi_enumerate_blocks_kernel<0>, (const void*)ambi_enumerate_blocks_kernel<1>, (const void*)ambi_enumerate_blocks_kernel<2>,
                                     (const void*)ambi_express_kernel, (const void*)ambi_lattice_kernel, (const void*)ambi_lattice_own_kernel, (const void*)ambi_search_kernel, (const void*)ambi_all_kernel,
                                     (const void*)ambi_all_lanes_kernel, (const void*)ambi_order_paths_kernel};
                for (const void* f : fns) { hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsMaxDynamic); if (e != hipSuccess) err = e; }
                if (err == hipSuccess) done |= 1ull << (L->device & 63);   // (a failed attempt is tried again by the next upload)
            }
            HIP_CK(err);
        }
        enum_classes_ = 0;
        for (const UnitIn& un : H.units) if (un.n_elem > 0) enum_classes_ |= 1 << enum_class_of(un.n_elem);
        avg_path_ = 0;
        for (const UnitIn& un : H.units) avg_path_ += un.path_cap;
        avg_path_ /= (double)(U > 0 ? U : 1);
        // slices: env AMBI_SLICES or the configuration; 0 = automatic (4 when the batch is large enough to fill the
        // chip four times over, else 1)
        {
            const char* env = ambi_env("AMBI_SLICES");
            int want = env ? atoi(env) : cfg.slices;
            // Measured on MI355X (profiles/r01_slices.md): the per-unit kernels already fill the issue slots of the chip,
            // so slicing brings nothing on this workload (2 slices: +3 %, 4 staggered slices: -17 %); default 1.
            if (want <= 0) want = 1;
            if (want > kMaxSlices) want = kMaxSlices;
            if (want > (int)U) want = (int)U > 0 ? (int)U : 1;
            n_slices_ = want;
            slice_lo_.assign(n_slices_ + 1, 0);
            for (int s = 0; s <= n_slices_; s++) slice_lo_[s] = (int)((int64_t)U * s / n_slices_);
            slice_base_.assign(n_slices_, 0); slice_bytes_.assign(n_slices_, 0);
            // (streams and events of the slices belong to the lease: created once, reused by every later batch)
            while ((int)L->slice_streams.size() < n_slices_ - 1) { hipStream_t st; HIP_CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); L->slice_streams.push_back(st); }
            while ((int)L->slice_events.size() < 2 * (n_slices_ - 1)) { hipEvent_t e; HIP_CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); L->slice_events.push_back(e); }
            side_.assign(L->slice_streams.begin(), L->slice_streams.begin() + (n_slices_ - 1));
            ev_join_.assign(L->slice_events.begin(), L->slice_events.begin() + (n_slices_ - 1));
            ev_stage_.assign(L->slice_events.begin() + (n_slices_ - 1), L->slice_events.begin() + 2 * (n_slices_ - 1));
            { const char* e2 = ambi_env("AMBI_STAGGER"); stagger_ = e2 ? atoi(e2) != 0 : true; }
            { const char* e4 = ambi_env("AMBI_ENUM_THREADS"); enum_threads_ = e4 ? atoi(e4) : 256; if (enum_threads_ != 512 && enum_threads_ != 1024) enum_threads_ = 256; }
            { const char* e3 = ambi_env("AMBI_ENUM_GRID"); enum_grid_ = e3 ? atoi(e3) : 16384; if (enum_grid_ < 1) enum_grid_ = 1; }   // >= work blocks: one block per workgroup, the rest exit (measured: 2048 -> 16384 workgroups = -8 % kernel time)
        }
        { const char* e9 = ambi_env("AMBI_BUILD_IN_EMIT"); build_in_emit_ = e9 ? (atoi(e9) != 0) : 1; }
        { const char* e9 = ambi_env("AMBI_BLOCK_DFS"); block_dfs_ = e9 ? (atoi(e9) != 0) : 1; }
        { const char* e9 = ambi_env("AMBI_EMIT_INTERLEAVE"); emit_interleave_ = e9 ? (atoi(e9) != 0) : 1; }
        { const char* e5 = ambi_env("AMBI_OVERLAP_BACK"); want_overlap_ = e5 ? atoi(e5) != 0 : true; }
        back_stream_ = nullptr; full_stream_ = nullptr; first_stream_ = nullptr; direct_n_ = 0; d_direct_cells_ = nullptr; direct_slots_ = 0;
        want_back_ = want_full_ = want_first_ = want_lattice_ = false;
        { const char* e = ambi_env("AMBI_STREAM_CLASSES"); classed_ = e ? atoi(e) != 0 : true; }
        std::vector<int32_t> dl;
        if (want_overlap_ && n_slices_ == 1) {
            // the stream of the lean finish kernel: default dispatch priority (AMBI_BACK_PRIORITY=1: lowest, round 1's setting
            // -- with the scan out of the way early the finish kernels have the whole enumerate kernel to hide behind, and
            // holding them back only lengthens the tail after it: 1.185 -> 1.168 ms per step, four interleaved runs)
            { const char* e8 = ambi_env("AMBI_BACK_PRIORITY"); const bool low = e8 ? atoi(e8) != 0 : false;
              if ((rc = lease_stream(L, 0, low ? 1 : 0, &back_stream_))) return rc; want_back_ = !low; }
            { const char* e = ambi_env("AMBI_FIRST_AHEAD"); first_ahead_ = e ? atoi(e) : 3; }
            // (direct full-stage launch: 512 threads with the path cells in device memory, 1024 with the cells in group memory --
            // measured, four interleaved runs: cells in group memory 1.151 ms per step; in device memory 256 / 512 / 1024
            // threads = 1.133 / 1.110 / 1.200)
            { const char* ee = ambi_env("AMBI_DIRECT_EXT"); direct_ext_ = ee ? atoi(ee) != 0 : true; }
            { const char* e = ambi_env("AMBI_FULL_THREADS"); full_threads_ = e ? atoi(e) : (direct_ext_ ? 512 : 1024); if (full_threads_ != 256 && full_threads_ != 512 && full_threads_ != 1024) full_threads_ = direct_ext_ ? 512 : 1024; }
            if (first_ahead_ >= 2) { if ((rc = lease_stream(L, 2, 2, &first_stream_))) return rc; want_first_ = true; }
            {   // units that go straight to the full finish stage (env AMBI_DIRECT_FULL=0: none, they pass through the lean stage first)
                const char* e7 = ambi_env("AMBI_DIRECT_FULL"); const bool on = e7 ? atoi(e7) != 0 : true;
                const char* e8 = ambi_env("AMBI_DIRECT_GRID"); direct_grid_ = e8 ? atoi(e8) : 1024; if (direct_grid_ < 1) direct_grid_ = 1;   // one workgroup per unit up to 1024 (measured: 64 / 128 / 256 / 512 workgroups for 512 units = 1.63 / 1.37 / 1.25 / 1.23 ms per step; without this launch 1.30)
                if (on && lean_finish_) for (size_t u2 = 0; u2 < U; u2++) if (H.units[u2].direct_full) dl.push_back((int32_t)u2);
                direct_n_ = (int)dl.size();
                if (direct_n_ > 0) {
                    // the direct full-finish stream: default priority, or the lowest (AMBI_FULL_PRIORITY=1: a few per cent on some
                    // boxes).  The stream belongs to the lease and is never destroyed -- with round 2's per-batch create / destroy of
                    // this priority stream a long soak showed stray writes into host memory.
                    const char* e9 = ambi_env("AMBI_FULL_PRIORITY"); const int fp = e9 ? atoi(e9) : 0;   // 0 default, 1 lowest, 2 highest priority
                    if ((rc = lease_stream(L, 1, fp == 1 ? 1 : (fp == 2 ? 2 : 0), &full_stream_))) return rc;
                    want_full_ = fp == 0;
                    if (direct_ext_) lds_finish_ext_ = (int)finish_work_bytes(H.max_n, H.max_m, H.max_bkp, 0, H.max_out);
                    { const char* ed = ambi_env("AMBI_DIRECT_EDIT"); direct_edit_ = (ed ? atoi(ed) != 0 : true) && direct_ext_; }
                    { const char* eg = ambi_env("AMBI_EDIT_GRID"); edit_grid_ = eg ? atoi(eg) : 1024; if (edit_grid_ < 1) edit_grid_ = 1; }
                    lds_finish_edit_ = (int)finish_edit_work_bytes(H.max_n, H.max_m, H.max_bkp);
                    if (lds_finish_edit_ > kLdsLimit) direct_edit_ = false;
                }
            }
            // AMBI_ENUM_LDS_FLOOR (experiments): make the enumerate kernel ask for more LDS than its image needs, i.e. fewer
            // of its workgroups per CU.  Measured (profiles/r01_slices.md): no floor is best -- the scan / finish
            // workgroups slip in as enumerate workgroups retire.
            { const char* e6 = ambi_env("AMBI_ENUM_LDS_FLOOR"); const int floor_lds = e6 ? atoi(e6) : 0; if (lds_blocks_ < floor_lds && floor_lds <= kLdsLimit) lds_blocks_ = floor_lds; }
        }
        {   // express path: small batches only, and only if a unit's whole working set fits one workgroup's group memory
            const char* e9 = ambi_env("AMBI_EXPRESS_UNITS"); express_units_ = e9 ? atoi(e9) : 32;
            { const char* et = ambi_env("AMBI_EXPRESS_THREADS"); express_threads_ = et ? atoi(et) : 512; if (express_threads_ != 256 && express_threads_ != 1024) express_threads_ = 512; }   // (measured, one 256-segment sample: 256 / 512 / 1024 threads = 79.7 / 76.8 / 79.2 us run -> results)
            lds_express_ = (int)express_work_bytes(H.max_n, H.max_m, H.max_k, H.max_bkp, finish_path_cells_, H.max_out) + 64;
            lds_lattice_ = (int)(64 * 8 + kPrepLatticeBytes + 64);
            lds_lattice_own_ = (int)lattice_own_bytes(H.max_k) + 64;
            const char* e8 = ambi_env("AMBI_SIDE_LATTICE");
            side_lattice_ = (e8 ? atoi(e8) != 0 : true) && (int)U <= express_units_ && n_slices_ == 1;
            if (side_lattice_ && (rc = lease_stream(L, 2, 0, &lattice_stream_))) return rc;
            want_lattice_ = side_lattice_;
        }
        // result mailbox in pinned host memory: batches that can take the express path, while the slots stay small
        mail_off_.assign(U, 0); mail_bytes_ = 0; mail_on_ = false; mail_valid_ = false;
        if ((int)U <= express_units_ && n_slices_ == 1) {
            for (size_t u2 = 0; u2 < U; u2++) { mail_off_[u2] = mail_bytes_; mail_bytes_ += mail_layout(H.units[u2].path_cap, H.units[u2].out_cap).total; }
            mail_on_ = mail_bytes_ <= kKeepMail;
            if (mail_on_ && (rc = lease_pinned_block(&L->h_mail, &L->dh_mail, &L->h_mail_bytes, mail_bytes_))) return rc;
        }
        // ---- the device block and the input image ----
        const int64_t total = layout(nullptr, dl);
        if ((rc = lease_device_block(&L->d_block, &L->d_block_bytes, total))) return rc;
        (void)layout(L->d_block, dl);
        if (H.inject.empty()) { d_inject_ = nullptr; d_inject_off_ = nullptr; }
        if (dl.empty()) d_direct_list_ = nullptr;
        if (direct_n_ > 0 && direct_ext_) {   // one path area per workgroup of the direct launch, guard words on both sides of each
            direct_stride_ = ((2ll * H.max_path + 16 + 15) & ~int64_t(15)) + 2 * kCellGuardBytes;
            direct_slots_ = direct_n_ < direct_grid_ ? direct_n_ : direct_grid_;
            if ((rc = lease_device_block(&L->d_cells, &L->d_cells_bytes, direct_stride_ * direct_slots_))) return rc;
            d_direct_cells_ = L->d_cells;
        }
        arena_bytes_ = cfg.order_arena_bytes > 0 ? cfg.order_arena_bytes : (int64_t)1 << 20;
        { const char* cap = ambi_env("AMBI_ARENA_MAX_BYTES"); const bool capped = cap && atoll(cap) > 0;
          // a configured size or a memory budget is taken literally; otherwise whatever the lease already holds is used
          if (cfg.order_arena_bytes <= 0 && !capped && L->d_arena_bytes > arena_bytes_) arena_bytes_ = L->d_arena_bytes;
          if (capped && arena_bytes_ > atoll(cap)) arena_bytes_ = atoll(cap); }
        if ((rc = lease_device_block(&L->d_arena, &L->d_arena_bytes, arena_bytes_))) return rc;
        d_arena_ = L->d_arena;
        for (int s = 0; s < n_slices_; s++) { slice_base_[s] = (arena_bytes_ / n_slices_ * s) & ~int64_t(order_align_ - 1); slice_bytes_[s] = (arena_bytes_ / n_slices_) & ~int64_t(order_align_ - 1); }
        // input image: same layout as the front of the device block; pinned staging (one asynchronous copy, queued by the
        // first run ahead of its kernels) or, for large batches, a plain buffer copied at once
        uint8_t* img;
        const bool pinned = in_bytes_ <= kKeepStage;
        if (pinned) { if ((rc = lease_pinned_block(&L->h_stage, &L->dh_stage, &L->h_stage_bytes, in_bytes_))) return rc; img = L->h_stage; }
        else { stage_big_.assign((size_t)in_bytes_, 0); img = stage_big_.data(); }
        auto put = [&](const void* dptr, const void* src, size_t bytes) { if (bytes) memcpy(img + (reinterpret_cast<const uint8_t*>(dptr) - L->d_block), src, bytes); };
        put(d_units_, H.units.data(), U * sizeof(UnitIn));
        put(d_seg_cn_, H.seg_cn.data(), H.seg_cn.size() * sizeof(double));
        put(d_junc_cn_, H.junc_cn.data(), H.junc_cn.size() * sizeof(double));
        put(d_junc_ends_, H.junc_ends.data(), H.junc_ends.size() * sizeof(JuncEnds));
        put(d_elems_, H.elems.data(), H.elems.size() * sizeof(Element));
        put(d_scratch_off_, H.scratch_off.data(), U * sizeof(int64_t));
        if (!dl.empty()) put(d_direct_list_, dl.data(), dl.size() * sizeof(int32_t));
        put(d_mail_off_, mail_off_.data(), U * sizeof(int64_t));
        put(d_run_slot_, H.run_slot.data(), (U + 1) * sizeof(int64_t));
        put(d_wide_index_, H.wide_index.data(), U * sizeof(int32_t));
        {
            std::vector<int32_t> wu;
            for (size_t u2 = 0; u2 < U; u2++) if (H.wide_index[u2] >= 0) wu.push_back((int32_t)u2);
            if (!wu.empty()) put(d_wide_units_, wu.data(), wu.size() * sizeof(int32_t));
        }
        if (!H.inject.empty()) { put(d_inject_, H.inject.data(), H.inject.size()); put(d_inject_off_, H.inject_off.data(), 2 * U * sizeof(int64_t)); }
        if (pinned) upload_pending_ = true;
        else {
            HIP_CK(hipMemcpy(L->d_block, img, (size_t)in_bytes_, hipMemcpyHostToDevice));
            HIP_CK(hipMemset(L->d_block + zero_off_, 0, (size_t)zero_bytes_));
            if (d_direct_cells_) { hipLaunchKernelGGL(ambi_guard_fill_kernel, dim3(direct_slots_), dim3(32), 0, nullptr, d_direct_cells_, direct_stride_, direct_slots_); HIP_CK(hipDeviceSynchronize()); }
            stage_big_.clear(); stage_big_.shrink_to_fit();
            upload_pending_ = false;
        }
        uploaded_ = true; arena_checked_ = false;
        return 0;
    }
    // the input image and the zero fill of a freshly uploaded batch, queued on the caller's stream ahead of the first kernels
    int flush_upload(hipStream_t st) {
        if (!upload_pending_) return 0;
        Lease* L = lease_;
        if (in_bytes_ + zero_bytes_ <= (8ll << 20) && L->dh_stage && !ambi_env("AMBI_NO_INGEST")) {   // (every part of the block is 256-byte aligned)
            const int64_t n16 = in_bytes_ / 16, z16 = zero_bytes_ / 16;
            int64_t blocks = (std::max(n16, z16) + 255) / 256;
            if (blocks > 1024) blocks = 1024;
            hipLaunchKernelGGL(ambi_ingest_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const uint4*)L->dh_stage, (uint4*)L->d_block, n16,
                               (uint4*)(L->d_block + zero_off_), z16);
        } else {
            HIP_CK(hipMemcpyAsync(L->d_block, L->h_stage, (size_t)in_bytes_, hipMemcpyHostToDevice, st));
            HIP_CK(hipMemsetAsync(L->d_block + zero_off_, 0, (size_t)zero_bytes_, st));
        }
        if (d_direct_cells_) hipLaunchKernelGGL(ambi_guard_fill_kernel, dim3(direct_slots_), dim3(32), 0, st, d_direct_cells_, direct_stride_, direct_slots_);
        upload_pending_ = false;
        return 0;
    }

    // whole-batch argument block (pack kernels, slow path); slice_args() narrows it to one slice
    void bind(uint32_t flags) {
        A_.n_units = (int32_t)hb().units.size(); A_.unit_base = 0; A_.arena_base = 0;
        A_.flags = flags; A_.first_budget = cfg_.first_budget; A_.target_lanes = cfg_.target_lanes;
        A_.enum_stack_lds = enum_stack_lds_; A_.enum_auto_lds = enum_auto_lds_;
        A_.block_lds = block_lds_; A_.block_scratch_lds = block_scratch_lds_; A_.block_max = block_max_; A_.build_in_emit = build_in_emit_; A_.emit_interleave = emit_interleave_; A_.order_align = order_align_; A_.block_dfs = block_dfs_; A_.finish_path_cells = finish_path_cells_; A_.unit_fallback = d_fallback_; A_.block_img = d_blk_img_; A_.block_hdr = d_blk_hdr_;
        A_.ideal_pos = d_ipos_; A_.auto_avail = d_aavail_; A_.auto_cnt = d_acnt_; A_.auto_cbase = d_acbase_; A_.auto_child = d_achild_; A_.auto_nblk = d_anblk_; A_.auto_depth = d_adepth_;
        A_.units = d_units_; A_.seg_cn = d_seg_cn_; A_.junc_cn = d_junc_cn_; A_.junc_ends = d_junc_ends_; A_.elems = d_elems_;
        A_.dags = d_dags_; A_.results = d_results_;
        A_.ideal_keys = d_ikeys_; A_.ideal_cnt = d_icnt_; A_.ideal_link = d_ilink_; A_.ideal_lvl_off = d_ilvl_off_; A_.ideal_counter = d_icounter_;
        A_.first_rows = d_first_rows_;
        A_.order_arena = d_arena_; A_.order_arena_bytes = arena_bytes_;
        A_.blk_off = d_blk_off_; A_.rows_per_lane = d_rows_; A_.n_pending = d_npending_; A_.orders_needed = d_needed_;
        A_.scratch_i32 = d_scratch_; A_.scratch_off = d_scratch_off_; A_.stage_clk = d_stage_clk_;
        A_.zero_pending = 0; A_.host_pending = nullptr; A_.host_needed = nullptr; A_.blocks_done = d_blocks_done_;
        A_.inject_valid = d_inject_; A_.inject_off = d_inject_off_; A_.refin_list = d_refin_list_; A_.refin_count = d_refin_count_; A_.direct_full_on = 0; A_.finish_retry = 0;
        { const char* ec = ambi_env("AMBI_EDIT_RUN_CAP"); A_.edit_cap_limit = ec ? atoi(ec) : 0; }
        A_.wide = n_wide_ > 0 ? d_wide_ : nullptr; A_.wide_index = n_wide_ > 0 ? d_wide_index_ : nullptr;
        A_.mail = mail_on_ ? lease_->dh_mail : nullptr; A_.mail_off = d_mail_off_;
        {   // the finish stages leave the final paths in run-length form in the block of this run's parity (the other one may still be on its way to the host)
            const int64_t U = (int64_t)hb().units.size();
            int32_t* blk = d_run_blk_[run_seq_ & 1];
            A_.run_cnt = blk; A_.run_cells = blk + U; A_.run_start = blk + 2 * U; A_.run_len = blk + 2 * U + run_total_; A_.run_slot = d_run_slot_;
        }
        A_.lat_R = d_lat_R_; A_.lat_status = d_lat_status_; A_.lat_sum = d_lat_sum_; A_.lat_seq = &lease_->dh_words->lat_seq; A_.lat_unsure = &lease_->dh_words->lat_unsure;
        A_.plan_seq = &lease_->dh_words->plan_seq; A_.late_flag = &lease_->dh_words->late_flag; A_.run_seq = run_seq_;
        A_.all_bits = d_all_bits_; A_.all_off = d_all_off_; A_.all_count = d_all_count_; A_.all_flags = d_all_flags_; A_.all_rank = all_rank_; A_.all_world = all_world_; { const char* e = ambi_env("AMBI_ALL_TABLE"); A_.all_rows_from_table = (e && atoi(e) != 0) ? 1 : 0; }
    }
    BatchArgs slice_args(int s) const {
        BatchArgs A = A_;
        A.unit_base = slice_lo_[s]; A.n_units = slice_lo_[s + 1] - slice_lo_[s];
        A.arena_base = slice_base_[s]; A.order_arena_bytes = slice_bytes_[s];
        A.blk_off = d_blk_off_ + slice_lo_[s] + s;        // n_units + 1 private entries
        A.orders_needed = d_needed_ + s;
        // the rows of the WHOLE batch are spread over target_lanes lanes: a slice gets its share
        A.target_lanes = cfg_.target_lanes / n_slices_ > 0 ? cfg_.target_lanes / n_slices_ : 1;
        return A;
    }
    hipStream_t slice_stream(int s) const { return s == 0 ? stream_ : side_[s - 1]; }

    // Per-kernel HIP events on the stream of the slice.  A ring of kTimingSlots event sets lets a timed region of many
    // runs be averaged without a host sync per run: run r records into slot r % kTimingSlots.
    static constexpr int kTimingSlots = 64, kTimedKernels = 7;
    void tick(const char* name, int slice, size_t idx, bool begin, hipStream_t on = (hipStream_t)-1) {
        if (!timing_ || !((timing_mask_ >> idx) & 1u)) return;
        const size_t slot = (size_t)(timed_runs_ % kTimingSlots);
        const size_t at = (slot * n_slices_ + slice) * kTimedKernels + idx;
        while (evs().size() <= at) {
            Ev e{name, nullptr, nullptr};
            (void)hipEventCreate(&e.a); (void)hipEventCreate(&e.b);
            evs().push_back(e);
        }
        evs()[at].name = name;
        (void)hipEventRecord(begin ? evs()[at].a : evs()[at].b, on == (hipStream_t)-1 ? slice_stream(slice) : on);
    }

    void fork() {   // the side streams start behind everything already queued on the caller's stream
        if (n_slices_ <= 1) return;
        (void)hipEventRecord(ev_fork_, stream_);
        for (auto st : side_) (void)hipStreamWaitEvent(st, ev_fork_, 0);
    }
    void join() {   // the caller's stream continues after all side streams
        for (int s = 1; s < n_slices_; s++) {
            (void)hipEventRecord(ev_join_[s - 1], side_[s - 1]);
            (void)hipStreamWaitEvent(stream_, ev_join_[s - 1], 0);
        }
    }

    void launch_front(int s, const BatchArgs& A) {   // prepare + plan of one slice
        hipStream_t st = slice_stream(s);
        first_launched_ = false;
        if (express_) {
            // express kernel (whole reconstruction of the units whose first order assembles; results complete at ev_express_),
            // then the lattice stage of every unit; the express kernel hands units over by status, not through the list
            BatchArgs Ax = A;
            Ax.refin_list = nullptr;
            Ax.express_left = dh_express_left_;
            Ax.express_seq = dh_express_left_ + 1;
            Ax.run_seq = run_seq_;
            h_express_left_[0] = 0;
            // the lattice kernel starts beside the express kernel: behind the input image of a fresh batch, else behind the END of the
            // previous run (an event recorded then: no marker in front of the express kernel now)
            const bool side = side_lattice_ && !lazy_;
            if (side) {
                lease_->h_words->lat_unsure = 0;
                if (flushed_) (void)hipEventRecord(ev_fork_, st);
                (void)hipStreamWaitEvent(lattice_stream_, flushed_ ? ev_fork_ : lease_->ev_tail, 0);
            }
            tick("ambi_express_kernel", s, 0, true);
            hipLaunchKernelGGL(ambi_express_kernel, dim3(A.n_units), dim3(express_threads_), lds_express_, st, Ax);
            tick("ambi_express_kernel", s, 0, false);
            (void)hipEventRecord(ev_express_, st);
            tick("ambi_plan_kernel", s, 1, true);
            if (side) {   // the lattice beside the express kernel, on a stream of its own; the plan kernel behind both
                hipLaunchKernelGGL(ambi_lattice_own_kernel, dim3(A.n_units), dim3(64), lds_lattice_own_, lattice_stream_, A);
                (void)hipEventRecord(lease_->ev_lat, lattice_stream_);
                (void)hipStreamWaitEvent(st, lease_->ev_lat, 0);
            } else
            hipLaunchKernelGGL(ambi_lattice_kernel, dim3(A.n_units), dim3(64), lds_lattice_, st, A);
        } else {
            tick("ambi_prepare_kernel", s, 0, true);
            hipLaunchKernelGGL(ambi_prepare_kernel, dim3(A.n_units), dim3(64), lds_prepare_, st, A);
            tick("ambi_prepare_kernel", s, 0, false);
            // The scan for the first valid order reads what the prepare stage left (first rows, DAG) and rewrites the status;
            // the plan stage reads and writes UnitOut::order_off only.  With first_ahead_ == 3 the scan therefore starts HERE,
            // beside the plan kernel (one workgroup, 25 us during which the chip is otherwise idle) and the ramp of the
            // enumerate kernel, instead of queueing behind 4096 enumerate workgroups for group memory.
            first_launched_ = false;
            static const bool one_event = [] { const char* e = ambi_env("AMBI_ONE_FRONT_EVENT"); return e && atoi(e) != 0; }();   // experiment: see below
            if (overlap_back_ && first_ahead_ == 3 && !(one_event && !lazy_)) {
                (void)hipEventRecord(ev_prep_, st);
                hipStream_t sf = first_stream_ ? first_stream_ : back_stream_;
                (void)hipStreamWaitEvent(sf, ev_prep_, 0);
                tick("ambi_first_kernel", s, 4, true, sf);
                hipLaunchKernelGGL(ambi_first_kernel, dim3(A.n_units), dim3(64), lds_first_, sf, A);
                tick("ambi_first_kernel", s, 4, false, sf);
                (void)hipEventRecord(ev_first_, sf);
                first_launched_ = true;
            }
            tick("ambi_plan_kernel", s, 1, true);
        }
        if (lazy_) {}   // no tables in this run: nothing to plan
        else if (express_ && side_lattice_) hipLaunchKernelGGL(ambi_plan_kernel, dim3(1), dim3(1024), 0, st, A);
        else { BatchArgs Ap = A; Ap.lat_R = nullptr; hipLaunchKernelGGL(ambi_plan_kernel, dim3(1), dim3(1024), 0, st, Ap); }
        tick("ambi_plan_kernel", s, 1, false);
        // (express chain: the lattice kernel reads the status, so the scan of the units the express kernel left stays behind
        // the plan kernel there)
        {
            static const bool one_event = [] { const char* e = ambi_env("AMBI_ONE_FRONT_EVENT"); return e && atoi(e) != 0; }();
            if (one_event && !express_ && !lazy_ && overlap_back_ && first_ahead_ == 3) {
                // experiment: ONE event on the caller's stream between prepare and the order-table kernel (behind the plan kernel) instead
                // of one on either side of the plan kernel; the scan then starts behind the plan kernel
                (void)hipEventRecord(ev_plan_, st);
                hipStream_t sf = first_stream_ ? first_stream_ : back_stream_;
                (void)hipStreamWaitEvent(sf, ev_plan_, 0);
                tick("ambi_first_kernel", s, 4, true, sf);
                hipLaunchKernelGGL(ambi_first_kernel, dim3(A.n_units), dim3(64), lds_first_, sf, A);
                tick("ambi_first_kernel", s, 4, false, sf);
                (void)hipEventRecord(ev_first_, sf);
                first_launched_ = true;
                return;
            }
        }
        if (overlap_back_) (void)hipEventRecord(first_launched_ ? ev_plan_ : ev_prep_, st);
    }
    void launch_build(int s, const BatchArgs& A) {   // block-emission images
        hipStream_t st = slice_stream(s);
        tick("ambi_blocks_build_kernel", s, 2, true);
        hipLaunchKernelGGL(ambi_blocks_build_kernel, dim3(A.n_units), dim3(256), lds_build_, st, A);
        tick("ambi_blocks_build_kernel", s, 2, false);
    }
    // Workgroups of the lean finish kernel.  Beside a long enumerate kernel the finish stage only has to be done when the
    // table is: the fewer of its workgroups are resident, the less they take from the enumerate workgroups (issue slots,
    // group-memory bandwidth), so the grid is sized to finish in time with a margin: too few workgroups and the finish
    // kernel becomes the tail of the step (a steep loss), too many cost a little (measured on one MI355X,
    // profiles/r01_slices.md: 112 / 128 / 144 / 160 / 176 / 256 / 4096 workgroups for the 4096 units of the bench workload
    // = 1.25 / 1.17 / 1.125 / 1.126 / 1.13 / 1.14 / 1.39 ms per step).  Without a long enumerate kernel (small order
    // tables, first run) every unit gets its own workgroup.
    int finish_grid_for(int U) const {
        if (finish_grid_ > 0) return U < finish_grid_ ? U : finish_grid_;          // env AMBI_FINISH_GRID
        if (!overlap_back_ || lazy_) return U;   // (no order table being written: nothing to hide behind, every unit its own workgroup)
        // order-table bytes of the previous run at the rate the enumerate kernel reaches: an optimistic 5.2 TB/s for rows of a byte
        // per node; rows of 5 bits per node (up to 32 nodes) leave at ~3.5 TB/s -- fewer bytes per row for the same work per row
        // (measured on the bench batch, 12-byte rows: 160 / 176 / 192 / 208 / 240 / 272 workgroups = 0.92-0.93 / 0.916-0.938 /
        // 0.903-0.919 / 0.914-0.933 / 0.93-0.94 / 0.95 ms per step with the 16-byte-group emission; with one row per lane the
        // table is done earlier and the finish kernels want more room: 192 / 208 / 224 / 256 / 288 = 0.887 / 0.864 / 0.857 / 0.876 /
        // 0.885 (means of 3-6 runs, the build before: 0.867); this rule gives 210 -> 224)
        const double enum_us = (double)last_needed_ / ((enum_classes_ & 3) ? 4.1e6 : 5.2e6);
        const double unit_us = 4.0 + avg_path_ / 900.0 + hb().max_m / 64.0;          // one unit through the lean finish stage (mean path capacity of the batch)
        if (enum_us < 8.0 * unit_us) return U;
        // (measured with the SV-carrying bench batch, 40 KB images: 160 / 192 / 256 / 320 workgroups = 1.24 / 1.17 / 1.20 / 1.24 ms
        // per step in round 2, profiles/r02_notes.md; with round 3's shorter lean stage 128 / 144 / 160 / 176 / 192 / 224 / 288 =
        // 1.16 / 1.14 / 1.09-1.11 / 1.087-1.091 / 1.10-1.13 / 1.10-1.14 / 1.13-1.15, profiles/r03_notes.md: the rule gives 164 there,
        // rounded up to a multiple of 16 = 176; round 2 rounded to 32 = 192)
        // (with the direct launch's units edited on their runs -- ambi_finish_edit_kernel, done well before the table -- the lean kernel
        // gets some of the room the full-stage workgroups used to take.  First build of that stage: 224 (this rule) / 256 / 288 / 320 / 384
        // workgroups = 0.877 / 0.836 / 0.802 / 0.792 / 0.813 ms per step on one box, against 0.844-0.855 with the full-stage launch; with the
        // shorter lean stage of the final code (hashed synthesis, run emission beside it): 192 / 208 / 224 / 240 / 256 / 272 / 288 / 304 / 320 /
        // 336 = 0.82-0.86 / 0.78-0.83 / 0.82 / 0.78-0.81 / 0.77-0.81 / 0.826 / 0.830 / 0.847 / 0.847 / 0.848 on two boxes, three runs each
        // (profiles/r04_notes.md): one lean workgroup per CU; a fifth more than the rule's own figure)
        const double room = (direct_edit_ && direct_n_ > 0 && direct_ext_ && d_direct_cells_) ? 1.2 : 1.0;
        int64_t grid = (int64_t)((double)U * unit_us * room / enum_us) + 1;
        grid = (grid + 15) & ~int64_t(15);
        if (grid < 32) grid = 32;
        return grid < U ? (int)grid : U;
    }
    void launch_back(int s, const BatchArgs& A) {    // enumerate, first valid order, finish
        hipStream_t st = slice_stream(s);
        const int U = A.n_units;
        const int grid = enum_grid_;
        // The scan for the first valid order (reads the first rows the prepare stage left, not the table).  Beside a long
        // enumerate kernel it is launched AHEAD of it (first_ahead_): behind the enumerate kernel in launch order its 4096
        // one-wave workgroups only get group memory as enumerate workgroups retire (0.06 ms alone, 0.5-0.6 ms beside), and
        // the finish kernels behind it then start half-way through the table and end after it.
        hipStream_t sb = overlap_back_ ? back_stream_ : st;
        auto launch_first = [&]() {
            if (first_launched_) {   // started beside the plan kernel (launch_front): the finish kernels wait for both
                (void)hipStreamWaitEvent(sb, ev_first_, 0); (void)hipStreamWaitEvent(sb, ev_plan_, 0);
                if (full_stream_) (void)hipStreamWaitEvent(full_stream_, ev_plan_, 0);
                return;
            }
            hipStream_t sf = (first_ahead_ == 2 && overlap_back_ && first_stream_) ? first_stream_ : sb;
            if (overlap_back_) (void)hipStreamWaitEvent(sf, ev_prep_, 0);
            tick("ambi_first_kernel", s, 4, true, sf);
            hipLaunchKernelGGL(ambi_first_kernel, dim3(U), dim3(64), lds_first_, sf, A);
            tick("ambi_first_kernel", s, 4, false, sf);
            if (overlap_back_) { (void)hipEventRecord(ev_first_, sf); if (sf != sb) (void)hipStreamWaitEvent(sb, ev_first_, 0); }
        };
        const bool ahead = (first_ahead_ > 0 && overlap_back_) || first_launched_;
        if (ahead) { launch_first(); if (first_ahead_ == 1) (void)hipStreamWaitEvent(st, ev_first_, 0); }
        tick("ambi_enumerate_kernel", s, 3, true);
        const int lds_emit = lds_blocks_;
        if (lazy_) {}   // the tables are written on demand (materialise_tables)
        else {
        if (enum_classes_ & 1) hipLaunchKernelGGL(ambi_enumerate_blocks_kernel<0>, dim3(grid), dim3(enum_threads_), lds_emit, st, A);
        if (enum_classes_ & 2) hipLaunchKernelGGL(ambi_enumerate_blocks_kernel<1>, dim3(grid), dim3(enum_threads_), lds_emit, st, A);
        if (enum_classes_ & 4) hipLaunchKernelGGL(ambi_enumerate_blocks_kernel<2>, dim3(grid), dim3(enum_threads_), lds_emit, st, A);
        if ((enum_classes_ & 1) && general_path_ != 0) hipLaunchKernelGGL(ambi_enumerate_kernel<0>, dim3(grid), dim3(256), lds_enum_, st, A);
        if ((enum_classes_ & 2) && general_path_ != 0) hipLaunchKernelGGL(ambi_enumerate_kernel<1>, dim3(grid), dim3(256), lds_enum_, st, A);
        if ((enum_classes_ & 4) && general_path_ != 0) hipLaunchKernelGGL(ambi_enumerate_kernel<2>, dim3(grid), dim3(256), lds_enum_, st, A);
        if (n_wide_ > 0) hipLaunchKernelGGL(ambi_enumerate_wide_kernel, dim3(64, n_wide_), dim3(256), 0, st, A, (const int32_t*)d_wide_units_, n_wide_);
        }
        tick("ambi_enumerate_kernel", s, 3, false);
        if (!ahead) launch_first();
        // units with deletion / duplication candidates go straight to the full finish stage, on a stream of their own beside
        // the lean kernel (both behind the scan, both beside the enumerate kernel); few workgroups, each taking units in turn
        if (direct_n_ > 0 && overlap_back_ && full_stream_) {
            (void)hipStreamWaitEvent(full_stream_, ev_first_, 0);
            const int dgrid = direct_n_ < direct_grid_ ? direct_n_ : direct_grid_;
            // path area of this launch: the capacity bound of the batch, or (experiment switch AMBI_DIRECT_CELLS, see wait())
            // fewer cells -- a path that does not fit then goes through the list kernel behind, which has the full area
            BatchArgs Ad = A;
            int lds_direct = lds_finish_;
            if (direct_cells_ > 0 && direct_cells_ < finish_path_cells_) {
                Ad.finish_path_cells = direct_cells_; Ad.finish_retry = 1;
                lds_direct = (int)finish_work_bytes(hb().max_n, hb().max_m, hb().max_bkp, direct_cells_, hb().max_out);
            }
            direct_retry_ = Ad.finish_retry != 0;
            tick("ambi_finish_ext_kernel", s, 6, true, full_stream_);
            if (direct_ext_ && d_direct_cells_ && direct_edit_) {
                // the edits of lone SVs on the runs of the path; what that stage hands on (chaining SVs, lists that outgrow their room)
                // to the launch with the path cells in device memory, over the list the first one leaves on the device
                direct_retry_ = false;
                const int egrid = direct_n_ < edit_grid_ ? direct_n_ : edit_grid_;
                hipLaunchKernelGGL(ambi_finish_edit_kernel, dim3(egrid), dim3(256), lds_finish_edit_, full_stream_, A, (const int32_t*)d_direct_list_, direct_n_, d_edit_list_, d_refin_count_ + 1);
                hipLaunchKernelGGL(ambi_finish_ext_kernel, dim3(dgrid < 64 ? dgrid : 64), dim3(full_threads_), lds_finish_ext_, full_stream_, A, (const int32_t*)d_edit_list_, 0, d_direct_cells_, direct_stride_,
                                   (const int32_t*)(d_refin_count_ + 1));
            } else
            if (direct_ext_ && d_direct_cells_) {   // path cells in device memory: a 13 KB workgroup that fits where a lean one fits
                direct_retry_ = false;
                hipLaunchKernelGGL(ambi_finish_ext_kernel, dim3(dgrid), dim3(full_threads_), lds_finish_ext_, full_stream_, A, (const int32_t*)d_direct_list_, direct_n_, d_direct_cells_, direct_stride_);
            } else
            hipLaunchKernelGGL(ambi_finish_kernel, dim3(dgrid), dim3(full_threads_), lds_direct, full_stream_, Ad, (const int32_t*)d_direct_list_, (const int32_t*)nullptr, direct_n_);
            tick("ambi_finish_ext_kernel", s, 6, false, full_stream_);
            (void)hipEventRecord(ev_full_, full_stream_);
        }
        tick("ambi_finish_kernel", s, 5, true, sb);
        const int fgrid = finish_grid_for(U);
        if (debug_) fprintf(stderr, "ambigram_hip: lean finish grid %d, image budget %d, mean path capacity %.0f, order bytes %lld\n", fgrid, block_lds_, avg_path_, (long long)last_needed_);
        if (lean_finish_) {
            if (lean_wave_) {
                const int wg = lean_wave_grid_ > 0 ? (lean_wave_grid_ < U ? lean_wave_grid_ : U) : (fgrid < U ? std::min(U, 4 * fgrid) : U);
                hipLaunchKernelGGL(ambi_finish_lean_wave_kernel, dim3(wg), dim3(64), lds_finish_lean_, sb, A);
            } else
            hipLaunchKernelGGL(ambi_finish_lean_kernel, dim3(fgrid), dim3(lean_threads_), lds_finish_lean_, sb, A, (const int32_t*)nullptr, 0);
            // units whose SVs chain or edit the path: the full stage right behind, over the list the lean kernel left on the
            // device (an empty list costs one launch of workgroups that exit at once)
            if (hb().any_sv) {
                if (direct_retry_ && direct_n_ > 0 && overlap_back_ && full_stream_) (void)hipStreamWaitEvent(sb, ev_full_, 0);   // the direct launch may add to the list
                hipLaunchKernelGGL(ambi_finish_kernel, dim3(U < 256 ? U : 256), dim3(256), lds_finish_, sb, A, (const int32_t*)d_refin_list_, (const int32_t*)d_refin_count_, -1);   // (at most one such workgroup fits a CU: more than 256 gain nothing)
            }
        } else hipLaunchKernelGGL(ambi_finish_kernel, dim3(U), dim3(256), lds_finish_, sb, A, (const int32_t*)nullptr, (const int32_t*)nullptr, -1);
        tick("ambi_finish_kernel", s, 5, false, sb);
        if (overlap_back_) {
            (void)hipEventRecord(ev_back_, sb); (void)hipStreamWaitEvent(st, ev_back_, 0);
            if (direct_n_ > 0 && full_stream_) (void)hipStreamWaitEvent(st, ev_full_, 0);
        }
    }

    // Everything of one run, queued on the caller's stream and the lease's side streams.  The FIRST run of a batch takes what
    // arena the lease has: the plan kernel reports what the tables need, and wait() grows the arena and runs the batch again
    // if that was not enough (round 2 ran prepare + plan, synchronised, sized the arena and only then queued the run --
    // two host round trips in front of every fresh batch).  AMBI_SLICES > 1 (an experiment) keeps the sizing pass.
    int64_t epoch_ = 0;
    int64_t results_epoch() const override { return epoch_; }
    void* own_stream() override {
        DeviceGuard dg_(device_);
        if (!lease_) return nullptr;
        if (!lease_->run_stream && hipStreamCreateWithFlags(&lease_->run_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return lease_->run_stream;
    }
    int run(uint32_t flags, void* stream) override {
        DeviceGuard dg_(device_);
        epoch_++;
        if (!uploaded_) return -32;
        if (ran_ && !tuned_ && tables_written_) { if (int rc = tune_after_first_run()) return rc; }
        stream_ = (hipStream_t)stream;
        if (classed_ && (want_back_ || want_full_ || want_first_ || want_lattice_)) {
            // side streams that dispatch beside THIS caller's stream (learnt once per stream and device: a few probe launches; a
            // stream seen before costs a look-up)
            static const bool first_prio = [] { const char* e = ambi_env("AMBI_FIRST_PRIORITY"); return e && atoi(e) != 0; }();
            if (stream_ != classed_for_) {
                hipStream_t sd[3] = {nullptr, nullptr, nullptr};
                const int rc = classified_side_streams(device_, stream_, (int)(lease_->uses & 1), sd);
                if (rc < 0) return rc;
                for (int k = 0; k < 3; k++) classed_streams_[k] = rc == 0 ? sd[k] : nullptr;   // (rc 1: keep the lease's streams, rounds 1-3's choice)
                classed_for_ = stream_;
            }
            if (want_back_ && classed_streams_[0]) back_stream_ = classed_streams_[0];
            if (want_full_ && classed_streams_[1]) full_stream_ = classed_streams_[1];
            static const bool first_on_back = [] { const char* e = ambi_env("AMBI_FIRST_ON_BACK"); return e && atoi(e) != 0; }();   // experiment: the scan on the lean finish kernel's stream (no event between the two)
            if (want_first_ && !first_prio && classed_streams_[2]) first_stream_ = first_on_back ? classed_streams_[0] : classed_streams_[2];
            if (want_lattice_ && classed_streams_[2]) lattice_stream_ = classed_streams_[2];
        }
        t_run_ = std::chrono::steady_clock::now();
        flushed_ = upload_pending_;
        if (int rc = flush_upload(stream_)) return rc;
        run_seq_ = ++lease_->seq;
        runs_parity_ = run_seq_ & 1;
        mail_valid_ = false;
        bind(flags);
        all_done_ = false;
        const int U = A_.n_units;
        lazy_ = (flags & FLAG_LAZY_ORDERS) != 0 && n_slices_ == 1;
        // one slice: no copy commands around the kernels (see BatchArgs::zero_pending)
        const bool direct = n_slices_ == 1 && dh_npending_ && dh_needed_;
        if (!direct) { HIP_CK(hipMemsetAsync(d_npending_, 0, sizeof(int32_t), stream_)); HIP_CK(hipMemsetAsync(d_refin_count_, 0, 2 * sizeof(int32_t), stream_)); }
        if (!arena_checked_ && n_slices_ > 1) {
            // first run of a sliced batch: size the arena regions of the slices from what their order tables need
            for (int s = 0; s < n_slices_; s++) launch_front(0, slice_args(s));   // all on the caller's stream
            HIP_CK(hipGetLastError());
            HIP_CK(hipMemcpyAsync(h_needed_, d_needed_, sizeof(int64_t) * n_slices_, hipMemcpyDeviceToHost, stream_));
            HIP_CK(hipStreamSynchronize(stream_));
            int64_t total = 0;
            for (int s = 0; s < n_slices_; s++) {
                slice_base_[s] = total;
                slice_bytes_[s] = (h_needed_[s] + (h_needed_[s] >> 4) + 4096 + order_align_ - 1) & ~int64_t(order_align_ - 1);
                total += slice_bytes_[s];
            }
            if (total > arena_bytes_) {
                (void)grow_arena(total);
                if (arena_bytes_ < total)
                    for (int s = 0; s < n_slices_; s++) { slice_base_[s] = (arena_bytes_ / n_slices_ * s) & ~int64_t(order_align_ - 1); slice_bytes_[s] = (arena_bytes_ / n_slices_) & ~int64_t(order_align_ - 1); }
                bind(flags);
            }
            arena_checked_ = true;
        }
        overlap_back_ = want_overlap_ && back_stream_ != nullptr && n_slices_ == 1;
        A_.direct_full_on = (overlap_back_ && direct_n_ > 0 && full_stream_ != nullptr) ? 1 : 0;
        express_ = n_slices_ == 1 && U <= express_units_ && lds_express_ <= kLdsMaxDynamic && dh_express_left_ != nullptr && direct && n_wide_ == 0;
        if (direct) { A_.zero_pending = 1; A_.host_pending = dh_npending_; A_.host_needed = dh_needed_; }
        // Software pipeline over the slices: slice s starts its latency-bound front (prepare, plan, image build) when
        // slice s-1 has finished its own and moves on to the HBM-bound enumerate kernel, so the two kinds of work
        // share the chip instead of alternating.
        fork();
        for (int s = 0; s < n_slices_; s++) {
            const BatchArgs A = slice_args(s);
            if (A.n_units <= 0) continue;
            if (s > 0 && stagger_) (void)hipStreamWaitEvent(slice_stream(s), ev_stage_[s - 1], 0);
            launch_front(s, A);
            if (lazy_) { tick("ambi_blocks_build_kernel", s, 2, true); tick("ambi_blocks_build_kernel", s, 2, false); }
            else if (!(build_in_emit_ && shared_units_ == 0)) launch_build(s, A);
            else { tick("ambi_blocks_build_kernel", s, 2, true); tick("ambi_blocks_build_kernel", s, 2, false); }   // nothing to build
            if (s + 1 < n_slices_) (void)hipEventRecord(ev_stage_[s], slice_stream(s));
            launch_back(s, A);
        }
        HIP_CK(hipGetLastError());
        join();
        if (!direct) {
            HIP_CK(hipMemcpyAsync(h_npending_, d_npending_, sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
            HIP_CK(hipMemcpyAsync(h_needed_, d_needed_, sizeof(int64_t) * n_slices_, hipMemcpyDeviceToHost, stream_));
        }
        if (side_lattice_) (void)hipEventRecord(lease_->ev_tail, stream_);   // where the NEXT run's side lattice kernel may start
        tables_written_ = !lazy_;
        ran_ = true; inflight_ = true;
        t_launched_ = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_run_).count();
        if (timing_) timed_runs_++;
        return 0;
    }
    // A larger arena: the new one first, the old one goes only when that worked -- a failed growth leaves the batch as it was,
    // and the units whose tables do not fit the old arena end with ORDERS_CAPACITY (plan kernel), one by one.
    // AMBI_ARENA_MAX_BYTES: an upper limit for the arena (a memory budget; the tests use it to reach this path).  Every stream idle.
    bool grow_arena(int64_t total) {
        int64_t want = total;
        { const char* cap = ambi_env("AMBI_ARENA_MAX_BYTES"); if (cap && atoll(cap) > 0 && want > atoll(cap)) want = atoll(cap); }
        bool grown = false;
        uint8_t* fresh = nullptr;
        want = (want + 4095) & ~int64_t(4095);
        if (want > arena_bytes_ && hipMalloc((void**)&fresh, (size_t)want) == hipSuccess) {
            (void)hipFree(lease_->d_arena);
            lease_->d_arena = fresh; lease_->d_arena_bytes = want;
            d_arena_ = fresh; arena_bytes_ = want;
            grown = true;
        } else if (want > arena_bytes_) (void)hipGetLastError();
        if (arena_bytes_ < total)
            fprintf(stderr, "ambigram_hip: order-table arena of %lld bytes not available; keeping %lld bytes (units beyond it report ORDERS_CAPACITY)\n",
                    (long long)total, (long long)arena_bytes_);
        return grown;
    }
    // FLAG_LAZY_ORDERS: the order tables of the last run, now (plan -> image build -> enumerate on the caller's stream, waited
    // for; the arena grows first if it must).  Every stream idle.
    int materialise_tables() {
        if (tables_written_ || !ran_) return 0;
        for (int attempt = 0; attempt < 2; attempt++) {
            BatchArgs A = slice_args(0);
            A.zero_pending = 1; A.host_pending = dh_npending_; A.host_needed = dh_needed_; A.lat_R = nullptr; A.plan_seq = nullptr; A.late_flag = nullptr;
            hipLaunchKernelGGL(ambi_plan_reset_kernel, dim3((A.n_units + 255) / 256), dim3(256), 0, stream_, A);
            hipLaunchKernelGGL(ambi_plan_kernel, dim3(1), dim3(1024), 0, stream_, A);
            hipLaunchKernelGGL(ambi_blocks_build_kernel, dim3(A.n_units), dim3(256), lds_build_, stream_, A);
            const int grid = enum_grid_;
            if (enum_classes_ & 1) hipLaunchKernelGGL(ambi_enumerate_blocks_kernel<0>, dim3(grid), dim3(enum_threads_), lds_blocks_, stream_, A);
            if (enum_classes_ & 2) hipLaunchKernelGGL(ambi_enumerate_blocks_kernel<1>, dim3(grid), dim3(enum_threads_), lds_blocks_, stream_, A);
            if (enum_classes_ & 4) hipLaunchKernelGGL(ambi_enumerate_blocks_kernel<2>, dim3(grid), dim3(enum_threads_), lds_blocks_, stream_, A);
            if (enum_classes_ & 1) hipLaunchKernelGGL(ambi_enumerate_kernel<0>, dim3(grid), dim3(256), lds_enum_, stream_, A);
            if (enum_classes_ & 2) hipLaunchKernelGGL(ambi_enumerate_kernel<1>, dim3(grid), dim3(256), lds_enum_, stream_, A);
            if (enum_classes_ & 4) hipLaunchKernelGGL(ambi_enumerate_kernel<2>, dim3(grid), dim3(256), lds_enum_, stream_, A);
            if (n_wide_ > 0) hipLaunchKernelGGL(ambi_enumerate_wide_kernel, dim3(64, n_wide_), dim3(256), 0, stream_, A, (const int32_t*)d_wide_units_, n_wide_);
            HIP_CK(hipGetLastError());
            HIP_CK(hipStreamSynchronize(stream_));
            const int64_t need = h_needed_[0];
            if (need <= slice_bytes_[0] || attempt == 1) break;
            const int64_t total = (need + (need >> 4) + 4096 + order_align_ - 1) & ~int64_t(order_align_ - 1);
            if (!grow_arena(total)) break;
            slice_base_[0] = 0; slice_bytes_[0] = arena_bytes_ & ~int64_t(order_align_ - 1);
            bind(A_.flags);
        }
        tables_written_ = true;
        return 0;
    }
    // the first complete run of a batch (every stream idle): did its tables fit the arena the lease had?
    int settle_first_run() {
        if (arena_checked_) return 0;
        arena_checked_ = true;
        if (n_slices_ != 1) return 0;
        const int64_t need = h_needed_[0];
        if (need > arena_bytes_) {
            const int64_t total = (need + (need >> 4) + 4096 + order_align_ - 1) & ~int64_t(order_align_ - 1);
            if (grow_arena(total)) {   // the whole batch again, now with room for every table (prepare resets what the first pass refused)
                slice_base_[0] = 0; slice_bytes_[0] = arena_bytes_ & ~int64_t(order_align_ - 1);
                lease_->h_words->late_flag = 0;
                ran_ = false;   // (the run below is still this batch's FIRST complete one: nothing is tuned from the refused pass)
                const bool t = timing_; timing_ = false;
                int rc = run(A_.flags, (void*)stream_);
                timing_ = t;
                if (rc) return rc;
                HIP_CK(hipStreamSynchronize(stream_));
                inflight_ = false;
            }
        }
        return 0;
    }
    // Launch parameters that depend on what the first run found (the batch is resident and immutable): which units need the
    // general enumerate kernel, whether any table is shared by several workgroups, the group-memory budget of the emission
    // workgroups.  Called before the SECOND run: a batch that is run once (one sample through the CLI) never pays for it.
    int tune_after_first_run() {
        if (int rc = wait()) return rc;
        tuned_ = true;
        {
            std::vector<int32_t> fb(hb().units.size());
            HIP_CK(hipMemcpy(fb.data(), d_fallback_, fb.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
            general_path_ = 0;
            for (int32_t f : fb) general_path_ += f != 0;
            if (direct_n_ > 0) {
                // Path area of the direct full-finish launch.  Default: the capacity bound of the batch (82 KB of group memory
                // per workgroup on the bench batch, one per CU).  AMBI_DIRECT_CELLS=n: n cells, -1: what the paths of these
                // units needed in this first run plus a quarter (53 KB there, three per CU; longer paths then go through the
                // list kernel) -- measured SLOWER, 1.18 vs 1.155 ms per step: more of these 16-wave workgroups get onto the
                // CUs while the table is being written and take the places of enumerate workgroups.
                const char* e = ambi_env("AMBI_DIRECT_CELLS");
                direct_cells_ = e ? atoi(e) : 0;
                if (direct_cells_ < 0) {
                    std::vector<UnitOut> hdr(hb().units.size());
                    HIP_CK(hipMemcpy(hdr.data(), d_results_, hdr.size() * sizeof(UnitOut), hipMemcpyDeviceToHost));
                    int mx = 0;
                    for (size_t u2 = 0; u2 < hdr.size(); u2++)
                        if (hb().units[u2].direct_full) { mx = std::max(mx, hdr[u2].path_len); mx = std::max(mx, hdr[u2].path_indel_len); }
                    direct_cells_ = mx > 0 ? ((mx + mx / 4 + 256 + 2047) & ~2047) : 0;
                }
            }
            // ... and whether any unit's table is shared by several workgroups (only those go through the build kernel
            // when single-block units build their image in the enumerate workgroup): if none, later runs do not launch it
            if (n_slices_ == 1) {
                std::vector<int64_t> bo(hb().units.size() + 1);
                HIP_CK(hipMemcpy(bo.data(), d_blk_off_, bo.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
                shared_units_ = 0;
                for (size_t u2 = 0; u2 + 1 < bo.size(); u2++) shared_units_ += (bo[u2 + 1] - bo[u2] > 1) ? 1 : 0;
                // Group memory per enumerate workgroup for the runs to come.  k workgroups share a CU's 160 KB and its store
                // bandwidth, so W work blocks take about ceil(W / (CUs * k)) * k "slot rounds": pick the k in 3..5 with the
                // fewest, among those whose budget 160 KB / k still holds every image of this batch (measured on the bench
                // batch, 4096 work blocks on 256 CUs: k = 3 (48 KB) 0.92-0.95 ms, k = 4 (40 KB) 0.86-0.87 ms).
                if (emit_lds_auto_ && general_path_ == 0 && bo.back() > 0) {
                    std::vector<int32_t> hd(hb().units.size() * 8);
                    HIP_CK(hipMemcpy(hd.data(), d_blk_hdr_, hd.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
                    int need = 0;
                    for (size_t u2 = 0; u2 < hb().units.size(); u2++) if (hd[8 * u2] > 0 && hd[8 * u2 + 3] > need) need = hd[8 * u2 + 3];
                    int ncu = 256;
                    { hipDeviceProp_t pr; int dev = 0; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ncu = pr.multiProcessorCount; }
                    const int64_t W = bo.back();
                    int best_k = 0; int64_t best = 0;
                    for (int k = 3; k <= 5; k++) {
                        const int budget = ((160 * 1024) / k) & ~15;
                        if (need <= 0 || need > budget) continue;
                        const int64_t rounds = ((W + (int64_t)ncu * k - 1) / ((int64_t)ncu * k)) * k;
                        if (!best_k || rounds < best) { best_k = k; best = rounds; }
                    }
                    if (debug_) fprintf(stderr, "ambigram_hip: image need %d bytes, %lld work blocks on %d CUs -> %d workgroups per CU\n", need, (long long)W, ncu, best_k);
                    // ... and ask for no more than the images need (512-byte granules): what the k enumerate workgroups leave of
                    // the CU's group memory is where the scan / finish workgroups run WITHOUT pushing an enumerate workgroup out
                    if (best_k) {
                        block_lds_ = ((160 * 1024) / best_k) & ~15;
                        const int tight = (need + 511) & ~511;
                        if (emit_lds_tight_ && tight < block_lds_) block_lds_ = tight;
                        lds_blocks_ = block_lds_;
                    }
                }
            }
        }
        return 0;
    }

    // parallel search for units whose sequential scan ran out of budget (rare)
    int slow_path() {
        epoch_++;
        const int U = A_.n_units;
        std::vector<UnitOut> hdr(U);
        HIP_CK(hipMemcpy(hdr.data(), d_results_, U * sizeof(UnitOut), hipMemcpyDeviceToHost));
        // two kinds of leftovers: units whose scan ran out of budget (parallel search, then the full finish stage) and units
        // the lean finish stage handed over (full finish stage only)
        std::vector<int32_t> pend, refin;
        for (int u = 0; u < U; u++) {
            if (hdr[u].status == ST_PENDING) {
                if (hdr[u].order_off < 0) {   // cannot happen (the scan only takes units the plan kernel gave rows): never search a table that is not there
                    const int32_t st = ST_ERR_ORDERS_CAPACITY;
                    HIP_CK(hipMemcpy(reinterpret_cast<uint8_t*>(d_results_) + sizeof(UnitOut) * (size_t)u + offsetof(UnitOut, status), &st, sizeof(int32_t), hipMemcpyHostToDevice));
                    continue;
                }
                pend.push_back(u);
            } else if (hdr[u].status == ST_REFINISH) refin.push_back(u);
        }
        if (pend.empty() && refin.empty()) return 0;
        const int np = (int)pend.size(), nfin = np + (int)refin.size(), chunk = 16;
        std::vector<int64_t> coff(np + 1, 0);
        for (int p = 0; p < np; p++) coff[p + 1] = coff[p] + (hdr[pend[p]].num_orders + chunk - 1) / chunk;
        std::vector<int32_t> fin(pend);
        fin.insert(fin.end(), refin.begin(), refin.end());
        DevBuf b_pend, b_coff, b_slots;
        HIP_CK(b_pend.alloc(nfin * sizeof(int32_t)));
        HIP_CK(b_coff.alloc((np + 1) * sizeof(int64_t)));
        HIP_CK(b_slots.alloc((np + 1) * sizeof(SearchSlot)));
        int32_t* d_pend = b_pend.as<int32_t>(); int64_t* d_coff = b_coff.as<int64_t>(); SearchSlot* d_slots = b_slots.as<SearchSlot>();
        HIP_CK(hipMemcpy(d_pend, fin.data(), nfin * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_CK(hipMemcpy(d_coff, coff.data(), (np + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
        int waves = (4 * lds_first_ <= 150 * 1024) ? 4 : 1;
        bool fwd = !(A_.flags & FLAG_REVERSED);
        // both passes are queued back to back: a unit resolved by pass 0 is skipped by pass 1 (its status is no longer PENDING)
        for (int pass = 0; pass < 2 && np > 0; pass++) {
            hipLaunchKernelGGL(ambi_search_init_kernel, dim3((np + 255) / 256), dim3(256), 0, stream_, d_slots, np);
            SearchArgs S{d_pend, d_coff, d_slots, np, chunk, fwd ? 1 : 0, lds_first_};
            int64_t nblk = (coff[np] + waves - 1) / waves;
            if (nblk > 65535 * 16) nblk = 65535 * 16;
            if (nblk < 1) nblk = 1;
            hipLaunchKernelGGL(ambi_search_kernel, dim3((unsigned)nblk), dim3(64 * waves), waves * lds_first_, stream_, A_, S);
            hipLaunchKernelGGL(ambi_resolve_kernel, dim3(np), dim3(64), lds_first_, stream_, A_, S, pass);
            HIP_CK(hipGetLastError());
            fwd = !fwd;
        }
        // finish: the units the search resolved go through the lean stage first (any path length) and through the full stage only
        // when their SVs chain or edit the path (the list the lean kernel leaves on the device); the units the main chain's lean
        // stage had handed over already go straight to the full stage
        if (np > 0) {
            BatchArgs Al = A_;
            Al.host_pending = nullptr; Al.direct_full_on = 0;
            HIP_CK(hipMemsetAsync(d_refin_count_, 0, sizeof(int32_t), stream_));
            hipLaunchKernelGGL(ambi_finish_lean_kernel, dim3(np < 1024 ? np : 1024), dim3(256), lds_finish_lean_, stream_, Al, (const int32_t*)d_pend, np);
            // ... with the path cells in device memory where the batch has such areas (units with deletion / duplication candidates:
            // any path length), else in group memory (up to kPathLdsCells cells)
            int32_t handed = 0;
            HIP_CK(hipMemcpyAsync(&handed, d_refin_count_, sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
            HIP_CK(hipStreamSynchronize(stream_));
            if (handed > 0 && d_direct_cells_ && direct_slots_ > 0)
                hipLaunchKernelGGL(ambi_finish_ext_kernel, dim3(handed < direct_slots_ ? handed : direct_slots_), dim3(full_threads_), lds_finish_ext_, stream_, Al,
                                   (const int32_t*)d_refin_list_, (int)handed, d_direct_cells_, direct_stride_);
            else if (handed > 0)
                hipLaunchKernelGGL(ambi_finish_kernel, dim3(handed < 256 ? handed : 256), dim3(256), lds_finish_, stream_, Al, (const int32_t*)d_refin_list_, (const int32_t*)d_refin_count_, -1);
        }
        if (nfin > np) hipLaunchKernelGGL(ambi_finish_kernel, dim3(nfin - np), dim3(256), lds_finish_, stream_, A_, (const int32_t*)(d_pend + np), (const int32_t*)nullptr, -1);
        HIP_CK(hipGetLastError());
        HIP_CK(hipStreamSynchronize(stream_));
        return 0;
    }

    int wait() override {
        DeviceGuard dg_(device_);
        if (!ran_) return 0;
        HIP_CK(hipStreamSynchronize(stream_));
        inflight_ = false;
        if (tables_written_) { if (int rc = settle_first_run()) return rc; }
        if (check_guards("wait")) return -31;
        if (tables_written_) { last_needed_ = 0; for (int s = 0; s < n_slices_; s++) last_needed_ += h_needed_[s]; }
        late_refusal_ = lease_->h_words->late_flag != 0;
        if (*h_npending_ > 0) {
            if (int rc = materialise_tables()) return rc;   // (FLAG_LAZY_ORDERS: the parallel search reads the tables)
            int rc = slow_path();
            if (rc) return rc;
            *h_npending_ = 0;
        }
        if ((A_.flags & FLAG_ALL) && !all_done_) {
            int rc = compute_all();
            if (rc) return rc;
            all_done_ = true;
        }
        // the mailbox of a small batch holds what the express kernel published; nothing behind it changed a header
        mail_valid_ = express_ && mail_on_ && !(A_.flags & FLAG_ALL) && *(volatile int32_t*)h_express_left_ == 0 && !late_refusal_;
        if (timing_ && timed_runs_ > 0) {
            // average duration of ONE launch of every kernel over the slices and the slots filled since timing was
            // switched on (every run launches each kernel once per slice)
            const long filled = timed_runs_ < kTimingSlots ? timed_runs_ : kTimingSlots;
            times_.clear();
            static const char* const kKernelNames[kTimedKernels] = {"ambi_prepare_kernel", "ambi_plan_kernel", "ambi_blocks_build_kernel",
                                                                    "ambi_enumerate_kernel", "ambi_first_kernel", "ambi_finish_kernel", "ambi_finish_ext_kernel"};
            for (int k = 0; k < kTimedKernels; k++) {
                double sum = 0, s0 = 0, s1 = 0; int cnt = 0, cnt_span = 0; const char* nm = kKernelNames[k];
                if (!((timing_mask_ >> k) & 1u)) { times_.push_back({nm, -1.0f}); continue; }
                for (long sl = 0; sl < filled; sl++) {
                    for (int sc = 0; sc < n_slices_; sc++) {
                        size_t at = ((size_t)sl * n_slices_ + sc) * kTimedKernels + k;
                        if (at >= evs().size() || !evs()[at].a) continue;
                        float ms = 0;
                        if (hipEventElapsedTime(&ms, evs()[at].a, evs()[at].b) == hipSuccess) { sum += ms; cnt++; nm = evs()[at].name; }
                        // where the kernel sits in its run: from the event in front of the run's first kernel (the events of one slot
                        // belong to one run; they sit on different streams of one device)
                        const size_t at0 = ((size_t)sl * n_slices_ + sc) * kTimedKernels;
                        float a = 0, b = 0;
                        if ((timing_mask_ & 1u) && at0 < evs().size() && evs()[at0].a && hipEventElapsedTime(&a, evs()[at0].a, evs()[at].a) == hipSuccess &&
                            hipEventElapsedTime(&b, evs()[at0].a, evs()[at].b) == hipSuccess) { s0 += a; s1 += b; cnt_span++; }
                    }
                }
                KernelTime kt{nm, cnt ? (float)(sum / cnt) : -1.0f};
                if (cnt_span) { kt.start_ms = (float)(s0 / cnt_span); kt.end_ms = (float)(s1 / cnt_span); }
                times_.push_back(kt);
            }
            (void)hipGetLastError();   // an event pair that was never recorded reports an error above: not one of ours
        }
        if (timing_ && (A_.flags & FLAG_ALL) && all_kernel_ms_ >= 0) {   // both passes of the --all evaluation of the last run
            if (!times_.empty() && !strcmp(times_.back().name, "ambi_all_kernel")) times_.pop_back();
            times_.push_back({"ambi_all_kernel", all_kernel_ms_});
        }
        return 0;
    }
    // Results complete (paths, breakpoints, output junctions of every unit) -- which, for a small batch on the express path,
    // is before the order tables behind them are written.  Returns early only when NOTHING behind the express kernel can
    // still void a result it published: no unit left to the ordinary kernels, and no unit that the lattice or plan stage
    // refuses afterwards (more order ideals than the table holds, an order table the arena has no room for).  For a batch
    // that has completed a run that is known (inputs are immutable); for a fresh batch the host also waits for the plan
    // kernel's verdict, which reaches it through pinned memory like the express kernel's.  Otherwise the same as wait().
    int wait_results() override {
        DeviceGuard dg_(device_);
        if (!ran_) return 0;
        static const bool lat = ambi_env("AMBI_DEBUG_LATENCY") != nullptr;   // diagnostics: when the two pinned words arrived, from the start of run()
        auto since = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_run_).count(); };
        if (express_ && !(A_.flags & FLAG_ALL) && ev_express_ && !late_refusal_ && (!lazy_ || arena_checked_)) {
            // the kernel's last workgroup stores the run's sequence number into pinned host memory: a short spin on it
            // returns microseconds before an event wait would; the event wait is the fallback
            volatile int32_t* seq = h_express_left_ + 1;
            bool seen = false;
            for (int spin = 0; spin < 400000 && !(seen = (*seq == run_seq_)); spin++) __builtin_ia32_pause();
            if (!seen) HIP_CK(hipEventSynchronize(ev_express_));
            const double t_express = lat ? since() : 0;
            if (*(volatile int32_t*)h_express_left_ == 0) {
                if (arena_checked_) { mail_valid_ = mail_on_; if (lat) fprintf(stderr, "ambigram latency: run() returned %.1f us, express word %.1f us\n", t_launched_, t_express); return 0; }
                // fresh batch: the verdict of the side lattice kernel (every lattice fine, all tables together fit the arena), or
                // else the plan kernel's
                volatile int32_t* pseq = side_lattice_ ? &lease_->h_words->lat_seq : &lease_->h_words->plan_seq;
                seen = false;
                for (int spin = 0; spin < 400000 && !(seen = (*pseq == run_seq_)); spin++) __builtin_ia32_pause();
                const bool fine = side_lattice_ ? *(volatile int32_t*)&lease_->h_words->lat_unsure == 0
                                                : (*(volatile int32_t*)&lease_->h_words->late_flag == 0 && *(volatile int64_t*)h_needed_ <= slice_bytes_[0]);
                if (seen && fine) {
                    arena_checked_ = true;     // every table of this batch fits the arena: true for all its runs
                    mail_valid_ = mail_on_;
                    if (lat) fprintf(stderr, "ambigram latency: run() returned %.1f us, express word %.1f us, plan word %.1f us (fresh batch)\n", t_launched_, t_express, since());
                    return 0;
                }
            }
        }
        return wait();
    }
    // header + final paths + output junctions of a unit on the HOST without a copy command: the pinned mailbox slot the
    // express kernel filled (nullptr: not available -- the caller downloads the result blob instead)
    const uint8_t* mail_slot(int unit) override {
        if (!mail_valid_ || !lease_ || unit < 0 || unit >= (int)mail_off_.size()) return nullptr;
        return lease_->h_mail + mail_off_[unit];
    }
    int download(std::vector<uint8_t>& blob) override {
        DeviceGuard dg_(device_);
        int rc = wait();
        if (rc) return rc;
        if (upload_pending_) { if ((rc = flush_upload(nullptr))) return rc; HIP_CK(hipStreamSynchronize(nullptr)); }   // never run: the blob is all zeroes
        blob.resize((size_t)hb().result_bytes);
        HIP_CK(hipMemcpy(blob.data(), d_results_, (size_t)hb().result_bytes, hipMemcpyDeviceToHost));
        if (d_stage_clk_) dump_stage_profile();
        return 0;
    }
    // AMBI_STAGE_PROFILE=1: mean shader-clock distance between consecutive marks of the per-unit stages (last run)
    void dump_stage_profile() {
        const size_t U = hb().units.size();
        std::vector<int64_t> clk(U * kStageSlots);
        if (hipMemcpy(clk.data(), d_stage_clk_, clk.size() * sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess) return;
        // marks 0-8 and 22-31: prepare (22-24 inside constructDAG, 26-28 inside the lattice); 9-12: scan; 16-21: finish.
        // Express kernel: 9 = start, 6 = junction side done, 7 = DAG built, 8 = order 0 found, 10 = both sides done (placement
        // included), 11 = imperfectFBI done, 12 = finish stage done
        fprintf(stderr, "ambigram_hip stage profile (mean cycles from the first mark of the stage, %zu units):", U);
        for (int s = 1; s < kStageSlots; s++) {
            const int base = ((s >= 13 && s <= 15) || s == 29 || s == 30) ? 31 : (s == 25 ? 9 : ((s <= 8 || s >= 22) ? 0 : (s <= 12 ? 9 : 16)));   // 13-15, 29, 30: image build (from 31)
            if (s == base) continue;
            double sum = 0; size_t cnt = 0;
            for (size_t u = 0; u < U; u++) {
                int64_t a = clk[u * kStageSlots + base];
                const int64_t b = clk[u * kStageSlots + s];
                if (!a && (base == 0 || base == 31)) a = clk[u * kStageSlots + 9];   // express kernel: the marks of the prepare pieces count from its own first mark
                if (a && b && b >= a) { sum += (double)(b - a); cnt++; }
            }
            if (cnt) fprintf(stderr, " [%d]=%.0f", s, sum / cnt);
        }
        fprintf(stderr, "\n");
    }
    int device_results(void** ptr, int64_t* bytes) override {
        if (ptr) *ptr = d_results_;
        if (bytes) *bytes = hb().result_bytes;
        return 0;
    }
    int pack_paths(int which, int32_t* dev_lengths, int32_t* dev_cells, int64_t cell_cap, int64_t* dev_total, void* stream) override {
        DeviceGuard dg_(device_);
        hipStream_t s = (hipStream_t)stream;
        bind(A_.flags);
        hipLaunchKernelGGL(ambi_pack_scan_kernel, dim3(1), dim3(1024), 0, s, A_, which, dev_lengths, d_pack_off_, dev_total);
        hipLaunchKernelGGL(ambi_pack_copy_kernel, dim3(A_.n_units), dim3(256), 0, s, A_, which, (const int64_t*)d_pack_off_, dev_cells, cell_cap);
        HIP_CK(hipGetLastError());
        return 0;
    }
    int pack_runs(int which, int32_t* dev_lengths, int32_t* dev_run_counts, int32_t* dev_run_start, int32_t* dev_run_len, int64_t run_cap,
                  int64_t* dev_totals, void* stream) override {
        DeviceGuard dg_(device_);
        hipStream_t s = (hipStream_t)stream;
        bind(A_.flags);
        hipLaunchKernelGGL(ambi_pack_runs_count_kernel, dim3(A_.n_units), dim3(256), 0, s, A_, which, dev_lengths, dev_run_counts);
        hipLaunchKernelGGL(ambi_pack_runs_scan_kernel, dim3(1), dim3(1024), 0, s, A_, (const int32_t*)dev_lengths, (const int32_t*)dev_run_counts,
                           d_pack_off_, dev_totals);
        hipLaunchKernelGGL(ambi_pack_runs_write_kernel, dim3(A_.n_units), dim3(256), 0, s, A_, which, (const int64_t*)d_pack_off_, dev_run_start,
                           dev_run_len, run_cap);
        HIP_CK(hipGetLastError());
        return 0;
    }
    // ---- final paths to the HOST in run-length form (the timed region ends with the "final path buffers on host";
    // the reference prints every path, LGM.cpp:3684-3689) ----
    // Slot layout (int32 words, device block and pinned mirror alike):
    //   [0..3] totals {runs, cells} as two int64 | lengths[U] | run_counts[U] | run_start[cap] | run_len[cap] | (headers: UnitOut[U])
    // runs_to_host queues, behind everything already on `stream`: the three pack kernels into the slot's device block, then ONE
    // device-to-host copy of the block on the lease's copy stream (so the caller's stream is free for the next run at once).
    // runs_wait waits for that copy; a slot whose capacity was too small is packed again with the capacity the totals name.
    int64_t runs_cap_[2] = {0, 0}; int runs_which_[2] = {0, 0}; bool runs_hdr_[2] = {false, false}, runs_queued_[2] = {false, false};
    int64_t runs_words(int64_t cap) const { return 4 + 2 * (int64_t)hb().units.size() + 2 * cap; }
    int runs_queue(int which, int slot, bool headers, void* stream, int64_t cap) {
        Lease* L = lease_;
        const int64_t U = (int64_t)hb().units.size();
        const int64_t words = runs_words(cap), hdr_bytes = headers ? U * (int64_t)sizeof(UnitOut) : 0;
        const int64_t bytes = ((words * 4 + 15) & ~int64_t(15)) + hdr_bytes;
        if (int rc = lease_device_block(&L->d_runs[slot], &L->d_runs_bytes[slot], bytes)) return rc;
        if (int rc = lease_pinned_block(&L->h_runs[slot], nullptr, &L->h_runs_bytes[slot], bytes)) return rc;
        if (!L->copy_stream) HIP_CK(hipStreamCreateWithFlags(&L->copy_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) {
            if (!L->ev_runs_packed[k]) HIP_CK(hipEventCreateWithFlags(&L->ev_runs_packed[k], hipEventDisableTiming));
            if (!L->ev_runs_done[k]) HIP_CK(hipEventCreateWithFlags(&L->ev_runs_done[k], hipEventDisableTiming));
        }
        hipStream_t s = (hipStream_t)stream;
        int32_t* w = reinterpret_cast<int32_t*>(L->d_runs[slot]);
        if (int rc = pack_runs(which, w + 4, w + 4 + U, w + 4 + 2 * U, w + 4 + 2 * U + cap, cap, reinterpret_cast<int64_t*>(w), stream)) return rc;
        HIP_CK(hipEventRecord(L->ev_runs_packed[slot], s));
        HIP_CK(hipStreamWaitEvent(L->copy_stream, L->ev_runs_packed[slot], 0));
        HIP_CK(hipMemcpyAsync(L->h_runs[slot], L->d_runs[slot], (size_t)(words * 4), hipMemcpyDeviceToHost, L->copy_stream));
        if (headers) HIP_CK(hipMemcpyAsync(L->h_runs[slot] + ((words * 4 + 15) & ~int64_t(15)), d_results_, (size_t)hdr_bytes, hipMemcpyDeviceToHost, L->copy_stream));
        HIP_CK(hipEventRecord(L->ev_runs_done[slot], L->copy_stream));
        runs_cap_[slot] = cap; runs_which_[slot] = which; runs_hdr_[slot] = headers; runs_queued_[slot] = true;
        return 0;
    }
    // which = 1 (the final path): the finish stages of the last run left the runs in the block of that run's parity (BatchArgs::run_*),
    // so there is nothing to compute -- ONE copy of that block on the copy stream, behind an event on the caller's stream.  which = 0, or
    // a batch with a unit whose runs overflowed its slots: the pack kernels (runs_queue).
    bool runs_direct_[2] = {false, false}; std::vector<int64_t> runs_off_[2];
    int runs_to_host(int which, int slot, int with_headers, void* stream) override {
        DeviceGuard dg_(device_);
        if (!ran_ || slot < 0 || slot > 1) return ST_ERR_BAD_INPUT;
        static const bool no_direct = ambi_env("AMBI_RUNS_PACK") != nullptr;   // env AMBI_RUNS_PACK=1: always through the pack kernels
        if (which == 1 && run_total_ > 0 && runs_parity_ >= 0 && !no_direct) {
            Lease* L = lease_;
            const int64_t U = (int64_t)hb().units.size();
            const int64_t words = 2 * U + 2 * run_total_, hdr_bytes = with_headers ? U * (int64_t)sizeof(UnitOut) : 0;
            const int64_t bytes = ((words * 4 + 15) & ~int64_t(15)) + hdr_bytes;
            if (int rc = lease_pinned_block(&L->h_runs[slot], nullptr, &L->h_runs_bytes[slot], bytes)) return rc;
            if (!L->copy_stream) HIP_CK(hipStreamCreateWithFlags(&L->copy_stream, hipStreamNonBlocking));
            for (int k = 0; k < 2; k++) {
                if (!L->ev_runs_packed[k]) HIP_CK(hipEventCreateWithFlags(&L->ev_runs_packed[k], hipEventDisableTiming));
                if (!L->ev_runs_done[k]) HIP_CK(hipEventCreateWithFlags(&L->ev_runs_done[k], hipEventDisableTiming));
            }
            HIP_CK(hipEventRecord(L->ev_runs_packed[slot], (hipStream_t)stream));
            HIP_CK(hipStreamWaitEvent(L->copy_stream, L->ev_runs_packed[slot], 0));
            HIP_CK(hipMemcpyAsync(L->h_runs[slot], d_run_blk_[runs_parity_], (size_t)(words * 4), hipMemcpyDeviceToHost, L->copy_stream));
            if (with_headers) HIP_CK(hipMemcpyAsync(L->h_runs[slot] + ((words * 4 + 15) & ~int64_t(15)), d_results_, (size_t)hdr_bytes, hipMemcpyDeviceToHost, L->copy_stream));
            HIP_CK(hipEventRecord(L->ev_runs_done[slot], L->copy_stream));
            runs_which_[slot] = which; runs_hdr_[slot] = with_headers != 0; runs_queued_[slot] = true; runs_direct_[slot] = true;
            return 0;
        }
        // capacity: what the last complete pack of this batch needed, else a bound from the breakpoint capacities (a run per
        // breakpoint pair, a few more where indelBFB edits the path)
        int64_t cap = std::max(runs_cap_[0], runs_cap_[1]);
        if (cap <= 0) { for (const UnitIn& un : hb().units) cap += un.bkp_cap / 2 + 8; cap += 64; }
        runs_direct_[slot] = false;
        return runs_queue(which, slot, with_headers != 0, stream, cap);
    }
    int runs_wait(int slot, RunsView* out) override {
        DeviceGuard dg_(device_);
        if (slot < 0 || slot > 1 || !runs_queued_[slot] || !out) return ST_ERR_BAD_INPUT;
        Lease* L = lease_;
        const int64_t U = (int64_t)hb().units.size();
        if (runs_direct_[slot]) {
            HIP_CK(hipEventSynchronize(L->ev_runs_done[slot]));
            const int32_t* w = reinterpret_cast<const int32_t*>(L->h_runs[slot]);
            int64_t nr = 0, nc = 0; bool over = false;
            for (int64_t u = 0; u < U; u++) { if (w[u] < 0) over = true; else nr += w[u]; nc += w[U + u]; }
            if (!over) {
                const int64_t words = 2 * U + 2 * run_total_;
                out->n_runs = nr; out->n_cells = nc;
                out->run_counts = w; out->lengths = w + U; out->run_start = w + 2 * U; out->run_len = w + 2 * U + run_total_;
                out->run_off = hb().run_slot.data();
                out->headers = runs_hdr_[slot] ? L->h_runs[slot] + ((words * 4 + 15) & ~int64_t(15)) : nullptr;
                out->bytes = (2 * U + 2 * nr) * 4 + (runs_hdr_[slot] ? U * (int64_t)sizeof(UnitOut) : 0);
                out->copied_bytes = words * 4 + (runs_hdr_[slot] ? U * (int64_t)sizeof(UnitOut) : 0);
                return 0;
            }
            // a unit's runs did not fit its slots: this batch through the pack kernels (results are still those of the last run)
            runs_direct_[slot] = false;
            int64_t cap = 0; for (const UnitIn& un : hb().units) cap += un.bkp_cap / 2 + 8; cap += 64;
            if (int rc = runs_queue(runs_which_[slot], slot, runs_hdr_[slot], stream_, cap)) return rc;
        }
        for (int attempt = 0; attempt < 2; attempt++) {
            HIP_CK(hipEventSynchronize(L->ev_runs_done[slot]));
            const int64_t* tot = reinterpret_cast<const int64_t*>(L->h_runs[slot]);
            if (tot[0] <= runs_cap_[slot]) break;
            if (attempt == 1) return ST_ERR_BAD_INPUT;
            // more runs than the slot holds (the kernels wrote none beyond it): once more with room for all of them
            if (int rc = runs_queue(runs_which_[slot], slot, runs_hdr_[slot], stream_, tot[0] + (tot[0] >> 3) + 64)) return rc;
        }
        const int32_t* w = reinterpret_cast<const int32_t*>(L->h_runs[slot]);
        const int64_t cap = runs_cap_[slot], words = runs_words(cap);
        out->n_runs = reinterpret_cast<const int64_t*>(w)[0]; out->n_cells = reinterpret_cast<const int64_t*>(w)[1];
        out->lengths = w + 4; out->run_counts = w + 4 + U; out->run_start = w + 4 + 2 * U; out->run_len = w + 4 + 2 * U + cap;
        runs_off_[slot].assign((size_t)U + 1, 0);
        for (int64_t u = 0; u < U; u++) runs_off_[slot][(size_t)u + 1] = runs_off_[slot][(size_t)u] + out->run_counts[u];
        out->run_off = runs_off_[slot].data();
        out->headers = runs_hdr_[slot] ? L->h_runs[slot] + ((words * 4 + 15) & ~int64_t(15)) : nullptr;
        out->bytes = (4 + 2 * U + 2 * out->n_runs) * 4 + (runs_hdr_[slot] ? U * (int64_t)sizeof(UnitOut) : 0);   // what the payload needs (the copy moves the slot's capacity)
        out->copied_bytes = words * 4 + (runs_hdr_[slot] ? U * (int64_t)sizeof(UnitOut) : 0);
        return 0;
    }
    int copy_orders(int unit, int64_t first, int64_t count, uint8_t* out) override {
        DeviceGuard dg_(device_);
        if (int rc = wait()) return rc;
        if (int rc = materialise_tables()) return rc;
        UnitOut h;
        HIP_CK(hipMemcpy(&h, d_results_ + sizeof(UnitOut) * (size_t)unit, sizeof(UnitOut), hipMemcpyDeviceToHost));
        if (h.order_off < 0 || first < 0 || first + count > h.num_orders) return ST_ERR_BAD_INPUT;
        // rows are Kpad bytes apart on the device; the caller gets count x K
        const int stride = row_stride(h.K);
        std::vector<uint8_t> tmp((size_t)(count * stride));
        HIP_CK(hipMemcpy(tmp.data(), d_arena_ + h.order_off + first * stride, tmp.size(), hipMemcpyDeviceToHost));
        // (5 bits per node up to 32 nodes, a byte above: row_node unpacks either)
        for (int64_t r = 0; r < count; r++) for (int d = 0; d < h.K; d++) out[r * h.K + d] = (uint8_t)row_node(tmp.data() + r * stride, h.K, d);
        return 0;
    }
    int copy_dag(int unit, Dag* out) override {
        DeviceGuard dg_(device_);
        HIP_CK(hipMemcpy(out, d_dags_ + unit, sizeof(Dag), hipMemcpyDeviceToHost));
        return 0;
    }
    int copy_dag_wide(int unit, int32_t* pat, int32_t* loop, uint64_t* succ2) override {
        DeviceGuard dg_(device_);
        if (unit < 0 || unit >= (int)hb().units.size() || hb().wide_index[unit] < 0) return ST_ERR_BAD_INPUT;
        const WideUnit* X = d_wide_ + hb().wide_index[unit];
        const uint8_t* base = reinterpret_cast<const uint8_t*>(X);
        HIP_CK(hipMemcpy(pat, base + offsetof(WideUnit, dag) + offsetof(WideDag, pat), sizeof(int32_t) * kWideNodeCap * 3, hipMemcpyDeviceToHost));
        HIP_CK(hipMemcpy(loop, base + offsetof(WideUnit, dag) + offsetof(WideDag, loop), sizeof(int32_t) * kWideNodeCap * 3, hipMemcpyDeviceToHost));
        HIP_CK(hipMemcpy(succ2, base + offsetof(WideUnit, succ), sizeof(WideSet) * kWideNodeCap, hipMemcpyDeviceToHost));
        return 0;
    }
    void set_timing(bool on) override { timing_ = on; timing_mask_ = ~0u; timed_runs_ = 0; }
    void set_timing_mask(uint32_t mask) override { timing_ = mask != 0; timing_mask_ = mask; timed_runs_ = 0; }
    const std::vector<KernelTime>& kernel_times() override { return times_; }
    int64_t order_bytes_written() const override { return last_needed_; }
    size_t object_bytes() const override { return sizeof(HipBackend); }
    int slice_count() const override { return n_slices_; }

    // ---- --all (LGM.cpp:3672-3695): every valid order of every unit, in the reference's print order ----
    // pass 0 = the first orientation (forward unless --reversed), pass 1 = the flipped one, run only for units whose LAST
    // order of pass 0 is invalid (LGM.cpp:3691-3695).  Device-side and batched: at wait() ONE launch per pass over all
    // (unit, 64-order chunk) work items (ambi_all_kernel: fused unrank + evaluate), validity bitmaps of R bits per pass
    // and unit stay in HBM, the host reads the per-unit counts (one copy) and a unit's bitmap only when its indices are
    // asked for; the paths are produced on demand by all_paths().
    int compute_all() {
        epoch_++;   // (the finalize kernel rewrites `evaluated` and the statuses of units with an undefined order)
        const int U = (int)hb().units.size();
        std::vector<UnitOut> hdr(U);
        HIP_CK(hipMemcpy(hdr.data(), d_results_, U * sizeof(UnitOut), hipMemcpyDeviceToHost));
        all_off_.assign(U + 1, 0);
        for (int u = 0; u < U; u++) {
            const bool live = hdr[u].status == ST_OK && hdr[u].num_orders > 0 && hdr[u].num_orders < (int64_t)kCountSat;
            all_off_[u + 1] = all_off_[u] + (live ? 2 * all_words(hdr[u].num_orders) : 0);
        }
        all_R_.resize(U);
        for (int u = 0; u < U; u++) all_R_[u] = hdr[u].num_orders;
        all_cache_.assign(U, {});
        all_idx_[0].assign(U, {}); all_idx_[1].assign(U, {});
        all_counts_.assign(2 * (size_t)U, 0);
        const int64_t words = all_off_[U], chunks = words / 2;
        // one pool: [bitmaps: words x 8 bytes][flags: U x 4 bytes] -- what ranks merge with ONE reduction when a wide sample's
        // orders are dealt over them (ambi_batch_all_device)
        const int64_t pool_words = words + ((int64_t)U + 1) / 2 + 1;
        if (d_all_bits_ && all_bits_cap_ < pool_words) { (void)hipFree(d_all_bits_); d_all_bits_ = nullptr; }
        if (!d_all_off_) { if (int rc = dalloc(&d_all_off_, (size_t)U + 1)) return rc; if (int rc = dalloc(&d_all_count_, 2 * (size_t)U)) return rc; }
        if (!d_all_bits_) { HIP_CK(hipMalloc((void**)&d_all_bits_, (size_t)pool_words * sizeof(uint64_t))); all_bits_cap_ = pool_words; }
        d_all_flags_ = reinterpret_cast<int32_t*>(d_all_bits_ + words);
        all_pool_bytes_ = (words + ((int64_t)U + 1) / 2) * (int64_t)sizeof(uint64_t);
        HIP_CK(hipMemcpyAsync(d_all_off_, all_off_.data(), (U + 1) * sizeof(int64_t), hipMemcpyHostToDevice, stream_));
        HIP_CK(hipMemsetAsync(d_all_bits_, 0, (size_t)pool_words * sizeof(uint64_t), stream_));
        HIP_CK(hipMemsetAsync(d_all_count_, 0, 2 * (size_t)U * sizeof(int32_t), stream_));
        bind(A_.flags);
        all_kernel_ms_ = -1.f;
        if (chunks > 0) {
            const int wave_lds = (int)((lds_first_ + 64 * kFirstRowStride + 15) & ~15);
            const int waves = (4 * wave_lds <= 64 * 1024) ? 4 : ((2 * wave_lds <= 150 * 1024) ? 2 : 1);
            int64_t nblk = (chunks + waves - 1) / waves;
            if (nblk > (1 << 20)) nblk = 1 << 20;   // grid-stride beyond
            EventPair ev;
            hipEvent_t& ea = ev.a; hipEvent_t& eb = ev.b;
            if (timing_) { HIP_CK(hipEventCreate(&ea)); HIP_CK(hipEventCreate(&eb)); HIP_CK(hipEventRecord(ea, stream_)); }
            // units with a short breakpoint path (the rule): one thread per order; the others: one wavefront per order
            int lane_cap = 0, lanes_units = 0, wave_units = 0;
            { const char* e = ambi_env("AMBI_ALL_LANES"); lane_cap = (e && atoi(e) == 0) ? 0 : kAllLaneMaxCells; }
            auto lane_unit = [&](int u) { return hb().units[u].bkp_cap <= lane_cap && hb().units[u].n_elem <= kMaxNodes; };
            for (int u = 0; u < U; u++) if (all_off_[u + 1] > all_off_[u]) { if (lane_unit(u)) lanes_units++; else wave_units++; }
            int max_lane_cells = 8;
            for (int u = 0; u < U; u++) if (lane_unit(u) && hb().units[u].bkp_cap > max_lane_cells) max_lane_cells = hb().units[u].bkp_cap;
            const int head_bytes = (int)((first_work_bytes(hb().max_n, 8) + 15) & ~15);
            // group memory for a staged copy of the unit's automaton (env AMBI_ALL_AUTO_LDS): 0 = the lanes unrank through L2.
            // Measured (4096 bench units): 0 / 4096 / 8192 bytes = 561 / 489 / 390 M orders/s -- the kernel lives on the number
            // of resident wavefronts (group-memory latency), and every KB of group memory costs some.
            int auto_bytes = 0;
            { const char* e = ambi_env("AMBI_ALL_AUTO_LDS"); if (e) auto_bytes = atoi(e) & ~15; if (auto_bytes < 0) auto_bytes = 0; }
            int max_k_lane = 1;
            for (const UnitIn& un : hb().units) if (un.n_elem <= kMaxNodes && un.n_elem > max_k_lane) max_k_lane = un.n_elem;   // (wide units: ambi_all_kernel)
            const int rows_bytes = 64 * ((max_k_lane + 3) & ~3);   // transposed orders: one 64-lane row per position
            const int lane_wave_lds = (head_bytes + rows_bytes + max_lane_cells * 64 * (int)sizeof(cell_t) + auto_bytes + 15) & ~15;
            int lane_waves = 1;   // wavefronts per workgroup (measured 1 / 2 / 4 = 563 / 553 / 379 M orders/s: group-memory allocation granularity)
            { const char* e = ambi_env("AMBI_ALL_WAVES"); if (e && atoi(e) >= 1 && atoi(e) <= 4) lane_waves = atoi(e); }
            int64_t lblk = (chunks + lane_waves - 1) / lane_waves;
            if (lblk > (1 << 20)) lblk = 1 << 20;
            for (int pass = 0; pass < 2; pass++) {
                if (lanes_units) hipLaunchKernelGGL(ambi_all_lanes_kernel, dim3((unsigned)lblk), dim3(64 * lane_waves), lane_waves * lane_wave_lds, stream_, A_, pass, lane_wave_lds, chunks, lane_cap, head_bytes, max_lane_cells, auto_bytes, rows_bytes);
                if (wave_units) hipLaunchKernelGGL(ambi_all_kernel, dim3((unsigned)nblk), dim3(64 * waves), waves * wave_lds, stream_, A_, pass, wave_lds, chunks, lane_cap);
            }
            if (timing_) HIP_CK(hipEventRecord(eb, stream_));
            hipLaunchKernelGGL(ambi_all_finalize_kernel, dim3((U + 255) / 256), dim3(256), 0, stream_, A_);
            HIP_CK(hipGetLastError());
            HIP_CK(hipMemcpyAsync(all_counts_.data(), d_all_count_, 2 * (size_t)U * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
            HIP_CK(hipStreamSynchronize(stream_));
            if (timing_) (void)hipEventElapsedTime(&all_kernel_ms_, ea, eb);
        }
        return 0;
    }
    // --all over several ranks: which chunks this rank evaluates; the pool the ranks merge; counts / headers after the merge
    int set_shard(int rank, int world) override {
        if (world < 1 || rank < 0 || rank >= world) return ST_ERR_BAD_INPUT;
        all_rank_ = rank; all_world_ = world;
        return 0;
    }
    int all_device(void** ptr, int64_t* bytes) override {
        if (ptr) *ptr = d_all_bits_;
        if (bytes) *bytes = d_all_bits_ ? all_pool_bytes_ : 0;
        return 0;
    }
    int all_finish() override {
        DeviceGuard dg_(device_);
        if (!d_all_bits_) return 0;
        const int U = (int)hb().units.size();
        hipLaunchKernelGGL(ambi_all_finalize_kernel, dim3((U + 255) / 256), dim3(256), 0, stream_, A_);
        HIP_CK(hipGetLastError());
        HIP_CK(hipMemcpyAsync(all_counts_.data(), d_all_count_, 2 * (size_t)U * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
        HIP_CK(hipStreamSynchronize(stream_));
        all_cache_.assign(U, {});
        all_idx_[0].assign(U, {}); all_idx_[1].assign(U, {});
        return 0;
    }
    // valid order indices of one pass of a unit, from its bitmap (fetched once per unit)
    int all_indices(int unit, int pass, const std::vector<int64_t>** out) {
        if (all_cache_[unit].empty() && all_off_[unit + 1] > all_off_[unit]) {
            all_cache_[unit].resize((size_t)(all_off_[unit + 1] - all_off_[unit]));
            HIP_CK(hipMemcpy(all_cache_[unit].data(), d_all_bits_ + all_off_[unit], all_cache_[unit].size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
            const int64_t nw = all_words(all_R_[unit]);
            for (int ps = 0; ps < 2; ps++) {
                auto& v = all_idx_[ps][unit];
                v.clear();
                for (int64_t w = 0; w < nw; w++) {
                    uint64_t x = all_cache_[unit][(size_t)(ps * nw + w)];
                    while (x) { v.push_back(w * 64 + __builtin_ctzll(x)); x &= x - 1; }
                }
            }
        }
        *out = &all_idx_[pass][unit];
        return 0;
    }
    int all_count(int unit, int pass, int64_t* count) override {
        if (pass < 0 || pass > 1 || unit < 0) { if (count) *count = 0; return ST_ERR_BAD_INPUT; }
        if (count) *count = (size_t)(2 * unit + pass) < all_counts_.size() ? (int64_t)all_counts_[2 * (size_t)unit + pass] : 0;
        return 0;
    }
    int all_orders(int unit, int pass, int64_t first, int64_t count, int64_t* idx) override {
        DeviceGuard dg_(device_);
        if (pass < 0 || pass > 1 || unit < 0 || unit >= (int)all_idx_[pass].size()) return ST_ERR_BAD_INPUT;
        const std::vector<int64_t>* vp = nullptr;
        if (int rc = all_indices(unit, pass, &vp)) return rc;
        const auto& v = *vp;
        if (first < 0 || count < 0 || first + count > (int64_t)v.size()) return ST_ERR_BAD_INPUT;
        for (int64_t i = 0; i < count; i++) idx[i] = v[first + i];
        return 0;
    }
    int all_paths(int unit, int pass, int64_t first, int64_t count, int32_t* lengths, int32_t* cells, int64_t stride) override {
        DeviceGuard dg_(device_);
        if (pass < 0 || pass > 1 || unit < 0 || unit >= (int)all_idx_[pass].size()) return ST_ERR_BAD_INPUT;
        const std::vector<int64_t>* vp = nullptr;
        if (int rc = all_indices(unit, pass, &vp)) return rc;
        const auto& v = *vp;
        if (first < 0 || count < 0 || first + count > (int64_t)v.size() || stride <= 0) return ST_ERR_BAD_INPUT;
        if (count == 0) return 0;
        const UnitIn& Uin = hb().units[unit];
        const bool fwd0 = !(A_.flags & FLAG_REVERSED), fwd = pass == 0 ? fwd0 : !fwd0;
        DevBuf b_idx, b_len, b_cells;
        HIP_CK(b_idx.alloc((size_t)count * sizeof(int64_t)));
        HIP_CK(b_len.alloc((size_t)count * sizeof(int32_t)));
        HIP_CK(b_cells.alloc((size_t)count * (size_t)stride * sizeof(int32_t)));
        int64_t* d_idx = b_idx.as<int64_t>(); int32_t* d_len = b_len.as<int32_t>(); int32_t* d_cells = b_cells.as<int32_t>();
        HIP_CK(hipMemcpy(d_idx, v.data() + first, (size_t)count * sizeof(int64_t), hipMemcpyHostToDevice));
        const int lds = (int)(first_work_bytes(Uin.n_seg, Uin.bkp_cap, Uin.n_elem > kMaxNodes) + 4ll * (Uin.bkp_cap / 2 + 2) + 16);
        hipLaunchKernelGGL(ambi_order_paths_kernel, dim3((unsigned)count), dim3(64), lds, stream_, A_, unit, fwd ? 1 : 0, (const int64_t*)d_idx, d_len,
                           d_cells, stride);
        HIP_CK(hipGetLastError());
        HIP_CK(hipMemcpyAsync(lengths, d_len, (size_t)count * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
        HIP_CK(hipMemcpyAsync(cells, d_cells, (size_t)count * (size_t)stride * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
        HIP_CK(hipStreamSynchronize(stream_));
        return 0;
    }
};

Backend* make_backend() { return new HipBackend(); }

// ILP entries on the device (LocalGenomicMap::BFB_ILP, LGM.cpp:4397-4752: int32 column + f64 coefficient per non-zero, 56.5 M
// non-zeros = 0.68 GB at 256 segments).  The non-zero space is cut into chunks of 1024 consecutive positions, one workgroup each, and
// every thread owns FOUR consecutive positions: one 16-byte store of columns and two of coefficients per thread, 1 KB + 2 KB
// contiguous per wavefront, every lane busy.  Rows differ in length by three orders of magnitude (2 n^2 / 3 entries for a segment row,
// one for the innermost nesting rows), so positions, not rows, are dealt out: the host names the first row of every chunk (a merge
// walk, O(rows + chunks)), the workgroup stages the offsets and descriptors of the rows that reach into its chunk in group memory
// (at most 1025), a thread finds its row by bisection there; when its four positions lie in one row (almost always) the row family is
// decided once and the four entries share the piece bounds and the triangle arithmetic (ilp_row_entries<4>), else entry by entry.
// What bounds it (profiles/r04_notes.md): NOT the stores -- a plain two-array store stream of this very shape runs at 5.76 TB/s
// (profiles/tools/hbm_write_two_streams.hip), this kernel at 3.4-3.6 -- but instructions per entry: 65.7 M vector instructions in the
// entry-by-entry form = 122 us of issue time alone.  Rounds 1-3 (one entry per thread, 4- / 8-byte stores) and a form with one
// wavefront-uniform row piece per step (scalar family dispatch, half the lanes idle on the short pieces) measure 0.190-0.199 and 0.22 ms.
#ifndef AMBI_ILP_PER
#define AMBI_ILP_PER 4
#endif
constexpr int kIlpChunk = 1024, kIlpStage = kIlpChunk + 4, kIlpPer = AMBI_ILP_PER, kIlpThreads = kIlpChunk / kIlpPer;   // entries per thread (4 or 8), threads per workgroup
__global__ __launch_bounds__(kIlpThreads) void ambi_ilp_fill_kernel(const IlpRowDesc* rows, const int64_t* row_ptr, const int32_t* chunk_row, int64_t nnz, IlpGeom G,
                                                            const int32_t* lit_col, const double* lit_val, int32_t* col, double* val) {
    __shared__ int32_t rel[kIlpStage];          // row offsets relative to the chunk's first position (rows are far shorter than 2^31)
    __shared__ IlpRowDesc desc[kIlpStage];
    const int64_t P0 = (int64_t)blockIdx.x * kIlpChunk;
    const int r_lo = chunk_row[blockIdx.x], R = chunk_row[blockIdx.x + 1] - r_lo + 1;   // rows r_lo .. r_lo + R - 1 reach into the chunk
    for (int i = threadIdx.x; i <= R; i += blockDim.x) {
        const int64_t d = row_ptr[r_lo + i] - P0;
        rel[i] = (int32_t)(d > (int64_t)0x3fffffff ? (int64_t)0x3fffffff : d);
    }
    for (int i = threadIdx.x; i < R; i += blockDim.x) desc[i] = rows[r_lo + i];
    __syncthreads();
    const int p = kIlpPer * (int)threadIdx.x;
    const int64_t left = nnz - P0;
    const int lim = left < (int64_t)kIlpChunk ? (int)left : kIlpChunk;   // positions of this chunk that exist
    if (p >= lim) return;
    int lo = 0, hi = R;                         // last staged row that starts at or before p
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (rel[mid] <= p) lo = mid; else hi = mid; }
    int r = lo;
    while (p >= rel[r + 1]) r++;                // (empty rows are stepped over)
    int32_t c4[kIlpPer]; double v4[kIlpPer];
#pragma unroll
    for (int k = 0; k < kIlpPer; k++) { c4[k] = 0; v4[k] = 0; }
    const int want = lim - p < kIlpPer ? lim - p : kIlpPer;
#if defined(AMBI_ILP_NOCOMPUTE)   // timing experiment: the stores without the entries
    if (true) { for (int k = 0; k < kIlpPer; k++) { c4[k] = p + k + r; v4[k] = 1.0; } }
#else
    if (want == kIlpPer && rel[r + 1] - p >= kIlpPer) ilp_row_entries<kIlpPer>(desc[r], G, p - rel[r], kIlpPer, lit_col, lit_val, c4, v4);
#endif
    else {
#pragma unroll
        for (int k = 0; k < kIlpPer; k++) {
            const int q = p + k;
            if (k < want) {
                while (q >= rel[r + 1]) r++;
                ilp_row_entry(desc[r], G, (int64_t)(q - rel[r]), lit_col, lit_val, &c4[k], &v4[k]);
            }
        }
    }
    if (want == kIlpPer) {
#pragma unroll
        for (int k = 0; k < kIlpPer; k += 4) *reinterpret_cast<int4*>(col + P0 + p + k) = make_int4(c4[k], c4[k + 1], c4[k + 2], c4[k + 3]);
#pragma unroll
        for (int k = 0; k < kIlpPer; k += 2) *reinterpret_cast<double2*>(val + P0 + p + k) = make_double2(v4[k], v4[k + 1]);
    } else {
        for (int k = 0; k < want; k++) { col[P0 + p + k] = c4[k]; val[P0 + p + k] = v4[k]; }
    }
}

int backend_expand_runs(const int32_t* run_start, const int32_t* run_len, const int64_t* cell_off, int64_t n_runs, int32_t* cells,
                        int64_t cell_cap, void* stream) {
    int64_t blocks = (n_runs + 3) / 4;
    if (blocks > 65536) blocks = 65536;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(ambi_expand_runs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, run_start, run_len, cell_off, n_runs, cells,
                       cell_cap);
    HIP_CK(hipGetLastError());
    return 0;
}

int backend_ilp_fill(const IlpRowDesc* rows, int64_t n_rows, const int64_t* row_ptr, int s, int e, const int32_t* lit_col, const double* lit_val,
                     int64_t n_lit, int32_t* col, double* val, float* kernel_ms) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return -30;   // AMBI_ERR_NO_DEVICE: no CPU fallback
    const int64_t nnz = row_ptr[n_rows];
    // the first row of every chunk of kIlpChunk positions (last row that starts at or before the chunk's first position), and the
    // last row behind the last chunk
    const int64_t n_chunks = (nnz + kIlpChunk - 1) / kIlpChunk;
    std::vector<int32_t> chunk_row((size_t)n_chunks + 1, 0);
    {
        int64_t r = 0;
        for (int64_t c = 0; c < n_chunks; c++) {
            const int64_t pos = c * kIlpChunk;
            while (r + 1 < n_rows && row_ptr[r + 1] <= pos) r++;
            chunk_row[(size_t)c] = (int32_t)r;
        }
        chunk_row[(size_t)n_chunks] = (int32_t)(n_rows > 0 ? n_rows - 1 : 0);
        for (int64_t c = 0; c < n_chunks; c++)
            if (chunk_row[(size_t)c + 1] - chunk_row[(size_t)c] + 2 > kIlpStage) return ST_ERR_BAD_INPUT;   // (more than 1024 rows inside 1024 positions: rows are not empty)
    }
    int32_t* d_chunk = nullptr;
    IlpRowDesc* d_rows = nullptr; int64_t* d_ptr = nullptr; int32_t* d_lc = nullptr; double* d_lv = nullptr; int32_t* d_col = nullptr; double* d_val = nullptr;
    HIP_CK(hipMalloc((void**)&d_rows, (size_t)(n_rows > 0 ? n_rows : 1) * sizeof(IlpRowDesc)));
    HIP_CK(hipMalloc((void**)&d_ptr, (size_t)(n_rows + 1) * sizeof(int64_t)));
    HIP_CK(hipMalloc((void**)&d_lc, (size_t)(n_lit > 0 ? n_lit : 1) * sizeof(int32_t)));
    HIP_CK(hipMalloc((void**)&d_lv, (size_t)(n_lit > 0 ? n_lit : 1) * sizeof(double)));
    HIP_CK(hipMalloc((void**)&d_col, (size_t)(nnz > 0 ? nnz : 1) * sizeof(int32_t)));
    HIP_CK(hipMalloc((void**)&d_val, (size_t)(nnz > 0 ? nnz : 1) * sizeof(double)));
    HIP_CK(hipMalloc((void**)&d_chunk, chunk_row.size() * sizeof(int32_t)));
    HIP_CK(hipMemcpy(d_chunk, chunk_row.data(), chunk_row.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_CK(hipMemcpy(d_rows, rows, (size_t)n_rows * sizeof(IlpRowDesc), hipMemcpyHostToDevice));
    HIP_CK(hipMemcpy(d_ptr, row_ptr, (size_t)(n_rows + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    if (n_lit > 0) { HIP_CK(hipMemcpy(d_lc, lit_col, (size_t)n_lit * sizeof(int32_t), hipMemcpyHostToDevice)); HIP_CK(hipMemcpy(d_lv, lit_val, (size_t)n_lit * sizeof(double), hipMemcpyHostToDevice)); }
    hipEvent_t ea, eb;
    HIP_CK(hipEventCreate(&ea)); HIP_CK(hipEventCreate(&eb));
    const int64_t grid = n_chunks > 0 ? n_chunks : 1;
    const int reps = kernel_ms ? 5 : 1;   // the timed figure is the mean of the last 4 of 5 launches
    float total = 0;
    for (int r = 0; r < reps; r++) {
        HIP_CK(hipEventRecord(ea, nullptr));
        if (n_chunks > 0)
            hipLaunchKernelGGL(ambi_ilp_fill_kernel, dim3((unsigned)grid), dim3(256), 0, nullptr, (const IlpRowDesc*)d_rows, (const int64_t*)d_ptr, (const int32_t*)d_chunk, nnz,
                               ilp_geom(s, e), (const int32_t*)d_lc, (const double*)d_lv, d_col, d_val);
        HIP_CK(hipEventRecord(eb, nullptr));
        HIP_CK(hipEventSynchronize(eb));
        float ms = 0;
        HIP_CK(hipEventElapsedTime(&ms, ea, eb));
        if (r > 0) total += ms;
    }
    HIP_CK(hipGetLastError());
    if (kernel_ms) *kernel_ms = reps > 1 ? total / (reps - 1) : 0;
    HIP_CK(hipMemcpy(col, d_col, (size_t)nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_CK(hipMemcpy(val, d_val, (size_t)nnz * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
    (void)hipFree(d_chunk); (void)hipFree(d_rows); (void)hipFree(d_ptr); (void)hipFree(d_lc); (void)hipFree(d_lv); (void)hipFree(d_col); (void)hipFree(d_val);
    return 0;
}

}  // namespace ambi
