// ambi_eval.hpp -- assembly of one candidate order into a fold-back palindrome breakpoint path.
//
// Restates the per-order body of LocalGenomicMap::getBFB (LGM.cpp:3519-3658) and imperfectFBI (:3431-3512) as
// SPMD code: the breakpoint path `bkp` (int16 signed local vertex ids) lives in the group's LDS; the reverse
// std::find with its parity / nesting tests becomes a strided scan + group max-reduction, the vector::insert
// of a loop's 4*cn breakpoints becomes a chunked LDS shift.  One wavefront evaluates one order.
#pragma once
#include "ambi_common.hpp"
#include "ambi_group.hpp"

namespace ambi {

typedef int16_t cell_t;

// fold-back map of the unit (getJuncCN's `inversions`): for local segment id i, the junction's two segment ids,
// or src[i]==0 when the segment has no entry.
struct InvMap {
    const int16_t* src;
    const int16_t* tgt;
};

template <class G>
AMBI_HD int find_last_slot(const G& g, const cell_t* bkp, int L, int target, bool less) {
    // LGM.cpp:3591-3602: last odd position f holding `target` that does not fail the nesting test
    int best = -1;
    for (int f = g.tid(); f < L; f += g.size()) {
        if ((f & 1) && bkp[f] == target) {
            bool skip = false;
            if (f < L - 2) {
                int x = iabs(bkp[f - 1]), y = iabs(bkp[f + 2]);
                skip = less ? (x < y) : (x > y);
            }
            if (!skip && f > best) best = f;
        }
    }
    return g.max_i32(best);
}

// open a gap of `count` cells at index `at` (cells [at,L) move up); group-cooperative, in place.  A round moves
// kShiftVec * size() cells: every thread reads its cells of the round (top-down, so nothing is overwritten before it was
// read by a LATER round), all wait, every thread writes -- two barriers per round instead of per size() cells.
constexpr int kShiftVec = 8;
template <class G, class T>
AMBI_HD void shift_up(const G& g, T* a, int L, int at, int count) {
    const int n = L - at, step = kShiftVec * g.size();
    for (int top = n; top > 0; top -= step) {
        T v[kShiftVec];
#pragma unroll
        for (int k = 0; k < kShiftVec; k++) {
            const int i = top - 1 - g.tid() - k * g.size();
            v[k] = i >= 0 ? a[at + i] : (T)0;
        }
        g.sync();
#pragma unroll
        for (int k = 0; k < kShiftVec; k++) {
            const int i = top - 1 - g.tid() - k * g.size();
            if (i >= 0) a[at + i + count] = v[k];
        }
        g.sync();
    }
}

// close the gap [at, at+count): cells [at+count, L) move down
template <class G, class T>
AMBI_HD void shift_down(const G& g, T* a, int L, int at, int count) {
    const int n = L - (at + count), step = kShiftVec * g.size();
    for (int base = 0; base < n; base += step) {
        T v[kShiftVec];
#pragma unroll
        for (int k = 0; k < kShiftVec; k++) {
            const int i = base + g.tid() + k * g.size();
            v[k] = i < n ? a[at + count + i] : (T)0;
        }
        g.sync();
#pragma unroll
        for (int k = 0; k < kShiftVec; k++) {
            const int i = base + g.tid() + k * g.size();
            if (i < n) a[at + i] = v[k];
        }
        g.sync();
    }
}

// LGM.cpp:3431-3512.  Returns false when the reference would have read out of bounds.
// The walk over pos is the reference's; inside a step nothing is left to one thread: the cells a step looks at are read
// by every thread (same addresses), the first later occurrence of -bkp[pos] is one flag rank per chunk of cells, the
// plain step's rewrite is computed by all and stored by one, and the rewrite of a palindrome -- in the reference a loop
// from the middle outwards -- runs one iteration per thread: iteration t touches the four cells p1, p1+1, p2-1, p2 with
// p1 = mid - 2t, p2 = mid + 1 + 2t, reads only bkp[p1], and no two iterations share a cell (parities), so they commute.
template <class G>
AMBI_HD bool imperfect_fbi(const G& g, cell_t* bkp, int L, const InvMap& inv) {
    int pos = 0;
    while (pos < L) {
        if (pos + 1 >= L) return false;
        const int c0 = bkp[pos], c1 = bkp[pos + 1];
        int r = L;
        if (pos + 3 < L) {
            const int want = -c0;
            for (int base = pos + 3; base < L; base += g.size()) {
                const int q = base + g.tid();
                const int hit = g.first_flag(q < L && bkp[q] == want);
                if (hit >= 0) { r = base + hit; break; }
            }
        }
        const int l = r - 1;
        const bool plain = (r == L) || (bkp[l] != -c1);
        g.sync();
        if (plain) {
            int n0 = c0, n1 = c1;
            {
                const int id = iabs(c1);
                const int s = inv.src[id];
                if (s != 0) {
                    const int t = inv.tgt[id];
                    n1 = c1 > 0 ? ((s < t) ? s : t) : ((s < t) ? -t : -s);
                }
            }
            if (pos > 0) {
                const int id = iabs(c0);
                const int s = inv.src[id];
                if (s != 0 && iabs(bkp[pos - 1]) == id) {
                    const int other = (s == id) ? inv.tgt[id] : s;
                    n0 = c0 > 0 ? other : -other;
                }
            }
            if (n0 > 0 && iabs(n0) > iabs(n1)) n1 = n0;
            if (n0 < 0 && iabs(n0) < iabs(n1)) n1 = n0;
            if (g.tid() == 0) { bkp[pos] = (cell_t)n0; bkp[pos + 1] = (cell_t)n1; }
            g.sync();
            pos += 2;
        } else {
            const int mid = pos + ((l - pos) / 2);
            int bad = 0;
            // iterations t = 0, 1, ..: p1 = mid - 2t while p1 >= pos - 1 and p1 > 0
            for (int t = g.tid();; t += g.size()) {
                const int p1 = mid - 2 * t, p2 = mid + 1 + 2 * t;
                if (!(p1 >= pos - 1 && p1 > 0)) break;
                const int v = bkp[p1];
                const int id = iabs(v);
                const int s = inv.src[id];
                if (s != 0) {
                    const int tt = inv.tgt[id];
                    if (p1 + 1 >= L) { bad = 1; break; }
                    int w0, w1;
                    if (v > 0) { if (s < tt) { w0 = s; w1 = -tt; } else { w0 = tt; w1 = -s; } }
                    else { if (s < tt) { w0 = -tt; w1 = s; } else { w0 = -s; w1 = tt; } }
                    bkp[p1] = (cell_t)w0; bkp[p1 + 1] = (cell_t)w1;
                    if (p2 != p1 + 1) {
                        if (p1 > pos - 1) { if (p2 >= L) { bad = 1; break; } bkp[p2] = (cell_t)-w0; }
                        bkp[p2 - 1] = (cell_t)-w1;
                    }
                }
            }
            g.sync();
            if (g.any(bad != 0)) return false;
            pos = r + 1;
        }
    }
    return true;
}

// Second part of the evaluation: imperfectFBI (LGM.cpp:3656, always before the validity test) and the verdict.
// `placed` = what eval_place returned (>= 0).
template <class G>
AMBI_HD int eval_finish(const G& g, int placed, int K, cell_t* bkp, int L, const InvMap& inv) {
    const bool ok = imperfect_fbi(g, bkp, L, inv);
    // imperfectFBI reading past the end of the breakpoint vector (LGM.cpp:3436-3442 with pos+1 == end) is undefined in the
    // reference.  On an order that did not place all its elements the outcome is "invalid" whatever that read returns
    // (validity is i == K, fixed before imperfectFBI runs, and the breakpoints of an invalid order are thrown away,
    // LGM.cpp:3522), so the order is simply invalid here as well -- what the reference reports whenever it survives the
    // read.  On a VALID order the printed path would depend on the stray value: that is refused.
    if (!ok && placed == K) return ST_ERR_REF_UB;
    return (placed == K) ? 1 : 0;
}

// Placement part of the evaluation of one order (LGM.cpp:3519-3646): seeds and places the elements; needs the DAG only.
// Returns the number of elements placed (== K: all) or a negative Status; *L_out = bkp length.
// DAG: anything with K, pat[][3], loop[][3] -- Dag (up to 63 nodes) or WideDag (ambi_wide.hpp, up to 127).
template <class G, class DAG>
AMBI_HD int eval_place(const G& g, const DAG& D, const uint8_t* ord, bool forward, cell_t* bkp, int cap, int* L_out) {
    const int K = D.K;
    int L = 0;
    int x = ord[0];
    bool isPat = D.pat[x][0] != 0, isLoop = D.loop[x][0] != 0;
    if (!isPat && !isLoop) { *L_out = 0; return ST_ERR_REF_UB; }   // reference indexes an empty vector
    {
        int s = isPat ? D.pat[x][0] : D.loop[x][0], e = isPat ? D.pat[x][1] : D.loop[x][1];
        int q0, q1, q2, q3, len;
        if (forward) { q0 = s; q1 = e; q2 = -e; q3 = -s; } else { q0 = -e; q1 = -s; q2 = s; q3 = e; }
        len = isPat ? 2 : 4 * D.loop[x][2];
        if (len > cap) { *L_out = 0; return ST_ERR_BKP_CAPACITY; }
        for (int i = g.tid(); i < len; i += g.size()) {
            int w = i & 3;
            bkp[i] = (cell_t)(w == 0 ? q0 : w == 1 ? q1 : w == 2 ? q2 : q3);
        }
        L = len;
        g.sync();
    }
    int i;
    for (i = 1; i < K; i++) {
        x = ord[i];
        if (D.pat[x][0] != 0) {   // LGM.cpp:3572-3585
            int s = D.pat[x][0], e = D.pat[x][1];
            if (L == 0) { *L_out = 0; return ST_ERR_REF_UB; }
            int back = bkp[L - 1];
            if (L + 2 > cap) { *L_out = L; return ST_ERR_BKP_CAPACITY; }
            g.sync();
            if (back == -s) { if (g.tid() == 0) { bkp[L] = (cell_t)s; bkp[L + 1] = (cell_t)e; } L += 2; }
            else if (back == e) { if (g.tid() == 0) { bkp[L] = (cell_t)-e; bkp[L + 1] = (cell_t)-s; } L += 2; }
            else break;
            g.sync();
        } else if (D.loop[x][0] != 0) {   // LGM.cpp:3586-3644
            int s = D.loop[x][0], e = D.loop[x][1], cn = D.loop[x][2];
            // LGM.cpp:3591-3602: the last slot holding -s (nesting test "<"), else the last slot holding e (">"): both
            // searches in one sweep and one reduction (a -s hit outranks every e hit)
            int best = -1;
            for (int q = g.tid(); q < L; q += g.size()) {
                if (!(q & 1)) continue;
                const int c = bkp[q];
                if (c != -s && c != e) continue;
                bool skip = false;
                if (q < L - 2) {
                    const int x = iabs(bkp[q - 1]), y = iabs(bkp[q + 2]);
                    skip = (c == -s) ? (x < y) : (x > y);
                }
                if (!skip) { const int enc = (c == -s) ? 0x10000 + q : q; if (enc > best) best = enc; }
            }
            best = g.max_i32(best);
            if (best < 0) break;
            const bool viaV1 = best >= 0x10000;
            const int f = viaV1 ? best - 0x10000 : best;
            int cnt = 4 * cn;
            if (L + cnt > cap) { *L_out = L; return ST_ERR_BKP_CAPACITY; }
            bool hasNext = (f + 1 != L);
            shift_up(g, bkp, L, f + 1, cnt);
            int q0, q1, q2, q3, fix0, fix1;
            if (viaV1) { q0 = s; q1 = e; q2 = -e; q3 = -s; fix0 = -s; fix1 = s; }
            else { q0 = -e; q1 = -s; q2 = s; q3 = e; fix0 = e; fix1 = -e; }
            for (int k = g.tid(); k < cnt; k += g.size()) {
                int w = k & 3;
                bkp[f + 1 + k] = (cell_t)(w == 0 ? q0 : w == 1 ? q1 : w == 2 ? q2 : q3);
            }
            if (g.tid() == 0) {
                bkp[f] = (cell_t)fix0;                       // *temp (LGM.cpp:3627 / :3640)
                if (hasNext) bkp[f + 1 + cnt] = (cell_t)fix1;   // *(temp+1), written before the insert shifts it
            }
            L += cnt;
            g.sync();
        }
        // both slots empty (possible after the library sort for K > 16): nothing is placed, the loop goes on
    }
    *L_out = L;
    return i;
}

// The placement part for ONE wavefront with the breakpoint cells in REGISTERS (device code): cell i lives in lane i & 63,
// register i >> 6 -- up to kRegCells cells.  A placement of the group-memory form above costs ~2 700 cycles on a lone wavefront
// (a strided scan, a max-reduction, a chunked shift through group memory with two barriers per round, the writes: ~25 dependent
// round trips); here the candidate test is a compare per lane with its neighbours fetched by lane shuffles, the last candidate a
// wave-wide maximum, and the insert's shift one lane rotation per register.  Same cells, same return values; kRegsGiveUp when the
// path outgrows the registers (the caller then runs eval_place).  Writes the cells to `bkp` (group memory) at the end.
constexpr int kRegCells = 256;
constexpr int kRegsGiveUp = -1000;
#if defined(__HIP_DEVICE_COMPILE__)
// NR registers per lane hold 64 * NR cells; everything that loops over the registers is unrolled at compile time
template <int NR>
struct RegCells {
    int c[NR];
    // cell i, i uniform over the wavefront
    __device__ inline int get_u(int i) const {
        int v = c[0];
#pragma unroll
        for (int r = 1; r < NR; r++) v = (i >> 6) == r ? c[r] : v;
        return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(i & 63));
    }
    __device__ inline void set_u(int i, int v, int lane) {
        const bool me = lane == (i & 63);
#pragma unroll
        for (int r = 0; r < NR; r++) c[r] = (me && (i >> 6) == r) ? v : c[r];
    }
};
template <int NR, class DAG>
__device__ inline int eval_place_regs_n(const DAG& D, const uint8_t* ord, bool forward, cell_t* bkp, int cap, int* L_out) {
    constexpr int kCells = 64 * NR;
    const int lane = (int)(threadIdx.x & 63u);
    WaveGroup g;
    const int K = D.K;
    if (K > 64) return kRegsGiveUp;
    int L = 0;
    // the elements in placement order, one per lane: the loop below fetches element i with v_readlane instead of a chain of
    // dependent reads (order -> node -> record)
    int e_p0 = 0, e_p1 = 0, e_l0 = 0, e_l1 = 0, e_l2 = 0;
    if (lane < K) { const int xj = ord[lane]; e_p0 = D.pat[xj][0]; e_p1 = D.pat[xj][1]; e_l0 = D.loop[xj][0]; e_l1 = D.loop[xj][1]; e_l2 = D.loop[xj][2]; }
    auto elem = [&](int v, int i) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(i)); };
    const bool isPat = elem(e_p0, 0) != 0, isLoop = elem(e_l0, 0) != 0;
    if (!isPat && !isLoop) { *L_out = 0; return ST_ERR_REF_UB; }
    RegCells<NR> R;
    {
        const int s = isPat ? elem(e_p0, 0) : elem(e_l0, 0), e = isPat ? elem(e_p1, 0) : elem(e_l1, 0);
        int q0, q1, q2, q3;
        if (forward) { q0 = s; q1 = e; q2 = -e; q3 = -s; } else { q0 = -e; q1 = -s; q2 = s; q3 = e; }
        const int len = isPat ? 2 : 4 * elem(e_l2, 0);
        if (len > cap) { *L_out = 0; return ST_ERR_BKP_CAPACITY; }
        if (len > kCells) return kRegsGiveUp;
        const int w = lane & 3, v = w == 0 ? q0 : w == 1 ? q1 : w == 2 ? q2 : q3;      // (64 is a multiple of 4: the pattern is the same in every register)
#pragma unroll
        for (int r = 0; r < NR; r++) R.c[r] = 64 * r + lane < len ? v : 0;
        L = len;
    }
    int i;
    for (i = 1; i < K; i++) {
        const int p0 = elem(e_p0, i), l0 = elem(e_l0, i);
        if (p0 != 0) {   // LGM.cpp:3572-3585
            const int s = p0, e = elem(e_p1, i);
            if (L == 0) { *L_out = 0; return ST_ERR_REF_UB; }
            const int back = R.get_u(L - 1);
            if (L + 2 > cap) { *L_out = L; return ST_ERR_BKP_CAPACITY; }
            if (L + 2 > kCells) return kRegsGiveUp;
            if (back == -s) { R.set_u(L, s, lane); R.set_u(L + 1, e, lane); L += 2; }
            else if (back == e) { R.set_u(L, -e, lane); R.set_u(L + 1, -s, lane); L += 2; }
            else break;
        } else if (l0 != 0) {   // LGM.cpp:3586-3644
            const int s = l0, e = elem(e_l1, i), cn = elem(e_l2, i);
            // candidates: odd slots holding -s or e that pass the nesting test |cell[q-1]| vs |cell[q+2]| (only below L-2); the
            // neighbours come from the lane below (same register: q is odd) and from two lanes above (the next register for lanes 62, 63)
            const int up = (lane + 2) & 63;
            const bool wrap = lane >= 62;
            int above[NR + 1];
#pragma unroll
            for (int r = 0; r < NR; r++) above[r] = __shfl(R.c[r], up, 64);
            above[NR] = 0;
            int best = -1;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const int q = 64 * r + lane, c = R.c[r];
                const bool cand = q < L && (q & 1) && (c == -s || c == e);
                const int below = __builtin_amdgcn_update_dpp(0, c, 0x111 /* row_shr:1 */, 0xf, 0xf, false);   // lane - 1 (candidates sit in odd lanes: never the first lane of a row)
                bool skip = false;
                if (q < L - 2) { const int xx = iabs(below), yy = iabs(wrap ? above[r + 1] : above[r]); skip = (c == -s) ? (xx < yy) : (xx > yy); }
                const int enc = (c == -s) ? 0x10000 + q : q;
                if (cand && !skip && enc > best) best = enc;
            }
            best = g.max_i32(best);
            if (best < 0) break;
            const bool viaV1 = best >= 0x10000;
            const int f = viaV1 ? best - 0x10000 : best;
            const int cnt = 4 * cn;
            if (L + cnt > cap) { *L_out = L; return ST_ERR_BKP_CAPACITY; }
            if (L + cnt > kCells) return kRegsGiveUp;
            const bool hasNext = (f + 1 != L);
            // cells [f+1, L) move up by cnt: cell i takes cell i - cnt -- a lane rotation by cnt & 63, the source register one lower for
            // the lanes the rotation wraps (and cnt >> 6 lower for long loops)
            {
                const int rot = cnt & 63, sl = (lane - rot) & 63, down = (cnt >> 6) + (lane < rot ? 1 : 0);
                int sh[NR];
#pragma unroll
                for (int r = 0; r < NR; r++) sh[r] = __shfl(R.c[r], sl, 64);
                int qq0, qq1, qq2, qq3;
                if (viaV1) { qq0 = s; qq1 = e; qq2 = -e; qq3 = -s; } else { qq0 = -e; qq1 = -s; qq2 = s; qq3 = e; }
                const int w = (lane - (f + 1)) & 3, ins = w == 0 ? qq0 : w == 1 ? qq1 : w == 2 ? qq2 : qq3;   // (the register base is a multiple of 4)
#pragma unroll
                for (int r = NR - 1; r >= 0; r--) {
                    const int ci = 64 * r + lane;
                    int moved = 0;
#pragma unroll
                    for (int t = 0; t < NR; t++) moved = (r - down == t) ? sh[t] : moved;
                    R.c[r] = ci >= f + 1 + cnt ? moved : (ci >= f + 1 ? ins : R.c[r]);
                }
            }
            R.set_u(f, viaV1 ? -s : e, lane);                          // *temp (LGM.cpp:3627 / :3640)
            if (hasNext) R.set_u(f + 1 + cnt, viaV1 ? s : -e, lane);   // *(temp+1): the cell that stood behind f, now behind the insert
            L += cnt;
        }
        // both slots empty (possible after the library sort for K > 16): nothing is placed, the loop goes on
    }
#pragma unroll
    for (int r = 0; r < NR; r++) if (64 * r + lane < L) bkp[64 * r + lane] = (cell_t)R.c[r];
    g.sync();
    *L_out = L;
    return i;
}
template <class DAG>
__device__ inline int eval_place_regs(const DAG& D, const uint8_t* ord, bool forward, cell_t* bkp, int cap, int* L_out) {
    if (cap <= 64) return eval_place_regs_n<1>(D, ord, forward, bkp, cap, L_out);
    if (cap <= 128) return eval_place_regs_n<2>(D, ord, forward, bkp, cap, L_out);
    if (cap <= 256) return eval_place_regs_n<4>(D, ord, forward, bkp, cap, L_out);
    return kRegsGiveUp;
}
#endif

// Evaluate one order.  Returns 1 valid / 0 invalid / negative Status on error.  *L_out = bkp length.
template <class G, class DAG>
AMBI_HD int eval_order(const G& g, const DAG& D, const uint8_t* ord, bool forward, const InvMap& inv,
                       cell_t* bkp, int cap, int* L_out, int64_t* clk = nullptr) {
    const int K = D.K;
    int L = 0;
    const int i = eval_place(g, D, ord, forward, bkp, cap, &L);
    *L_out = L;
    if (i < 0) return i;
    clk_mark(g, clk, 25);
    return eval_finish(g, i, K, bkp, L, inv);
}


// LGM.cpp:3661-3670: breakpoint pairs -> per-segment path (int16 local signed ids).  `offs` = scratch of
// L/2+1 ints.  Returns P or a negative Status.
template <class G, class GP = int16_t>
AMBI_HD int expand_bkp(const G& g, const cell_t* bkp, int L, cell_t* path, int pcap, int32_t* offs, GP* gpath = nullptr,
                       int seg_base = 0) {
    int np = L / 2;
    int carry = 0;
    for (int base = 0; base < np; base += g.size()) {
        int j = base + g.tid();
        int len = 0;
        if (j < np) {
            int a = bkp[2 * j], b = bkp[2 * j + 1];
            if (a > 0) { len = iabs(b) - a + 1; } else { len = (-a) - iabs(b) + 1; }
            if (len < 0) len = 0;
        }
        int tot;
        int ex = g.exscan_i32(len, &tot);
        if (j < np) offs[j] = carry + ex;
        carry += tot;
    }
    const int P = carry;
    if (g.tid() == 0) offs[np] = P;
    g.sync();
    if (P > pcap) return ST_ERR_PATH_CAPACITY;
    // run-major: a sub-group of up to 64 threads (one wavefront on the GPU) writes the run of one pair; both the
    // '+' run a, a+1, .. and the '-' run -|a|, -(|a|-1), .. are  a + k.  The absolute-id copy for the result blob
    // (gpath, optional) leaves in the same pass.
    const int lanes = g.size() < 64 ? g.size() : 64;
    const int sub = g.tid() / lanes, nsub = g.size() / lanes, lane = g.tid() - sub * lanes;
    for (int j = sub; j < np; j += nsub) {
        const int a = bkp[2 * j], o0 = offs[j], len = offs[j + 1] - o0;
        for (int k = lane; k < len; k += lanes) {
            const int v = a + k;
            if (path) path[o0 + k] = (cell_t)v;
            // the result blob holds local ids in 2 bytes (readers add the base); 4-byte callers (--all paths) get absolute ids
            if (gpath) gpath[o0 + k] = sizeof(GP) == 2 ? (GP)v : (GP)(v > 0 ? v + seg_base : v - seg_base);
        }
    }
    g.sync();
    return P;
}

}  // namespace ambi
