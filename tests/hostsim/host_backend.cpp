// tests/hostsim/host_backend.cpp -- TEST INFRASTRUCTURE ONLY.
// A Backend that executes the engine's SPMD stage code (ambigram_amd/csrc/ambi_stages.hpp -- the very source the
// HIP kernels instantiate) with the 1-thread HostGroup on heap memory.  It lets `pytest -m "not gpu"` check the
// stage logic, the packing, the result-blob layout and the C ABI against the oracle without a GPU, and lets the
// sanitizers run over the kernel code on the CPU.  It is built into tests/hostsim/libambigram_hostsim.so only;
// the product library libambigram_hip.so does not contain it and nothing in ambigram_amd/ loads it by default.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../ambigram_amd/csrc/ambi_backend.hpp"
#include "../../ambigram_amd/csrc/ambi_ilp_rows.hpp"
#include "../../ambigram_amd/csrc/ambi_stages.hpp"

namespace ambi {

class HostSimBackend : public Backend {
    HostBatch hb_;
    EngineConfig cfg_;
    std::vector<UnitIn> units_;
    std::vector<Dag> dags_;
    std::vector<uint8_t> results_, arena_, first_rows_;
    std::vector<uint64_t> ikeys_, icnt_, aavail_, acnt_;
    std::vector<uint32_t> anblk_; std::vector<uint8_t> adepth_;
    std::vector<uint16_t> achild_;
    std::vector<uint32_t> ilink_;
    std::vector<int32_t> ilvl_off_, icounter_, ipos_, acbase_, rows_per_lane_, scratch_;
    std::vector<int64_t> blk_off_;
    int32_t n_pending_ = 0, refin_count_ = 0;
    std::vector<int32_t> refin_list_;
    int64_t orders_needed_ = 0;
    std::vector<KernelTime> times_;
    std::vector<WideUnit> wide_;          // working sets of the units with 64..127 nodes (ambi_wide.hpp)
    std::vector<int32_t> run_blk_;        // final paths in run-length form as the finish stages leave them (BatchArgs::run_*)
    BatchArgs A_{};

  public:
    const char* name() const override { return "hostsim"; }
    int device_count(int* n) override { if (n) *n = 0; return 0; }
    int set_device(int) override { return 0; }

    int upload(const HostBatch& hb, const EngineConfig& cfg) override {
        hb_ = hb; cfg_ = cfg;
        units_ = hb.units;
        const size_t U = units_.size();
        dags_.assign(U, Dag{});
        results_.assign((size_t)hb.result_bytes, 0);
        ikeys_.assign((size_t)hb.ideal_slots, 0); icnt_.assign((size_t)hb.ideal_slots, 0);
        ipos_.assign((size_t)hb.ideal_slots, 0);
        ilink_.assign((size_t)hb.ideal_slots * 4 + 8, 0);
        ilvl_off_.assign(U * (kMaxNodes + 3), 0); icounter_.assign(2 * U, 0);
        aavail_.assign((size_t)hb.ideal_slots / 2 + 1, 0); acnt_.assign((size_t)hb.ideal_slots / 2 + 1, 0);
        anblk_.assign((size_t)hb.ideal_slots / 2 + 1, 0); adepth_.assign((size_t)hb.ideal_slots / 2 + 8, 0);
        acbase_.assign((size_t)hb.ideal_slots / 2 + U + 1, 0); achild_.assign((size_t)hb.ideal_slots * 4 + 8, 0);
        rows_per_lane_.assign(U, 1); blk_off_.assign(U + 1, 0);
        scratch_.assign((size_t)hb.scratch_ints + 8, 0);
        // env AMBI_HOSTSIM_TABLE_SCAN=1: the scan for the first valid order reads the order table (the other supported source)
        if (!ambi_env("AMBI_HOSTSIM_TABLE_SCAN")) first_rows_.assign(U * (size_t)(cfg.first_budget > 0 ? cfg.first_budget : 1) * kFirstRowStride, 0);
        else first_rows_.clear();
        arena_.assign((size_t)(cfg.order_arena_bytes > 0 ? cfg.order_arena_bytes : 0), 0);
        wide_.assign((size_t)hb.n_wide, WideUnit{});
        run_blk_.assign(2 * U + 2 * (size_t)(hb.run_slot.empty() ? 0 : hb.run_slot.back()) + 1, 0);
        return 0;
    }

    void bind(uint32_t flags) {
        A_.n_units = (int32_t)units_.size(); A_.unit_base = 0; A_.arena_base = 0;   // one slice
        A_.flags = flags; A_.order_align = 4096 /* as HipBackend's default */; A_.first_budget = cfg_.first_budget; A_.target_lanes = cfg_.target_lanes;
        { const char* envm = ambi_env("AMBI_BLOCK_MAX"); int bm = envm ? atoi(envm) : cfg_.block_max;   // as HipBackend::upload
          if (bm < 1) bm = 1; if (bm > kBlockMaxLimit) bm = kBlockMaxLimit; A_.block_max = bm; }
        A_.ideal_pos = ipos_.data(); A_.auto_avail = aavail_.data(); A_.auto_cnt = acnt_.data();
        A_.auto_cbase = acbase_.data(); A_.auto_child = achild_.data(); A_.auto_nblk = anblk_.data(); A_.auto_depth = adepth_.data();
        A_.units = units_.data(); A_.seg_cn = hb_.seg_cn.data(); A_.junc_cn = hb_.junc_cn.data(); A_.junc_ends = hb_.junc_ends.data(); A_.elems = hb_.elems.data();
        A_.dags = dags_.data(); A_.results = results_.data();
        A_.ideal_keys = ikeys_.data(); A_.ideal_cnt = icnt_.data(); A_.ideal_link = ilink_.data();
        A_.ideal_lvl_off = ilvl_off_.data(); A_.ideal_counter = icounter_.data();
        A_.first_rows = first_rows_.empty() ? nullptr : first_rows_.data();
        A_.order_arena = arena_.data(); A_.order_arena_bytes = (int64_t)arena_.size();
        A_.blk_off = blk_off_.data(); A_.rows_per_lane = rows_per_lane_.data();
        A_.n_pending = &n_pending_; A_.orders_needed = &orders_needed_;
        A_.scratch_i32 = scratch_.data(); A_.scratch_off = hb_.scratch_off.data(); A_.stage_clk = nullptr;
        refin_list_.assign(units_.size() + 1, 0); refin_count_ = 0;
        A_.refin_list = refin_list_.data(); A_.refin_count = &refin_count_; A_.direct_full_on = 0; A_.finish_retry = 0;
        { const char* ec = ambi_env("AMBI_EDIT_RUN_CAP"); A_.edit_cap_limit = ec ? atoi(ec) : 0; }
        A_.wide = wide_.empty() ? nullptr : wide_.data(); A_.wide_index = wide_.empty() ? nullptr : hb_.wide_index.data();
        A_.inject_valid = hb_.inject.empty() ? nullptr : hb_.inject.data();
        A_.inject_off = hb_.inject.empty() ? nullptr : hb_.inject_off.data();
        {
            const size_t U = units_.size(); const int64_t tot = hb_.run_slot.empty() ? 0 : hb_.run_slot.back();
            A_.run_cnt = run_blk_.data(); A_.run_cells = run_blk_.data() + U; A_.run_start = run_blk_.data() + 2 * U; A_.run_len = run_blk_.data() + 2 * U + tot;
            A_.run_slot = hb_.run_slot.data();
        }
    }

    void enumerate_all() {
        HostGroup g;
        const int64_t total = blk_off_[units_.size()];
        std::vector<uint8_t> stacks((size_t)enum_stack_bytes(64));
        const char* env = ambi_env("AMBI_BLOCK_LDS");
        const int64_t block_lds = env ? atoll(env) : cfg_.block_lds;
        const int block_max = A_.block_max;
        std::vector<uint8_t> image((size_t)(block_lds > 64 ? block_lds : 64));
        std::vector<uint8_t> bscratch((size_t)cfg_.block_scratch_lds);
        int built_unit = -1;
        BlockImageHeader H{};
        bool fast = false, dfs = false;
        BuildTables Bt{};
        std::vector<uint16_t> dfs_stack(64);
        std::vector<uint32_t> dfs_pw(20);
        for (int64_t b = 0; b < total; b++) {
            int lo = 0, hi = (int)units_.size();
            while (hi - lo > 1) { int mid = (lo + hi) / 2; if (blk_off_[mid] <= b) lo = mid; else hi = mid; }
            const int u = lo;
            UnitOut* out = unit_out(A_.results, u);
            const int K = out->K, T = rows_per_lane_[u];
            const int64_t R = out->num_orders, base_rank = (b - blk_off_[u]) * 256ll * T;
            IdealTable tbl = unit_ideal_table(A_, u);
            AutoView V = auto_view(tbl);
            uint8_t* rows = A_.order_arena + out->order_off;
            if (K > kMaxNodes) {   // ambi_enumerate_wide_kernel: every row unranked from the wide unit's counts (once per unit)
                if (b == blk_off_[u]) {
                    const WideUnit& X = A_.wide[A_.wide_index[u]];
                    const int stride = row_stride(K);
                    for (int64_t r = 0; r < R; r++) {
                        uint8_t row[kWideNodeCap];
                        unrank_wide(X, (uint64_t)r, row);
                        for (int d = K; d < stride; d++) row[d] = 0xFF;
                        memcpy(rows + r * stride, row, (size_t)stride);
                    }
                }
                continue;
            }
            if (A_.first_rows && R <= A_.first_budget) {   // as ambi_enumerate_blocks_kernel: the table is a copy of the first rows
                copy_first_rows(g, A_.first_rows + (int64_t)u * A_.first_budget * kFirstRowStride, K, R, rows);
                continue;
            }
            if (u != built_unit) {   // ambi_blocks_build_kernel: once per unit
                fast = build_block_image(g, tbl, K, row_stride(K) / 4, R, block_max, bscratch.data(), (int64_t)bscratch.size(),
                                         image.data(), block_lds, H);
                dfs = false;
                if (ambi_env("AMBI_HOSTSIM_TRACE")) { BuildTables d2; fprintf(stderr, "hostsim: unit %d K=%d R=%lld nI=%d nC=%d nB=%d suf_words=%d image_bytes=%d scratch=%lld\n", u, K, (long long)R, H.nI, H.nC, H.nB, H.suf_words, H.image_bytes, (long long)carve_build_tables(bscratch.data(), H.nI, H.nC, d2)); }
                if (!fast) {   // ambi_blocks_build_kernel's second form: tables + suffix rows, walked at emission
                    BuildTables dummy;
                    const int64_t scr = carve_build_tables(bscratch.data(), tbl.counter[0], tbl.counter[1], dummy);
                    const int64_t budget = block_lds - kDfsStateBytes - scr;
                    for (int bm = block_max; bm >= 8 && !fast && budget > 0 && scr <= (int64_t)bscratch.size(); bm >>= 1)
                        fast = build_block_image(g, tbl, K, row_stride(K) / 4, R, bm, bscratch.data(), (int64_t)bscratch.size(),
                                                 image.data(), budget, H, nullptr, false);
                    if (fast) { dfs = true; (void)carve_build_tables(bscratch.data(), tbl.counter[0], tbl.counter[1], Bt); }
                    if (ambi_env("AMBI_HOSTSIM_TRACE")) fprintf(stderr, "hostsim: unit %d (K=%d, R=%lld): %s\n", u, K, (long long)R, fast ? "directory-free block walk" : "general path");
                }
                built_unit = u;
            }
            for (int w = 0; w < 4; w++) {
                const int64_t wlo = base_rank + (int64_t)w * 64 * T;
                int64_t whi = wlo + 64ll * T;
                if (whi > R) whi = R;
                if (wlo >= R) break;
                if (fast && dfs) {
                    emit_blocks_dfs_dispatch<-1>(Bt, reinterpret_cast<const uint32_t*>(image.data()), K, H.block_max, (uint32_t)wlo, (uint32_t)whi, rows,
                                                 dfs_stack.data(), dfs_pw.data(), 0, 64);
                } else if (fast) {
                    emit_blocks_dispatch<-1>(reinterpret_cast<const uint32_t*>(image.data()), H.nB, K, (uint32_t)wlo, (uint32_t)whi, rows, 0, 64);
                } else {
                    GlobalAuto ga{V};
                    for (int lane = 0; lane < 64; lane++)
                        enumerate_lane_dispatch<-1>(ga, ga, V, K, R, wlo + (int64_t)lane * T, T, stacks.data(), lane, 64, rows);
                }
            }
        }
    }

    // slow path of the first-valid search (LGM.cpp:3519-3696): the engine's own chunked search (stage_search_chunk /
    // stage_resolve), the chunks taken in the order AMBI_HOSTSIM_SEARCH_ORDER asks for -- "asc" (default), "desc" (a late
    // chunk always reports before an early one) or "shuffle" -- which is what concurrency on the GPU amounts to.
    void search_pending() {
        HostGroup g;
        const char* ord = ambi_env("AMBI_HOSTSIM_SEARCH_ORDER");
        const int mode = !ord ? 0 : (!strcmp(ord, "desc") ? 1 : (!strcmp(ord, "shuffle") ? 2 : 0));
        const int chunk = 16;
        for (size_t u = 0; u < units_.size(); u++) {
            UnitOut* out = unit_out(A_.results, (int)u);
            if (out->status != ST_PENDING) continue;
            if (out->order_off < 0) continue;   // no table to search (the plan stage had no room): the finish stage behind turns this into ORDERS_CAPACITY, as on the device, where the finish kernels run before the search
            const UnitIn& U = units_[u];
            const bool wide = U.n_elem > kMaxNodes;
            std::vector<uint8_t> work((size_t)first_work_bytes(U.n_seg, U.bkp_cap, wide));
            FirstWork W = carve_first(work.data(), U.n_seg, U.bkp_cap, wide);
            load_first_work(g, A_, (int)u, W);
            const int64_t R = out->num_orders, nchunks = (R + chunk - 1) / chunk;
            std::vector<int64_t> order((size_t)nchunks);
            for (int64_t c = 0; c < nchunks; c++) order[(size_t)c] = mode == 1 ? nchunks - 1 - c : c;
            if (mode == 2) { uint64_t x = 88172645463325252ull + u; for (int64_t c = nchunks - 1; c > 0; c--) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; std::swap(order[(size_t)c], order[(size_t)(x % (uint64_t)(c + 1))]); } }
            bool fwd = !(A_.flags & FLAG_REVERSED);
            for (int pass = 0; pass < 2 && out->status == ST_PENDING; pass++) {
                SearchSlot slot{kSearchNone, kSearchNone};
                for (int64_t c : order) {
                    if (c * chunk >= search_limit(slot.found, slot.err_key)) continue;   // as ambi_search_kernel
                    stage_search_chunk(g, A_, (int)u, W, c * chunk, chunk, fwd, &slot);
                }
                std::vector<uint8_t> work2((size_t)first_work_bytes(U.n_seg, U.bkp_cap, wide));
                stage_resolve(g, A_, (int)u, work2.data(), &slot, fwd, pass);
                fwd = !fwd;
            }
        }
    }

    int run(uint32_t flags, void*) override {
        HostGroup g;
        bind(flags);
        n_pending_ = 0;
        const int Un = (int)units_.size();
        // small batches: the express stage (whole reconstruction of units whose first order assembles) in front, the lattice
        // stage behind it, as the HIP backend launches them (env AMBI_EXPRESS_UNITS, default 32; 0 = never)
        const char* ex = ambi_env("AMBI_EXPRESS_UNITS");
        const int express_units = ex ? atoi(ex) : 32;
        const bool express = Un <= express_units && hb_.n_wide == 0;   // (as HipBackend::run)
        if (express)
            for (int u = 0; u < Un; u++) {
                const UnitIn& U = units_[u];
                std::vector<uint8_t> work((size_t)express_work_bytes(U.n_seg, U.n_junc, U.n_elem, U.bkp_cap, lds_path_cap(A_, U.path_cap), U.out_cap) + 64);
                stage_express(g, g, -1, A_, u, work.data());
            }
        for (int attempt = 0; attempt < 2; attempt++) {
            for (int u = 0; u < Un; u++) {
                if (express) {
                    std::vector<uint8_t> work((size_t)(64 * 8 + kPrepLatticeBytes + 64));
                    stage_lattice(g, A_, u, work.data());
                    continue;
                }
                std::vector<uint8_t> work((size_t)prepare_work_bytes(units_[u].n_seg, units_[u].n_junc, units_[u].n_elem));
                stage_prepare(g, A_, u, work.data());
            }
            plan_serial(A_);
            if (orders_needed_ <= (int64_t)arena_.size()) break;
            int64_t want = orders_needed_;
            { const char* cap = ambi_env("AMBI_ARENA_MAX_BYTES"); if (cap && atoll(cap) > 0 && want > atoll(cap)) want = atoll(cap); }   // as HipBackend::run
            if (want <= (int64_t)arena_.size()) break;   // cannot grow further: the units beyond it end with ORDERS_CAPACITY
            arena_.assign((size_t)want, 0);              // grow the arena and redo (first run only)
            bind(flags);
            if (express)   // the lattice stage does not rewrite the header: take back what the plan pass decided against the small arena
                for (int u = 0; u < Un; u++) {
                    UnitOut* o = unit_out(A_.results, u);
                    if (o->order_off != kOrderOffNone && o->num_orders < (int64_t)kCountSat) o->order_off = kOrderOffWanted;   // offsets and refusals alike
                }
        }
        enumerate_all();
        for (int u = 0; u < Un; u++) {
            std::vector<uint8_t> work((size_t)first_work_bytes(units_[u].n_seg, units_[u].bkp_cap));
            stage_first(g, A_, u, work.data());
        }
        if (n_pending_ > 0) search_pending();
        // lean finish stage first, the full one for the units it hands over (AMBI_HOSTSIM_LEAN_FINISH=0: full stage only)
        const char* lf = ambi_env("AMBI_HOSTSIM_LEAN_FINISH");
        const bool lean = lf ? atoi(lf) != 0 : true;
        for (int u = 0; u < Un && lean; u++) {
            const UnitIn& U = units_[u];
            std::vector<uint8_t> work((size_t)finish_lean_work_bytes(U.n_seg, U.n_junc, U.bkp_cap));
            stage_finish_lean(g, A_, u, work.data());
        }
        for (int u = 0; u < Un; u++) {
            const UnitIn& U = units_[u];
            if (unit_out(A_.results, u)->reserved) continue;   // done by the express stage
            if (lean && unit_out(A_.results, u)->status != ST_REFINISH) continue;
            // units with deletion / duplication candidates: the form with the path cells outside the work area (what the HIP
            // backend's direct full-stage launch runs), in buffers of exactly its sizes; AMBI_HOSTSIM_EXT_PATH=0: the ordinary form
            const char* ep = ambi_env("AMBI_HOSTSIM_EXT_PATH");
            const char* ed = ambi_env("AMBI_HOSTSIM_EDIT");
            if (U.direct_full && !(ed && atoi(ed) == 0)) {   // what the HIP backend's direct launch runs first: the edits on the runs of the path
                std::vector<uint8_t> work((size_t)finish_edit_work_bytes(U.n_seg, U.n_junc, U.bkp_cap));
                stage_finish_edit(g, A_, u, work.data());
                const bool handed_on = unit_out(A_.results, u)->status == ST_REFINISH;
                if (ambi_env("AMBI_HOSTSIM_DEBUG")) fprintf(stderr, "hostsim: unit %d through the edit stage: %s\n", u, handed_on ? "handed on" : "done");
#if defined(AMBI_EDIT_TRACE_ON)
                fprintf(stderr, "edit counts unit %d: table builds %ld sweeps %ld chain scans %ld (runs %d)\n", u, g_edit_count[0], g_edit_count[1], g_edit_count[2], 0); g_edit_count[0] = g_edit_count[1] = g_edit_count[2] = 0;
#endif
                if (!handed_on) continue;   // (handed on: the full stage below)
            }
            if (U.direct_full && !(ep && atoi(ep) == 0)) {
                std::vector<uint8_t> work((size_t)finish_work_bytes(U.n_seg, U.n_junc, U.bkp_cap, 0, U.out_cap));
                std::vector<cell_t> cells((size_t)U.path_cap + 8 + 8);   // 2 * path_cap + 16 bytes, aligned to 16 below
                cell_t* base = cells.data();
                while (reinterpret_cast<uintptr_t>(base) & 15) base++;
                stage_finish<true>(g, A_, u, work.data(), base);
                continue;
            }
            std::vector<uint8_t> work((size_t)finish_work_bytes(U.n_seg, U.n_junc, U.bkp_cap, lds_path_cap(A_, U.path_cap), U.out_cap));
            stage_finish(g, A_, u, work.data());
        }
        if (flags & FLAG_ALL) compute_all();
        return 0;
    }
    // --all: the engine's fused unrank + evaluate stage (stage_all_chunk), one 64-order chunk after the other; bitmaps,
    // counts and flags as the HIP backend keeps them
    std::vector<std::vector<int64_t>> all_idx_[2];
    std::vector<uint64_t> all_bits_; std::vector<int64_t> all_off_; std::vector<int32_t> all_count_;
    int64_t all_pool_bytes_ = 0; int all_rank_ = 0, all_world_ = 1;
    int set_shard(int rank, int world) override { if (world < 1 || rank < 0 || rank >= world) return ST_ERR_BAD_INPUT; all_rank_ = rank; all_world_ = world; return 0; }
    int all_device(void** ptr, int64_t* bytes) override { if (ptr) *ptr = all_bits_.empty() ? nullptr : all_bits_.data(); if (bytes) *bytes = all_bits_.empty() ? 0 : all_pool_bytes_; return 0; }
    int all_finish() override { if (!all_bits_.empty()) finalize_all(); return 0; }
    void compute_all() {
        HostGroup g;
        const int Un = (int)units_.size();
        all_idx_[0].assign(Un, {}); all_idx_[1].assign(Un, {});
        all_off_.assign(Un + 1, 0);
        for (int u = 0; u < Un; u++) {
            const UnitOut* out = unit_out(A_.results, u);
            const bool live = out->status == ST_OK && out->num_orders > 0 && out->num_orders < (int64_t)kCountSat;
            all_off_[u + 1] = all_off_[u] + (live ? 2 * all_words(out->num_orders) : 0);
        }
        // one pool as in the HIP backend: [bitmaps][flags]
        const int64_t words = all_off_[Un];
        all_bits_.assign((size_t)(words + (Un + 1) / 2 + 1), 0); all_count_.assign(2 * (size_t)Un, 0);
        all_pool_bytes_ = (words + (Un + 1) / 2) * 8;
        A_.all_bits = all_bits_.data(); A_.all_off = all_off_.data(); A_.all_count = all_count_.data();
        A_.all_flags = reinterpret_cast<int32_t*>(all_bits_.data() + words);
        A_.all_rank = all_rank_; A_.all_world = all_world_; A_.all_rows_from_table = 0;
        for (int pass = 0; pass < 2; pass++) {
            for (int u = 0; u < Un; u++) {
                if (all_off_[u + 1] == all_off_[u]) continue;
                const UnitIn& U = units_[u];
                const int64_t R = unit_out(A_.results, u)->num_orders;
                if (pass == 1 && all_pass0_last_valid(A_, u, R)) continue;
                const bool wide = U.n_elem > kMaxNodes;
                std::vector<uint8_t> work((size_t)first_work_bytes(U.n_seg, U.bkp_cap, wide)), rows(64 * kFirstRowStride);
                FirstWork W = carve_first(work.data(), U.n_seg, U.bkp_cap, wide);
                load_first_work(g, A_, u, W);
                // as the HIP backend: one thread per order for units with a short breakpoint path, the wavefront form otherwise
                const char* el = ambi_env("AMBI_ALL_LANES");
                const bool lanes = !(el && atoi(el) == 0) && U.bkp_cap <= kAllLaneMaxCells && !wide;
                std::vector<cell_t> cells(lanes ? (size_t)U.bkp_cap * 64 : 1);
                for (int64_t c = 0; c < all_words(R); c++) {
                    if (!all_chunk_is_mine(A_, all_off_[u] / 2 + c, c, R)) continue;   // another rank's chunk
                    if (lanes) stage_all_chunk_lanes(g, A_, u, W, rows.data(), cells.data(), c, pass);
                    else stage_all_chunk(g, A_, u, W, rows.data(), c, pass);
                }
            }
        }
        finalize_all();
    }
    void finalize_all() {
        const int Un = (int)units_.size();
        all_idx_[0].assign(Un, {}); all_idx_[1].assign(Un, {});
        for (int u = 0; u < Un; u++) {
            all_finalize_unit(A_, u);
            if (all_off_[u + 1] == all_off_[u]) continue;
            const int64_t nw = all_words(unit_out(A_.results, u)->num_orders);
            for (int ps = 0; ps < 2; ps++)
                for (int64_t w = 0; w < nw; w++) {
                    uint64_t x = all_bits_[(size_t)(all_off_[u] + ps * nw + w)];
                    while (x) { all_idx_[ps][u].push_back(w * 64 + __builtin_ctzll(x)); x &= x - 1; }
                }
        }
    }
    int all_count(int unit, int pass, int64_t* count) override {
        if (pass < 0 || pass > 1 || unit < 0) return ST_ERR_BAD_INPUT;
        if (count) *count = (size_t)(2 * unit + pass) < all_count_.size() ? (int64_t)all_count_[2 * (size_t)unit + pass] : 0;
        return 0;
    }
    int all_orders(int unit, int pass, int64_t first, int64_t count, int64_t* idx) override {
        if (pass < 0 || pass > 1 || unit < 0 || unit >= (int)all_idx_[pass].size()) return ST_ERR_BAD_INPUT;
        const auto& v = all_idx_[pass][unit];
        if (first < 0 || count < 0 || first + count > (int64_t)v.size()) return ST_ERR_BAD_INPUT;
        for (int64_t i = 0; i < count; i++) idx[i] = v[first + i];
        return 0;
    }
    int all_paths(int unit, int pass, int64_t first, int64_t count, int32_t* lengths, int32_t* cells, int64_t stride) override {
        if (pass < 0 || pass > 1 || unit < 0 || unit >= (int)all_idx_[pass].size()) return ST_ERR_BAD_INPUT;
        const auto& v = all_idx_[pass][unit];
        if (first < 0 || count < 0 || first + count > (int64_t)v.size() || stride <= 0) return ST_ERR_BAD_INPUT;
        HostGroup g;
        const UnitIn& U = units_[unit];
        const bool wide = U.n_elem > kMaxNodes;
        std::vector<uint8_t> work((size_t)first_work_bytes(U.n_seg, U.bkp_cap, wide));
        std::vector<int32_t> offs((size_t)U.bkp_cap / 2 + 2);
        FirstWork W = carve_first(work.data(), U.n_seg, U.bkp_cap, wide);
        load_first_work(g, A_, unit, W);
        const bool fwd0 = !(A_.flags & FLAG_REVERSED), fwd = pass == 0 ? fwd0 : !fwd0;
        for (int64_t j = 0; j < count; j++) {
            int L = 0;
            const int ok = eval_indexed(g, A_, unit, W, v[first + j], fwd, &L);
            lengths[j] = ok == 1 ? expand_bkp(g, W.bkp, L, (cell_t*)nullptr, (int)(stride < U.path_cap ? stride : U.path_cap), offs.data(),
                                              cells + j * stride, U.seg_base)
                                 : -1;
        }
        return 0;
    }
    int wait() override { return 0; }
    int download(std::vector<uint8_t>& blob) override { blob = results_; return 0; }
    int device_results(void** ptr, int64_t* bytes) override {
        if (ptr) *ptr = results_.data();
        if (bytes) *bytes = (int64_t)results_.size();
        return 0;
    }
    int pack_paths(int which, int32_t* lengths, int32_t* cells, int64_t cap, int64_t* total, void*) override {
        int64_t off = 0;
        for (size_t u = 0; u < units_.size(); u++) {
            const UnitOut* h = unit_out(results_.data(), (int)u);
            const UnitIn& U = units_[u];
            UnitLayout L = unit_layout(U.n_seg, U.bkp_cap, U.path_cap, U.out_cap);
            int len = which ? h->path_indel_len : h->path_len;
            const rcell_t* src = reinterpret_cast<const rcell_t*>(results_.data() + U.res_off + ((which && h->path_ind_stored) ? L.path_ind : L.path));
            lengths[u] = len;
            for (int i = 0; i < len && off + i < cap; i++) cells[off + i] = abs_cell(src[i], U.seg_base);
            off += len;
        }
        if (total) *total = off;
        return 0;
    }
    int pack_runs(int which, int32_t* lengths, int32_t* run_counts, int32_t* run_start, int32_t* run_len, int64_t cap, int64_t* totals,
                  void*) override {
        int64_t off = 0, cells = 0;
        for (size_t u = 0; u < units_.size(); u++) {
            const UnitOut* h = unit_out(results_.data(), (int)u);
            const UnitIn& U = units_[u];
            UnitLayout L = unit_layout(U.n_seg, U.bkp_cap, U.path_cap, U.out_cap);
            const int len = which ? h->path_indel_len : h->path_len;
            const rcell_t* src = reinterpret_cast<const rcell_t*>(results_.data() + U.res_off + ((which && h->path_ind_stored) ? L.path_ind : L.path));
            lengths[u] = len;
            int n = 0;
            for (int i = 0; i < len; i++) {
                if (i == 0 || src[i] != src[i - 1] + 1) { if (off + n < cap) { run_start[off + n] = abs_cell(src[i], U.seg_base); run_len[off + n] = 0; } n++; }
                if (off + n - 1 < cap) run_len[off + n - 1]++;
            }
            run_counts[u] = n;
            off += n; cells += len;
        }
        if (totals) { totals[0] = off; totals[1] = cells; }
        return 0;
    }
    int copy_orders(int unit, int64_t first, int64_t count, uint8_t* out) override {
        const UnitOut* h = unit_out(results_.data(), unit);
        if (h->order_off < 0 || first < 0 || first + count > h->num_orders) return ST_ERR_BAD_INPUT;
        const int stride = row_stride(h->K);
        for (int64_t r = 0; r < count; r++)
            for (int d = 0; d < h->K; d++) out[r * h->K + d] = (uint8_t)row_node(arena_.data() + h->order_off + (first + r) * stride, h->K, d);
        return 0;
    }
    int copy_dag(int unit, Dag* out) override { *out = dags_[unit]; return 0; }
    int copy_dag_wide(int unit, int32_t* pat, int32_t* loop, uint64_t* succ2) override {
        if (unit < 0 || unit >= (int)units_.size() || hb_.wide_index[unit] < 0) return ST_ERR_BAD_INPUT;
        const WideUnit& X = wide_[(size_t)hb_.wide_index[unit]];
        memcpy(pat, X.dag.pat, sizeof(X.dag.pat)); memcpy(loop, X.dag.loop, sizeof(X.dag.loop)); memcpy(succ2, X.succ, sizeof(X.succ));
        return 0;
    }
    int runs_to_host(int which, int, int, void*) override { return which == 1 ? 0 : ST_ERR_BAD_INPUT; }   // (the host simulation has the runs where the stages wrote them)
    int runs_wait(int, RunsView* out) override {
        const int64_t U = (int64_t)units_.size(), tot = hb_.run_slot.empty() ? 0 : hb_.run_slot.back();
        const int32_t* w = run_blk_.data();
        int64_t nr = 0, nc = 0;
        for (int64_t u = 0; u < U; u++) { if (w[u] < 0) return ST_ERR_BAD_INPUT; nr += w[u]; nc += w[U + u]; }
        *out = RunsView{nr, nc, w + U, w, w + 2 * U, w + 2 * U + tot, hb_.run_slot.data(), results_.data(), (2 * U + 2 * nr) * 4, (2 * U + 2 * tot) * 4};
        return 0;
    }
    void set_timing(bool) override {}
    const std::vector<KernelTime>& kernel_times() override { return times_; }
    int64_t order_bytes_written() const override { return orders_needed_; }
    size_t object_bytes() const override { return sizeof(HostSimBackend); }
};

Backend* make_backend() { return new HostSimBackend(); }

int backend_expand_runs(const int32_t* run_start, const int32_t* run_len, const int64_t* cell_off, int64_t n_runs, int32_t* cells,
                        int64_t cell_cap, void*) {
    for (int64_t r = 0; r < n_runs; r++)
        for (int k = 0; k < run_len[r]; k++) if (cell_off[r] + k < cell_cap) cells[cell_off[r] + k] = run_start[r] + k;
    return 0;
}

int backend_stream_probe(void*, void*, float* us) { if (us) *us = 0; return 0; }   // one thread: nothing runs side by side

int backend_ilp_fill(const IlpRowDesc* rows, int64_t n_rows, const int64_t* row_ptr, int s, int e, const int32_t* lit_col, const double* lit_val,
                     int64_t, int32_t* col, double* val, float* kernel_ms) {
    ilp_fill_rows(rows, row_ptr, 0, n_rows, 1, ilp_geom(s, e), lit_col, lit_val, 0, 1, col, val);
    if (kernel_ms) *kernel_ms = 0;
    return 0;
}

}  // namespace ambi
