// ambi_capi.cpp -- implementation of the C ABI declared in include/ambigram_hip.h on top of a Backend.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/ambigram_hip.h"
#include "ambi_backend.hpp"
#include "ambi_pack.hpp"
#include "ambi_wide.hpp"
#include "lh_graph.hpp"

using namespace ambi;

struct ambi_graph { LhGraph g; };

// One device's share of a batch (ambi_batch_run_sharded): its own packed inputs and backend, RESIDENT on its device from the first
// call on, and a host thread of its own that lives as long as the share (parked between calls).  What comes back from the device
// after a run is what a caller of many units wants first -- every unit's header and its final path in run-length form, in pinned
// memory (Backend::runs_to_host) -- the whole result blob of a share only when a getter asks for something else.
struct Shard {
    HostBatch hb;
    std::unique_ptr<Backend> be;
    std::vector<uint8_t> blob; bool blob_there = false;    // downloaded on demand
    RunsView view{}; bool view_there = false;
    std::vector<int> units;     // local unit -> unit of the batch
    int device = 0, rc = 0;
    bool uploaded = false;
    // the worker: go / done are counted, flags / cfg belong to the call in flight
    std::thread th; std::mutex mu; std::condition_variable cv;
    std::atomic<long> go{0}, done{0}; std::atomic<bool> quit{false}; uint32_t flags = 0; const EngineConfig* cfg = nullptr;
    // (both sides spin for a moment before they block: a parked thread takes 50-100 us to wake, a call takes ~1 ms)
    template <class F> static bool spin_for(F&& cond, int us) {
        const auto t0 = std::chrono::steady_clock::now();
        while (!cond()) { if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(us)) return false; __builtin_ia32_pause(); }
        return true;
    }
    void work() {
        rc = 0; view_there = false; blob_there = false;
        if (units.empty()) return;
        if ((rc = be->set_device(device))) return;
        if (!uploaded) { if ((rc = be->upload(hb, *cfg))) return; uploaded = true; }
        void* st = be->own_stream();
        if ((rc = be->run(flags, st))) return;
        // headers + run-length final paths follow the run's kernels to the host; ONE blocking wait (for that copy: the kernels are
        // complete when it is), then wait() on an idle stream for what the host still has to look at (the first run of a batch done
        // again with a larger arena, units finished by the parallel search, --all) -- if that changed results, their copy once more
        for (int attempt = 0; attempt < 3; attempt++) {
            const int64_t e0 = be->results_epoch();
            if ((rc = be->runs_to_host(1, 0, 1, st))) return;
            if ((rc = be->runs_wait(0, &view))) return;
            if ((rc = be->wait())) return;
            if (be->results_epoch() == e0) break;
        }
        view_there = view.headers != nullptr;
        if (!view_there) { rc = be->download(blob); blob_there = rc == 0; }   // (a backend without the pinned view: the blob at once)
    }
    void loop() {
        for (;;) {
            if (!spin_for([&] { return quit || go.load() > done.load(); }, 2000)) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return quit || go.load() > done.load(); });
            }
            if (quit) return;
            work();
            { std::lock_guard<std::mutex> lk(mu); done++; }
            cv.notify_all();
        }
    }
    void start(uint32_t f, const EngineConfig* c) {
        { std::lock_guard<std::mutex> lk(mu); flags = f; cfg = c; go++; }
        if (!th.joinable()) th = std::thread([this] { loop(); });
        cv.notify_all();
    }
    void finish() {
        if (spin_for([&] { return done.load() >= go.load(); }, 5000)) return;
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done.load() >= go.load(); });
    }
    int fetch_blob() {   // the share's whole result blob, once per run
        if (blob_there) return 0;
        const int r = be->download(blob);   // (the backend selects its own device for the call)
        blob_there = r == 0;
        return r;
    }
    ~Shard() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

struct ambi_batch {
    HostBatch hb;
    EngineConfig cfg;
    std::unique_ptr<Backend> be;
    std::vector<std::unique_ptr<Shard>> shards;        // empty unless ambi_batch_run_sharded built them
    std::vector<std::pair<int, int>> where;             // unit -> (shard, local unit)
    // the backend that holds a unit's device-side state (DAG, order table, --all bitmaps) and the unit's index there
    Backend* owner(int unit, int* local) const {
        if (shards.empty()) { *local = unit; return be.get(); }
        *local = where[unit].second;
        return shards[where[unit].first]->be.get();
    }
    std::vector<uint8_t> blob;
    RunsView runs[2] = {};                              // ambi_batch_runs_wait: what arrived in the slot
    bool uploaded = false, downloaded = false;
    bool sharded_ready = false;   // ambi_batch_run_sharded has run: results are read where the shares left them (headers + run-length paths; a share's blob on demand)
    bool mail_view = false;   // header / final paths / output junctions are read from the backend's pinned mailbox (ambi_batch_fetch_paths)
};

static int64_t copy_text(const std::string& s, char* buf, int64_t cap) {
    if (buf && cap > 0) {
        int64_t n = (int64_t)s.size() < cap - 1 ? (int64_t)s.size() : cap - 1;
        memcpy(buf, s.data(), (size_t)n);
        buf[n] = 0;
    }
    return (int64_t)s.size();
}

extern "C" {

const char* ambi_error_string(int code) {
    switch (code) {
        case 0: return "ok";
        case AMBI_ST_SHORTCUT: return "no fold-back inversion (reference path)";
        case AMBI_ST_INFEASIBLE: return "ILP is unsolvable";
        case AMBI_ST_NO_VALID_ORDER: return "no valid BFB order";
        case AMBI_ERR_TOO_MANY_NODES: return "more than 127 selected patterns/loops in one chromosome";
        case AMBI_ERR_NO_ELEMENTS: return "ILP solution selects no pattern or loop";
        case AMBI_ERR_REF_UB: return "reference behaviour undefined on this input (out-of-bounds read)";
        case AMBI_ERR_BKP_CAPACITY: return "breakpoint path capacity exceeded";
        case AMBI_ERR_PATH_CAPACITY: return "path capacity exceeded";
        case AMBI_ERR_ORDERS_CAPACITY: return "topological-order table does not fit the arena";
        case AMBI_ERR_IDEALS_CAPACITY: return "order-ideal table capacity exceeded";
        case AMBI_ERR_BAD_INPUT: return "bad unit input";
        case AMBI_ERR_OUTJUNC_CAPACITY: return "output junction capacity exceeded";
        case AMBI_ERR_NO_DEVICE: return "no HIP device available (libambigram_hip has no CPU fallback)";
        case AMBI_ERR_HIP: return "HIP runtime error";
        case AMBI_ERR_STATE: return "call order violated";
        case AMBI_ERR_ARG: return "bad argument";
        default: return lh_error_string(code);
    }
}

int ambi_abi_version(void) { return AMBI_ABI_VERSION; }

const char* ambi_backend_name(void) {
    static std::string nm;
    if (nm.empty()) { std::unique_ptr<Backend> b(make_backend()); nm = b->name(); }
    return nm.c_str();
}
int ambi_device_count(int* count) { std::unique_ptr<Backend> b(make_backend()); return b->device_count(count); }
int ambi_set_device(int device) {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);   // (only honoured when this is the process's first HIP call: see INTEGRATION.md)
    std::unique_ptr<Backend> b(make_backend());
    return b->set_device(device);
}

int ambi_debug_stream_probe(void* stream_a, void* stream_b, float* us) { return us ? backend_stream_probe(stream_a, stream_b, us) : AMBI_ERR_ARG; }

// ---- graph ----
int ambi_graph_read_lh(const char* lh_path, ambi_graph_t** out) {
    if (!lh_path || !out) return AMBI_ERR_ARG;
    std::unique_ptr<ambi_graph> g(new ambi_graph());
    int rc = read_lh(lh_path, g->g);
    if (rc != LH_OK) return rc;
    *out = g.release();
    return 0;
}
void ambi_graph_destroy(ambi_graph_t* g) { delete g; }
int ambi_graph_sizes(const ambi_graph_t* g, int32_t* n_seg, int32_t* n_junc, int32_t* n_chr) {
    if (!g) return AMBI_ERR_ARG;
    if (n_seg) *n_seg = g->g.n_seg();
    if (n_junc) *n_junc = g->g.n_junc();
    if (n_chr) *n_chr = g->g.n_chr();
    return 0;
}
int ambi_graph_segments(const ambi_graph_t* g, int32_t* id, int32_t* chr_id, int32_t* start, int32_t* end, double* cov, double* cn) {
    if (!g) return AMBI_ERR_ARG;
    const LhGraph& G = g->g;
    for (int i = 0; i < G.n_seg(); i++) {
        if (id) id[i] = G.seg_id[i];
        if (chr_id) chr_id[i] = G.seg_chr[i];
        if (start) start[i] = G.seg_start[i];
        if (end) end[i] = G.seg_end[i];
        if (cov) cov[i] = G.seg_cov[i];
        if (cn) cn[i] = G.seg_cn[i];
    }
    return 0;
}
int ambi_graph_junctions(const ambi_graph_t* g, int32_t* src, int8_t* sdir, int32_t* tgt, int8_t* tdir, double* cov,
                         double* cn, uint8_t* inferred, uint8_t* bounded) {
    if (!g) return AMBI_ERR_ARG;
    const LhGraph& G = g->g;
    for (int i = 0; i < G.n_junc(); i++) {
        if (src) src[i] = G.j_src[i];
        if (sdir) sdir[i] = G.j_sdir[i];
        if (tgt) tgt[i] = G.j_tgt[i];
        if (tdir) tdir[i] = G.j_tdir[i];
        if (cov) cov[i] = G.j_cov[i];
        if (cn) cn[i] = G.j_cn[i];
        if (inferred) inferred[i] = G.j_inferred[i];
        if (bounded) bounded[i] = G.j_bounded[i];
    }
    return 0;
}
int ambi_graph_chromosome(const ambi_graph_t* g, int32_t chr, int32_t* source_id, int32_t* sink_id) {
    if (!g || chr < 0 || chr >= g->g.n_chr()) return AMBI_ERR_ARG;
    if (source_id) *source_id = g->g.source_ids[chr];
    if (sink_id) *sink_id = g->g.sink_ids[chr];
    return 0;
}
int ambi_graph_read_juncs(ambi_graph_t* g, const char* juncs_path) {
    if (!g) return AMBI_ERR_ARG;
    return read_juncs(g->g, juncs_path ? juncs_path : "");
}
int64_t ambi_graph_log(const ambi_graph_t* g, char* buf, int64_t cap) {
    if (!g) return AMBI_ERR_ARG;
    std::string s;
    for (auto& l : g->g.log) { s += l; s += '\n'; }
    return copy_text(s, buf, cap);
}
int ambi_graph_props(const ambi_graph_t* g, int32_t* ins_mode, int32_t* con_mode, char* main_chr, int64_t cap) {
    if (!g) return AMBI_ERR_ARG;
    if (ins_mode) *ins_mode = g->g.ins_mode;
    if (con_mode) *con_mode = g->g.con_mode;
    copy_text(g->g.main_chr, main_chr, cap);
    return 0;
}
int ambi_graph_trx_before(const ambi_graph_t* g, int32_t* original_of, int32_t cap) {
    if (!g) return AMBI_ERR_ARG;
    if (!g->g.trx) return 0;
    const auto& m = g->g.trx->original_of;
    if (original_of) for (size_t i = 0; i < m.size() && (int)i < cap; i++) original_of[i] = m[i];
    return (int)m.size();
}
int ambi_graph_trx_original(const ambi_graph_t* g, ambi_graph_t** out) {
    if (!g || !out) return AMBI_ERR_ARG;
    if (!g->g.trx) return AMBI_ERR_STATE;
    ambi_graph* o = new ambi_graph();
    o->g = g->g.trx->original;
    *out = o;
    return 0;
}
int ambi_graph_trx_restore(const ambi_graph_t* g, int32_t* path, int32_t len, int32_t cap, char* text, int64_t text_cap, int64_t* text_len) {
    if (!g || !path || len < 0) return AMBI_ERR_ARG;
    std::vector<int32_t> p(path, path + len);
    std::vector<std::string> lines;
    int rc = trx_restore_path(g->g, p, lines);
    if (rc != LH_OK) return rc;
    std::string s;
    for (auto& l : lines) { s += l; s += '\n'; }
    const int64_t n = copy_text(s, text, text_cap);
    if (text_len) *text_len = n;
    for (size_t i = 0; i < p.size() && (int)i < cap; i++) path[i] = p[i];
    return (int)p.size();
}
int ambi_graph_write_lh(ambi_graph_t* g, const char* lh_path) {
    if (!g || !lh_path) return AMBI_ERR_ARG;
    return write_lh(g->g, lh_path);
}
int ambi_graph_recalculate(ambi_graph_t* g) {
    if (!g) return AMBI_ERR_ARG;
    int rc = hap_depth(g->g);
    if (rc != LH_OK) return rc;
    copy_num(g->g);
    return 0;
}
int ambi_graph_components(const ambi_graph_t* g, int32_t* ids, int32_t ids_cap, int32_t* offsets, int32_t off_cap) {
    if (!g) return AMBI_ERR_ARG;
    int32_t at = 0, c = 0;
    for (auto& comp : g->g.components) {
        if (offsets && c < off_cap) offsets[c] = at;
        for (int32_t v : comp) { if (ids && at < ids_cap) ids[at] = v; at++; }
        c++;
    }
    if (offsets && c < off_cap) offsets[c] = at;
    return c;
}

// ---- batch ----
int ambi_batch_create(ambi_batch_t** out) {
    if (!out) return AMBI_ERR_ARG;
    ambi_batch* b = new ambi_batch();
    b->be.reset(make_backend());
    *out = b;
    if (ambi_env("AMBI_DEBUG_SIZES")) {   // diagnostics: heap objects the engine creates and frees per batch
        static bool once = false;
        if (!once) fprintf(stderr, "ambigram sizes: ambi_batch %zu, HostBatch %zu, backend %zu, BatchArgs (kernel argument block) %zu, UnitIn %zu, UnitOut %zu, Dag %zu\n",
                           sizeof(ambi_batch), sizeof(HostBatch), b->be->object_bytes(), sizeof(BatchArgs), sizeof(UnitIn), sizeof(UnitOut), sizeof(Dag));
        once = true;
    }
    return 0;
}
// AMBI_DEBUG_QUARANTINE=1 (diagnostics): destroyed batches are not returned to the allocator but filled with a pattern and
// checked at every later destroy -- a write into a destroyed batch or backend object shows up with its offset.
extern "C++" {
namespace {
struct Quarantined { unsigned char* p; size_t n; const char* what; };
std::vector<Quarantined>& quarantine() { static std::vector<Quarantined> q; return q; }
void check_quarantine() {
    for (auto& q : quarantine())
        for (size_t i = 0; i < q.n; i++)
            if (q.p[i] != 0xAB) { fprintf(stderr, "ambigram_hip QUARANTINE: destroyed %s object %p modified at offset %zu: 0x%02x\n", q.what, (void*)q.p, i, q.p[i]); q.p[i] = 0xAB; }
}
}  // namespace
}  // extern "C++"
void ambi_batch_destroy(ambi_batch_t* b) {
    static const bool quarantine_on = ambi_env("AMBI_DEBUG_QUARANTINE") != nullptr;
    if (!quarantine_on || !b) { delete b; return; }
    check_quarantine();
    Backend* raw = b->be.release();
    if (raw) {
        const size_t n = raw->object_bytes();
        raw->~Backend();
        memset((void*)raw, 0xAB, n);
        quarantine().push_back({reinterpret_cast<unsigned char*>(raw), n, "backend"});
    }
    b->~ambi_batch();
    memset((void*)b, 0xAB, sizeof(ambi_batch));
    quarantine().push_back({reinterpret_cast<unsigned char*>(b), sizeof(ambi_batch), "batch"});
}

int ambi_batch_add_chromosome(ambi_batch_t* b, const ambi_graph_t* g, int32_t chr, int32_t n_cols, const int32_t* col,
                              const int32_t* val, int32_t infeasible) {
    if (!b || !g || b->uploaded) return b && b->uploaded ? AMBI_ERR_STATE : AMBI_ERR_ARG;
    SolFile s;
    s.infeasible = infeasible != 0;
    for (int i = 0; i < n_cols; i++) { s.col.push_back(col[i]); s.val.push_back(val[i]); }
    return b->hb.add_graph_chr(g->g, chr, &s);
}
int ambi_batch_add_chromosome_sol(ambi_batch_t* b, const ambi_graph_t* g, int32_t chr, const char* sol_path) {
    if (!b || !g || !sol_path) return AMBI_ERR_ARG;
    if (b->uploaded) return AMBI_ERR_STATE;
    SolFile s;
    int rc = read_sol(sol_path, s);
    if (rc != LH_OK) return rc;
    return b->hb.add_graph_chr(g->g, chr, &s);
}
int ambi_batch_add_chromosome_sol_block(ambi_batch_t* b, const ambi_graph_t* g, int32_t chr, const char* sol_path, int32_t block, int32_t n_blocks) {
    if (!b || !g || !sol_path || n_blocks < 1 || block < 0 || block >= n_blocks) return AMBI_ERR_ARG;
    if (b->uploaded) return AMBI_ERR_STATE;
    SolFile s;
    int rc = read_sol(sol_path, s);
    if (rc != LH_OK) return rc;
    return b->hb.add_graph_chr(g->g, chr, &s, block, n_blocks);
}
int ambi_batch_add_unit(ambi_batch_t* b, int32_t n_seg, int32_t seg_base, const double* seg_cn, int32_t n_junc,
                        const int32_t* j_src, const int32_t* j_tgt, const int8_t* j_sdir, const int8_t* j_tdir,
                        const double* j_cn, int32_t n_elem, const int32_t* e_is_loop, const int32_t* e_a,
                        const int32_t* e_b, const int32_t* e_cn, int32_t infeasible, int32_t has_components) {
    if (!b) return AMBI_ERR_ARG;
    if (b->uploaded) return AMBI_ERR_STATE;
    return b->hb.add_unit(n_seg, seg_base, seg_cn, n_junc, j_src, j_tgt, j_sdir, j_tdir, j_cn, n_elem, e_is_loop, e_a, e_b,
                          e_cn, infeasible, has_components);
}
int ambi_batch_size(const ambi_batch_t* b, int32_t* n_units) {
    if (!b || !n_units) return AMBI_ERR_ARG;
    *n_units = (int32_t)b->hb.units.size();
    return 0;
}
int ambi_batch_configure(ambi_batch_t* b, int64_t order_arena_bytes, int32_t ideal_cap, int32_t first_budget, int32_t target_lanes) {
    if (!b) return AMBI_ERR_ARG;
    if (b->uploaded) return AMBI_ERR_STATE;
    if (order_arena_bytes >= 0) b->cfg.order_arena_bytes = order_arena_bytes;
    if (ideal_cap > 0) {
        int c = 2;
        while (c < ideal_cap) c <<= 1;
        b->hb.ideal_cap = c;
    }
    if (first_budget > 0) b->cfg.first_budget = first_budget;
    if (target_lanes >= 64) b->cfg.target_lanes = target_lanes;
    return 0;
}
int ambi_batch_debug_inject_validity(ambi_batch_t* b, int32_t unit, const int8_t* verdicts, int64_t count) {
    if (!b || unit < 0 || unit >= (int)b->hb.units.size() || count < 0 || (count > 0 && !verdicts)) return AMBI_ERR_ARG;
    if (b->uploaded) return AMBI_ERR_STATE;
    if (b->hb.inject_unit.size() < b->hb.units.size()) b->hb.inject_unit.resize(b->hb.units.size());
    b->hb.inject_unit[unit].assign(verdicts, verdicts + count);
    return 0;
}
int ambi_batch_upload(ambi_batch_t* b) {
    if (!b) return AMBI_ERR_ARG;
    if (b->hb.units.empty() || !b->shards.empty()) return AMBI_ERR_STATE;
    b->hb.finalize();
    int rc = b->be->upload(b->hb, b->cfg);
    if (rc == 0) b->uploaded = true;
    return rc;
}
int ambi_batch_run(ambi_batch_t* b, uint32_t flags, void* hip_stream) {
    if (!b) return AMBI_ERR_ARG;
    if (!b->uploaded) return AMBI_ERR_STATE;
    b->downloaded = false; b->mail_view = false;
    return b->be->run(flags, hip_stream);
}
int ambi_batch_wait(ambi_batch_t* b) { return b ? b->be->wait() : AMBI_ERR_ARG; }
int ambi_batch_wait_results(ambi_batch_t* b) { return b ? b->be->wait_results() : AMBI_ERR_ARG; }
int ambi_batch_download(ambi_batch_t* b) {
    if (!b) return AMBI_ERR_ARG;
    if (!b->uploaded) return AMBI_ERR_STATE;
    int rc = b->be->download(b->blob);
    if (rc == 0) { b->downloaded = true; b->mail_view = false; }
    return rc;
}
int ambi_batch_fetch_paths(ambi_batch_t* b) {
    if (!b) return AMBI_ERR_ARG;
    if (!b->uploaded) return AMBI_ERR_STATE;
    int rc = b->be->wait_results();
    if (rc) return rc;
    if (b->downloaded) return 0;
    if (b->be->mail_slot(0)) { b->mail_view = true; return 0; }
    return ambi_batch_download(b);
}
// Multi-GPU below Python (`ambi_bfb_reconstruct_batch(..., device_or_minus1_for_all)`; the units are the iterations
// of the loop localhap.cpp:111-265 and do not depend on each other): one host thread per device, the units dealt round-robin,
// every shard through its own backend (upload -> run -> download), the result blobs merged into the batch's on the host.
int ambi_batch_run_sharded(ambi_batch_t* b, uint32_t flags, const int32_t* devices, int32_t n_devices) {
    if (!b) return AMBI_ERR_ARG;
    if (b->hb.units.empty() || b->uploaded) return AMBI_ERR_STATE;   // (a batch is either uploaded to one device or sharded)
    std::vector<int> devs;
    if (n_devices > 0 && devices) devs.assign(devices, devices + n_devices);
    else {
        int n = 0;
        b->be->device_count(&n);
        if (n_devices > 0 && n_devices < n) n = n_devices;
        if (n <= 0) return AMBI_ERR_NO_DEVICE;
        for (int d = 0; d < n; d++) devs.push_back(d);
    }
    const int N = (int)devs.size(), U = (int)b->hb.units.size();
    // the shares of an earlier call are kept (inputs resident on their devices) only for the same devices AND the same units:
    // units added since then (ambi_batch_add_* stays open on a sharded batch) mean new shares
    bool same = (int)b->shards.size() == N && (int)b->where.size() == U;
    for (int k = 0; same && k < N; k++) same = b->shards[k]->device == devs[k];
    if (!same) {
        b->shards.clear(); b->where.assign(U, {0, 0});
        for (int k = 0; k < N; k++) {
            std::unique_ptr<Shard> s(new Shard());
            s->device = devs[k];
            s->hb.ideal_cap = b->hb.ideal_cap;
            s->be.reset(make_backend());
            b->shards.push_back(std::move(s));
        }
        for (int u = 0; u < U; u++) {
            Shard& s = *b->shards[u % N];
            int l = s.hb.add_unit_from(b->hb, u);
            if (l < 0) { b->shards.clear(); b->where.clear(); return l; }   // no half-dealt shares for the next call to find
            s.units.push_back(u);
            b->where[u] = {u % N, l};
        }
        for (auto& s : b->shards) s->hb.finalize();
    }
    b->hb.finalize();
    b->downloaded = false; b->mail_view = false; b->sharded_ready = false;
    // every share on its own (resident) thread, the first one too: set_device changes the current device of the thread that calls it,
    // and the caller's thread keeps the device it had.  Nothing is merged: the getters read a unit where its share left it.
    for (auto& s : b->shards) s->start(flags, &b->cfg);
    for (auto& s : b->shards) s->finish();
    for (auto& s : b->shards) if (s->rc) return s->rc;
    b->blob.clear();
    b->sharded_ready = true;
    b->downloaded = true;
    return 0;
}
int ambi_batch_device_results(ambi_batch_t* b, void** dev_ptr, int64_t* bytes) {
    if (!b || !b->uploaded) return b ? AMBI_ERR_STATE : AMBI_ERR_ARG;
    return b->be->device_results(dev_ptr, bytes);
}
int ambi_batch_pack_runs(ambi_batch_t* b, int32_t which, int32_t* dev_lengths, int32_t* dev_run_counts, int32_t* dev_run_start,
                         int32_t* dev_run_len, int64_t run_cap, int64_t* dev_totals, void* hip_stream) {
    if (!b || !b->uploaded) return b ? AMBI_ERR_STATE : AMBI_ERR_ARG;
    if (!dev_lengths || !dev_run_counts || !dev_run_start || !dev_run_len || run_cap < 0) return AMBI_ERR_ARG;
    return b->be->pack_runs(which, dev_lengths, dev_run_counts, dev_run_start, dev_run_len, run_cap, dev_totals, hip_stream);
}
int ambi_batch_runs_to_host(ambi_batch_t* b, int32_t which, int32_t slot, void* hip_stream) {
    if (!b || !b->uploaded) return b ? AMBI_ERR_STATE : AMBI_ERR_ARG;
    if (slot < 0 || slot > 1 || which < 0 || which > 1) return AMBI_ERR_ARG;
    b->runs[slot] = RunsView{};
    return b->be->runs_to_host(which, slot, 0, hip_stream);
}
int ambi_batch_runs_wait(ambi_batch_t* b, int32_t slot, ambi_runs_view_t* out) {
    if (!b || !b->uploaded) return b ? AMBI_ERR_STATE : AMBI_ERR_ARG;
    if (slot < 0 || slot > 1) return AMBI_ERR_ARG;
    RunsView v{};
    int rc = b->be->runs_wait(slot, &v);
    if (rc) return rc;
    b->runs[slot] = v;
    if (out) { out->n_runs = v.n_runs; out->n_cells = v.n_cells; out->bytes = v.bytes; out->copied_bytes = v.copied_bytes;
               out->lengths = v.lengths; out->run_counts = v.run_counts; out->run_start = v.run_start; out->run_len = v.run_len; out->run_off = v.run_off; }
    return 0;
}
int ambi_batch_runs_unit_path(ambi_batch_t* b, int32_t slot, int32_t unit, int32_t* out, int32_t cap) {
    if (!b || slot < 0 || slot > 1 || unit < 0 || unit >= (int)b->hb.units.size()) return AMBI_ERR_ARG;
    const RunsView& v = b->runs[slot];
    if (!v.lengths) return AMBI_ERR_STATE;
    int at = 0;
    for (int64_t r = v.run_off[unit]; r < v.run_off[unit] + v.run_counts[unit]; r++)
        for (int k = 0; k < v.run_len[r]; k++, at++) if (out && at < cap) out[at] = v.run_start[r] + k;
    return at;
}
int ambi_expand_runs(const int32_t* dev_run_start, const int32_t* dev_run_len, const int64_t* dev_cell_off, int64_t n_runs,
                     int32_t* dev_cells, int64_t cell_cap, void* hip_stream) {
    if (n_runs < 0 || cell_cap < 0 || (n_runs > 0 && (!dev_run_start || !dev_run_len || !dev_cell_off || !dev_cells))) return AMBI_ERR_ARG;
    if (n_runs == 0) return 0;
    return ambi::backend_expand_runs(dev_run_start, dev_run_len, dev_cell_off, n_runs, dev_cells, cell_cap, hip_stream);
}
int ambi_batch_pack_paths(ambi_batch_t* b, int32_t which, int32_t* dev_lengths, int32_t* dev_cells, int64_t cell_cap,
                          int64_t* dev_total_cells, void* hip_stream) {
    if (!b || !b->uploaded) return b ? AMBI_ERR_STATE : AMBI_ERR_ARG;
    return b->be->pack_paths(which, dev_lengths, dev_cells, cell_cap, dev_total_cells, hip_stream);
}

// Where a unit's results are: the batch's downloaded blob, or -- after ambi_batch_run_sharded -- its share's (the header from the pinned
// view that came back with the run, the blob itself fetched from the share's device the first time something else is asked for).
struct UnitRef { const UnitOut* h; const uint8_t* blob; const UnitIn* U; Shard* shard; int local; };
static bool unit_ref(const ambi_batch_t* b, int unit, UnitRef& r, bool need_blob) {
    if (!b || !b->downloaded || unit < 0 || unit >= (int)b->hb.units.size()) return false;
    if (!b->sharded_ready) {
        r = UnitRef{reinterpret_cast<const UnitOut*>(b->blob.data()) + unit, b->blob.data(), &b->hb.units[unit], nullptr, unit};
        return true;
    }
    Shard* S = b->shards[b->where[unit].first].get();
    const int l = b->where[unit].second;
    if ((need_blob || !S->view_there) && S->fetch_blob() != 0) return false;
    const UnitOut* h = S->view_there ? reinterpret_cast<const UnitOut*>(S->view.headers) + l : reinterpret_cast<const UnitOut*>(S->blob.data()) + l;
    r = UnitRef{h, S->blob_there ? S->blob.data() : nullptr, &S->hb.units[l], S, l};
    return true;
}
static const UnitOut* header(const ambi_batch_t* b, int unit) {
    UnitRef r;
    return unit_ref(b, unit, r, false) ? r.h : nullptr;
}
// The parts of a unit's results that ambi_batch_fetch_paths makes available: from the downloaded blob, or straight from the
// backend's pinned mailbox (MailLayout) when the batch took the express path.
struct PathView { const UnitOut* h; const rcell_t* path; const rcell_t* path_ind; const OutJunc* out; bool mail; };
static bool path_view(const ambi_batch_t* b, int unit, PathView& v) {
    if (!b || unit < 0 || unit >= (int)b->hb.units.size()) return false;
    const UnitIn& U = b->hb.units[unit];
    if (b->downloaded) {
        UnitRef R;
        if (!unit_ref(b, unit, R, true)) return false;
        const UnitIn& Q = *R.U;
        const UnitLayout L = unit_layout(Q.n_seg, Q.bkp_cap, Q.path_cap, Q.out_cap);
        const uint8_t* r = R.blob + Q.res_off;
        v = PathView{R.h, reinterpret_cast<const rcell_t*>(r + L.path),
                     reinterpret_cast<const rcell_t*>(r + L.path_ind), reinterpret_cast<const OutJunc*>(r + L.out_junc), false};
        return true;
    }
    if (!b->mail_view) return false;
    const uint8_t* slot = b->be->mail_slot(unit);
    if (!slot) return false;
    const MailLayout M = mail_layout(U.path_cap, U.out_cap);
    v = PathView{reinterpret_cast<const UnitOut*>(slot), reinterpret_cast<const rcell_t*>(slot + M.path), reinterpret_cast<const rcell_t*>(slot + M.path_ind),
                 reinterpret_cast<const OutJunc*>(slot + M.out_junc), true};
    return true;
}

int ambi_batch_unit_result(const ambi_batch_t* b, int32_t unit, ambi_unit_result_t* out) {
    PathView pv;
    pv.mail = false;
    if (b && b->sharded_ready) {   // the header came back with the run: no blob needed
        pv.h = header(b, unit);
        if (!pv.h || !out) return AMBI_ERR_ARG;
    } else
    if (!path_view(b, unit, pv) || !out) return b && !b->downloaded && !b->mail_view ? AMBI_ERR_STATE : AMBI_ERR_ARG;
    const UnitOut* h = pv.h;
    out->status = h->status; out->bias = h->bias; out->n_nodes = h->K; out->bkp_len = h->bkp_len;
    out->path_len = h->path_len; out->path_indel_len = h->path_indel_len; out->indel_printed = h->indel_printed;
    out->n_out_junc = h->n_out_junc; out->first_forward = h->first_forward; out->evaluated = h->evaluated;
    out->num_orders = h->num_orders; out->first_valid = h->first_valid; out->inv_cn_sum = h->inv_cn_sum;
    out->path_indel_stored = h->path_ind_stored; out->reserved = 0;
    if (pv.mail) out->num_orders = -1;   // the order count comes from the lattice stage, behind the express kernel: not in the mailbox
    return 0;
}
int ambi_batch_unit_path(const ambi_batch_t* b, int32_t unit, int32_t which, int32_t* out, int32_t cap) {
    if (b && b->sharded_ready) {
        // the final path came back in run-length form with the run; it is also the path before indelBFB unless that stage edited it
        UnitRef R;
        if (!unit_ref(b, unit, R, false)) return AMBI_ERR_ARG;
        if (R.shard->view_there && (which || !R.h->path_ind_stored)) {
            const RunsView& v = R.shard->view;
            int at = 0;
            for (int64_t r = v.run_off[R.local]; r < v.run_off[R.local] + v.run_counts[R.local]; r++)
                for (int k = 0; k < v.run_len[r]; k++, at++) if (out && at < cap) out[at] = v.run_start[r] + k;
            return at;
        }
    }
    PathView pv;
    if (!path_view(b, unit, pv)) return b && !b->downloaded && !b->mail_view ? AMBI_ERR_STATE : AMBI_ERR_ARG;
    const UnitOut* h = pv.h;
    const UnitIn& U = b->hb.units[unit];
    // the path after indelBFB is stored separately only when indelBFB changed it
    const rcell_t* src = (which && h->path_ind_stored) ? pv.path_ind : pv.path;
    int len = which ? h->path_indel_len : h->path_len;
    if (out) for (int i = 0; i < len && i < cap; i++) out[i] = abs_cell(src[i], U.seg_base);   // the blob holds local ids
    return len;
}
int ambi_batch_unit_bkp(const ambi_batch_t* b, int32_t unit, int32_t* out, int32_t cap) {
    const UnitOut* h = header(b, unit);
    if (!h) return b && !b->downloaded ? AMBI_ERR_STATE : AMBI_ERR_ARG;
    UnitRef R;
    if (!unit_ref(b, unit, R, true)) return AMBI_ERR_STATE;
    const UnitIn& U = *R.U;
    UnitLayout L = unit_layout(U.n_seg, U.bkp_cap, U.path_cap, U.out_cap);
    const int16_t* src = reinterpret_cast<const int16_t*>(R.blob + U.res_off + L.bkp);
    if (out) for (int i = 0; i < h->bkp_len && i < cap; i++) { int v = src[i]; out[i] = v > 0 ? v + U.seg_base : v - U.seg_base; }
    return h->bkp_len;
}
int ambi_batch_unit_prepare(const ambi_batch_t* b, int32_t unit, double* junc_cn, double* seg_cn, int32_t* target_cn,
                            int32_t* inv_junc_global) {
    const UnitOut* h = header(b, unit);
    if (!h) return b && !b->downloaded ? AMBI_ERR_STATE : AMBI_ERR_ARG;
    UnitRef R;
    if (!unit_ref(b, unit, R, true)) return AMBI_ERR_STATE;
    const UnitIn& U = *R.U;
    UnitLayout L = unit_layout(U.n_seg, U.bkp_cap, U.path_cap, U.out_cap);
    const uint8_t* r = R.blob + U.res_off;
    const int n = U.n_seg;
    if (junc_cn) memcpy(junc_cn, r + L.junc_cn, sizeof(double) * 2 * (n + 1));
    if (seg_cn) memcpy(seg_cn, r + L.seg_cn, sizeof(double) * (n + 1));
    if (target_cn) memcpy(target_cn, r + L.target_cn, sizeof(int32_t) * (n + 1));
    if (inv_junc_global) {
        const int32_t* ij = reinterpret_cast<const int32_t*>(r + L.inv_junc);
        const std::vector<int32_t>& map = b->hb.junc_global[unit];
        for (int i = 0; i <= n; i++) inv_junc_global[i] = (ij[i] >= 0 && ij[i] < (int)map.size()) ? map[ij[i]] : ij[i];
    }
    return n + 1;
}
int ambi_batch_unit_dag(const ambi_batch_t* b, int32_t unit, int32_t* node2pat, int32_t* node2loop, uint64_t* succ) {
    const UnitOut* h = header(b, unit);
    if (!h) return b && !b->downloaded ? AMBI_ERR_STATE : AMBI_ERR_ARG;
    int local = unit;
    Backend* be = b->owner(unit, &local);
    const int base = b->hb.units[unit].seg_base;
    if (h->K > kMaxNodes) {   // a wide unit (64..255 nodes): succ[] gets the LOW word of every successor set (ambi_batch_unit_dag_nwords has all)
        std::vector<int32_t> p(kWideNodeCap * 3), l(kWideNodeCap * 3);
        std::vector<uint64_t> s2((size_t)kWideNodeCap * kWideWords);
        int rc = be->copy_dag_wide(local, p.data(), l.data(), s2.data());
        if (rc) return rc;
        for (int i = 0; i < h->K; i++) {
            for (int c = 0; c < 3; c++) {
                if (node2pat) node2pat[3 * i + c] = p[3 * i + c] + ((c < 2 && p[3 * i] != 0) ? base : 0);
                if (node2loop) node2loop[3 * i + c] = l[3 * i + c] + ((c < 2 && l[3 * i] != 0) ? base : 0);
            }
            if (succ) succ[i] = s2[(size_t)kWideWords * i];
        }
        return h->K;
    }
    Dag D;
    int rc = be->copy_dag(local, &D);
    if (rc) return rc;
    for (int i = 0; i < h->K; i++) {
        for (int c = 0; c < 3; c++) {
            if (node2pat) node2pat[3 * i + c] = D.pat[i][c] + ((c < 2 && D.pat[i][0] != 0) ? base : 0);
            if (node2loop) node2loop[3 * i + c] = D.loop[i][c] + ((c < 2 && D.loop[i][0] != 0) ? base : 0);
        }
        if (succ) succ[i] = D.succ[i];
    }
    return h->K;
}
int ambi_batch_unit_dag_nwords(const ambi_batch_t* b, int32_t unit, int32_t nwords, uint64_t* succ) {
    const UnitOut* h = header(b, unit);
    if (!h || !succ || nwords < 1) return b && !b->downloaded ? AMBI_ERR_STATE : AMBI_ERR_ARG;
    int local = unit;
    Backend* be = b->owner(unit, &local);
    if (h->K > kMaxNodes) {
        std::vector<int32_t> p(kWideNodeCap * 3), l(kWideNodeCap * 3);
        std::vector<uint64_t> s2((size_t)kWideNodeCap * kWideWords);
        int rc = be->copy_dag_wide(local, p.data(), l.data(), s2.data());
        if (rc) return rc;
        for (int i = 0; i < h->K; i++)
            for (int w = 0; w < nwords; w++) succ[(size_t)nwords * i + w] = w < kWideWords ? s2[(size_t)kWideWords * i + w] : 0;
        return h->K;
    }
    Dag D;
    int rc = be->copy_dag(local, &D);
    if (rc) return rc;
    for (int i = 0; i < h->K; i++) for (int w = 0; w < nwords; w++) succ[(size_t)nwords * i + w] = w == 0 ? D.succ[i] : 0;
    return h->K;
}
int ambi_batch_unit_dag_words(const ambi_batch_t* b, int32_t unit, uint64_t* succ2) { return ambi_batch_unit_dag_nwords(b, unit, 2, succ2); }
int ambi_batch_unit_out_juncs(const ambi_batch_t* b, int32_t unit, int32_t* u, int32_t* v, int32_t* count, int32_t cap) {
    PathView pv;
    if (!path_view(b, unit, pv)) return b && !b->downloaded && !b->mail_view ? AMBI_ERR_STATE : AMBI_ERR_ARG;
    const UnitOut* h = pv.h;
    const OutJunc* src = pv.out;
    for (int i = 0; i < h->n_out_junc && i < cap; i++) {
        if (u) u[i] = src[i].u;
        if (v) v[i] = src[i].v;
        if (count) count[i] = src[i].count;
    }
    return h->n_out_junc;
}
int ambi_batch_unit_orders(ambi_batch_t* b, int32_t unit, int64_t first, int64_t count, uint8_t* out) {
    if (!b || !(b->uploaded || !b->shards.empty()) || !out || unit < 0 || unit >= (int)b->hb.units.size()) return AMBI_ERR_ARG;
    int local = unit;
    Backend* be = b->owner(unit, &local);
    return be->copy_orders(local, first, count, out);
}
int ambi_batch_all_count(ambi_batch_t* b, int32_t unit, int32_t pass, int64_t* count) {
    if (!b || !(b->uploaded || !b->shards.empty()) || !count || unit < 0 || unit >= (int)b->hb.units.size()) return AMBI_ERR_ARG;
    int local = unit;
    Backend* be = b->owner(unit, &local);
    return be->all_count(local, pass, count);
}
int ambi_batch_all_orders(ambi_batch_t* b, int32_t unit, int32_t pass, int64_t first, int64_t count, int64_t* order_idx) {
    if (!b || !(b->uploaded || !b->shards.empty()) || (!order_idx && count > 0) || unit < 0 || unit >= (int)b->hb.units.size()) return AMBI_ERR_ARG;
    int local = unit;
    Backend* be = b->owner(unit, &local);
    return be->all_orders(local, pass, first, count, order_idx);
}
int ambi_batch_all_paths(ambi_batch_t* b, int32_t unit, int32_t pass, int64_t first, int64_t count, int32_t* lengths, int32_t* cells,
                         int64_t stride) {
    if (!b || !(b->uploaded || !b->shards.empty()) || !lengths || !cells || unit < 0 || unit >= (int)b->hb.units.size()) return AMBI_ERR_ARG;
    int local = unit;
    Backend* be = b->owner(unit, &local);
    return be->all_paths(local, pass, first, count, lengths, cells, stride);
}
int ambi_batch_all_set_shard(ambi_batch_t* b, int32_t rank, int32_t world) { return b ? b->be->set_shard(rank, world) : AMBI_ERR_ARG; }
int ambi_batch_all_device(ambi_batch_t* b, void** ptr, int64_t* bytes) { return (b && b->uploaded) ? b->be->all_device(ptr, bytes) : AMBI_ERR_ARG; }
int ambi_batch_all_finish(ambi_batch_t* b) { return (b && b->uploaded) ? b->be->all_finish() : AMBI_ERR_ARG; }
int ambi_batch_set_timing(ambi_batch_t* b, int32_t on) { if (!b) return AMBI_ERR_ARG; b->be->set_timing(on != 0); return 0; }
int ambi_batch_set_timing_mask(ambi_batch_t* b, uint32_t mask) { if (!b) return AMBI_ERR_ARG; b->be->set_timing_mask(mask); return 0; }
int ambi_batch_slices(const ambi_batch_t* b) { return b ? b->be->slice_count() : AMBI_ERR_ARG; }
int ambi_batch_kernel_count(const ambi_batch_t* b) { return b ? (int)b->be->kernel_times().size() : AMBI_ERR_ARG; }
int ambi_batch_kernel_time(const ambi_batch_t* b, int32_t idx, const char** name, float* ms) {
    if (!b) return AMBI_ERR_ARG;
    const auto& kt = b->be->kernel_times();
    if (idx < 0 || idx >= (int)kt.size()) return AMBI_ERR_ARG;
    if (name) *name = kt[idx].name;
    if (ms) *ms = kt[idx].ms;
    return 0;
}
int ambi_batch_kernel_span(const ambi_batch_t* b, int32_t idx, float* start_ms, float* end_ms) {
    if (!b) return AMBI_ERR_ARG;
    const auto& kt = b->be->kernel_times();
    if (idx < 0 || idx >= (int)kt.size()) return AMBI_ERR_ARG;
    if (start_ms) *start_ms = kt[idx].start_ms;
    if (end_ms) *end_ms = kt[idx].end_ms;
    return 0;
}
int ambi_batch_traffic(const ambi_batch_t* b, int64_t* input_bytes, int64_t* order_bytes, int64_t* result_bytes) {
    if (!b) return AMBI_ERR_ARG;
    const HostBatch& hb = b->hb;
    if (input_bytes) *input_bytes = (int64_t)(hb.seg_cn.size() * 8 + hb.juncs.size() * sizeof(Junction) + hb.elems.size() * sizeof(Element) + hb.units.size() * sizeof(UnitIn));
    if (order_bytes) *order_bytes = b->be->order_bytes_written();
    if (result_bytes) *result_bytes = hb.result_bytes;
    return 0;
}

// ---- whole-sample helpers ----
int64_t ambi_format_path(const ambi_graph_t* g, const int32_t* path, int32_t len, char* buf, int64_t cap) {
    if (!g) return AMBI_ERR_ARG;
    return copy_text(format_path(g->g, path, len), buf, cap);
}
int ambi_translocation_bfb(const ambi_graph_t* g, int32_t* paths, const int64_t* offsets, int32_t n_chr, int32_t* out, int32_t cap) {
    if (!g || !paths || !offsets) return AMBI_ERR_ARG;
    std::vector<std::vector<int32_t>> pp(n_chr);
    for (int c = 0; c < n_chr; c++) pp[c].assign(paths + offsets[c], paths + offsets[c + 1]);
    std::vector<int32_t> res;
    translocation_bfb(g->g, pp, res);
    for (int c = 0; c < n_chr; c++)   // the per-chromosome paths may have been reverse-complemented in place
        for (size_t i = 0; i < pp[c].size(); i++) paths[offsets[c] + i] = pp[c][i];
    if (out) for (size_t i = 0; i < res.size() && (int)i < cap; i++) out[i] = res[i];
    return (int)res.size();
}

}  // extern "C"

// ---- ILP model of one chromosome (host; LocalGenomicMap::BFB_ILP, LGM.cpp:4397-4752) ----
#include "ambi_ilp.hpp"
#include "ambi_ilp_rows.hpp"
struct ambi_ilp { ambi::IlpModel m; };

extern "C" {

int ambi_ilp_build(const ambi_graph_t* g, int32_t chr, const double* seg_cn, const double* junc_cn, int32_t bias,
                   double max_cn_total, int32_t juncs_info, ambi_ilp_t** out) {
    if (!g || !seg_cn || !junc_cn || !out || chr < 0 || chr >= g->g.n_chr()) return AMBI_ERR_ARG;
    const int s = g->g.source_ids[chr], e = g->g.sink_ids[chr], n = e - s + 1;
    std::vector<double> cn(n), fold(n);
    for (int i = 0; i < n; i++) { cn[i] = seg_cn[i + 1]; fold[i] = junc_cn[2 * (i + 1) + 1]; }
    std::vector<std::vector<int32_t>> comps;
    for (auto& c : g->g.components)
        if (!c.empty() && c[0] >= 1 && c[0] <= g->g.n_seg() && g->g.seg_partition[c[0] - 1] == chr) comps.push_back(c);
    ambi_ilp* p = new ambi_ilp();
    ambi::build_bfb_ilp(s, e, cn.data(), fold.data(), bias, max_cn_total, comps, juncs_info != 0, p->m);
    *out = p;
    return 0;
}
int ambi_ilp_build_device(const ambi_graph_t* g, int32_t chr, const double* seg_cn, const double* junc_cn, int32_t bias,
                          double max_cn_total, int32_t juncs_info, float* kernel_ms, ambi_ilp_t** out) {
    if (!g || !seg_cn || !junc_cn || !out || chr < 0 || chr >= g->g.n_chr()) return AMBI_ERR_ARG;
    const int s = g->g.source_ids[chr], e = g->g.sink_ids[chr], n = e - s + 1;
    std::vector<double> cn(n), fold(n);
    for (int i = 0; i < n; i++) { cn[i] = seg_cn[i + 1]; fold[i] = junc_cn[2 * (i + 1) + 1]; }
    std::vector<std::vector<int32_t>> comps;
    for (auto& c : g->g.components)
        if (!c.empty() && c[0] >= 1 && c[0] <= g->g.n_seg() && g->g.seg_partition[c[0] - 1] == chr) comps.push_back(c);
    std::unique_ptr<ambi_ilp> p(new ambi_ilp());
    std::vector<ambi::IlpRowDesc> rows; std::vector<int32_t> lc; std::vector<double> lv;
    ambi::build_bfb_ilp_rows(s, e, cn.data(), fold.data(), bias, max_cn_total, comps, juncs_info != 0, p->m, rows, lc, lv);
    int rc = ambi::backend_ilp_fill(rows.data(), (int64_t)rows.size(), p->m.row_ptr.data(), s, e, lc.data(), lv.data(), (int64_t)lc.size(),
                                    p->m.col.data(), p->m.val.data(), kernel_ms);
    if (rc) return rc;
    *out = p.release();
    return 0;
}
int ambi_ilp_build_sc(const ambi_graph_t* g0, int32_t chr, int32_t n_graphs, const double* seg_cn, const double* fold_cn, ambi_ilp_t** out) {
    if (!g0 || !seg_cn || !fold_cn || !out || n_graphs < 1 || chr < 0 || chr >= g0->g.n_chr()) return AMBI_ERR_ARG;
    const int s = g0->g.source_ids[chr], e = g0->g.sink_ids[chr];
    std::vector<std::pair<int, int>> evolution;     // localhap.cpp:430-434: every pair i < j
    for (int i = 0; i < n_graphs; i++) for (int j = i + 1; j < n_graphs; j++) evolution.push_back({i, j});
    ambi_ilp* p = new ambi_ilp();
    ambi::build_bfb_ilp_sc(s, e, n_graphs, seg_cn, fold_cn, evolution, p->m);
    *out = p;
    return 0;
}
void ambi_ilp_destroy(ambi_ilp_t* p) { delete p; }
int ambi_ilp_sizes(const ambi_ilp_t* p, int64_t* n_rows, int64_t* nnz, int32_t* n_cols, int32_t* n_int) {
    if (!p) return AMBI_ERR_ARG;
    if (n_rows) *n_rows = p->m.n_rows();
    if (nnz) *nnz = p->m.nnz();
    if (n_cols) *n_cols = p->m.n_cols;
    if (n_int) *n_int = p->m.n_int;
    return 0;
}
int ambi_ilp_copy(const ambi_ilp_t* p, int64_t* row_ptr, int32_t* col, double* val, double* row_lo, double* row_up,
                  double* col_lo, double* col_up, double* obj) {
    if (!p) return AMBI_ERR_ARG;
    const ambi::IlpModel& m = p->m;
    if (row_ptr) memcpy(row_ptr, m.row_ptr.data(), m.row_ptr.size() * sizeof(int64_t));
    if (col) memcpy(col, m.col.data(), m.col.size() * sizeof(int32_t));
    if (val) memcpy(val, m.val.data(), m.val.size() * sizeof(double));
    if (row_lo) memcpy(row_lo, m.row_lo.data(), m.row_lo.size() * sizeof(double));
    if (row_up) memcpy(row_up, m.row_up.data(), m.row_up.size() * sizeof(double));
    if (col_lo) memcpy(col_lo, m.col_lo.data(), m.col_lo.size() * sizeof(double));
    if (col_up) memcpy(col_up, m.col_up.data(), m.col_up.size() * sizeof(double));
    if (obj) memcpy(obj, m.obj.data(), m.obj.size() * sizeof(double));
    return 0;
}
int64_t ambi_graph_chrom_name(const ambi_graph_t* g, int32_t seg_id, char* buf, int64_t cap) {
    if (!g || seg_id < 1 || seg_id > g->g.n_seg()) return AMBI_ERR_ARG;
    return copy_text(g->g.seg_chrom[seg_id - 1], buf, cap);
}
int ambi_ilp_write_lp(const ambi_ilp_t* p, const char* path) {
    if (!p || !path) return AMBI_ERR_ARG;
    return ambi::write_lp(path, p->m) ? 0 : AMBI_ERR_OPEN;
}
int ambi_ilp_write_mps(const ambi_ilp_t* p, const char* path) {
    if (!p || !path) return AMBI_ERR_ARG;
    return ambi::write_mps(path, p->m) ? 0 : AMBI_ERR_OPEN;
}

}  // extern "C"
