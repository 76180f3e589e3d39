// ambi_backend.hpp -- what the C-ABI layer (ambi_capi.cpp) needs from an execution backend.
// The product links exactly one implementation: the HIP engine (ambi_engine.hip).  tests/hostsim links the same
// C-ABI layer against a 1-thread host simulation of the SAME stage code for CPU-only checks; it is never shipped
// in ambigram_amd/ and never selected at run time.
#pragma once
#include <stdint.h>
#include <vector>

#include "ambi_pack.hpp"

namespace ambi {

struct EngineConfig {
    int64_t order_arena_bytes = 0;   // 0: sized from the first run
    int32_t first_budget = 64;
    int32_t block_lds = 49152;       // LDS bytes per workgroup for a unit's block-emission image (env AMBI_BLOCK_LDS overrides)
    int32_t block_scratch_lds = 16384;   // LDS of the image-build kernel (automaton copy; env AMBI_BLOCK_SCRATCH_LDS overrides)
    int32_t block_max = 256;         // largest suffix block in rows (env AMBI_BLOCK_MAX overrides; rows of a byte per node: 128 / 192 / 256 / 320 measure the same, profiles/r01_slices.md; 5-bit rows: 64 / 96 / 128 / 192 / 256 / 320 / 384 / 512 = 1.47 / 1.01 / 0.918 / 0.92 / 0.900 / 0.903 / 0.914 / 1.00 ms per step, profiles/r03_notes.md; an image that does not fit the LDS budget sends the unit down the general path)
    int32_t target_lanes = 524288;   // enumerate kernel: rows of the batch are spread over about this many lanes
    int32_t slices = 0;              // unit ranges run on separate streams (0: automatic; env AMBI_SLICES overrides)
};

struct KernelTime { const char* name; float ms; float start_ms = -1.f, end_ms = -1.f; };   // start / end: from the start of the run's first kernel (mean over the timed runs; -1: not known)
// the final paths of a batch in run-length form, in HOST memory (Backend::runs_wait): per unit its cells and runs, then the runs of
// all units one after the other (start value = absolute signed segment id, length); headers: UnitOut[U] or nullptr
struct RunsView {
    int64_t n_runs, n_cells;
    const int32_t* lengths; const int32_t* run_counts; const int32_t* run_start; const int32_t* run_len;
    const int64_t* run_off;      // [U+1] unit u's runs are run_start / run_len [run_off[u] .. run_off[u] + run_counts[u])
    const void* headers;
    int64_t bytes, copied_bytes;
};

class Backend {
  public:
    virtual ~Backend() {}
    virtual const char* name() const = 0;
    virtual int device_count(int* n) = 0;
    virtual int set_device(int d) = 0;
    virtual int upload(const HostBatch& hb, const EngineConfig& cfg) = 0;
    virtual int run(uint32_t flags, void* stream) = 0;
    // a stream of the backend's own on its device (process lifetime), for callers that have none to pass: nullptr = none (the host
    // simulation); and a counter that moves whenever the results on the device change after run() has returned (the first run of a
    // batch done again with a larger arena, units finished by the parallel search at wait())
    virtual void* own_stream() { return nullptr; }
    virtual int64_t results_epoch() const { return 0; }
    virtual int wait() = 0;
    virtual int wait_results() { return wait(); }   // results complete; order tables may still be in flight (express path)
    virtual int download(std::vector<uint8_t>& blob) = 0;
    // Header, final path(s) and output junctions of a unit where the HOST can read them without a copy command (MailLayout,
    // ambi_batch.hpp), valid after wait_results() / wait() of a small batch; nullptr: download the blob instead
    virtual const uint8_t* mail_slot(int unit) { (void)unit; return nullptr; }
    virtual int device_results(void** ptr, int64_t* bytes) = 0;
    virtual int pack_runs(int which, int32_t* dev_lengths, int32_t* dev_run_counts, int32_t* dev_run_start, int32_t* dev_run_len,
                          int64_t run_cap, int64_t* dev_totals, void* stream) = 0;
    virtual int pack_paths(int which, int32_t* dev_lengths, int32_t* dev_cells, int64_t cell_cap, int64_t* dev_total,
                           void* stream) = 0;
    // final paths (which: 0 getBFB, 1 after indelBFB) packed into run-length form and copied to pinned host memory behind `stream`,
    // without blocking it: slot 0 / 1 alternate so that the copy of one run travels while the next run computes
    virtual int runs_to_host(int which, int slot, int with_headers, void* stream) { (void)which; (void)slot; (void)with_headers; (void)stream; return ST_ERR_BAD_INPUT; }
    virtual int runs_wait(int slot, RunsView* out) { (void)slot; (void)out; return ST_ERR_BAD_INPUT; }
    virtual int copy_orders(int unit, int64_t first, int64_t count, uint8_t* out) = 0;
    virtual int copy_dag(int unit, Dag* out) = 0;
    // the same for a wide unit (64..127 nodes): node records [K][3] each, successor sets as [K][2] 64-bit words
    virtual int copy_dag_wide(int unit, int32_t* pat, int32_t* loop, uint64_t* succ2) { (void)unit; (void)pat; (void)loop; (void)succ2; return ST_ERR_BAD_INPUT; }
    virtual void set_timing(bool on) = 0;
    virtual void set_timing_mask(uint32_t mask) { set_timing(mask != 0); }
    virtual const std::vector<KernelTime>& kernel_times() = 0;
    virtual int64_t order_bytes_written() const = 0;
    virtual size_t object_bytes() const = 0;   // sizeof the concrete backend (diagnostics: ambi_batch_destroy's quarantine mode)
    virtual int slice_count() const { return 1; }   // launches of every kernel per run
    // --all (run with FLAG_ALL, after wait): valid orders of pass 0 (first orientation) / pass 1 (flipped orientation,
    // empty unless the last order of pass 0 is invalid), and the paths of a range of them (cells: count x stride int32)
    virtual int all_count(int unit, int pass, int64_t* count) = 0;
    virtual int all_orders(int unit, int pass, int64_t first, int64_t count, int64_t* idx) = 0;
    // --all with the orders of a unit dealt over `world` ranks (see BatchArgs::all_rank): set before run; after wait() every
    // rank merges the pool all_device() names (bitmaps + flags, bytes: MAX over the ranks) and calls all_finish()
    virtual int set_shard(int rank, int world) = 0;
    virtual int all_device(void** ptr, int64_t* bytes) = 0;
    virtual int all_finish() = 0;
    virtual int all_paths(int unit, int pass, int64_t first, int64_t count, int32_t* lengths, int32_t* cells, int64_t stride) = 0;
};

Backend* make_backend();   // defined by the linked backend
// diagnostics: microseconds until a tiny kernel on stream b has run while a kernel with a long backlog of workgroups occupies stream a
int backend_stream_probe(void* stream_a, void* stream_b, float* us);

// ILP entries (ambi_ilp_rows.hpp) written by the linked backend: the HIP engine launches ambi_ilp_fill_kernel and copies
// col/val back, the host simulation runs the same entry function on the CPU.  kernel_ms: device time of the fill (0 on the host).
struct IlpRowDesc;
// runs -> cells in the backend's memory space (ambi_expand_runs of the C ABI)
int backend_expand_runs(const int32_t* run_start, const int32_t* run_len, const int64_t* cell_off, int64_t n_runs, int32_t* cells,
                        int64_t cell_cap, void* stream);
int backend_ilp_fill(const IlpRowDesc* rows, int64_t n_rows, const int64_t* row_ptr, int start_id, int end_id, const int32_t* lit_col,
                     const double* lit_val, int64_t n_lit, int32_t* col, double* val, float* kernel_ms);

}  // namespace ambi
