/* ambigram_hip.h -- C ABI of libambigram_hip.so, the MI355X-native BFB path-reconstruction engine.
 *
 * Drop-in boundary for the `--op bfb` path of deepomicslab/Ambigram.  The reference has no plugin/FFI interface
 * (SURVEY.md 8b): the path is reached through C++ member calls on pointer-graph objects from main()
 * (localhap.cpp:49-388).  Each entry point below names the reference call(s) it replaces; INTEGRATION.md shows
 * the binding a maintainer adds to localhap.cpp and the ctypes stub used by ambigram_amd/.
 *
 * Conventions: plain pointers and sizes, caller-owned arrays, int32 unless noted; return 0 on success or a negative
 * code (ambi_error_string); no exceptions cross the boundary; a handle is not thread-safe, different handles are.
 * Vertices are signed ABSOLUTE segment ids: +id = (id,'+'), -id = (id,'-').
 */
#ifndef AMBIGRAM_HIP_H
#define AMBIGRAM_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMBI_ABI_VERSION 1

/* run flags */
#define AMBI_FLAG_REVERSED 1u /* --reversed (localhap.cpp:37,55) */
#define AMBI_FLAG_ALL 2u      /* --all      (localhap.cpp:38,56) */
/* Extension (no counterpart in the reference, which always materialises every topological order, LocalGenomicMap.cpp:3380-3409):
 * the run does not WRITE the order tables.  Results are identical (the scan for the first valid order reads the first orders the
 * lattice stage unranks, --all unranks in the kernel); the tables are written on demand -- when a unit's scan runs out of
 * budget and the parallel search needs them, or when ambi_batch_unit_orders asks for rows.  bench.py reports the step with
 * and without the tables (91 % of the default step's HBM bytes are this by-product). */
#define AMBI_FLAG_LAZY_ORDERS 4u

/* per-unit status (ambi_unit_result_t.status); negative values are errors */
#define AMBI_ST_OK 0
#define AMBI_ST_SHORTCUT 1        /* no fold-back inversion: path 1+..n+ (localhap.cpp:164-170) */
#define AMBI_ST_INFEASIBLE 2      /* .sol Infeasible: path 1+..n+ + "ILP is unsolvable." (localhap.cpp:213-220) */
#define AMBI_ST_NO_VALID_ORDER 3  /* no order assembles in either orientation (reference leaves the path empty) */
#define AMBI_ERR_TOO_MANY_NODES (-10)  /* more than 255 selected patterns + loops in one chromosome */
#define AMBI_ERR_NO_ELEMENTS (-11)
#define AMBI_ERR_REF_UB (-12)     /* the reference reads out of bounds on this input; refused instead of guessed */
#define AMBI_ERR_BKP_CAPACITY (-13)
#define AMBI_ERR_PATH_CAPACITY (-14)
#define AMBI_ERR_ORDERS_CAPACITY (-15)  /* the unit's order table does not fit the arena (it could not grow: out of memory, or the
                                        * limit AMBI_ARENA_MAX_BYTES of the environment); units are given rows in unit order, the
                                        * others end with this status and carry no results */
#define AMBI_ERR_IDEALS_CAPACITY (-16)
#define AMBI_ERR_BAD_INPUT (-17)
#define AMBI_ERR_OUTJUNC_CAPACITY (-18)
/* host / runtime errors */
#define AMBI_ERR_OPEN (-1)
#define AMBI_ERR_MALFORMED (-2)
#define AMBI_ERR_UNKNOWN_SEG (-3)
#define AMBI_ERR_SOURCE_SINK (-4)
#define AMBI_ERR_PLOIDY (-5)
#define AMBI_ERR_SEG_IDS (-6)
#define AMBI_ERR_SOL_OPEN (-7)
#define AMBI_ERR_LINE_TOO_LONG (-8)
#define AMBI_ERR_UNSUPPORTED (-9)
#define AMBI_ERR_NO_DEVICE (-30)  /* no HIP device: the engine has NO CPU fallback */
#define AMBI_ERR_HIP (-31)
#define AMBI_ERR_STATE (-32)      /* call order violated (e.g. run before upload) */
#define AMBI_ERR_ARG (-33)

const char* ambi_error_string(int code);
int ambi_abi_version(void);
/* "hip" for libambigram_hip.so.  (The test-only host simulation built under tests/hostsim reports "hostsim".) */
const char* ambi_backend_name(void);
int ambi_device_count(int* count);
int ambi_set_device(int device);
/* Diagnostics: do two HIP streams dispatch side by side?  A kernel with ~100 us of workgroups waiting is put on stream_a and one tiny
 * workgroup on stream_b; *us = microseconds from the start of the former to the end of the latter -- a few microseconds when the
 * streams feed different dispatch pipes, ~the whole backlog when they share one (the engine asks the same question about the
 * caller's stream when it picks its side streams). */
int ambi_debug_stream_probe(void* stream_a, void* stream_b, float* us);

/* ------------------------------------------------------------------------------------------------
 * Graph: replaces `new Graph(lh)` + calculateHapDepth + calculateCopyNum + readBFBProps
 * (localhap.cpp:65-75; Graph.cpp:36-49,109-237,312-405; LocalGenomicMap.cpp:3941-3987).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ambi_graph ambi_graph_t;
int ambi_graph_read_lh(const char* lh_path, ambi_graph_t** out);
void ambi_graph_destroy(ambi_graph_t* g);
int ambi_graph_sizes(const ambi_graph_t* g, int32_t* n_seg, int32_t* n_junc, int32_t* n_chr);
/* any output pointer may be NULL */
int ambi_graph_segments(const ambi_graph_t* g, int32_t* id, int32_t* chr_id, int32_t* start, int32_t* end, double* cov, double* cn);
int ambi_graph_junctions(const ambi_graph_t* g, int32_t* src, int8_t* sdir, int32_t* tgt, int8_t* tdir, double* cov,
                         double* cn, uint8_t* inferred, uint8_t* bounded);
int ambi_graph_chromosome(const ambi_graph_t* g, int32_t chr, int32_t* source_id, int32_t* sink_id);
/* chromosome name of segment `seg_id` (the SEG line's H:id:<chrom>:start:end); returns the length needed */
int64_t ambi_graph_chrom_name(const ambi_graph_t* g, int32_t seg_id, char* buf, int64_t cap);
/* replaces LocalGenomicMap::readComponents (LocalGenomicMap.cpp:5096-5156, localhap.cpp:102); may add junctions */
int ambi_graph_read_juncs(ambi_graph_t* g, const char* juncs_path);
/* stdout lines the reference prints while loading (progress, SEG echoes, .juncs breakpoints), '\n' separated.
 * Returns the number of bytes needed (excluding the terminator); copies at most cap-1 bytes. */
int64_t ambi_graph_log(const ambi_graph_t* g, char* buf, int64_t cap);
/* PROP line: ins_mode / con_mode as in localhap.cpp:72-75, main chromosome name copied into main_chr[cap] */
int ambi_graph_props(const ambi_graph_t* g, int32_t* ins_mode, int32_t* con_mode, char* main_chr, int64_t cap);
/* Components collected by ambi_graph_read_juncs (the `res` of readComponents, LocalGenomicMap.cpp:5096-5156, after its
 * sort/unique): component c = ids[offsets[c] .. offsets[c+1]).  Returns the number of components; copies at most
 * ids_cap ids and off_cap offsets (offsets needs count+1 entries). */
/* calculateHapDepth + calculateCopyNum once more (`--op sc_bfb` does that to its first graph, localhap.cpp:438-439): entries
 * whose copy number is still <= 0 are recomputed and echoed again; new lines are appended to the graph's log. */
int ambi_graph_recalculate(ambi_graph_t* g);
/* TRX-BFB, `PROP I1:...` / `PROP C1:...` (a translocation BEFORE the BFB cycles; localhap.cpp:79-88 and :263).  ambi_graph_load does
 * what the reference does between the PROP line and the chromosome loop: insertBeforeBFB / concatBeforeBFB (LocalGenomicMap.cpp:
 * 4195-4295 / 4297-4395) rebuild the graph -- so every other entry point (chromosomes, batch units, ILP) sees the REBUILT graph, and
 * ambi_graph_log holds the lines the reference prints while rebuilding.  The reference's `new Graph(mSegs, mJuncs, mSources, mSinks)`
 * assigns through pointers it never initialises (Graph.cpp:25-34); its evident meaning -- a graph made of copies of the four
 * vectors -- is what is implemented.
 * ambi_graph_trx_before: 0 for an ordinary graph; else the number of entries of the map rebuilt segment id -> id in the file
 *   (entry 0 unused), copied to original_of.
 * ambi_graph_trx_restore = LocalGenomicMap::virusBFB (LocalGenomicMap.cpp:3839-3939), called on every reconstructed chromosome's
 *   path after indelBFB (localhap.cpp:263): path (signed rebuilt ids, `len` cells, room for `cap`) becomes the path over the
 *   segments of the file; text receives the lines the reference prints (caption + path of the first stage and, when an unused
 *   junction cuts the path, of the second stage), *text_len its full length.  Returns the new number of cells, or
 *   AMBI_ERR_UNSUPPORTED where the reference aborts or leaves a vertex of the rebuilt graph in the path.
 * ambi_graph_trx_original: a handle of its own (ambi_graph_destroy) on the graph of the FILE -- chromosome names and coordinates of
 *   the vertices of restored paths (the reference's simulation_sv.txt rows, localhap.cpp:326-337). */
int ambi_graph_trx_before(const ambi_graph_t* g, int32_t* original_of, int32_t cap);
int ambi_graph_trx_original(const ambi_graph_t* g, ambi_graph_t** out);
int ambi_graph_trx_restore(const ambi_graph_t* g, int32_t* path, int32_t len, int32_t cap, char* text, int64_t text_cap, int64_t* text_len);
/* replaces Graph::writeGraph (Graph.cpp:239-266): the graph as it stands -- after the copy-number maths and any junctions
 * ambi_graph_read_juncs added -- as .lh text, byte for byte what the reference writes (fixed SAMPLE_NAME TEST, its seven
 * header keys, numbers in %g form, a 'B' behind every segment); "write seg", which the reference prints to stdout, is
 * appended to the graph's log. */
int ambi_graph_write_lh(ambi_graph_t* g, const char* lh_path);

int ambi_graph_components(const ambi_graph_t* g, int32_t* ids, int32_t ids_cap, int32_t* offsets, int32_t off_cap);

/* ------------------------------------------------------------------------------------------------
 * Batch of units.  A unit = one chromosome of one sample = one iteration of the loop localhap.cpp:111-265:
 *   getJuncCN (:136-139), bias (:141-146), getIndelBias (:147), no-FBI shortcut (:164-170), [ILP + cbc on the host],
 *   targetCN (:222-232), constructDAG (:236), allTopologicalOrders (:254), getBFB (:261), indelBFB (:262)
 * plus the unit's share of the output-junction loop (:267-289).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ambi_batch ambi_batch_t;
int ambi_batch_create(ambi_batch_t** out);
void ambi_batch_destroy(ambi_batch_t* b);

/* Adds chromosome `chr` of `g` with the ILP solution of that chromosome: n_cols (column index, value) pairs as read
 * from `<prefix>.sol` (column numbering of localhap.cpp:122-133; epsilon/bias columns and non-positive values are
 * ignored), or infeasible != 0.  Returns the unit index (>= 0) or a negative code. */
int ambi_batch_add_chromosome(ambi_batch_t* b, const ambi_graph_t* g, int32_t chr, int32_t n_cols, const int32_t* col,
                              const int32_t* val, int32_t infeasible);
/* Same, reading the .sol text (localhap.cpp:184-212).  A missing file returns AMBI_ERR_SOL_OPEN. */
int ambi_batch_add_chromosome_sol(ambi_batch_t* b, const ambi_graph_t* g, int32_t chr, const char* sol_path);
/* `--op sc_bfb` (localhap.cpp:390-679): several graphs share ONE joint .sol per chromosome; graph `block` of `n_blocks`
 * owns the columns [block*numComp, (block+1)*numComp) (localhap.cpp:540-566).  The unit reconstructs with that block's
 * elements and never takes the no-fold-back shortcut by itself (the driver decides it from the first graph, :505-512). */
int ambi_batch_add_chromosome_sol_block(ambi_batch_t* b, const ambi_graph_t* g, int32_t chr, const char* sol_path, int32_t block,
                                        int32_t n_blocks);
/* Raw unit: LOCAL segment ids 1..n_seg (absolute id = local + seg_base), junctions with both ends inside the unit,
 * elements (is_loop, a, b, cn) of the decomposition. */
int ambi_batch_add_unit(ambi_batch_t* b, int32_t n_seg, int32_t seg_base, const double* seg_cn, int32_t n_junc,
                        const int32_t* j_src, const int32_t* j_tgt, const int8_t* j_sdir, const int8_t* j_tdir,
                        const double* j_cn, int32_t n_elem, const int32_t* e_is_loop, const int32_t* e_a,
                        const int32_t* e_b, const int32_t* e_cn, int32_t infeasible, int32_t has_components);
int ambi_batch_size(const ambi_batch_t* b, int32_t* n_units);

/* Tunables (before upload): order-table arena bytes (0 = size it from the first run), ideal-table slots per unit,
 * first-valid scan budget, number of lanes the enumerate kernel spreads the order-table rows over. */
int ambi_batch_configure(ambi_batch_t* b, int64_t order_arena_bytes, int32_t ideal_cap, int32_t first_budget, int32_t target_lanes);

/* Diagnostics hook, not part of the drop-in path (the reference has nothing like it).  Replaces, for one unit, the
 * OUTCOME of evaluating an order (getBFB's per-order body, LGM.cpp:3519-3658) by a given verdict, so that the control
 * flow around it -- scan budget, parallel search for the minimum index, orientation flip (LGM.cpp:3686-3695), --all --
 * can be exercised at places no known input reaches.  verdicts[0..R) = orders in the "forward seed"
 * orientation, verdicts[R..2R) = "reversed seed"; 1 valid, 0 invalid, a negative AMBI_ERR_* code, 127 = evaluate as
 * usual.  Call after the unit was added and before ambi_batch_upload; the breakpoints of a "valid" order are those the
 * real assembly leaves. */
int ambi_batch_debug_inject_validity(ambi_batch_t* b, int32_t unit, const int8_t* verdicts, int64_t count);

/* Packs the units and hands the inputs to the current device (they stay resident in HBM across runs).  Streams, events,
 * pinned words and device blocks come from a per-device pool with process lifetime (created on first use, reused by every
 * later batch, never destroyed per batch).  Small batches: the input image is written into pinned host memory here and its
 * ONE copy to HBM is queued by the first ambi_batch_run ahead of that run's kernels. */
int ambi_batch_upload(ambi_batch_t* b);
/* Enqueues one pass of the whole pipeline over the batch on `hip_stream` (a hipStream_t; NULL = default stream; it must
 * stay alive until the run has been waited for).  Asynchronous: the first run of a batch takes the order-table arena the
 * pool holds, and ambi_batch_wait grows it and repeats the run if the tables did not fit. */
int ambi_batch_run(ambi_batch_t* b, uint32_t flags, void* hip_stream);
int ambi_batch_wait(ambi_batch_t* b);
/* Returns as soon as the RECONSTRUCTION results of the enqueued run are complete in HBM (paths, breakpoints, output
 * junctions, headers): for small batches the engine reconstructs every unit whose first order assembles in one kernel and
 * builds the order tables (LocalGenomicMap::allTopologicalOrders' by-product) behind it, still in flight when this call
 * returns.  ambi_batch_wait / ambi_batch_download / the next ambi_batch_run wait for everything.  Same as ambi_batch_wait
 * whenever the fast path does not apply. */
int ambi_batch_wait_results(ambi_batch_t* b);
/* Copies the result blob to the host (implies wait).  After this the getters below are valid. */
int ambi_batch_download(ambi_batch_t* b);
/* What the caller of ONE sample needs -- localhap.cpp:261-289: the path of getBFB, the path after indelBFB, the output
 * junctions -- on the host with the least traffic (implies ambi_batch_wait_results).  A small batch on the fast path has
 * had them written into pinned host memory by the reconstruction kernel itself: no copy command at all.  Otherwise the same
 * as ambi_batch_download.  Afterwards ambi_batch_unit_result (num_orders = -1 when the order count is not part of what
 * was fetched), ambi_batch_unit_path and ambi_batch_unit_out_juncs are valid; the other getters need ambi_batch_download. */
int ambi_batch_fetch_paths(ambi_batch_t* b);

/* Several GPUs from C (`ambi_bfb_reconstruct_batch(..., device_or_minus1_for_all)`; north star: "independent .lh
 * samples shard embarrassingly across the 8 GPUs of one node").  The units of a batch are the iterations of the loop
 * localhap.cpp:111-265 and do not depend on each other: this call deals them round-robin over the devices, runs every share
 * on its device from a host thread of its own (upload -> run(flags) -> download) and merges the results on the host.  Takes
 * the place of ambi_batch_upload + _run + _download: all getters are valid afterwards (the per-unit device-side ones -- DAG,
 * order rows, --all lists -- are answered by the device that holds the unit).  devices: n_devices device ordinals (an ordinal
 * may repeat: several shares on one device), or NULL for the first n_devices visible devices (all of them if n_devices <= 0).
 * May be called again (the shares stay resident).  A batch is either uploaded to one device or sharded, not both. */
int ambi_batch_run_sharded(ambi_batch_t* b, uint32_t flags, const int32_t* devices, int32_t n_devices);

/* Device-side view for collectives: the result blob (header [n_units] + per-unit arrays) lives in device memory.  Its
 * layout is the engine's own (csrc/ambi_batch.hpp: UnitOut, unit_layout; path cells are 2-byte LOCAL signed segment ids --
 * absolute id = id +- the chromosome's first segment id - 1); portable consumers use ambi_batch_pack_paths / _pack_runs, which
 * deliver absolute ids, or the getters after ambi_batch_download. */
int ambi_batch_device_results(ambi_batch_t* b, void** dev_ptr, int64_t* bytes);
/* Packs the final paths of all units into caller-provided DEVICE buffers: lengths[n_units] (int32) and the
 * concatenation of the paths (int32, absolute signed ids) -- the payload of the end-of-batch RCCL gather.
 * which: 0 = getBFB path, 1 = path after indelBFB.  total_cells receives the number of cells written (device int64). */
int ambi_batch_pack_paths(ambi_batch_t* b, int32_t which, int32_t* dev_lengths, int32_t* dev_cells, int64_t cell_cap,
                          int64_t* dev_total_cells, void* hip_stream);

/* The same payload in run-length form: a path is a sequence of runs whose cells count up by one (a stretch of
 * consecutive segments on one strand: `3+4+5+` = start 3, length 3; `5-4-3-` = start -5, length 3), a few dozen runs for
 * thousands of cells, so the exchange moves kilobytes instead of megabytes per sample and the receiving rank expands
 * the runs in its own memory (ambi_expand_runs).  Per unit: dev_lengths[u] = cells, dev_run_counts[u] = runs; the runs
 * of all units follow each other in dev_run_start / dev_run_len (int32 each, at most run_cap).  dev_totals (device,
 * int64[2]) receives {runs, cells}. */
int ambi_batch_pack_runs(ambi_batch_t* b, int32_t which, int32_t* dev_lengths, int32_t* dev_run_counts, int32_t* dev_run_start,
                         int32_t* dev_run_len, int64_t run_cap, int64_t* dev_totals, void* hip_stream);
/* The final paths of every unit ON THE HOST, in that run-length form, without stopping the stream (the reference
 * ends with its paths in host memory and prints every one, LocalGenomicMap.cpp:3684-3689, localhap.cpp:262).
 * ambi_batch_runs_to_host queues, behind the work already on hip_stream, ONE device-to-host copy of {lengths, run counts, runs} into
 * pinned memory on a copy stream of the engine's own -- the caller's stream is free for the next ambi_batch_run at once.  For the
 * final path (which = 1) nothing is computed for it: the finish kernels write every unit's runs beside its cells, into one of two
 * blocks that alternate from run to run; which = 0 (the path before indelBFB) goes through the packing kernels first.  slot 0 / 1: two blocks, so that the copy of step i travels while
 * step i+1 computes.  ambi_batch_runs_wait blocks until that copy has arrived and describes it (pointers into the engine's pinned
 * block, valid until the slot is queued again or the batch is destroyed); bytes = what the payload needs, copied_bytes = what
 * the copy moved (the slot's capacity).  ambi_batch_runs_unit_path expands one unit's runs into cells (absolute signed ids):
 * exactly what ambi_batch_unit_path returns after a download. */
typedef struct {
    int64_t n_runs, n_cells, bytes, copied_bytes;
    const int32_t* lengths;      /* [n_units] cells of the path */
    const int32_t* run_counts;   /* [n_units] runs of the path */
    const int32_t* run_start;    /* first cell of a run (absolute signed segment id) */
    const int32_t* run_len;      /* cells of the run, counting up by one */
    const int64_t* run_off;      /* [n_units + 1] unit u's runs are entries run_off[u] .. run_off[u] + run_counts[u] - 1 of run_start / run_len
                                  * (the finish kernels leave every unit's runs in slots of its own: the entries are not packed) */
} ambi_runs_view_t;
int ambi_batch_runs_to_host(ambi_batch_t* b, int32_t which, int32_t slot, void* hip_stream);
int ambi_batch_runs_wait(ambi_batch_t* b, int32_t slot, ambi_runs_view_t* out);
int ambi_batch_runs_unit_path(ambi_batch_t* b, int32_t slot, int32_t unit, int32_t* out, int32_t cap);

/* Expands runs into cells: run r writes dev_cells[dev_cell_off[r] + k] = dev_run_start[r] + k, k < dev_run_len[r]
 * (dev_cell_off = exclusive prefix sum of the lengths, int64).  All pointers are device memory of the current device. */
int ambi_expand_runs(const int32_t* dev_run_start, const int32_t* dev_run_len, const int64_t* dev_cell_off, int64_t n_runs,
                     int32_t* dev_cells, int64_t cell_cap, void* hip_stream);

typedef struct {
    int32_t status;
    int32_t bias;             /* localhap.cpp:141-146 */
    int32_t n_nodes;          /* K */
    int32_t bkp_len;
    int32_t path_len;         /* getBFB result */
    int32_t path_indel_len;   /* after indelBFB */
    int32_t indel_printed;    /* reference prints the indel caption + path */
    int32_t n_out_junc;
    int32_t first_forward;    /* 1 forward seed / 0 reversed seed / -1 */
    int32_t evaluated;        /* order evaluations of the reference's sequential scan */
    int64_t num_orders;       /* R */
    int64_t first_valid;      /* index of the first valid order, -1 if none */
    double inv_cn_sum;        /* localhap.cpp:150-153 */
    int32_t path_indel_stored; /* 1: indelBFB changed the path (a second path is held); 0: it equals the getBFB path */
    int32_t reserved;
} ambi_unit_result_t;
int ambi_batch_unit_result(const ambi_batch_t* b, int32_t unit, ambi_unit_result_t* out);
/* which: 0 = getBFB path (LocalGenomicMap.cpp:3660-3671), 1 = after indelBFB (:3746-3837). Returns length or <0. */
int ambi_batch_unit_path(const ambi_batch_t* b, int32_t unit, int32_t which, int32_t* out, int32_t cap);
int ambi_batch_unit_bkp(const ambi_batch_t* b, int32_t unit, int32_t* out, int32_t cap);
/* per local segment id 0..n (slot 0 unused): junction CN (2 per id), CN after getIndelBias, targetCN, fold-back map */
int ambi_batch_unit_prepare(const ambi_batch_t* b, int32_t unit, double* junc_cn, double* seg_cn, int32_t* target_cn,
                            int32_t* inv_junc_global);
/* DAG of the unit: node2pat / node2loop as [K][3] (absolute ids; a==0 empty), successor bit masks [K] */
int ambi_batch_unit_dag(const ambi_batch_t* b, int32_t unit, int32_t* node2pat, int32_t* node2loop, uint64_t* succ);
/* successor sets of the unit's K nodes as [K][nwords] 64-bit words (bit j of word j/64): a unit may have up to 255 nodes (64..255: a
 * "wide" unit, served by the plain four-word form of the DAG / lattice stages of LocalGenomicMap.cpp:3276-3409; ambi_batch_unit_dag's
 * succ[] carries the low word).  ambi_batch_unit_dag_words = the same with nwords = 2 (enough up to 127 nodes; kept from round 3). */
int ambi_batch_unit_dag_nwords(const ambi_batch_t* b, int32_t unit, int32_t nwords, uint64_t* succ);
int ambi_batch_unit_dag_words(const ambi_batch_t* b, int32_t unit, uint64_t* succ2);
int ambi_batch_unit_out_juncs(const ambi_batch_t* b, int32_t unit, int32_t* u, int32_t* v, int32_t* count, int32_t cap);
/* --all (localhap.cpp:38, LocalGenomicMap.cpp:3672-3695): after ambi_batch_run(b, AMBI_FLAG_ALL, ...) + wait, every
 * valid order of a unit in the reference's print order.  pass 0 = the first orientation (forward unless
 * AMBI_FLAG_REVERSED), pass 1 = the flipped orientation, which the reference runs only when the LAST order of pass 0
 * is invalid (count 0 otherwise).  all_orders: indices into the order table; all_paths: the expanded paths (absolute
 * signed ids) of valid orders [first, first+count) of the pass, cells[j*stride ...], lengths[j] (negative: capacity).
 * The unit's ordinary results (first valid order, path, indelBFB, output junctions) are those of the default mode;
 * ambi_unit_result_t.evaluated counts every order of the executed passes. */
int ambi_batch_all_count(ambi_batch_t* b, int32_t unit, int32_t pass, int64_t* count);
int ambi_batch_all_orders(ambi_batch_t* b, int32_t unit, int32_t pass, int64_t first, int64_t count, int64_t* order_idx);
int ambi_batch_all_paths(ambi_batch_t* b, int32_t unit, int32_t pass, int64_t first, int64_t count, int32_t* lengths, int32_t* cells,
                         int64_t stride);
/* --all for ONE wide sample on several GPUs (orders of a chromosome block-partitioned over ranks).  Every
 * rank holds the same batch; ambi_batch_all_set_shard(rank, world) before ambi_batch_run(AMBI_FLAG_ALL) makes a rank evaluate
 * the 64-order chunks c with c % world == rank (plus the last chunk of every unit, so that every rank knows by itself
 * whether the orientation flips, LocalGenomicMap.cpp:3691-3695).  After ambi_batch_wait the ranks merge the pool that
 * ambi_batch_all_device names -- validity bitmaps and undefined-order flags, as BYTES with MAX (disjoint contributions,
 * zero elsewhere): one all-reduce over RCCL -- and call ambi_batch_all_finish, which recounts and finalises the headers;
 * ambi_batch_all_count / _orders / _paths then answer on every rank as after a single-GPU run. */
int ambi_batch_all_set_shard(ambi_batch_t* b, int32_t rank, int32_t world);
int ambi_batch_all_device(ambi_batch_t* b, void** ptr, int64_t* bytes);
int ambi_batch_all_finish(ambi_batch_t* b);
/* Rows [first,first+count) of the unit's order table -- the reference's `orders` (LocalGenomicMap.cpp:3380-3409), row r = the r-th
 * topological order it pushes -- as count x K uint8 node ids.  (On the device a row holds 5 bits per node up to 32 nodes, 6 up to 63,
 * a byte above; this call unpacks.) */
int ambi_batch_unit_orders(ambi_batch_t* b, int32_t unit, int64_t first, int64_t count, uint8_t* out);

/* Per-kernel timing (HIP events on the streams the kernels are launched on, milliseconds): the average duration of
 * ONE launch of the kernel over the runs since timing was enabled; names are static strings.  A run launches every
 * kernel once per slice (ambi_batch_slices: contiguous unit ranges whose kernel chains run on separate HIP streams so
 * that the HBM-bound and the latency-bound kernels overlap; set with the environment variable AMBI_SLICES, default
 * automatic).  Enable with ambi_batch_set_timing(b, 1) before run. */
int ambi_batch_slices(const ambi_batch_t* b);
int ambi_batch_set_timing(ambi_batch_t* b, int32_t on);
/* The same for a subset of the kernels: bit k of `mask` = kernel index k of ambi_batch_kernel_time (0 prepare, 1 plan,
 * 2 image build, 3 enumerate, 4 scan, 5 finish (lean + list), 6 direct full finish); the others report -1.  Every pair of events is a marker in the
 * stream, and twelve of them per run cost ~4 % on the bench workload -- a timed region that needs one kernel's duration
 * asks for that kernel only. */
int ambi_batch_set_timing_mask(ambi_batch_t* b, uint32_t mask);
int ambi_batch_kernel_count(const ambi_batch_t* b);
int ambi_batch_kernel_time(const ambi_batch_t* b, int32_t idx, const char** name, float* ms);
/* Where the kernel sits in its run: start / end in ms from the start of the run's first kernel (mean over the timed runs; the
 * kernels run on several streams side by side, so these spans -- not the durations -- give the step's critical path); -1 when the
 * first kernel is not among the timed ones. */
int ambi_batch_kernel_span(const ambi_batch_t* b, int32_t idx, float* start_ms, float* end_ms);
/* bytes: inputs resident in HBM, order-table bytes written by the last run, result blob bytes */
int ambi_batch_traffic(const ambi_batch_t* b, int64_t* input_bytes, int64_t* order_bytes, int64_t* result_bytes);

/* ------------------------------------------------------------------------------------------------
 * Whole-sample helpers used by the CLI (host side, tiny): path text (printBFB, LocalGenomicMap.cpp:3411-3429),
 * BFB-TRX stitching (translocationBFB, :4052-4193), output-junction merge (localhap.cpp:267-316).
 * ---------------------------------------------------------------------------------------------- */
int64_t ambi_format_path(const ambi_graph_t* g, const int32_t* path, int32_t len, char* buf, int64_t cap);
/* paths: concatenated per-chromosome paths with offsets[n_chr+1]; result written to out (cap cells). Returns length. */
int ambi_translocation_bfb(const ambi_graph_t* g, int32_t* paths, const int64_t* offsets, int32_t n_chr, int32_t* out, int32_t cap);

/* ------------------------------------------------------------------------------------------------
 * ILP model of one chromosome (host side; the solve stays with the external `cbc`, localhap.cpp:179-181).
 * Replaces LocalGenomicMap::BFB_ILP (LocalGenomicMap.cpp:4397-4752): same rows, same row order, same in-row entry
 * order, generated in O(nnz) instead of the reference's O(n * numPat^2) assembly loop (:4464-4477).
 * seg_cn / junc_cn: the arrays ambi_batch_unit_prepare returns for the chromosome (local ids, slot 0 unused);
 * max_cn_total: sum of the CN of ALL segments of the graph at that point (:4708-4711).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ambi_ilp ambi_ilp_t;
int ambi_ilp_build(const ambi_graph_t* g, int32_t chr, const double* seg_cn, const double* junc_cn, int32_t bias,
                   double max_cn_total, int32_t juncs_info, ambi_ilp_t** out);
/* The same model with its entries written ON THE DEVICE: the host lists the rows (O(rows)),
 * ambi_ilp_fill_kernel writes the 12 bytes per non-zero (int32 column + f64 coefficient) with one thread per entry,
 * the arrays come back to the host.  Bit-identical to ambi_ilp_build.  kernel_ms (optional): mean device time of the
 * fill kernel, for the roofline (algorithmic bytes = 12 * nnz). */
int ambi_ilp_build_device(const ambi_graph_t* g, int32_t chr, const double* seg_cn, const double* junc_cn, int32_t bias,
                          double max_cn_total, int32_t juncs_info, float* kernel_ms, ambi_ilp_t** out);
/* Joint model of `--op sc_bfb` (LocalGenomicMap::BFB_ILP_SC, LocalGenomicMap.cpp:4754-5093) for chromosome `chr` of
 * n_graphs graphs with the same segmentation: seg_cn / fold_cn are n_graphs x n (graph-major; n = segments of the
 * chromosome): the segment copy numbers (first graph after its getIndelBias, localhap.cpp:497) and the fold-back copy
 * numbers juncCN[i][1] of every graph's own getJuncCN (LocalGenomicMap.cpp:4794).  Linking rows for every pair of graphs
 * (localhap.cpp:430-434).  The matrix is the one the reference builds, including its use of the running row counter in
 * the epsilon columns (LocalGenomicMap.cpp:4815) -- see DESIGN.md. */
int ambi_ilp_build_sc(const ambi_graph_t* g0, int32_t chr, int32_t n_graphs, const double* seg_cn, const double* fold_cn, ambi_ilp_t** out);
void ambi_ilp_destroy(ambi_ilp_t* p);
int ambi_ilp_sizes(const ambi_ilp_t* p, int64_t* n_rows, int64_t* nnz, int32_t* n_cols, int32_t* n_int);
/* CSR copy-out; infinity is +-DBL_MAX (OsiClp getInfinity()); any pointer may be NULL */
int ambi_ilp_copy(const ambi_ilp_t* p, int64_t* row_ptr, int32_t* col, double* val, double* row_lo, double* row_up,
                  double* col_lo, double* col_up, double* obj);
/* writes <path> as CPLEX-LP text with CoinUtils column names x<j> (what `cbc <prefix>.lp solve solu <prefix>.sol` reads) */
int ambi_ilp_write_lp(const ambi_ilp_t* p, const char* path);
/* <prefix>.mps beside <prefix>.lp, as the reference leaves it (LGM.cpp:4749): free-format MPS of the same model. */
int ambi_ilp_write_mps(const ambi_ilp_t* p, const char* path);

#ifdef __cplusplus
}
#endif
#endif /* AMBIGRAM_HIP_H */
