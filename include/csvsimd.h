/*
 * csvsimd.h — C ABI of libcsvsimd_hip.so: MI355X (gfx950) stage-1 CSV structural indexer.
 *
 * Drop-in boundary for the reference crate's stage 1.  The reference has no FFI of its own; its
 * seam is the Rust function
 *
 *     pub fn read(memmap: &Mmap) -> StructureIndex            (reference src/reader.rs:150)
 *
 * parameterised per ISA by `trait Stage1<T>` (src/stage1.rs:156-167), called from
 * `csv_simd::create` (src/lib.rs:61-74) which then builds `TapeCore`/`Tape`
 * (src/tape.rs:303-347, 83-94).  Every entry point below cites the reference interface it
 * replaces.  INTEGRATION.md shows the ~30-line Rust `extern "C"` binding that swaps
 * `reader::read` for `csvsimd_stage1_index`.
 *
 * Conventions: plain pointers and sizes; caller allocates and frees every buffer; no callbacks;
 * thread-safe for distinct contexts; all functions return 0 or a negative CSVSIMD_ERR_* code.
 * There is NO CPU fallback: without a HIP device every compute entry point fails with
 * CSVSIMD_ERR_NO_DEVICE.
 */
#ifndef CSVSIMD_H
#define CSVSIMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes: -1..-4 mirror reference StructureError (src/error.rs:7-21) ------------- */
#define CSVSIMD_OK 0
#define CSVSIMD_ERR_IO (-1)                 /* StructureError::Io               */
#define CSVSIMD_ERR_MISSING_VALUE (-2)      /* StructureError::MissingValue     */
#define CSVSIMD_ERR_INVALID_STATE (-3)      /* StructureError::InvalidState     */
#define CSVSIMD_ERR_INVALID_CSV_FORMAT (-4) /* StructureError::InvalidCsvFormat */
#define CSVSIMD_ERR_INVALID_ARG (-9)
#define CSVSIMD_ERR_TAPE_CAPACITY (-11) /* tape_cap too small; *tape_len holds the needed size */
#define CSVSIMD_ERR_HIP (-12)           /* HIP runtime error: csvsimd_last_error() has the text */
#define CSVSIMD_ERR_NO_DEVICE (-13)
#define CSVSIMD_ERR_INTERNAL (-14) /* in-kernel look-back spin bound hit (~1 s without progress of a
                                      predecessor tile: never seen; conceivable only if the GPU is
                                      preempted for that long by another process — retry); also a host-side
                                      C++ exception (out of host memory, thread creation) caught at the ABI:
                                      csvsimd_last_error() has the text */
#define CSVSIMD_ERR_RCCL (-15)     /* librccl missing or a collective failed */

const char* csvsimd_strerror(int code);
const char* csvsimd_last_error(void); /* thread-local text of the last HIP failure */
int csvsimd_device_count(void);       /* number of HIP devices visible, 0 if none */
uint32_t csvsimd_abi_version(void); /* 4: frequency counts take n_records / scratch_bytes, + _async, ingest plan / phases (3: + columnar consumers, copy yardstick; 2: csvsimd_stitch grew an error field;
                                       device-side stitch / re-emit).  Bindings should check it when they load the library */
/* bytes one workgroup indexes per look-back step (informational: tests probe sizes around it) */
uint32_t csvsimd_tile_bytes(void);

/* ---- context: one per (thread, device); owns the look-back scratch.  All launches that use one
 * context must be ordered with respect to each other (same stream, or explicit dependencies): they
 * share the scratch block.  Use several contexts for concurrent streams. ---------------------- */
typedef struct csvsimd_ctx csvsimd_ctx;
int csvsimd_ctx_create(int device, csvsimd_ctx** out);
void csvsimd_ctx_destroy(csvsimd_ctx* ctx);


/* What one stage-1 pass over a shard reports (device- or host-resident, 64 bytes).
 * (quote_parity, count_enter_outside, count_enter_inside) is the composable shard descriptor
 * of SURVEY.md §8e; the reference carries the same two quantities between 64-byte blocks as
 * `inside_str` and `array_idx` (src/reader.rs:217-218). */
typedef struct csvsimd_shard_result {
    uint64_t count;               /* structural entries under the entering state of this pass (in_quote_in_used) */
    uint64_t count_enter_outside; /* ... had the shard been entered outside a quoted string   */
    uint64_t count_enter_inside;  /* ... had it been entered inside one                       */
    uint32_t quote_parity;        /* number of '"' bytes in the shard, mod 2                  */
    uint32_t in_quote_out;        /* in_quote_in_used ^ quote_parity                          */
    uint32_t error;               /* 0, or CSVSIMD_ERR_INTERNAL's in-kernel flag              */
    uint32_t escape_out;          /* escape dialects only: the byte after the shard is escaped */
    uint64_t written;             /* min(count, tape_cap): entries actually stored            */
    uint32_t in_quote_in_used;    /* the entering state this pass ran with: the one passed, or — with
                                     CSVSIMD_ENTER_GUESS — the one the kernel chose itself       */
    uint32_t reserved0;
    uint64_t reserved1;
} csvsimd_shard_result;

/* in_quote_in of the device entry points: 0 = the shard starts outside a quoted string, 1 = inside one,
 * CSVSIMD_ENTER_GUESS = unknown (a shard cut out of the middle of a file, README.md:24 "splitting work without first
 * knowing record breaks"): the kernel indexes the shard under the entering state for which its first EIGHT tiles
 * (2 MiB; their quote parities and counts composed in order) hold more entries — read with the wrong quote parity, text
 * outside strings looks quoted and nearly every separator disappears — and reports the choice in in_quote_in_used.  `count`, `in_quote_out` and the tape are those of that
 * state; the two hypothesis counts and quote_parity are state independent as always.  A wrong guess is found by the
 * stitch (csvsimd_stitch.reemit) and costs the re-emit launch, exactly like a wrong in_quote_in = 0 speculation.
 * Progress (round 5): a GUESS launch needs no particular number of resident workgroups.  A workgroup that holds a counted
 * tile it cannot resolve yet, because the choice does not exist, gives that tile up (its aggregate and vote are in; it is
 * counted again later) and draws the next ticket — the next voter — so even a grid of ONE workgroup completes the vote by
 * itself (tests/test_gpu_guess_small_grid.py: 1 and 2 workgroups, both geometries, three contexts sharing the GPU).  Until
 * round 4 the vote needed 4 (dense geometry: 16) workgroups of the launch resident at once. */
#define CSVSIMD_ENTER_OUTSIDE 0u
#define CSVSIMD_ENTER_INSIDE 1u
#define CSVSIMD_ENTER_GUESS 2u

/* ---- stage 1, device-resident (the timed path) ---------------------------------------------
 * Replaces the hot loop of reader::read (src/reader.rs:229-290) = SimdInput::structure
 * (src/avx/stage1.rs:193-430) + Stage1::crush_set_bits (src/stage1.rs:162-296) for the bytes
 * dbuf[0..len): writes, ascending, base_off + i for every i with dbuf[i] in {',', CR, LF} and
 * an even number of '"' at positions <= i (counting in_quote_in as one earlier quote), as
 * uint64_t (reference CodeUnitPos(usize), src/stage1.rs:67) to dtape[0..min(count,tape_cap)).
 * No sentinel is written here (the reference's leading 0, src/reader.rs:216, belongs to the
 * whole file, not to a shard).  dtape may be NULL with tape_cap 0 for a count-only pass.
 * Asynchronous on hip_stream (a hipStream_t, NULL = default stream); d_result is DEVICE memory
 * (>= sizeof(csvsimd_shard_result), 16-byte aligned), valid once the stream has drained.  No
 * allocation and no synchronisation happens inside as long as the context scratch is large
 * enough for len (grow it up front with csvsimd_ctx_reserve), so the call may be captured
 * into a hipGraph.  Any dbuf alignment is accepted; 128-byte alignment (any allocator's) is fastest,
 * other alignments cost ~2 % (wave loads then straddle L2 lines).  dtape needs 8-byte alignment only:
 * the kernel aligns its wave stores to 128-byte lines itself. */
int csvsimd_ctx_reserve(csvsimd_ctx* ctx, uint64_t max_len);
int csvsimd_stage1_index_device_async(csvsimd_ctx* ctx, const void* dbuf, uint64_t len,
                                      uint64_t base_off, uint32_t in_quote_in, void* dtape,
                                      uint64_t tape_cap, void* d_result, void* hip_stream);
/* Same, synchronous, result copied to host. Returns CSVSIMD_ERR_TAPE_CAPACITY if
 * count > tape_cap and dtape != NULL (result still filled). */
int csvsimd_stage1_index_device(csvsimd_ctx* ctx, const void* dbuf, uint64_t len,
                                uint64_t base_off, uint32_t in_quote_in, void* dtape,
                                uint64_t tape_cap, csvsimd_shard_result* result, void* hip_stream);

/* ---- stage 1, many device-resident buffers in ONE launch --------------------------------------------------------
 * A launch has a fixed cost of ~20 us (its fill and drain: DESIGN.md); a 128-MiB buffer indexed alone runs at 40 % of
 * HBM read bandwidth where a 1-GiB one reaches 60 %.  Many files — csv_simd::create per file, src/lib.rs:61-74, is the
 * reference's unit of work — should not pay it per file: n_items independent buffers (any sizes, any alignments, each
 * with its own tape, base offset and entering state 0 / 1) are indexed by ONE persistent kernel whose workgroups draw
 * tiles of all buffers from one ticket; a tile's look-back stops at its buffer's first tile.  d_results = n_items
 * csvsimd_shard_result records in DEVICE memory (16-byte aligned), record i for items[i], valid once the stream has
 * drained: exactly what csvsimd_stage1_index_device_async would have written for that buffer alone (same tape, same
 * counts).  items is HOST memory (read before the call returns).  Reference dialect only.  Asynchronous on hip_stream:
 * nothing it enqueues is waited for; it allocates / synchronises only when the context's scratch or its buffer table has
 * to grow, and it may wait for the table upload of the call before the previous one (two staging blocks alternate).
 * Not capturable into a hipGraph (the table travels by a host-staged copy): capture
 * csvsimd_stage1_index_device_async launches instead. */
typedef struct csvsimd_batch_item {
    const void* dbuf;
    uint64_t len;
    uint64_t base_off;
    void* dtape;
    uint64_t tape_cap;
    uint32_t in_quote_in; /* CSVSIMD_ENTER_OUTSIDE / CSVSIMD_ENTER_INSIDE */
    uint32_t reserved;
} csvsimd_batch_item;
int csvsimd_stage1_index_batch_device_async(csvsimd_ctx* ctx, const csvsimd_batch_item* items, uint32_t n_items,
                                            void* d_results, void* hip_stream);

/* ---- stage 1, host buffer: the drop-in for `reader::read(&Mmap) -> StructureIndex` -----------
 * (src/reader.rs:150).  buf = the mmap; tape = caller-owned uint64_t[tape_cap]; on return
 * tape[0] = 0 (sentinel, src/reader.rs:216) followed by every structural offset relative to
 * buf[0]; *tape_len = entries including the sentinel (== StructureIndex.len()).
 * csvsimd_stage1_bound gives a capacity that can never be exceeded (len + 1).  If tape_cap is
 * too small nothing past it is written, *tape_len holds the exact size needed and the call
 * returns CSVSIMD_ERR_TAPE_CAPACITY (retry once).  tape == NULL, tape_cap == 0 is a count-only
 * call.  Streams the file through pinned staging in chunks; accepts any len >= 0 (the reference
 * itself is only defined for len >= 64). */
int csvsimd_stage1_bound(uint64_t len, uint64_t* max_entries);
int csvsimd_stage1_index(csvsimd_ctx* ctx, const uint8_t* buf, uint64_t len, uint64_t* tape,
                         uint64_t tape_cap, uint64_t* tape_len, uint32_t* in_quote_out);

/* ---- stage 1, MANY host buffers in one call (round 5) -------------------------------------------
 * The reference's unit of work is a file (csv_simd::create, src/lib.rs:61-74; its own inputs, the .csv files under res/, are 96 to 623
 * bytes).  One csvsimd_stage1_index call per small file pays a launch and a wait each (30 us against 1.3 us on one CPU
 * core for res/sample.csv); this call indexes n files for one pipeline's worth of them: files are packed into pinned
 * groups of up to 4 MiB, each group crosses PCIe as one copy and is indexed by ONE batched launch (every file a buffer of
 * its own, entered outside a quoted string; a quote left open in one file does not leak into the next), tapes and records
 * come back as the kernel's own stores.  Per item, exactly what csvsimd_stage1_index(buf, len, tape, tape_cap) would have
 * produced: tape[0] = 0, then the structural offsets relative to buf[0]; tape_len = entries incl. the sentinel;
 * status = CSVSIMD_OK, or CSVSIMD_ERR_TAPE_CAPACITY when tape_cap is too small (tape_len then holds the size needed and
 * nothing past tape_cap was written).  tape == NULL, tape_cap == 0: count only.  Files above 1 MiB (and files with more
 * than an entry per 4 bytes) take csvsimd_stage1_index's own path inside the call.  Returns CSVSIMD_OK,
 * CSVSIMD_ERR_TAPE_CAPACITY if any item reported it (all others are complete), or an error that voids the whole call.
 * Reference dialect.  items is read and written by the call; buffers and tapes stay the caller's. */
typedef struct csvsimd_host_batch_item {
    const uint8_t* buf;
    uint64_t len;
    uint64_t* tape;
    uint64_t tape_cap;
    uint64_t tape_len;      /* out */
    uint32_t in_quote_out;  /* out */
    int32_t status;         /* out */
} csvsimd_host_batch_item;
int csvsimd_stage1_index_batch(csvsimd_ctx* ctx, csvsimd_host_batch_item* items, uint32_t n_items);

/* How csvsimd_stage1_index cuts a buffer of `len` bytes into the chunks it streams through the GPU (the host side owns
 * the chunking; reference: csv_simd::create maps the whole file and streaming is a TODO, src/lib.rs:61-74, README.md:23):
 * cuts[0] = 0 < cuts[1] < ... = len, chunk i = [cuts[i], cuts[i + 1]).  No chunk exceeds the 32-MiB slot; the plan starts
 * at len / 32 (1 ... 4 MiB), doubles up to full slots and halves down to len / 16 (1 ... 8 MiB) — a call waits for the
 * staging of its first chunk and for the way back of its last one, everything between overlaps — and no chunk of a
 * multi-chunk plan is shorter than 512 KiB (a stub is folded into its neighbour).  2 GiB: 4, 8, 16, 32 ... 32, 16, 8 MiB;
 * 32 MiB: 1, 2, 4, 8, 3, 8, 4, 2; 4 MiB: 1, 2, 1.  *n_cuts = entries needed; CSVSIMD_ERR_TAPE_CAPACITY if cap is smaller
 * (nothing is written). */
int csvsimd_ingest_chunk_plan(uint64_t len, uint64_t* cuts, uint64_t cap, uint64_t* n_cuts);

/* Where the wall time of the calling thread's most recent csvsimd_stage1_index[_dialect] call went, by thread of its
 * pipeline (seconds).  A file of three or more chunks runs on three host threads — stager (user buffer -> pinned
 * slots, ahead of the H2D copies; the first chunk is staged by the caller), submitter (the caller: H2D copies, launches,
 * polling the records the packing kernel publishes in pinned memory), expander (32-bit offsets in the pinned slots -> the
 * caller's tape; the last chunk is expanded by the caller); a shorter one runs the same steps in turn on the caller's
 * thread.  The two extra threads belong to the context and sleep between calls. */
typedef struct csvsimd_ingest_phases {
    uint64_t bytes, chunks;
    uint32_t host_threads;       /* 3 or 1 */
    uint32_t reserved;
    double wall;                 /* the whole call */
    double stage_copy;           /* stager: copying into the pinned slots ... */
    double stage_wait;           /* ... and waiting for a slot to come free (H2D of the chunk four before) */
    double expand_copy;          /* expander: widening offsets into the caller's tape */
    double submit;               /* submitter: enqueueing copies, kernels, events */
    double wait_staged;          /* submitter: waiting for the stager (the link idles meanwhile unless copies are queued) */
    double wait_record;          /* submitter: waiting for a chunk's result record (H2D + kernels two chunks back) */
    double wait_expanded;        /* submitter: waiting for the expander to release a pinned tape slot */
} csvsimd_ingest_phases;
int csvsimd_ingest_last_phases(csvsimd_ingest_phases* out);

/* ---- dialect extension -------------------------------------------------
 * NOT reference behaviour: the reference hard-wires ',' and '"' (src/avx/stage1.rs:392-394);
 * its class table already knows backslash and space (src/stage1.rs:41-48) but nothing uses them
 * and escapes are a README TODO (README.md:32).  The entry points above never take this path, so
 * the default stays bit-exact with reader::read.  Semantics (checked against
 * oracle_dialect_index): the byte after an unescaped `escape` byte is literal; an unescaped
 * `quote` byte toggles the in-string state; an unescaped `delimiter`, CR or LF outside a string
 * is structural.  quote = 0 / escape = 0 switch that feature off.  delimiter must be non-zero and
 * all special bytes pairwise distinct (CSVSIMD_ERR_INVALID_ARG otherwise).
 * Shards: escape_in = "the first byte of this buffer is escaped" (the buffer before it ended in an
 * odd run of escape bytes = that shard's result.escape_out).  Unlike the quote state it cannot be
 * speculated away, but it is known from one byte run at the shard boundary before launching. */
typedef struct csvsimd_dialect {
    uint8_t delimiter; /* ',' */
    uint8_t quote;     /* '"', 0 = no quoting */
    uint8_t escape;    /* 0 = none (reference), e.g. '\\' */
    uint8_t escape_in; /* 0 / 1 */
    uint32_t reserved; /* 0 */
} csvsimd_dialect;
int csvsimd_dialect_init(csvsimd_dialect* d); /* fills in the reference dialect */
int csvsimd_stage1_index_device_dialect_async(csvsimd_ctx* ctx, const csvsimd_dialect* dialect,
                                              const void* dbuf, uint64_t len, uint64_t base_off,
                                              uint32_t in_quote_in, void* dtape, uint64_t tape_cap,
                                              void* d_result, void* hip_stream);
int csvsimd_stage1_index_dialect(csvsimd_ctx* ctx, const csvsimd_dialect* dialect, const uint8_t* buf,
                                 uint64_t len, uint64_t* tape, uint64_t tape_cap, uint64_t* tape_len,
                                 uint32_t* in_quote_out);

/* Entries per input byte of the data this context is about to index (entries structural bytes in `bytes` bytes of
 * text like it; bytes == 0: forget).  The device entry points choose between two instantiations of the stage-1 kernel
 * by it — above 0.125 (delimiter-dense files: BASELINE config 5 holds 0.2) the one whose emit path spends fewer
 * instructions per entry — and produce the same tape either way (reference: crush_set_bits is one routine for every
 * density, src/stage1.rs:162-296).  The synchronous entry points and csvsimd_stage1_index learn the figure themselves
 * from the records they read; a caller of the _async entry points, which return before any record exists, may say it.
 * csvsimd_ctx_kernel_name: the kernel the context's next emitting launch of `dialect` (NULL: the reference's) runs. */
int csvsimd_ctx_hint_density(csvsimd_ctx* ctx, uint64_t entries, uint64_t bytes);
/* Test / diagnostic knob (round 5): caps the grid of every following stage-1 launch of this context at n workgroups
 * (0 = back to the default, two per CU).  Results never depend on it — every launch, CSVSIMD_ENTER_GUESS included, makes
 * progress with any number of resident workgroups — which is what tests/test_gpu_guess_small_grid.py uses it to show. */
int csvsimd_ctx_limit_workgroups(csvsimd_ctx* ctx, uint32_t n);
const char* csvsimd_ctx_kernel_name(const csvsimd_ctx* ctx, const csvsimd_dialect* dialect);


/* ---- multi-GPU stitch (host arithmetic; the exchange itself is one all-gather of these
 * descriptors over RCCL, done by the caller's communicator) -----------------------------------
 * New relative to the reference (single-threaded; README.md:24 lists it as a TODO).  Given the
 * per-shard results of a first pass in rank order — each run with in_quote_in = 0 (speculation), 1, or
 * CSVSIMD_ENTER_GUESS (the kernel's own choice, reported in in_quote_in_used) — computes for shard
 * `rank` its true entering state, whether that differs from the state its pass ran with (reemit), and the
 * global tape index of its first entry (sentinel included), plus the whole-file totals. */
typedef struct csvsimd_stitch {
    uint32_t in_quote_in;      /* TRUE entering state of this shard (the re-emit launch reads it) */
    uint32_t in_quote_final;   /* state after the last shard                            */
    uint64_t count;            /* this shard's entry count under its true entering state */
    uint64_t tape_index_base;  /* global index of this shard's first entry (>= 1)       */
    uint64_t total_entries;    /* whole file, sentinel included                         */
    uint32_t error;            /* 1 if any shard's record carried its error flag        */
    uint32_t reemit;           /* 1 if this shard's pass ran with another entering state than the true one
                                  (results[rank].in_quote_in_used != in_quote_in): its tape must be re-emitted */
} csvsimd_stitch;
int csvsimd_stitch_shards(const csvsimd_shard_result* results, uint32_t n_shards, uint32_t rank,
                          uint32_t file_in_quote_in, csvsimd_stitch* out);
/* The same arithmetic as a one-lane kernel on hip_stream: d_results = n_shards records in DEVICE memory
 * (the all-gather's receive buffer), d_stitch = DEVICE csvsimd_stitch (8-byte aligned).  With
 * csvsimd_stage1_reemit_device_async below the sharded step never waits for the host:
 *     first csvsimd_stage1_index_device_async (rank 0: the file's entering state; other ranks:
 *     CSVSIMD_ENTER_GUESS, or 0 to speculate)  ->  all-gather of the records
 *     ->  csvsimd_stitch_shards_device_async  ->  csvsimd_stage1_reemit_device_async
 * csvsimd_stage1_reemit_device_async is csvsimd_stage1_index_device_async whose in_quote_in is read on the
 * device from d_stitch when the kernel starts: if d_stitch->reemit is 0 the launch returns at once and
 * leaves tape and d_result exactly as the first pass wrote them (which is then final); if it is 1
 * the shard is indexed again, for real, with d_stitch->in_quote_in (README.md:24 of the
 * reference: "requires toggling interpretation if/when start in quoted text"). */
int csvsimd_stitch_shards_device_async(const void* d_results, uint32_t n_shards, uint32_t rank,
                                       uint32_t file_in_quote_in, void* d_stitch, void* hip_stream);
int csvsimd_stage1_reemit_device_async(csvsimd_ctx* ctx, const void* dbuf, uint64_t len, uint64_t base_off,
                                       const void* d_stitch, void* dtape, uint64_t tape_cap,
                                       void* d_result, void* hip_stream);

/* Native form of the same step for hosts without torch.distributed: one communicator per rank
 * (one process per GPU).  Rank 0 obtains an id (ncclGetUniqueId) and hands its 128 bytes to the
 * other ranks by whatever channel the application has; every rank then creates its communicator
 * (ncclCommInitRank).  csvsimd_stage1_index_sharded = first pass (rank 0: file_in_quote_in; other ranks:
 * CSVSIMD_ENTER_GUESS) -> ONE ncclAllGather of the 64-byte result records over xGMI -> stitch on the device -> re-emit
 * launch that does nothing unless this rank's pass ran with the wrong entering state -> the final record and the
 * stitch are copied to the host; one synchronisation at the very end.  The tape stays sharded (entries
 * of this rank's bytes, absolute offsets).  Every rank reaches the collective even if its own launch
 * fails (it contributes a record with the error flag set and all ranks return an error).  RCCL is
 * resolved with dlopen at first use: CSVSIMD_ERR_RCCL if it is absent. */
#define CSVSIMD_COMM_ID_BYTES 128
typedef struct csvsimd_comm csvsimd_comm;
int csvsimd_comm_unique_id(uint8_t id[CSVSIMD_COMM_ID_BYTES]);
int csvsimd_comm_create(const uint8_t id[CSVSIMD_COMM_ID_BYTES], int rank, int world, int device,
                        csvsimd_comm** out);
void csvsimd_comm_destroy(csvsimd_comm* comm);
int csvsimd_stage1_index_sharded(csvsimd_ctx* ctx, csvsimd_comm* comm, const void* dbuf, uint64_t len,
                                 uint64_t base_off, uint32_t file_in_quote_in, void* dtape,
                                 uint64_t tape_cap, csvsimd_shard_result* result,
                                 csvsimd_stitch* stitch, void* hip_stream);

/* ---- one host file -> the GPUs of ONE process ----------------------------------------------------------------------
 * csv_simd::create maps a file and hands the whole mapping to reader::read (src/lib.rs:61-74); streaming and splitting
 * are README TODOs (README.md:23-24).  Here the host side owns the chunking: the buffer is cut into n_shards contiguous
 * byte ranges (csvsimd_multi_shard_range: cut i at i * len / n_shards, rounded down to 64 bytes), shard g is streamed
 * into shards[g].dbuf on the device of shards[g].ctx (one host thread, one H2D stream and two pinned staging slots per
 * shard, all shards concurrently), indexed there — shard 0 under the file's entering state, every other shard under the
 * state its own first tiles speak for (CSVSIMD_ENTER_GUESS) — the G result records are stitched (csvsimd_stitch_shards)
 * and only a shard that guessed wrong is indexed again.  The bytes and the tapes STAY on the devices, sharded in order
 * (absolute offsets: concatenated behind the sentinel 0 they are reader::read's index), ready for the consumers above.
 * In: ctx (one context per shard; contexts may share a device), dbuf (>= the shard's bytes), dtape / tape_cap.
 * The call works on the contexts' PRIVATE streams: dbuf and dtape must be idle when it starts (work the caller has
 * enqueued on them — a fill, a previous consumer — must have completed: synchronise first), and are complete when it
 * returns.  The calling thread's current HIP device is left as it was found.
 * Out: begin / end (the shard's byte range), result (its final record), stitch (entering state, tape_index_base,
 * totals).  CSVSIMD_ERR_TAPE_CAPACITY if some shard's tape_cap was too small (its result.count says what it needs).
 * Multi-process jobs (one rank per GPU, torch.distributed / RCCL) use csvsimd_stage1_index_sharded instead. */
typedef struct csvsimd_multi_shard {
    csvsimd_ctx* ctx;
    void* dbuf;
    void* dtape;
    uint64_t tape_cap;
    uint64_t begin, end;
    csvsimd_shard_result result;
    csvsimd_stitch stitch;
} csvsimd_multi_shard;
int csvsimd_multi_shard_range(uint64_t len, uint32_t n_shards, uint32_t i, uint64_t* begin, uint64_t* end);
int csvsimd_stage1_index_multi(const uint8_t* buf, uint64_t len, csvsimd_multi_shard* shards, uint32_t n_shards,
                               uint32_t file_in_quote_in);

/* ---- tape: host-side, after stage 1 (reference src/tape.rs, src/record_source.rs) ------------ */
typedef struct csvsimd_tape csvsimd_tape;
#define CSVSIMD_NEWLINE_LF 0   /* NewLine::LF   (src/stage1.rs:470-480) */
#define CSVSIMD_NEWLINE_CRLF 1 /* NewLine::CRLF */

/* Header::new (src/tape.rs:226-273) + TapeCore::create/init + Tape::from_core
 * (src/tape.rs:303-347, 83-94).  Borrows bytes and index (they must outlive the tape).
 * Fails with CSVSIMD_ERR_INVALID_CSV_FORMAT when (index_len-1) % jump != 0. */
int csvsimd_tape_create(const uint8_t* bytes, uint64_t len, const uint64_t* index,
                        uint64_t index_len, csvsimd_tape** out);
void csvsimd_tape_destroy(csvsimd_tape* t);
uint32_t csvsimd_tape_field_cnt(const csvsimd_tape* t);        /* Header.field_cnt      */
uint32_t csvsimd_tape_record_cnt(const csvsimd_tape* t);       /* Tape.record_cnt (u32) */
uint64_t csvsimd_tape_record_jump_size(const csvsimd_tape* t); /* Tape.record_jump_size */
uint32_t csvsimd_tape_record_offset(const csvsimd_tape* t);    /* Header.record_offset  */
int csvsimd_tape_new_line(const csvsimd_tape* t);              /* CSVSIMD_NEWLINE_*     */
/* header name i, trimmed (src/tape.rs:259-262); returns length, copies <= cap bytes */
int64_t csvsimd_tape_header_name(const csvsimd_tape* t, uint32_t i, char* dst, uint64_t cap);
/* RecordSource::seek_record / seek_field (src/record_source.rs:70-140): [begin, end) delimit
 * the field/record text in bytes; found == 0 mirrors Ok(None). */
int csvsimd_tape_seek_record(const csvsimd_tape* t, uint32_t record_idx, uint64_t* begin,
                             uint64_t* end, int* found);
int csvsimd_tape_seek_field(const csvsimd_tape* t, uint32_t record_idx, uint32_t field_idx,
                            uint64_t* begin, uint64_t* end, int* found);
/* boundaries(task_size, job_count) (src/tape.rs:385-428): writes <= job_count (start,len)
 * pairs, *n_out = how many; task_size == 0 or job_count == 0 -> CSVSIMD_ERR_INVALID_STATE
 * (the reference returns None, which Tape::chunks maps to InvalidState, src/tape.rs:99-100). */
typedef struct csvsimd_boundary {
    uint64_t start, len;
} csvsimd_boundary;
int csvsimd_boundaries(uint32_t task_size, uint8_t job_count, csvsimd_boundary* out,
                       uint32_t* n_out);
/* Tape::chunks(num) (src/tape.rs:95-140): (id, start, end, record_cnt) in index-key units */
typedef struct csvsimd_chunk {
    uint8_t id;
    uint64_t start, end; /* KeyToPos */
    uint32_t record_cnt;
} csvsimd_chunk;
int csvsimd_tape_chunks(const csvsimd_tape* t, uint8_t num, csvsimd_chunk* out, uint32_t* n_out);

/* csv_simd::create(filename) (src/lib.rs:61-74): open + mmap + Header::new + stage 1 on the
 * GPU + tape.  The returned tape owns the mapping and the index. */
int csvsimd_create(csvsimd_ctx* ctx, const char* filename, csvsimd_tape** out);
const uint64_t* csvsimd_tape_index(const csvsimd_tape* t, uint64_t* index_len);
const uint8_t* csvsimd_tape_bytes(const csvsimd_tape* t, uint64_t* len);

/* ---- device-side consumers of a finished, device-resident tape ---------
 * The reference's plan for the tape: "use the result to run frequency counts, and function search"
 * (design_notes_1.md:1-4), in parallel over the `Chunk {start, end, record_cnt}` records of Tape::chunks
 * ("Atomic representation of how to utilize the tape in a parallel-processing context", src/tape.rs:12-19,
 * 95-140).  The reference stops at the chunk list; these entry points are the consumers, driven by the very
 * csvsimd_chunk records csvsimd_tape_chunks returns.  Everywhere: dindex = the tape WITH its leading sentinel
 * (dindex[0] == 0), index_len entries, in device memory; field / record arithmetic is
 * RecordSource::seek_field's (src/record_source.rs:106-140); records are numbered as seek_field numbers them
 * (0 = first data row).  A ragged index -> CSVSIMD_ERR_INVALID_CSV_FORMAT (TapeCore::init,
 * src/tape.rs:327,342-344); a chunk that is not whole rows of this tape -> CSVSIMD_ERR_INVALID_ARG.
 *
 * Bulk seek_field: for records [first_record, first_record + n_records) and field field_idx writes
 * d_begin[i], d_end[i] such that bytes[begin..end) is the field text — the same pair
 * csvsimd_tape_seek_field returns.  *n_valid = how many of the requested records exist (seek_field's
 * Ok(None) cases are simply not written). */
int csvsimd_tape_field_spans_device(const void* dindex, uint64_t index_len, uint32_t field_cnt, int new_line,
                                    uint32_t field_idx, uint64_t first_record, uint64_t n_records,
                                    void* d_begin, void* d_end, uint64_t* n_valid, void* hip_stream);
/* RecordSource::seek_record (src/record_source.rs:70-104) for a range of records: bytes[begin..end)
 * is the whole row without its line end — the pair csvsimd_tape_seek_record returns. */
int csvsimd_tape_record_spans_device(const void* dindex, uint64_t index_len, uint32_t field_cnt, int new_line,
                                     uint64_t first_record, uint64_t n_records, void* d_begin, void* d_end,
                                     uint64_t* n_valid, void* hip_stream);
/* The same for one column of one chunk: *n_records = chunk->record_cnt spans, for the chunk's rows in order. */
int csvsimd_chunk_field_spans_device(const void* dindex, uint64_t index_len, uint32_t field_cnt, int new_line,
                                     const csvsimd_chunk* chunk, uint32_t field_idx, void* d_begin, void* d_end,
                                     uint64_t* n_records, void* hip_stream);
/* Copies the text of each span into row i of d_dst (n_records x stride bytes, truncated to stride,
 * zero padded); d_len[i] (uint32, may be NULL) = untruncated length.  dbytes[0..bytes_len) is the file.
 * 16 bytes per lane per step when stride % 16 == 0 and d_dst is 16-byte aligned (any field alignment). */
int csvsimd_gather_fields_device(const void* dbytes, uint64_t bytes_len, const void* d_begin, const void* d_end,
                                 uint64_t n_records, void* d_dst, uint32_t stride, void* d_len,
                                 void* hip_stream);

/* Frequency count of column field_idx over the given chunks (all of csvsimd_tape_chunks' output = the whole
 * file): one entry per DISTINCT field text with the number of records that hold it.  EXACT (bytes are compared, see
 * csvsimd_columnar_frequency_device below: this call gathers the column into fixed-stride rows — spans, then
 * csvsimd_gather_fields_device at the stride of its longest field — and counts it with that ONE implementation).
 * Definition checked against: collections.Counter over seek_field.
 *   d_scratch   : scratch_bytes >= csvsimd_column_frequency_scratch_bytes(n_records, n_chunks, max_field_bytes), 256-byte
 *                 aligned; n_records = the chunks' records together.  The column is gathered at the largest stride the
 *                 scratch holds — size it for the longest field you expect, not generously: a stride of 256 bytes for
 *                 8-byte values moves 32 x the bytes.  If some field is longer than that stride the call returns
 *                 CSVSIMD_ERR_TAPE_CAPACITY with status->max_field_bytes set (the longest field of the column, reported
 *                 by every call that got as far as gathering the column: a scratch too small even for 16-byte rows is
 *                 refused before that, with max_field_bytes 0) — size the scratch from that and call again.  The stride
 *                 is capped at 4 096 bytes however large the scratch is.  At most 2^28 records per call
 *   d_entries   : entries_cap csvsimd_freq_entry, unordered (sort by first_record for a deterministic order);
 *                 status->n_distinct of them are valid (CSVSIMD_ERR_TAPE_CAPACITY if more exist than fit)
 * Synchronous on hip_stream (one wait, at the end).  The slow path next to the columnar one: it touches one or two
 * 64-byte sectors of the file plus a slice of tape per record. */
typedef struct csvsimd_freq_entry {
    uint64_t first_record; /* first record (seek_field numbering) that holds this value */
    uint64_t begin, end;   /* bytes[begin..end) = the value's text (of that record)      */
    uint64_t count;        /* records holding it                                         */
} csvsimd_freq_entry;
typedef struct csvsimd_freq_status {
    uint64_t n_records, n_distinct, max_field_bytes, overflow;
} csvsimd_freq_status;
uint64_t csvsimd_column_frequency_scratch_bytes(uint64_t n_records, uint32_t n_chunks, uint64_t max_field_bytes);
int csvsimd_column_frequency_device(csvsimd_ctx* ctx, const void* dbytes, const void* dindex, uint64_t index_len,
                                    uint32_t field_cnt, int new_line, const csvsimd_chunk* chunks,
                                    uint32_t n_chunks, uint32_t field_idx, void* d_scratch, uint64_t scratch_bytes,
                                    void* d_entries, uint64_t entries_cap, csvsimd_freq_status* status,
                                    void* hip_stream);

/* Search in column field_idx of one chunk: bit i of d_bitmap (uint64 words, (record_cnt + 63) / 64 of them,
 * bit i of word i / 64) is set iff the chunk's i-th record matches; *n_matches = how many do.  dbytes[0..bytes_len)
 * is the file (fields are read 8 bytes at a time, never past bytes_len).  needle is HOST
 * memory, at most 256 bytes.  Definitions checked against Python's ==, bytes.startswith, `needle in field`. */
#define CSVSIMD_SEARCH_EQUALS 0
#define CSVSIMD_SEARCH_STARTS_WITH 1
#define CSVSIMD_SEARCH_CONTAINS 2
int csvsimd_column_search_device(csvsimd_ctx* ctx, const void* dbytes, uint64_t bytes_len, const void* dindex,
                                 uint64_t index_len,
                                 uint32_t field_cnt, int new_line, const csvsimd_chunk* chunk, uint32_t field_idx,
                                 const void* needle, uint32_t needle_len, int mode, void* d_bitmap,
                                 uint64_t* n_matches, void* hip_stream);
/* Bitmap -> ascending record ids: d_out[k] = first_record + (position of the k-th set bit); *n_out = number of
 * set bits among the first n_rows (CSVSIMD_ERR_TAPE_CAPACITY if > out_cap; the first out_cap are written).
 * d_scratch: csvsimd_bitmap_select_scratch_bytes(n_rows) bytes, 8-byte aligned. */
uint64_t csvsimd_bitmap_select_scratch_bytes(uint64_t n_rows);
int csvsimd_bitmap_select_device(const void* d_bitmap, uint64_t n_rows, uint64_t first_record, void* d_scratch,
                                 void* d_out, uint64_t out_cap, uint64_t* n_out, void* hip_stream);

/* ---- row-major CSV -> columns in ONE pass, and the two consumers on a column ------------------------------------
 * The per-column entry points above re-read the row-major file once per column (a 32-byte field of a 528-byte row
 * costs one or two 64-byte sectors plus a slice of tape per record).  csvsimd_chunk_to_columns_device reads a chunk's
 * bytes and its slice of the tape ONCE (whole rows staged through LDS) and writes every requested column:
 *     field fields[c] of the chunk's i-th record -> d_cols[(c * n + i) * stride ...), truncated to stride, zero padded,
 *     its untruncated length -> d_lens[c * n + i] (uint32; d_lens may be NULL),      n = *n_records = chunk->record_cnt.
 * Field text is RecordSource::seek_field's (src/record_source.rs:106-140), quotes included.  fields = HOST array of
 * n_fields <= 1024 field ids; fields == NULL: the first n_fields columns (n_fields 0 = all field_cnt of them).  stride: a
 * multiple of 16, <= 4096; d_cols 16-byte aligned.  Asynchronous on hip_stream.  With fields == NULL nothing is waited
 * for, allocated or copied (capturable into a hipGraph); with a field list the call stages it through one of the
 * context's two pinned blocks first, so it may wait for the upload of the call before last and may (re)allocate that
 * block: not capturable, like csvsimd_stage1_index_batch_device_async.  Frequency count and search then run on a column
 * with contiguous 16-byte loads (below), and any columnar engine can take the buffers as they are. */
int csvsimd_chunk_to_columns_device(csvsimd_ctx* ctx, const void* dbytes, uint64_t bytes_len, const void* dindex,
                                    uint64_t index_len, uint32_t field_cnt, int new_line, const csvsimd_chunk* chunk,
                                    const uint32_t* fields, uint32_t n_fields, void* d_cols, uint32_t stride, void* d_lens,
                                    uint64_t* n_records, void* hip_stream);
/* Frequency count of one column of such a copy (d_col = n_records x stride bytes, d_len = its lengths or NULL for
 * fixed-width keys): one entry per DISTINCT value (length + bytes) with the number of records that hold it and the FIRST
 * record that does (first_record + its position in the column; the value's text is that row of the column).  EXACT by
 * construction: two records count as one value only if their bytes are equal (hash bits choose where values meet, a
 * representative record is compared), so a hash collision costs a comparison, never a wrong count — no verification
 * pass, no retry.  design_notes_1.md:1-4; definition checked against collections.Counter over seek_field.
 * Two launches, no table in device memory, nothing cleared per call (round 4): pass 1 aggregates each slab of 8 192
 * records in LDS and writes its surviving (first record, count, hash) tuples partitioned by hash into d_scratch; pass 2
 * merges each partition in LDS and writes its entries.
 *   d_scratch : csvsimd_columnar_frequency_scratch_bytes(n_records) bytes (~12 per record), 16-byte aligned; no need
 *               to clear it
 *   d_entries : entries_cap csvsimd_colfreq_entry, unordered; n_distinct of the status are valid (the first entries_cap
 *               are written if there are more)
 *   d_status  : (_async) one csvsimd_colfreq_status in DEVICE memory, 8-byte aligned, written by the launches
 * csvsimd_columnar_frequency_device_async enqueues the launches on hip_stream (two; three from 4 Mi records on, where a
 * streaming kernel — one workgroup per CU, one table kept over its whole share of the column — counts columns of few values
 * first and leaves the rest to pass 1) and returns: nothing is waited for, allocated or copied — capturable into a hipGraph;
 * the caller reads d_status when it needs it.
 * csvsimd_columnar_frequency_device = the same + the status copied to the host + one synchronisation; it returns
 * CSVSIMD_ERR_TAPE_CAPACITY if status->n_distinct > entries_cap, if status->truncated records are longer than the stride
 * (their counts would merge values that differ past it: transpose with a larger stride), or if status->overflow (more
 * than 8 192 distinct values share 21 hash bits: not a property of real data).  n_records <= 2^28 per call
 * (CSVSIMD_ERR_INVALID_ARG above: count slices and merge the entry lists). */
typedef struct csvsimd_colfreq_entry {
    uint64_t first_record, count;
} csvsimd_colfreq_entry;
typedef struct csvsimd_colfreq_status {
    uint64_t n_records, n_distinct, truncated, overflow;
} csvsimd_colfreq_status;
uint64_t csvsimd_columnar_frequency_scratch_bytes(uint64_t n_records);
int csvsimd_columnar_frequency_device_async(csvsimd_ctx* ctx, const void* d_col, const void* d_len, uint64_t n_records,
                                            uint32_t stride, uint64_t first_record, void* d_scratch, uint64_t scratch_bytes,
                                            void* d_entries, uint64_t entries_cap, void* d_status, void* hip_stream);
int csvsimd_columnar_frequency_device(csvsimd_ctx* ctx, const void* d_col, const void* d_len, uint64_t n_records,
                                      uint32_t stride, uint64_t first_record, void* d_scratch, uint64_t scratch_bytes,
                                      void* d_entries, uint64_t entries_cap, csvsimd_colfreq_status* status,
                                      void* hip_stream);
/* csvsimd_column_search_device on a column of the columnar copy: bit i of d_bitmap = record i matches; same modes and
 * definitions.  CSVSIMD_ERR_TAPE_CAPACITY (bitmap and *n_matches still written) if some record is longer than the stride.
 * The call is one launch and a wait on hip_stream (the needle — host memory, up to 256 bytes — travels in the kernel's arguments,
 * the counts come back through pinned memory); n_records < 2^36 (CSVSIMD_ERR_INVALID_ARG above). */
int csvsimd_columnar_search_device(csvsimd_ctx* ctx, const void* d_col, const void* d_len, uint64_t n_records,
                                   uint32_t stride, const void* needle, uint32_t needle_len, int mode, void* d_bitmap,
                                   uint64_t* n_matches, void* hip_stream);

/* Trims every span [d_begin[i], d_end[i]) in place: CSVSIMD_TRIM_SPACE drops leading / trailing
 * 0x20 bytes — the reference's class 4, whose legend says `todo: trim " xx "`
 * (src/stage1.rs:41-48; extension) — then CSVSIMD_TRIM_QUOTES drops one
 * enclosing pair of `quote` bytes.  seek_field itself never trims (src/record_source.rs:106-140),
 * so without this call the spans stay exactly the reference's. */
#define CSVSIMD_TRIM_SPACE 1u
#define CSVSIMD_TRIM_QUOTES 2u
int csvsimd_trim_spans_device(const void* dbytes, void* d_begin, void* d_end, uint64_t n_records,
                              uint32_t flags, uint8_t quote, void* hip_stream);

/* ---- UTF-8 validation of device-resident bytes (extension) --------------
 * The reference carries a UTF-8 checker (src/avx/utf8check.rs) that reader::read never calls;
 * this is the MI355X counterpart as a separate pass, off the stage-1 path.  RFC 3629 rules
 * (shortest form, no surrogates, <= U+10FFFF, no truncated tail).  first_invalid = offset of the
 * first byte that does not start or continue a well-formed sequence (what Python's bytes.decode
 * reports as UnicodeDecodeError.start), UINT64_MAX if the whole buffer is valid.  d_result is
 * DEVICE memory, 16 bytes, 8-byte aligned; asynchronous on hip_stream; any dbuf alignment. */
typedef struct csvsimd_utf8_result {
    uint64_t first_invalid;
    uint64_t reserved;
} csvsimd_utf8_result;
int csvsimd_utf8_validate_device_async(csvsimd_ctx* ctx, const void* dbuf, uint64_t len, void* d_result,
                                       void* hip_stream);
int csvsimd_utf8_validate_device(csvsimd_ctx* ctx, const void* dbuf, uint64_t len,
                                 csvsimd_utf8_result* result, void* hip_stream);

/* ---- utilities used by the bench / tests (device-side, no reference counterpart) ------------- */
/* Synthetic corpus bytes [file_off, file_off+len) of the cols x width shape. */
int csvsimd_synth_fill_device(void* dbuf, uint64_t file_off, uint64_t len, uint32_t cols,
                              uint32_t width, uint64_t seed, uint32_t quote_pct, void* hip_stream);
/* Order-sensitive checksum of dtape[0..n) whose first element has global index first_index;
 * d_out = device uint64_t[2] (accumulated into: zero it first). */
int csvsimd_tape_checksum_device(const void* dtape, uint64_t n, uint64_t first_index, void* d_out,
                                 void* hip_stream);
/* Wavefront-primitive self test (DPP scans, ballots, ordered descriptor reduction). */
int csvsimd_selftest_device(int device);
/* Average duration in ms of `iters` back-to-back stage-1 launches measured with hipEvents on
 * hip_stream itself (bench.py's roofline leg: torch events only see torch's own stream). */
int csvsimd_stage1_time_device(csvsimd_ctx* ctx, const void* dbuf, uint64_t len, void* dtape,
                               uint64_t tape_cap, void* d_result, void* hip_stream, int warmup,
                               int iters, float* avg_ms);
/* Name of the kernel a stage-1 launch runs (emit = a tape is written; dialect NULL = the reference's),
 * as rocprofv3 prints it: bench.py reports what it timed instead of a string of its own. */
const char* csvsimd_stage1_kernel_name(int emit, const csvsimd_dialect* dialect);
/* 1 if this library was built with -DCSVSIMD_DEV_PROBES (environment-driven ablation hooks in
 * csvsimd_stage1_time_device, scripts/probe*.py): never the product library; bench.py refuses such a build. */
uint32_t csvsimd_build_has_probes(void);

/* HBM streaming probe: the stage-1 traffic shape with none of its work, so bench.py can report the
 * ceiling this GPU actually reaches next to the 8 TB/s spec peak.  Reads dbuf[0..len) (16-byte aligned; whole
 * 128-KiB tiles are streamed) with non-temporal loads and writes write_per16 bytes per 16 bytes read to dout
 * (>= len * write_per16 / 16 bytes, 16-byte aligned) in line-aligned 1-KiB wave stores: 0 = read only, 4 = the
 * 64-col corpus's share of tape writes, 25 = the dense corpus's (1.56 B written per byte read).
 * blocks_per_cu x 4 waves per CU (4 = the stage-1 kernel's 16 waves; 2 is where a bare stream peaks).
 * Average ms of `iters` launches by hipEvents on hip_stream. */
int csvsimd_hbm_probe_device(csvsimd_ctx* ctx, const void* dbuf, uint64_t len, void* dout, int write_per16,
                             int blocks_per_cu, void* hip_stream, int warmup, int iters, float* avg_ms);

/* Independent yardstick for the probe above: a plain copy of dsrc[0..len) to ddst (both 16-byte aligned, len a
 * multiple of 16 is copied) — mode 0 = hipMemcpyDtoDAsync, 1 = the textbook kernel (one 16-byte element per thread, grid
 * as large as the buffer; the shape behind the "6.29 TB/s float4 copy" of the MI355X guide), 2 = the same kernel with
 * non-temporal loads and stores.  Average ms of `iters` back-to-back copies by hipEvents on hip_stream.  The copy moves
 * 2 x len bytes: bench.py prints read + write TB/s next to the probe's 1:1 row. */
int csvsimd_copy_probe_device(csvsimd_ctx* ctx, const void* dsrc, void* ddst, uint64_t len, int mode, void* hip_stream,
                              int warmup, int iters, float* avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* CSVSIMD_H */
