/* include/fasim_hip.h -- C-ABI of libfasim_hip.so: the MI355X (gfx950) drop-in for the striped
 * Smith-Waterman hot path of Fasim-LongTarget.
 *
 * The reference has no plugin registry; its native boundary for this path is the extern "C" block of
 * ssw.h (ssw.h:19-197) plus, one level up, calc_score_once() (stats.h:879) and fastSIM() (fastsim.h:158)
 * called from LongTarget() (Fasim-LongTarget.cpp:379-598).  Each entry point below names what it replaces.
 * Plain pointers and sizes only; all pointers are HOST pointers unless the name says `_dev`.
 * Every function returns 0 on success or a negative FASIM_E_* code; fasim_last_error() gives the text.
 * There is NO CPU fallback behind this ABI: without a usable HIP device fasim_engine_create() fails.
 */
#ifndef FASIM_HIP_H
#define FASIM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FASIM_OK            0
#define FASIM_E_NODEVICE   -1   /* no HIP device / HIP runtime error at init                      */
#define FASIM_E_ARG        -2   /* bad argument (null, empty query, letters outside ACGTUN, ...)   */
#define FASIM_E_HIP        -3   /* HIP runtime error (message has the hipError string)             */
#define FASIM_E_OVERFLOW   -4   /* a score left the 16-bit range the path computes in              */
#define FASIM_E_UNSUPPORTED -5  /* input hits behaviour the reference leaves undefined (see DESIGN) */
#define FASIM_E_NOMEM      -6

typedef struct fasim_engine fasim_engine;

/* Defaults of initEnv() (Fasim-LongTarget.cpp:284-303); field meaning = struct para (fastsim.h:22-45). */
typedef struct fasim_params {
	int32_t rule;            /* -r   0 = all rules                         */
	int32_t cutLength;       /* -c   5000                                  */
	int32_t strand;          /* -t   0 both, 1 parallel only, -1 anti only */
	int32_t overlapLength;   /* -o   100                                   */
	int32_t ntMin;           /* -ni  20                                    */
	int32_t ntMax;           /* -na  100000                                */
	float   scoreMin;        /*      0                                     */
	float   minIdentity;     /* -i   60   (parsed with atoi by the CLI)    */
	float   minStability;    /* -S   1    (parsed with atoi by the CLI)    */
	int32_t penaltyT;        /* -pt  -1000                                 */
	int32_t penaltyC;        /* -pc  0                                     */
	int32_t cDistance;       /* -ds  15                                    */
	int32_t cLength;         /* -lg  50                                    */
	int32_t classicSim;      /* -F   0; 1 = classic SIM instead of fastSIM (doFastSim = false, Fasim-LongTarget.cpp:360-362) */
} fasim_params;

void fasim_params_default(fasim_params* p);

/* ---- engine ---------------------------------------------------------------------------------- */
/* Process-wide side effects of the first fasim_engine_create() (each can be switched off by its environment variable):
 *   GPU_MAX_HW_QUEUES=8 is exported if unset (the batches in flight need their own hardware queues);
 *   hipSetDeviceFlags(hipDeviceScheduleBlockingSync) for `device`, so that host threads sleep instead of polling in stream
 *     synchronisation (no effect if the host application created the device's context first; FASIM_BLOCKING_SYNC=0);
 *   mallopt(): freed host blocks of up to 32 MB stay in the heap instead of being unmapped -- unmapping host memory while HIP
 *     queues are live stalls the running kernels (FASIM_MALLOPT=0). */
int  fasim_engine_create(int device, fasim_engine** out);
/* flags = FASIM_CREATE_NO_PROCESS_TUNING: none of the three process-wide settings above is touched (what a host application that
 * only embeds single calls wants; the `ssw.h` symbols create their engine this way).  The worker engines of a scan inherit nothing:
 * they are created by the scan through fasim_engine_create, so a throughput run should create its engine without the flag. */
#define FASIM_CREATE_NO_PROCESS_TUNING 1
int  fasim_engine_create_ex(int device, int32_t flags, fasim_engine** out);
void fasim_engine_destroy(fasim_engine* e);
const char* fasim_last_error(const fasim_engine* e);   /* e may be NULL: last global error */

/* Tuning knobs (optional).  key "workers": batches kept in flight by fasim_scan (default 10, env FASIM_WORKERS);
 * key "seg_batch": segments per batch (default 384, env FASIM_SEG_BATCH); value <= 0 restores the default.
 * key "taper": percent of the segments scanned in half-size batches at the end (default: 25 for a single-lncRNA scan whose batches are fitted to the record, else 0);
 * key "heavy_gate": k_scan / k_align_fwd launches in flight at once (default 3, env FASIM_HEAVY_GATE; 0 = no gate);
 * -1 restores the default of the last two.
 * key "host_threads": host threads for the host side of the batches, all workers of this engine together (default 3/8 of the
 * cores, at most 96, env FASIM_HOST_THREADS); several engines in one process should share the cores.
 * key "hazard_chunks" (1) / "hazard_snapshots" (0) / "hazard_chunk_cols" (200) / "hazard_hot_weight" (2): the stripe-faithful
 * re-run of the units the reference's signed lazy-F exit can touch: in parallel column chunks from checkpoints (0: one
 * sequential run per unit), checkpoint pass continued from pipeline snapshots of the main scan (0: from column 0; the snapshots
 * cost 64 KB of HBM writes per unit), cost target of a chunk in columns, price of a column whose maximum is >= 144
 * (env FASIM_HAZARD_CHUNKS, FASIM_HAZARD_SNAP).
 * key "numa_affinity" (1): for the duration of a scan the calling thread and the threads the scan starts are pinned to the CPUs of
 * the GPU's NUMA node (/sys/bus/pci/devices/<bus id>/local_cpulist); the caller's affinity is restored afterwards; no effect on a
 * single-node machine.
 * key "band" (1): the banded forward pass of stage 3 (csrc/band.hip); 0 = every window try runs over the whole query (env FASIM_BAND).
 * Results do not depend on any of them. */
int fasim_set_option(fasim_engine* e, const char* key, int32_t value);

/* Replaces ssw_init()/init_destroy() (ssw.h:78,83) and init_work() (stats.h:386): the lncRNA is
 * encoded once and stays resident on the device (the reference rebuilds its profile on every call). */
int fasim_set_query(fasim_engine* e, const char* rna, int32_t len);

/* ---- single-problem drop-ins of the native kernels (each launches the HIP path) --------------- */
/* calc_score_once() (stats.h:879-956): exact max local score of the query vs `target`.            */
int fasim_calc_score_once(fasim_engine* e, const char* target, int32_t n, int32_t* score);
/* ssw_pre_align() (ssw.h:128, sswNew.cpp:1309) behind Aligner::preAlign's base translation
 * (ssw_cpp.cpp:394-415): out_cols[n] = per-column maxima incl. the reference's Q1/Q2/Q3 behaviour. */
int fasim_ssw_pre_align(fasim_engine* e, const char* target, int32_t n, int32_t* out_cols);
/* column maxima of the reference's 16-bit kernels (sw_sse2_word, sswNew.cpp:893-1069; same DP as the unreachable
 * sw_sse2_word_once, :698): 8 stripes, no overflow rule, no signed-compare quirk.  Used by the ssw.h shim
 * (include/ssw.h) for the sub-optimal score of alignments whose score overflows 8 bits.                */
int fasim_ssw_colmax_word(fasim_engine* e, const char* target, int32_t n, int32_t* out_cols);
/* peak picking of Aligner::preAlign (ssw_cpp.cpp:427-572) on a column-max array (host logic).     */
int fasim_pick_candidates(const int32_t* cols, int32_t n, int32_t threshold,
                          int32_t* out_score, int32_t* out_pos, int32_t cap, int32_t* count);
/* Self-check of the host record path (pure host code, no device): `n` random alignments (random CIGARs over a random segment
 * and lncRNA, every rule encoding) through the numbers-only conversion the scan uses and through the reference-shaped
 * string conversion, then both record types through the dedup (sort / unique with the reference's comparators, which are
 * not strict weak orderings: the order must match exactly).  *mismatches = number of differences (0 expected).          */
int fasim_selfcheck_records(uint64_t seed, int32_t n, int32_t* mismatches);
/* ssw_align() (ssw.h:118, sswNew.cpp:1446) behind Aligner::Align (ssw_cpp.cpp:599-643).           */
typedef struct fasim_alignment {
	int32_t sw_score, ref_begin, ref_end, query_begin, query_end;
	int32_t cigar_len;                 /* number of BAM-encoded ops in cigar[]; -1 (with sw_score 0): the
	                                      reference's ssw_align returns NULL here (banded_sw finds no path,
	                                      sswNew.cpp:1535-1538), which its callers treat as score 0      */
	uint32_t cigar[256];               /* (len<<4)|op, op 0=M 1=I 2=D (ssw.h:174)                   */
} fasim_alignment;
int fasim_ssw_align(fasim_engine* e, const char* window, int32_t n, fasim_alignment* out);
/* batched forms (the shapes the engine uses internally): nprob problems, targets concatenated      */
int fasim_pre_align_batch(fasim_engine* e, const char* targets, const int64_t* offsets, const int32_t* lens,
                          int32_t nprob, int32_t* out_cols /* concatenated like targets */, int32_t* out_stage1);
int fasim_align_batch(fasim_engine* e, const char* windows, const int64_t* offsets, const int32_t* lens,
                      int32_t nprob, fasim_alignment* out);
/* transferString()/reverseSeq()/complement() (rules.h:59-318) for encoding enc in [0,48), canonical
 * execution order of LongTarget(): target[n] and src[n] (src is NUL-padded if letters were dropped). */
int fasim_encode_unit(const char* seg, int32_t n, int32_t enc, char* target, char* src);

/* ---- row f3: the -F path, classic SIM (sim.h:410-1143) ------------------------------------------- */
/* The forward sweep of SIM() (sim.h:506-571) on the GPU: local alignment scores with start points over the whole
 * (query x target) matrix, every cell above `min_score` fed to the K = 50 node list in row-major order (addnode,
 * sim.h:99-148).  Returns the node list the sweep leaves, in list order: what the reference holds when its traceback
 * loop starts (sim.h:572).  Scores are the reference's x10 values; min_score is compared unscaled, as the reference
 * does (sim.h:567).  fasim_scan with params.classicSim = 1 runs the whole -F path: this sweep on the GPU, then K rounds in
 * lock step over the units of a batch -- best node, linear-space traceback and triplex record on host threads
 * (csrc/host_sim.cpp), the backward / growing / forward re-sweeps of the influenced rectangles (sim.h:884-1141) on the GPU
 * (k_sim_resweep, csrc/sim.hip).
 * Query and targets: ACGT (other letters score as mismatches; the reference reads an uninitialised table there), at most
 * 65534 long (16-bit start fields in the 64-bit DP keys; a re-sweep starts lines at row M + 1 / column N + 1). */
typedef struct fasim_sim_node { int64_t score, stari, starj, endi, endj, top, bot, left, right; } fasim_sim_node;
#define FASIM_SIM_K 50
typedef struct fasim_result fasim_result;      /* defined below */
/* Host half of the -F path for ONE unit (pure host code, no device): everything SIM() does after its first sweep
 * (sim.h:572-1141) for segment `seg` under encoding `enc`, starting from the node list of the forward sweep.  The records
 * are the unit's triplexes as SIM() appends them (before LongTarget()'s tail filter). */
int fasim_sim_finish_unit(const char* rna, int32_t m, const char* seg, int32_t n, int32_t enc, int64_t dna_start, int64_t min_score,
                          const fasim_params* p, const fasim_sim_node* nodes, int32_t nnodes, fasim_result** out);
int fasim_sim_forward_batch(fasim_engine* e, const char* targets, const int64_t* offsets, const int32_t* lens, int32_t nprob,
                            const int64_t* min_scores, fasim_sim_node* nodes /* [nprob][FASIM_SIM_K] */, int32_t* counts /* [nprob] */);

/* ---- the batched body of LongTarget() (Fasim-LongTarget.cpp:379-598) -------------------------- */
/* One record per triplex that survives fastSIM()'s own filter and LongTarget()'s tail filter, in the
 * reference's order (segment, encoding, fastSIM rank).  Strings live in `pool` (NUL-terminated).   */
typedef struct fasim_triplex {
	int32_t stari, endi, starj, endj, strand, reverse, rule, nt;
	float   score, identity, tri_score;
	int32_t seg, enc;                   /* provenance: segment index, encoding index                 */
	int32_t genome_shift;               /* added to start_genome for THIS record's genome coordinates; 0 from
	                                       fasim_scan.  Only the B1-compatible reader (--accumulate-records) sets it:
	                                       the reference patches each FASTA record's triplexes with that record's
	                                       own start (Fasim-LongTarget.cpp:141-149)                         */
	int64_t tfo_off, tts_off;           /* offsets of stri_align / strj_align in pool                */
} fasim_triplex;

#define FASIM_KERNEL_FAMILIES 10
typedef struct fasim_scan_stats {
	int64_t segments, segments_skipped, units;
	int64_t candidates, align_calls;
	int64_t align_word_reruns;          /* window alignments repeated with the 16-bit pass (8-bit maximum >= 251) */
	int64_t stage2_overflow_units, stage1_word_reruns;
	int64_t logical_cells;              /* m * sum(len(segment)) * n_enc                  */
	double  t_total_s, t_stage1_s, t_stage2_s, t_stage3_s, t_host_s;   /* host wall clock per phase       */
	/* HIP-event time of the kernels, summed over launches (each on the stream it was launched on; with several
	 * batches in flight the durations include sharing the GPU).  index: 0 k_scan (fused stage 1+2), 1 k_striped
	 * stage-1/2 (hazard re-runs, long queries), 2 k_align_fwd, 3 k_finish_lds, 4 encode/hits/post/stream,
	 * 5 k_striped stage 3 (exact replays, exact reverse passes), 6 k_finish (global scratch) + k_banded, 7 k_sim_forward (-F) */
	double  kernel_ms[FASIM_KERNEL_FAMILIES];      /* ... 8 k_align_band (banded stage-3 forward), 9 k_band_select */
	int64_t kernel_launches[FASIM_KERNEL_FAMILIES];
	int64_t cells_stage1, cells_stage2, cells_stage3;   /* DP cells actually executed (stage 3: fwd + rev)   */
	int64_t hazard_units;               /* units re-run by the stripe-faithful kernel (possible Q2)  */
	int64_t rev_exact;                  /* window tries whose reverse pass ran on the stripe-faithful kernel */
	int64_t exact_replays;              /* candidates replayed try by try on the stripe-faithful kernels */
	int64_t tries_skipped;              /* window tries of the reference whose result cannot matter and that were not run */
	int64_t band_tries;                 /* forward passes run on a row band (k_align_band), second attempts included */
	int64_t band_proven;                /* window tries whose band result was proven to be the full-height result */
	int64_t band_cells;                 /* DP cells executed by k_align_band (part of cells_stage3) */
	int64_t rev_bound_passes;           /* window tries that took the full-height reverse pass (bounds for the band passes) */
} fasim_scan_stats;

struct fasim_result {
	fasim_triplex* recs; int64_t count;
	char* pool; int64_t pool_len;
	fasim_scan_stats stats;
};

/* Scans segments [seg_first, seg_first+seg_count) of `dna` (whole sequence of ONE FASTA record, host
 * memory, upper-case ACGTN).  Coordinates in the records are relative to the whole `dna` exactly as in
 * the reference.  seg_count < 0 = all remaining.  Used unsharded (1 GPU) or per rank (multi-GPU).   */
int fasim_scan(fasim_engine* e, const char* dna, int64_t dna_len, int64_t seg_first, int64_t seg_count,
               const fasim_params* p, fasim_result** out);
/* Multi-lncRNA batch (BASELINE config 4: many lncRNAs x one genome).  The reference holds ONE RNA per run
 * (readRna(), Fasim-LongTarget.cpp:174-200) and is started once per lncRNA; here the DNA record stays resident
 * and every (lncRNA, batch of segments) pair is one work item of the same queue, so consecutive lncRNAs overlap
 * instead of paying a ramp-up and a drain each.  out[q] receives the records of rnas[q] exactly as fasim_scan
 * would return them for that query alone (free each with fasim_result_free).  dna == NULL scans the resident
 * record (fasim_load_dna).  Afterwards the engine's current query is rnas[nq-1].  Per-result stats.t_total_s is
 * the wall clock from the start of that query's first batch to the end of its last one (neighbours overlap). */
int fasim_scan_queries(fasim_engine* e, const char* const* rnas, const int32_t* rna_lens, int32_t nq,
                       const char* dna, int64_t dna_len, int64_t seg_first, int64_t seg_count,
                       const fasim_params* p, fasim_result** out /* [nq] */);
/* Host half of the path's one exchange step: concatenates the records of `nparts` shards in the
 * order given and rebases their pool offsets.  Shards are contiguous segment ranges, so rank order IS the
 * reference's canonical (segment, encoding, fastSIM rank) order, which cluster_triplex()'s unstable sort needs.
 * The result owns its memory (fasim_result_free); stats are zero. */
int fasim_merge_results(const fasim_triplex* const* recs, const int64_t* counts, const char* const* pools,
                        const int64_t* pool_lens, int32_t nparts, fasim_result** out);
/* In-place form for a gather that receives every shard's records and pool directly at their final positions of one
 * flat buffer (what gather_results() of the Python package does over RCCL): adds `delta` (= the bytes of pool that
 * precede this shard's pool) to the pool offsets of `count` records. */
int fasim_rebase_offsets(fasim_triplex* recs, int64_t count, int64_t delta);
/* Optional: upload a DNA record once and keep it resident in HBM (and a host copy for the string work);
 * afterwards fasim_scan(e, NULL, 0, ...) scans the resident record without any H2D copy of the sequence. */
int fasim_load_dna(fasim_engine* e, const char* dna, int64_t dna_len);
void fasim_result_free(fasim_result* r);
int64_t fasim_segment_count(int64_t dna_len, const fasim_params* p);   /* cutSequence(): fastsim.h:71 */

/* ---- host tail: main() coordinates + cluster_triplex() + printResult() ------------------------- */
/* (Fasim-LongTarget.cpp:141-149, 600-691, 797-829).  Takes the concatenation of all shards' records
 * in canonical order and returns the bytes of the -TFOsorted file (malloc'd; free with fasim_free).  */
int fasim_tfosorted(const fasim_triplex* recs, int64_t count, const char* pool, int64_t pool_len,
                    const char* chr, int64_t start_genome, const fasim_params* p,
                    char** text, int64_t* text_len);
/* print_cluster() (Fasim-LongTarget.cpp:694-795): the bytes of the -TFOclass<level>-<ds>-<lg> bedGraph file for
 * class `level` (the reference writes levels 1 and 2, :832).  Clusters the records itself; dna_len = record length. */
int fasim_tfoclass(const fasim_triplex* recs, int64_t count, int32_t level, const char* chr, int64_t start_genome,
                   int64_t dna_len, const char* rna_name, const fasim_params* p, char** text, int64_t* text_len);
/* The same two with a flags word.  FASIM_TAIL_CLAMP_CLUSTER: when a triplex mid-point lies within -ds of the query
 * start, cluster_triplex() of the reference indexes std::map<size_t,...> with a negative number and never terminates
 * (Fasim-LongTarget.cpp:641-688; needs -lg below about 2*-ds).  Without the flag such input is refused
 * (FASIM_E_UNSUPPORTED); with it the positions before the query start are treated as non-existent, which is what the
 * reference's arithmetic means and gives a defined, terminating result (identical to the reference wherever the
 * reference terminates). */
#define FASIM_TAIL_CLAMP_CLUSTER 1
int fasim_tfosorted_ex(const fasim_triplex* recs, int64_t count, const char* pool, int64_t pool_len,
                       const char* chr, int64_t start_genome, const fasim_params* p, int32_t flags,
                       char** text, int64_t* text_len);
int fasim_tfoclass_ex(const fasim_triplex* recs, int64_t count, int32_t level, const char* chr, int64_t start_genome,
                      int64_t dna_len, const char* rna_name, const fasim_params* p, int32_t flags,
                      char** text, int64_t* text_len);
/* printResult() as a whole (Fasim-LongTarget.cpp:797-836): the -TFOsorted text and the two -TFOclass texts (levels 1 and 2)
 * from ONE clustering of the records (the separate calls above each cluster again).  Free each text with fasim_free. */
int fasim_tail_outputs(const fasim_triplex* recs, int64_t count, const char* pool, int64_t pool_len, const char* chr,
                       int64_t start_genome, int64_t dna_len, const char* rna_name, const fasim_params* p, int32_t flags,
                       char** tfosorted, int64_t* tfosorted_len, char** class1, int64_t* class1_len,
                       char** class2, int64_t* class2_len);
void fasim_free(void* p);
/* ingest helper: upper-cases a DNA record in place (soft-masked genomes such as UCSC hg38 carry repeats in lower
 * case; the reference does not upper-case and treats such letters as unknown, rules.h:286-312, 82-83). */
void fasim_upper_case(char* seq, int64_t n);

/* deterministic synthetic DNA (splitmix64, 2 bits/base; same stream as tools/synth.py) */
void fasim_synth_dna(char* out, int64_t n, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
