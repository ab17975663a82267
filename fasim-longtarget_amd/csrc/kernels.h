// Host-callable launchers of the gfx950 kernels (defined in kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "device_types.h"

namespace fasim {

enum StripedMode { MODE_PRE = 0, MODE_MAX1 = 1, MODE_ALIGN = 2, MODE_REV = 3 };

constexpr int HAZARD_MAX_CHUNKS = 512;   // column chunks (and checkpoints) per hazard unit in the chunked re-run
struct StripedLaunch {
	const uint8_t* tcodes;      // device: target codes
	const uint8_t* qcodes;      // device: query codes (stage-1 or stage-2 coding), length q_total
	const StripedProb* probs;   // device
	int32_t nprob;
	uint32_t* counter;          // device: work-queue head, zeroed by the launcher
	ScoreLut lut;
	int32_t max_qlen;           // largest q_len in the batch (sizes the LDS stripes)
	uint8_t* colmax;            // MODE_PRE: u8 column maxima, same indexing as tcodes
	uint16_t* colmax_w = nullptr;   // MODE_PRE with word == true: u16 column maxima instead
	int32_t* max_out;           // MODE_PRE / MODE_MAX1: per problem (slot = prob.unit) score, 255 = byte overflow
	AlignEnds* ends;            // MODE_ALIGN: per problem (slot = index in probs)
	// chunked hazard re-run (MODE_PRE, byte mode): see StripedArgs in kernels.hip
	const uint16_t* state = nullptr; int32_t state_rows = 0; const int32_t* chunk_cols = nullptr; const int32_t* chunk_base = nullptr; uint8_t* chunk_rows = nullptr; int32_t row_stride = 0;
	int32_t* chunk_out = nullptr;
	bool spread = false;        // one 256-thread workgroup per CU (LDS request padded past half a CU): every wave gets a SIMD of its own
};

// word == false: 8-bit semantics (16 stripes); word == true: 16-bit semantics (8 stripes)
// quirk: reproduce the signed lazy-F exit test of the SSW byte kernels (sswNew.cpp:369,590)
hipError_t launch_striped(StripedMode mode, bool word, bool quirk, const StripedLaunch& a, hipStream_t st);

// target codes of every (segment, encoding) unit: tcodes[(seg*nenc + k)*tstride + c]
hipError_t launch_encode(const uint8_t* dna_dev, const int32_t* seg_start, const int32_t* seg_len, int32_t nseg,
	const int32_t* enc_ids, int32_t nenc, const uint8_t* enc_lut /*[48][256] -> code*/, uint8_t* tcodes,
	int32_t tstride, hipStream_t st);

// hits above threshold, ordered by column, per unit.  hits[] entries = (pos << 8) | score
// unit_ids == nullptr: units 0..nunit-1 ; otherwise nunit entries of unit_ids are processed
hipError_t launch_hits(const uint8_t* colmax, const int32_t* unit_ids, const int32_t* unit_len, const int32_t* stage1, int32_t nunit,
	int32_t tstride, uint32_t* hits, uint32_t hits_cap, uint32_t* hits_total, int32_t* hit_off, int32_t* hit_cnt,
	int32_t* thr_out, hipStream_t st);

// chunked hazard re-run (kernels.hip): chunk plan of every hazard unit, and the merge of the groups' private rows
hipError_t launch_hazard_plan(const int32_t* unit_ids, int32_t nlist, const int32_t* unit_len, const int32_t* unit_first, const uint16_t* colmax16,
	int32_t tstride, int32_t target, int32_t hot_thr, int32_t hot_w, int32_t* chunk_cols, hipStream_t st);
hipError_t launch_hazard_merge(const uint16_t* colmax16, uint8_t* colmax, const int32_t* unit_ids, int32_t nlist, const int32_t* unit_len,
	const int32_t* chunk_cols, const int32_t* chunk_base, const int32_t* src_chunk, const int32_t* zero_from, const uint8_t* chunk_rows, int32_t row_stride, int32_t tstride, hipStream_t st);

hipError_t launch_banded(const uint8_t* tcodes, const uint8_t* qcodes, const BandProb* probs, int32_t nprob,
	uint8_t* scratch, BandOut* out, hipStream_t st);

// ---- scan.hip: fused stage-1 + stage-2 systolic kernel ---------------------------------------------
constexpr int SCAN_SNAP_STEPS = 1024;
// one work item of the checkpoint pass: continue unit `unit` from pipeline step `step0` (0, or a multiple of SCAN_SNAP_STEPS:
// from the snapshot taken there) and leave the DP state after columns dump_cols[first .. first + count)
struct ScanDumpItem { int32_t unit, step0, first, count; };
struct ScanLaunch {
	const uint8_t* tcodes; const int32_t* unit_ids; const int32_t* unit_len; int32_t nwork; int32_t tstride;
	uint32_t* counter; const uint8_t* qcodes; int32_t m; int8_t score[25]; uint16_t* colmax16;
	uint2* boundary;      // [unit][tstride] hand-over rows between query tiles; needed when systolic_tiles(m) > 1
	int32_t coarse;       // 1: coarse Q2 test (FASIM_Q2_COARSE=1, for measurements)
	int32_t* unit_hz;     // [unit], zeroed by the caller: |= 1 when the unit needs the stripe-faithful re-run; may be NULL
	int32_t* unit_first = nullptr;          // [unit], preset to INT_MAX by the caller: first step at which a Q2 taint could arise
	// pipeline snapshots: the main pass leaves its whole wave state every SCAN_SNAP_STEPS steps, so that the checkpoint pass of
	// the chunked hazard re-run can start in the middle of a unit (single-tile queries only)
	uint32_t* snap = nullptr;               // [unit][snap_per_unit][systolic_snap_dwords(m)][64 lanes]; NULL: none taken / none to read
	int32_t snap_per_unit = 0;
	// checkpoint variant (dump_items != NULL): work item = one window of one hazard unit
	const ScanDumpItem* dump_items = nullptr;
	const int32_t* dump_cols = nullptr;     // [chunk]: the column after which the DP state of all rows is wanted (ascending within an item)
	uint16_t* dump_state = nullptr;         // [chunk][2][16 * ceil(m/16)]: H, then the reference's E
	// block maxima for the banded stage 3 (band.hip): [unit][tile][ublk_blocks][64 lanes] x 2 bytes; NULL: not wanted
	uint16_t* ublk = nullptr;
	int32_t ublk_blocks = 0;
};
constexpr int SCAN_UBLK_STEPS = 64;       // pipeline steps per block of maxima
inline int scan_ublk_blocks(int tstride) { return (tstride + 127 + SCAN_UBLK_STEPS - 1) / SCAN_UBLK_STEPS; }
int systolic_vs(int m);
int systolic_tiles(int m);     // query tiles of 128 virtual lanes x <= 24 rows (1 for m <= 3072)
bool systolic_fits(int m);
int systolic_snap_dwords(int m); // dwords per lane of one pipeline snapshot
hipError_t launch_scan(const ScanLaunch& L, hipStream_t st);      // hipErrorInvalidValue: query too long for this kernel
hipError_t launch_scan_post(const uint16_t* colmax16, const int32_t* unit_ids, int32_t nwork, const int32_t* unit_len,
	int32_t tstride, const int32_t* stage1_in, uint32_t* hits, uint32_t hits_cap, uint32_t* hits_total, int32_t* hit_off,
	int32_t* hit_cnt, int32_t* thr_out, int32_t* stage1_out, int32_t* flags, const int32_t* unit_hz, hipStream_t st);
hipError_t launch_max16(const uint16_t* colmax16, const int32_t* unit_ids, int32_t nwork, const int32_t* unit_len,
	int32_t tstride, int32_t* out, hipStream_t st);

// ---- align.hip: stage 3 ---------------------------------------------------------------------------
struct FwdLaunch {
	const uint8_t* stream; const FwdProb* probs; const int32_t* task_first; int32_t ntask; uint32_t* counter;
	const uint8_t* qcodes; int32_t m; FwdOut* out;
	uint4* boundary;      // [stream position] hand-over between query tiles; needed when systolic_tiles(m) > 1
	int32_t word;         // 1: the reference's 16-bit pass (no overflow rule, no Q2): plain variant without taint tracking
	// reverse pass (word == 1, lane_ub != NULL): reversed query against reversed windows; leaves per window (slot ub_slot[k], -1:
	// none) and zone 0..3 the per-lane maxima [slot][4][128 * tiles] and writes no FwdOut
	uint16_t* lane_ub = nullptr; const int32_t* ub_slot = nullptr;
};
// zones: NULL, or (reverse pass) per window the lengths of the candidate's next three tries (bytes 0..2; 0 = none): the stream then
// holds the REVERSED window with the zone of every column in bits 5-6
hipError_t launch_build_stream(const uint8_t* tcodes, const FwdProb* probs, int32_t nprob, uint8_t* stream, const uint32_t* zones, hipStream_t st);
hipError_t launch_align_fwd(const FwdLaunch& L, hipStream_t st);     // hipErrorInvalidValue: query too long
hipError_t launch_finish(const uint8_t* tcodes, const uint8_t* qcodes, const FwdProb* probs, const FwdOut* fwd, const int32_t* order,
	int32_t nprob, uint8_t* dirs, AlignOutDev* out, uint32_t* cigar_pool, uint32_t pool_cap, uint32_t* pool_count, hipStream_t st);

hipError_t launch_finish_mid(const uint8_t* tcodes, const uint8_t* qcodes, const FwdProb* probs, const FwdOut* fwd, const int32_t* idx_list,
	int32_t nlist, uint8_t* dirs, AlignOutDev* out, uint32_t* cigar_pool, uint32_t pool_cap, uint32_t* pool_count, hipStream_t st);
hipError_t launch_finish_big(const uint8_t* tcodes, const uint8_t* qcodes, const FwdProb* probs, const FwdOut* fwd,
	const int32_t* idx_list, int32_t nlist, uint8_t* scratch, int32_t scratch_cap, AlignOutDev* out, uint32_t* cigar_pool,
	uint32_t pool_cap, uint32_t* pool_count, hipStream_t st);

// ---- band.hip: banded forward pass of stage 3 -------------------------------------------------------------
// one selected try: the band kernel sweeps rows [r0, r0 + 48 G) of window `prob`; its result is the reference's iff the
// score reaches theta_min; nq = length of the try's column stream in groups of 4 columns
struct BandTry { int32_t prob, r0, theta_min, nq; };
constexpr int BAND_SLOT_COLS = 208;       // 16-bit stream words per try (window <= 200 columns + 2 void columns, rounded up to 4)
constexpr int BAND_MAX_ZONES = 16;
// counters of one selection pass: [class * BAND_MAX_ZONES + zone] tries, then stream columns per class, then (debug) the tries left
// unbanded because a bound reaches 148 / because no band proves the target
constexpr int BAND_COUNT_COLS = 3 * BAND_MAX_ZONES, BAND_COUNT_HOT = BAND_COUNT_COLS + 3, BAND_COUNT_NOBAND = BAND_COUNT_COLS + 4, BAND_COUNTS = 64;
struct BandSelLaunch {
	const FwdProb* probs; const int32_t* target; const int32_t* idx; int32_t n, tstride;
	const uint16_t* ublk; int32_t ublk_blocks; int32_t m; const uint8_t* tcodes;
	// start-based bounds from the reverse pass of the candidate (k_align_fwd's lane maxima over the reversed problem):
	// prev[k] = slot * 4 + zone (zone 0..3) or -1 (then the block maxima of k_scan bound the try); NULL: none
	const uint16_t* prev_ub = nullptr; const int32_t* prev = nullptr;
	BandTry* list[3]; uint16_t* slots[3]; uint32_t list_cap;
	int4* dec;                // [n] decisions (decide -> emit)
	uint32_t* counts;         // [BAND_COUNTS], zeroed and filled by launch_band_decide
	uint32_t* cursors;        // [3 * BAND_MAX_ZONES] first slot of every (class, zone) segment, consumed by launch_band_emit
	FwdOut* out; int32_t class_mask;
	int32_t debug = 0;        // 1: count the tries left unbanded by reason
};
// one workgroup of a band launch: profile lanes [zbase, zbase + lc) staged in LDS, tries list[first .. first + count) shared by
// the `nwg` workgroups of the zone (this one is number `wg`)
struct BandZoneTab { int32_t zbase, first, count, wg, nwg; };
struct BandLaunch { const BandTry* list; const uint16_t* slots; const BandZoneTab* tab; int32_t nwg, cls; const uint8_t* qcodes; int32_t m; FwdOut* out; };
int band_profile_lanes(int m);
int band_zone_stride(int m);          // profile lanes per zone (== band_profile_lanes(m) for queries that need one zone)
int band_classes(int m);              // bit c set: class c (sub-pipelines of 8 << c lanes = 384 << c rows) is available for this query
std::vector<BandZoneTab> band_plan(int m, int cls, const uint32_t* zone_count, const uint32_t* zone_first);
hipError_t launch_band_decide(const BandSelLaunch& L, hipStream_t st);
hipError_t launch_band_emit(const BandSelLaunch& L, hipStream_t st);
hipError_t launch_align_band(const BandLaunch& L, hipStream_t st);

// ---- sim.hip: forward sweep of classic SIM (-F), one wave per unit ---------------------------------------
hipError_t launch_sim_forward(const SimFwdArgs& a, int32_t nunit, hipStream_t st);
hipError_t launch_sim_resweep(const SimResweepArgs& a, int32_t nunit, bool few_units, hipStream_t st);

} // namespace fasim
