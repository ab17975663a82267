// Descriptors shared by the host engine and the gfx950 kernels.
#pragma once
#include <stdint.h>

namespace fasim {

constexpr int GAP_OPEN = 16;   // cost of the first gap residue (ssw_cpp.cpp:244)
constexpr int GAP_EXT = 4;     // each further residue (ssw_cpp.cpp:245)
constexpr int BIAS = 4;        // |min(matrix)|, ssw_init (sswNew.cpp:1283-1286) / init_work (stats.h:404-419)

// target / query letter codes on the device: A=0 C=1 G=2 T=3 other=4 ; 5 = pad row (score 0)
constexpr int CODE_N = 4;
constexpr int CODE_PAD = 5;

// one striped Smith-Waterman problem = one 16-lane group
struct StripedProb {
	int64_t tbase;     // offset of the unit's first column in the target-code buffer
	int32_t t0;        // first column of the window inside the unit
	int32_t ref_len;   // number of columns
	int32_t q_len;     // number of query rows (forward pass)
	int32_t unit;      // output slot / provenance
	int32_t aux;       // MODE_REV: the forward score the reverse pass terminates on
	int32_t pad;
};

// result of the forward+reverse passes of ssw_align (sswNew.cpp:1446-1525)
struct AlignEnds {
	int32_t score_fwd;   // 255 in byte mode = overflow -> must be re-run in word mode
	int32_t ref_end;
	int32_t read_end;
	int32_t score_rev;
	int32_t ref_begin;
	int32_t read_begin;
};

// banded traceback problem (sswNew.cpp:1071-1259); one thread each
struct BandProb {
	int64_t tbase;       // target codes of the first column of the alignment rectangle
	int32_t q_begin;     // first query row
	int32_t ref_len, read_len;
	int32_t score;
	int64_t scratch_off; // byte offset of this problem's scratch region
	int32_t scratch_cap; // bytes available
	int32_t pad;
};

struct BandOut {
	int32_t status;      // 0 ok, 1 = traceback error (reference returns NULL), 2 = scratch too small, 3 = undefined behaviour in the reference
	int32_t cigar_len;
	uint32_t cigar[62];
};

constexpr int MAX_CIGAR_DEV = 62;

// ---- stage 3 v2 (align.hip) ------------------------------------------------------------------------
struct FwdProb {
	int64_t tbase;        // target-code offset of the window's first column
	int32_t len;          // window length L
	int32_t stream_off;   // offset of the window's first (void) byte in the column stream
};
// flags bit 0: possible Q2 in the forward pass -> stripe-faithful re-run; bit 1: (ref_begin, read_begin) already hold
// the result of an exact reverse pass and `score` is min(forward, reverse)
struct FwdOut { int32_t score, ref_end, read_end, flags, ref_begin, read_begin; };
constexpr int ALIGN_MAX_CIGAR = 48;
struct AlignOutDev {           // 32-byte header; the CIGAR ops go to a compact pool
	int32_t sw_score, ref_begin, ref_end, query_begin, query_end;
	int32_t status;       // 0 ok; 1 traceback error (reference returns NULL); 3 undefined behaviour in the reference;
	                      // 2/4/10/11: re-run on the stripe-faithful path (scratch, cigar pool, hazard, reverse pass undecided)
	int32_t cigar_len;
	uint32_t cigar_off;   // offset in the cigar pool
};

// ---- row f3: forward sweep of classic SIM (sim.hip) ---------------------------------------------------
struct SimEvent { uint32_t j, pad; uint64_t key; };     // a cell above the threshold: column (the row is implied by the buffer), (score + 2^23) << 32 | start_row << 16 | start_col
struct SimNodeDev { int64_t score, stari, starj, endi, endj, top, bot, left, right; };   // = fasim_sim_node (fasim_hip.h)
constexpr int SIM_K = 50;
struct SimFwdArgs {
	const uint8_t* tcodes;      // [unit][tstride] target letters as codes A0 C1 G2 T3 other 4
	const int32_t* unit_len;
	int32_t tstride;
	const uint8_t* qcodes;      // [m] query letters, same coding
	int32_t m;
	const int64_t* min_score;   // [unit] threshold of the first sweep (the reference compares the x10 scores with it, sim.h:567)
	uint64_t* rowbuf;           // [unit][2][row_stride]: C and D of the last finished strip's bottom row
	int64_t row_stride;
	SimEvent* events;           // [unit][64 rows of the strip in flight][event_cap]: the cells above the threshold, row by row
	uint32_t event_cap;         // >= the longest unit: a row cannot overflow
	SimNodeDev* nodes;          // [unit][SIM_K]: the node list after the sweep
	int32_t* node_count;        // [unit]
};

// ---- row f3: the re-sweeps of classic SIM between the K rounds (sim.hip, k_sim_resweep) --------------------------------------
struct SimRoundReq { int32_t active, m1, mm, n1, nn, floor_score, pairs_first, pairs_count, unit, node_count; };   // active: 1 = new round (bounding box of its node, its traceback's used pairs), 2 = carry on
struct SimRoundOut { int32_t node_count, floor_score, pending, pad; };   // floor_score = the reference's `min` after the round (0, then 1); pending: out of time, to be continued
struct SimSweepState { int32_t phase, i, m1, n1, rl, cl, floor_score, nround, grow_rows, grow_cols, positive, pad; };   // an unfinished re-sweep between two launches
struct SimResweepArgs {
	const uint8_t* tcodes; const int32_t* unit_len; int32_t tstride;
	const uint8_t* qcodes; int32_t m;
	const SimRoundReq* req;         // [slot]: the units this launch works on (unit = index into the per-unit arrays below)
	const SimNodeDev* nodes_in;     // [slot][SIM_K]: node list of a unit that starts a round (req.node_count entries)
	SimNodeDev* nodes_out;          // [slot][SIM_K] + out[slot]: what the host reads back
	SimRoundOut* out;
	const uint32_t* pairs;          // (query row << 16) | target column, 1-based, of all units' requests
	uint16_t* usedc;                // [unit][col_stride][SIM_K]: the same per target column (the query row aligned to it)
	uint16_t* used; int32_t* used_cnt;   // [unit][m + 2][SIM_K]: per query row and round the target column aligned to it (0 = none); [unit]: rounds swept so far
	uint64_t* colS; uint64_t* colG; // [unit][col_stride]: per target column the DP state across the sweep line (CC/RR/EE and DD/SS/FF of sim.h as one key each)
	uint64_t* rowS; uint64_t* rowG; // [unit][row_stride]: per query row (HH/II/JJ and WW/XX/YY)
	int64_t col_stride, row_stride;
	SimNodeDev* nodes; int32_t* node_count;      // [unit][SIM_K]: where a suspended unit's list waits for the next launch
	SimSweepState* state;           // [unit]
	int32_t budget;                 // at most this many 64-cell steps per unit and launch ...
	int64_t slice_ticks;            // ... and this much time (100 MHz ticks): what normally ends a unit's share of a launch
	uint64_t* debug;                // FASIM_SIM_DEBUG=1: 3 x (count, 100 MHz ticks) summed over the units: backward steps, forward steps, events; replay passes, outranking events
};

// packed 4-bit score table: entry q (0..5) of row t = score(t,q)+BIAS
struct ScoreLut { uint32_t row[5]; };

} // namespace fasim
