// fasim-longtarget_amd/csrc/scan.hip -- the fused stage-1 + stage-2 "scan" kernel for gfx950.
//
// Textbook Gotoh recurrence over the whole (query x 5 kb segment) matrix of a
// unit, as a systolic pipeline inside ONE wave64:
//   * 128 virtual lanes = 64 lanes x 2 packed 16-bit halves; virtual lane v owns RP consecutive query
//     rows (H and E of those rows live in VGPRs as packed u16 pairs) and processes column (step - v)
//     at pipeline step `step`, so every dependency (diagonal H, vertical F, running column maximum,
//     target code) arrives from virtual lane v-1 one step earlier: one DPP wave_shr + v_alignbit each.
//   * all cell arithmetic is packed 16-bit VALU (v_pk_add_u16 / v_pk_max_i16 / v_pk_sub_u16 clamp):
//     two DP cells per instruction, ~10 instructions per pair.  No MFMA: this is integer DP.
//   * the int16 query profile (5 target codes x rows) is staged once per workgroup in LDS, laid out
//     [code][lane][half][24 rows] with a 112-byte lane stride so the ds_read_b128 fetches (8 rows each)
//     are bank-conflict free; the two halves of a lane sit on different columns, so each fetches its own
//     code's rows and a v_perm_b32 merges them.
//   * the segment's target codes are read 64 columns at a time (one coalesced byte per lane) and fed to
//     virtual lane 0 with v_readlane; per-column maxima leave virtual lane 127 as one u16 per step.
//
// What it replaces: calc_score_once() (stats.h:879-956) and sw_sse2_byte_once() (sswNew.cpp:255-464) for
// every unit in which the reference's layout-dependent behaviour cannot show: the column maxima are the
// textbook ones up to the reference's overflow column (Q1, applied in k_scan_post) unless the signed
// lazy-F exit (Q2, sswNew.cpp:369) can trigger.  Q2 needs an F value >= 132 to cross one of the 15 stripe
// boundaries k*ceil(m/16) of the reference's striped layout.  The rows are laid out so that each of the
// reference's 16 stripes is exactly `vs` virtual lanes: the F entering virtual lane vs*k IS the F crossing
// boundary k.  Zero-score pad rows (Q3) are part of the profile.
//
// Taint tracking.  Every DP value is carried DOUBLED, with bit 0 = "the reference may hold a smaller value here
// because of Q2".  Doubled scores and gap costs are even, so the bit rides through add / saturating subtract for
// free, and max() keeps it exactly when the winning operand carries it (ties go to the tainted operand, which is
// the safe side).  The hazard test sets the bit on the cells whose value the reference's early lazy-F exit would
// have withheld; a unit needs the stripe-faithful re-run (kernels.hip) only if a tainted value ends up as a column
// maximum that matters (above the threshold or at the overflow cut: k_scan_post).  Second-order effects stay on the
// safe side: the reference's values are never larger than the textbook ones, so a chain that is hot there is hot
// here; a tainted H counts as "< 144" in the arming test; and a tainted H or F can only produce tainted results.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kernels.h"

namespace fasim {

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v2s as_s(v2u x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ v2u as_u(v2s x) { return __builtin_bit_cast(v2u, x); }
__device__ __forceinline__ v2s s_from(int x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ int to_int(v2s x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ int to_int(v2u x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ v2u u_fromi(int x) { return __builtin_bit_cast(v2u, x); }

// shift a packed pair down the virtual-lane pipeline: out.lo = x.hi of lane-1 (lane 0: inject), out.hi = x.lo
__device__ __forceinline__ int vshift(int x, int inject_hi)
{
	const int up = __builtin_amdgcn_update_dpp(inject_hi, x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
	return __builtin_amdgcn_alignbit(x, up, 16);
}

// Packed 16-bit helpers for the (rare) hazard branch, written as inline asm so that the compiler neither rewrites
// them into per-half compares/selects nor hoists them into the main path of every step.
__device__ __forceinline__ v2u pk_subs(v2u a, v2u b) { v2u r; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2u pk_subs_k(v2u a, uint32_t k) { v2u r; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "s"(k)); return r; }
__device__ __forceinline__ v2u pk_ksubs(uint32_t k, v2u a) { v2u r; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "s"(k), "v"(a)); return r; }
__device__ __forceinline__ v2u pk_minu(v2u a, v2u b) { v2u r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2u pk_minu_k(v2u a, uint32_t k) { v2u r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(k)); return r; }
__device__ __forceinline__ v2u pk_maxu(v2u a, v2u b) { v2u r; asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// x |= bits with input and output tied to ONE register: a value that the rare branch may modify then needs no copy at
// the join with the common path (plain C would let the allocator put the two versions into different registers and
// pay a v_mov per row in every step)
__device__ __forceinline__ void or_in_place(int& x, int bits) { asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(bits)); }

// the same shift with zeros entering virtual lane 0 (bound_ctrl: an out-of-range source lane reads as 0): no register has to
// be preloaded with the value to inject
__device__ __forceinline__ int vshift0(int x)
{
	const int up = __builtin_amdgcn_mov_dpp(x, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
	return __builtin_amdgcn_alignbit(x, up, 16);
}

// two carried maxima (2 * value + taint each) -> two bytes min(value, 255)
__device__ __forceinline__ uint16_t ublk_pack(v2u acc)
{
	const v2u v = __builtin_elementwise_min(acc >> (v2u){ 1, 1 }, (v2u){ 255, 255 });
	return (uint16_t)(v[0] | (v[1] << 8));
}

constexpr int SCAN_RS = 24;                 // storage rows per virtual lane in the LDS profile
constexpr int SCAN_LANE_STRIDE = 112;       // bytes: 2 halves x 24 rows x 2 B + 16 B pad (bank-conflict-free b128)
constexpr int SCAN_CODE_STRIDE = 64 * SCAN_LANE_STRIDE;   // 7168 B, a multiple of the 256-B bank row
constexpr int SCAN_DEAD = -16384;           // score of rows beyond the padded query: can never be chosen

struct ScanArgs {
	const uint8_t* tcodes;       // [unit][tstride]
	const int32_t* unit_ids;     // work list (indices into tcodes/unit_len), nwork entries
	const int32_t* unit_len;
	int32_t nwork;
	int32_t tstride;
	uint32_t* counter;
	const uint8_t* qcodes;       // query codes for this scoring (0..4)
	int32_t m;                   // query length
	int32_t m_pad;               // 16 * ceil(m/16): rows [m, m_pad) score 0 (Q3)
	int32_t seg_len16;           // ceil(m/16): stripe length of the reference's byte kernels
	int8_t score[25];            // score[t*5+q]
	uint16_t* colmax16;          // [unit][tstride] : 2 * column maximum + taint bit (32767 = saturated)
	int32_t* unit_hz;            // [unit] |= 1: the whole unit needs the stripe-faithful re-run (NULL: not tracked)
	int32_t coarse;              // 1: unit-level test only (any F[b] >= 132), no row analysis
	int32_t vs;                  // virtual lanes per reference stripe (multiple of 8): 16*vs virtual lanes in all
	int32_t tile;                // this launch handles virtual lanes [128*tile, 128*tile+128)
	int32_t ntiles;
	uint2* boundary;             // [unit][tstride]: per column {hbot | fbot<<16, cm | fpo<<16} handed from tile to tile
	int32_t* unit_first;         // [unit] atomicMin: first pipeline step at which a Q2 taint could arise (NULL: not tracked)
	uint32_t* snap;              // pipeline snapshots [unit][snap_per_unit][2 * RP + 6][64] (main pass: written; DUMP: read); may be NULL
	int32_t snap_per_unit;
	// DUMP variant only (checkpoints for the chunked hazard re-run): work item w = dump_items[w] continues a unit from a
	// snapshot (or from step 0) and leaves H and the reference's E of every row after columns dump_cols[first .. first + count)
	// in dump_state[chunk][2][rows_total] (values as carried: 2 * value + taint)
	const ScanDumpItem* dump_items;
	const int32_t* dump_cols;
	uint16_t* dump_state;
	int32_t rows_total;          // 16 * ceil(m/16)
	// block maxima for the banded stage 3 (band.hip): per unit, query tile and block of SCAN_UBLK_STEPS pipeline steps two bytes per
	// lane = the maximum H (capped at 255: anything from 148 on only means "not banded") over the rows of the lane's two virtual
	// lanes and the steps of the block; virtual lane v of the tile sees column c at step c + v.  [unit][tile][block][64 lanes]; NULL: not wanted
	uint16_t* ublk;
	int32_t ublk_blocks;         // blocks per (unit, tile) in the buffer
};

// rows owned by global virtual lane v (stripe-aligned layout): stripe s = v / vs gets its ceil(m/16) rows spread over
// vs virtual lanes, the first (segLen % vs) of them one row longer
__device__ __forceinline__ void lane_rows(int v, int seg_len, int vs, int* row0, int* rows)
{
	const int s = v / vs, j = v - s * vs, q = seg_len / vs, rem = seg_len - q * vs;
	*rows = q + (j < rem ? 1 : 0);
	*row0 = s * seg_len + j * q + (j < rem ? j : rem);
}

// score of target code t against the r-th row of global virtual lane v (doubled: bit 0 is the taint bit)
__device__ __forceinline__ int scan_cell_score(const ScanArgs& a, int t, int v, int r)
{
	int row0, rows_v;
	lane_rows(v, a.seg_len16, a.vs, &row0, &rows_v);
	if (r >= rows_v) return SCAN_DEAD;
	const int row = row0 + r;
	if (row < a.m) return 2 * a.score[t * 5 + a.qcodes[row]];
	return 0;                                 // rows [m, 16*segLen) are the reference's zero-score pad rows (Q3)
}

// One 256-thread workgroup shares the per-code int16 profile (35 KB); the two halves of a lane fetch their own code's rows and a
// v_perm_b32 per row merges them.  (A variant with the profile stored per PAIR of codes -- no perm, 109 KB of LDS, 1024-thread
// workgroups -- was 5 % faster alone and 1-4 % slower with ten batches in flight; it was removed in round 3.)
template <int RP, bool DUMP = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DUMP ? 1 : 4, DUMP ? 2 : 4))) k_scan(ScanArgs a)
{
	extern __shared__ __align__(16) uint8_t prof[];
	const int lane = threadIdx.x & 63;

	{
		// ---- stage the int16 profile: prof[t][lane][half][r] -------------------------------------------
		// stripe-aligned layout: the reference's stripe s = rows [s*segLen, (s+1)*segLen) is spread over `vs` virtual
		// lanes, so every stripe boundary is a virtual-lane boundary; this launch stages the rows of its tile only
		for (int idx = threadIdx.x; idx < 5 * 128 * SCAN_RS; idx += blockDim.x) {
			const int r = idx % SCAN_RS;
			const int v = (idx / SCAN_RS) % 128;
			const int t = idx / (SCAN_RS * 128);
			const int sc = scan_cell_score(a, t, 128 * a.tile + v, r);
			*reinterpret_cast<int16_t*>(prof + t * SCAN_CODE_STRIDE + (v >> 1) * SCAN_LANE_STRIDE + (v & 1) * 48 + r * 2) = (int16_t)sc;
		}
	}
	__syncthreads();

	// ---- per-lane constants ---------------------------------------------------------------------------
	// fthr: a half whose virtual lane starts a stripe (v = vs*k, k >= 1) sees F[b] of the boundary row b directly as
	//       its incoming F; Q2 needs F[b] >= 132 (doubled: > 263), other halves never flag.
	// act : 0xFFFF where the half owns RP rows, 0 where it owns RP-1 (its last register row is transparent)
	uint32_t fthr = 0xFFFFFFFFu, act = 0, startbits = 0;
	for (int h = 0; h < 2; h++) {
		const int v = 128 * a.tile + 2 * lane + h;        // global virtual lane
		int row0, rows_v;
		lane_rows(v, a.seg_len16, a.vs, &row0, &rows_v);
		if (v % a.vs == 0 && v > 0) { fthr = (fthr & ~(0xFFFFu << (16 * h))) | (263u << (16 * h)); startbits |= 0xFFFFu << (16 * h); }
		if (rows_v == RP) act |= 0xFFFFu << (16 * h);
	}
	const v2u fthr2 = __builtin_bit_cast(v2u, fthr);
	const v2u actm = __builtin_bit_cast(v2u, act);
	// stripe-start halves take the crossing F as the start of a propagation chain
	const v2u startm = __builtin_bit_cast(v2u, startbits);
	// the refined test needs F to die inside one stripe (F <= 234 decays by 4 per row)
	const bool lvl2 = a.seg_len16 >= 96 && !a.coarse;
	const uint8_t* pl = prof + lane * SCAN_LANE_STRIDE;

	for (;;) {
		int w = 0;
		if (lane == 0) w = (int)atomicAdd(a.counter, 1u);
		w = __builtin_amdgcn_readfirstlane(w);
		if (w >= a.nwork) break;
		ScanDumpItem item = { 0, 0, 0, 0 };
		if constexpr (DUMP) item = a.dump_items[w];
		const int unit = DUMP ? item.unit : a.unit_ids[w];
		const int n = a.unit_len[unit];
		const uint8_t* tc_unit = a.tcodes + (int64_t)unit * a.tstride;
		uint16_t* out = a.colmax16 + (int64_t)unit * a.tstride;
		uint2* bnd = a.boundary + (int64_t)unit * a.tstride;
		const bool first_tile = a.tile == 0, last_tile = a.tile == a.ntiles - 1;
		uint2 bchunk = make_uint2(0u, 0u);

		// H and E are kept as plain 32-bit registers (two packed u16 halves each) and only viewed as vectors inside the
		// arithmetic: vector-typed loop-carried values get split into halves and re-packed by the compiler
		int H[RP], E[RP];
#pragma unroll
		for (int r = 0; r < RP; r++) { H[r] = 0; E[r] = 0; }
		int tc = 0x00040004;          // target codes of my two halves (N = neutral while the pipeline fills)
		int hbot = 0, fbot = 0, cm = 0, recv_h_last = 0, fpo = 0;
		v2u hzacc = (v2u){ 0, 0 };      // != 0: the unit goes to the stripe-faithful kernel (coarse test of short queries)
		int chunk = CODE_N;
		int first_enter = 0x7fffffff;          // first step of this unit in which the hazard branch ran (wave-uniform)
		v2u ubacc = (v2u){ 0, 0 };             // running maximum of the current block of steps (banded stage 3)
		// DUMP only: the reference's OWN E.  Its lazy-F loop corrects H but not E (sswNew.cpp:355: "don't update E"), so E follows
		// the H of the main pass, which only knows the F chain restarted at the top of each stripe (Fm).  The H values are the
		// same either way (a gap pair in the order down-right scores what right-down scores), the E array is not, and a
		// checkpoint has to be the reference's exact state.
		int Es[RP]; int fmbot = 0;
#pragma unroll
		for (int r = 0; r < RP; r++) Es[r] = 0;
		// checkpoint columns are ascending and a half's column moves up by one per step: each half only watches its NEXT one
		int dlast = -1, dnext[2] = { 0x7fffffff, 0x7fffffff }, dnx[2] = { 0, 0 };
		int drow0[2] = { 0, 0 }, drows[2] = { 0, 0 };
		const int32_t* dc = nullptr;
		constexpr int SNAP_DW = 2 * RP + 6;
		if constexpr (DUMP) {
			dc = a.dump_cols + item.first;
			if (item.count > 0) { dlast = dc[item.count - 1]; dnext[0] = dnext[1] = dc[0]; }
			for (int h = 0; h < 2; h++) lane_rows(128 * a.tile + 2 * lane + h, a.seg_len16, a.vs, &drow0[h], &drows[h]);
			if (item.step0 > 0) {
				// continue from the snapshot the main pass took at the top of step step0.  The reference's E starts as the textbook
				// E and is exact 64 columns later (a surplus decays by the gap extension, 4 a column, from at most 250; the
				// restarted F chain follows within the column): the host only asks for columns >= step0 + 64 here.
				const uint32_t* sp = a.snap + ((size_t)unit * a.snap_per_unit + (item.step0 / SCAN_SNAP_STEPS - 1)) * (SNAP_DW * 64) + lane;
#pragma unroll
				for (int r = 0; r < RP; r++) { H[r] = (int)sp[(2 * r) * 64]; E[r] = (int)sp[(2 * r + 1) * 64]; Es[r] = E[r]; }
				tc = (int)sp[(2 * RP) * 64]; hbot = (int)sp[(2 * RP + 1) * 64]; fbot = (int)sp[(2 * RP + 2) * 64]; cm = (int)sp[(2 * RP + 3) * 64];
				recv_h_last = (int)sp[(2 * RP + 4) * 64]; fpo = (int)sp[(2 * RP + 5) * 64];
			}
		}
		// (the DUMP variant only has to reach its last checkpoint column)
		const int nsteps = DUMP ? (dlast < 0 ? 0 : (dlast + 1 < n ? dlast + 1 : n) + 127) : n + 127;
		// The step body is instantiated twice per loop iteration: within one step the new H column is written into the
		// registers the profile rows were loaded into (the old H column is still being read), so consecutive steps
		// alternate between two register banks; with a single copy of the body the compiler has to move the whole column
		// back at the end of every step (~16 v_mov_b64 of ~230 instructions).
		auto do_step = [&](const int step) __attribute__((always_inline)) {
			if ((step & 63) == 0) {
				const int c = step + lane;
				chunk = c < n ? (int)tc_unit[c] : CODE_N;
				if (!first_tile) bchunk = c < n ? bnd[c] : make_uint2(0u, 0u);     // bottom row of the previous tile
			}
			const int newcode = __builtin_amdgcn_readlane(chunk, step & 63);
			// hand-over from virtual lane v-1 (computed one step ago; lane 0 takes zeros or the previous tile's bottom row)
			tc = vshift(tc, newcode << 16);
			int recv_h, recv_f, recv_cm, recv_fp;
			if (first_tile) {
				recv_h = vshift0(hbot); recv_f = vshift0(fbot); recv_cm = vshift0(cm); recv_fp = vshift0(fpo);
			} else {
				const uint32_t bx = (uint32_t)__builtin_amdgcn_readlane((int)bchunk.x, step & 63);
				const uint32_t by = (uint32_t)__builtin_amdgcn_readlane((int)bchunk.y, step & 63);
				recv_h = vshift(hbot, (int)(bx << 16)); recv_f = vshift(fbot, (int)(bx & 0xffff0000u));
				recv_cm = vshift(cm, (int)(by << 16)); recv_fp = vshift(fpo, (int)(by & 0xffff0000u));
			}
			const int t_lo = tc & 0xff, t_hi = (tc >> 16) & 0xff;
			const uint8_t* pa = pl + t_lo * SCAN_CODE_STRIDE;
			const uint8_t* pb = pl + t_hi * SCAN_CODE_STRIDE + 48;
			const int hdiag0 = recv_h_last;           // H[i0-1][c-1]
			recv_h_last = recv_h;
			v2u f = __builtin_bit_cast(v2u, recv_f);
			v2u fm = (v2u){ 0, 0 };
			if constexpr (DUMP) fm = __builtin_bit_cast(v2u, vshift0(fmbot)) & ~startm;      // the main pass starts every stripe with F = 0
			v2s lmx[4] = { (v2s){ 0, 0 }, (v2s){ 0, 0 }, (v2s){ 0, 0 }, (v2s){ 0, 0 } };   // independent chains: no back-to-back dependent v_pk_max
			constexpr int ROWS_PER_LOAD = 8;
			constexpr int NLOAD = (RP + ROWS_PER_LOAD - 1) / ROWS_PER_LOAD;
			v4i PA[NLOAD], PB[NLOAD];
#pragma unroll
			for (int g = 0; g < NLOAD; g++) {
				PA[g] = *reinterpret_cast<const v4i*>(pa + 16 * g);
				PB[g] = *reinterpret_cast<const v4i*>(pb + 16 * g);
			}
			auto score_of = [&](int r) -> int {
				const int g = r / ROWS_PER_LOAD, k = r % ROWS_PER_LOAD;
				return __builtin_amdgcn_perm(PB[g][k >> 1], PA[g][k >> 1], (k & 1) ? 0x07060302 : 0x05040100);
			};
			// Row r needs the OLD H[r-1] (diagonal) and writes the NEW H[r].  The add of row r+1 (old H[r] + its score) is
			// issued before H[r] is overwritten, so the new value can go into the same register: no second copy of the H
			// column and no register moves at the end of the step.
			// (the sum is forced into the register of the score, which dies there: the compiler would otherwise add in place
			//  over the old H, keep that register busy until the next row and have to park the new H somewhere else)
			auto diag_plus_score = [](int hold, int sc) -> v2s { asm("v_pk_add_i16 %0, %1, %0 clamp" : "+v"(sc) : "v"(hold)); return s_from(sc); };
			v2s t = diag_plus_score(hdiag0, score_of(0));
#pragma unroll
			for (int r = 0; r < RP; r++) {
				v2s tnext = t;
				if (r + 1 < RP) tnext = diag_plus_score(H[r], score_of(r + 1));
				if constexpr (DUMP) {
					const v2s hm = __builtin_elementwise_max(__builtin_elementwise_max(t, s_from(Es[r])), as_s(fm));
					const v2u hom = __builtin_elementwise_sub_sat(as_u(hm), (v2u){ 2 * GAP_OPEN, 2 * GAP_OPEN });
					Es[r] = to_int(__builtin_elementwise_max(__builtin_elementwise_sub_sat(u_fromi(Es[r]), (v2u){ 2 * GAP_EXT, 2 * GAP_EXT }), hom));
					const v2u fmn = __builtin_elementwise_max(__builtin_elementwise_sub_sat(fm, (v2u){ 2 * GAP_EXT, 2 * GAP_EXT }), hom);
					fm = (r == RP - 1) ? ((fmn & actm) | (fm & ~actm)) : fmn;
				}
				v2s h = __builtin_elementwise_max(t, s_from(E[r]));
				// h = max(h, f), written into the register that held the OLD H[r] (tied dummy operand; its last real use was
				// the sum for row r+1 above): the H column stays where it is from step to step
				// (tnext is passed as an unused operand only to order this after the sum above, which still reads the old H[r])
				{ int hn; asm("v_pk_max_i16 %0, %2, %3" : "=v"(hn) : "0"(H[r]), "v"(to_int(h)), "v"(to_int(f)), "v"(to_int(tnext))); h = s_from(hn); }
				H[r] = to_int(h);
				const v2u ho = __builtin_elementwise_sub_sat(as_u(h), (v2u){ 2 * GAP_OPEN, 2 * GAP_OPEN });
				E[r] = to_int(__builtin_elementwise_max(__builtin_elementwise_sub_sat(u_fromi(E[r]), (v2u){ 2 * GAP_EXT, 2 * GAP_EXT }), ho));
				const v2u fnew = __builtin_elementwise_max(__builtin_elementwise_sub_sat(f, (v2u){ 2 * GAP_EXT, 2 * GAP_EXT }), ho);
				if (r == RP - 1) {
					// a half that owns only RP-1 rows passes F and its bottom H through unchanged
					f = (fnew & actm) | (f & ~actm);
					lmx[r & 3] = __builtin_elementwise_max(lmx[r & 3], as_s(as_u(h) & actm));
					if (RP > 1) hbot = (to_int(h) & (int)act) | (H[RP > 1 ? RP - 2 : 0] & ~(int)act);
					else hbot = to_int(h);
				} else {
					f = fnew;
					lmx[r & 3] = __builtin_elementwise_max(lmx[r & 3], h);
				}
				t = tnext;
			}
			const v2s lmax = __builtin_elementwise_max(__builtin_elementwise_max(lmx[0], lmx[1]), __builtin_elementwise_max(lmx[2], lmx[3]));
			// pin the reduction here: if the compiler sinks it below the hazard branch, the 22 pre-branch H values stay alive
			// next to the (possibly tainted) ones and the common path pays a register copy per row at the join
			asm volatile("" :: "v"(to_int(lmax)));
			if constexpr (!DUMP) ubacc = __builtin_elementwise_max(ubacc, as_u(lmax));
			fbot = to_int(f);
			if constexpr (DUMP) fmbot = to_int(fm);
			// ---- Q2 hazard.  The reference's lazy-F loop leaves early (signed compare, sswNew.cpp:369) only while a
			// stripe's propagated boundary value Fp = F[b] - 4j is >= 132 and the H it has just corrected is < 144
			// (then vF >= 128 reads as negative, vH < 128 as positive).  From then on the rows of that stripe whose H is
			// exactly the propagated value hold something smaller in the reference: those cells get the taint bit.
			// fp = F crossing a stripe start, carried down the rows of the stripe; the row loop runs only in the (rare)
			// steps where some lane holds fp >= 132 or an armed chain.  All values are doubled (see the header).
			fpo = 0;
			if (lvl2) {
				// fpo halves: bits 0..14 = propagated value, bit 15 = "an early exit was possible at an earlier row"
				// cheap superset first (3 VALU): a stripe-start lane receives F >= 132, or some lane receives a live chain;
				// the exact condition is only evaluated behind it
				const v2u cand = pk_subs_k(__builtin_bit_cast(v2u, recv_f) & startm, 263u * 0x10001u) | __builtin_bit_cast(v2u, recv_fp);
				bool enter = false;
				v2u fp_in = (v2u){ 0, 0 }, arm_in = (v2u){ 0, 0 };
				if (__builtin_amdgcn_ballot_w64(to_int(cand) != 0) != 0ull) {
					const v2u fpraw = (__builtin_bit_cast(v2u, recv_f) & startm) | (__builtin_bit_cast(v2u, recv_fp) & ~startm);
					fp_in = fpraw & (v2u){ 0x7fff, 0x7fff };
					arm_in = (fpraw >> (v2u){ 15, 15 }) & ~startm;
					const v2u hot = pk_subs_k(fp_in, 263u * 0x10001u) | pk_minu(arm_in, fp_in);
					enter = __builtin_amdgcn_ballot_w64(to_int(hot) != 0) != 0ull;
				}
				if (enter) {
					first_enter = first_enter < step ? first_enter : step;
					v2u fp = fp_in, arm = arm_in;
					const v2u one = (v2u){ 1, 1 };
					constexpr uint32_t K1 = 0x00010001u, K263 = 263u * 0x10001u, K288 = 288u * 0x10001u, KE = (2u * GAP_EXT) * 0x10001u,
						KO = (2u * GAP_OPEN) * 0x10001u;
					// pass 1 only READS H and E and collects the taint decisions as bit r of (dh, de)[r / 16]; pass 2 ORs them
					// in.  Every read of the old value thus precedes the in-place update, so H[r] / E[r] stay in the registers
					// the common path uses and the join needs no copies.
					uint32_t dh[2] = { 0u, 0u }, de[2] = { 0u, 0u };
#pragma unroll
					for (int r = 0; r < RP; r++) {
						const v2u hr = u_fromi(H[r]);
						const v2u ge = pk_minu_k(pk_subs_k(fp, K263), K1);                     // Fp >= 132
						// H < 144; a tainted H may be smaller in the reference, so it counts as "< 144" too
						v2u lt = pk_minu_k(pk_ksubs(K288, hr), K1) | (hr & one);
						// the reference keeps a smaller H only where, after a possible early exit, H is exactly the propagated value
						v2u eq = pk_ksubs(K1, pk_subs(hr, fp));                                // H <= Fp (H >= Fp always)
						v2u nfp = pk_subs_k(fp, KE);
						if (r == RP - 1) { lt &= actm; eq &= actm; nfp = (nfp & actm) | (fp & ~actm); }
						const v2u dev = pk_minu(pk_minu(eq, fp), arm);                         // 0 / 1
						// E of this row was just derived from the untainted H: taint it when it came from H (or ties with it)
						const v2u ho = pk_subs_k(hr, KO);
						const v2u efrom = pk_minu(pk_ksubs(K1, pk_subs(u_fromi(E[r]), ho)), ho);
						dh[r >> 4] |= (uint32_t)to_int(dev) << (r & 15);
						de[r >> 4] |= (uint32_t)to_int(pk_minu(efrom, dev)) << (r & 15);
						arm = pk_maxu(arm, pk_minu(ge, lt));
						fp = nfp;
					}
#pragma unroll
					for (int r = 0; r < RP; r++) {
						or_in_place(H[r], (int)((dh[r >> 4] >> (r & 15)) & K1));
						or_in_place(E[r], (int)((de[r >> 4] >> (r & 15)) & K1));
					}
					if (RP > 1) hbot = (H[RP - 1] & (int)act) | (H[RP > 1 ? RP - 2 : 0] & ~(int)act);
					else hbot = H[0];
					fpo = to_int(fp | (arm << (v2u){ 15, 15 }));
				}
			} else {
				// short stripes (ceil(m/16) < 96): a chain may cross several stripes, the row analysis does not hold; any
				// F[b] >= 132 sends the unit to the stripe-faithful kernel
				hzacc |= __builtin_elementwise_sub_sat(__builtin_bit_cast(v2u, recv_f), fthr2);
			}
			if constexpr (DUMP) {
				// my lo half has just finished column step - 2*lane, my hi half column step - 2*lane - 1
				const int col_lo = step - 2 * lane, col_hi = col_lo - 1;
				if (__builtin_amdgcn_ballot_w64(col_lo == dnext[0] || col_hi == dnext[1]) != 0ull) {
					if (col_lo == dnext[0]) {
						uint16_t* sp = a.dump_state + (size_t)(item.first + dnx[0]) * 2 * a.rows_total + drow0[0];
#pragma unroll
						for (int r = 0; r < RP; r++) if (r < drows[0]) { sp[r] = (uint16_t)H[r]; sp[a.rows_total + r] = (uint16_t)Es[r]; }
						dnx[0]++;
						dnext[0] = dnx[0] < item.count ? dc[dnx[0]] : 0x7fffffff;
					}
					if (col_hi == dnext[1]) {
						uint16_t* sp = a.dump_state + (size_t)(item.first + dnx[1]) * 2 * a.rows_total + drow0[1];
#pragma unroll
						for (int r = 0; r < RP; r++) if (r < drows[1]) { sp[r] = (uint16_t)((uint32_t)H[r] >> 16); sp[a.rows_total + r] = (uint16_t)((uint32_t)Es[r] >> 16); }
						dnx[1]++;
						dnext[1] = dnx[1] < item.count ? dc[dnx[1]] : 0x7fffffff;
					}
				}
			}
			cm = to_int(__builtin_elementwise_max(__builtin_bit_cast(v2u, recv_cm), as_u(lmax)));
			const int cdone = step - 127;
			if (lane == 63 && cdone >= 0) {
				if (last_tile) out[cdone] = (uint16_t)((uint32_t)cm >> 16);
				else bnd[cdone] = make_uint2(((uint32_t)hbot >> 16) | ((uint32_t)fbot & 0xffff0000u), ((uint32_t)cm >> 16) | ((uint32_t)fpo & 0xffff0000u));
			}
		};
		int step = DUMP ? item.step0 : 0;
		for (; step + 1 < nsteps; step += 2) {
			if constexpr (!DUMP) {
				// pipeline snapshot (every SCAN_SNAP_STEPS steps, all lanes at once: ~50 stores per 1024 steps of ~230 instructions)
				if (a.snap && step != 0 && (step & (SCAN_SNAP_STEPS - 1)) == 0 && step / SCAN_SNAP_STEPS <= a.snap_per_unit) {
					uint32_t* sp = a.snap + ((size_t)unit * a.snap_per_unit + (step / SCAN_SNAP_STEPS - 1)) * (SNAP_DW * 64) + lane;
#pragma unroll
					for (int r = 0; r < RP; r++) { sp[(2 * r) * 64] = (uint32_t)H[r]; sp[(2 * r + 1) * 64] = (uint32_t)E[r]; }
					sp[(2 * RP) * 64] = (uint32_t)tc; sp[(2 * RP + 1) * 64] = (uint32_t)hbot; sp[(2 * RP + 2) * 64] = (uint32_t)fbot; sp[(2 * RP + 3) * 64] = (uint32_t)cm;
					sp[(2 * RP + 4) * 64] = (uint32_t)recv_h_last; sp[(2 * RP + 5) * 64] = (uint32_t)fpo;
				}
			}
			do_step(step); do_step(step + 1);
			if constexpr (!DUMP) {
				// (steps come in pairs: a block of SCAN_UBLK_STEPS steps ends after an odd step)
				if (((step + 1) & (SCAN_UBLK_STEPS - 1)) == SCAN_UBLK_STEPS - 1) {
					if (a.ublk) a.ublk[(((size_t)unit * a.ntiles + a.tile) * a.ublk_blocks + (step / SCAN_UBLK_STEPS)) * 64 + lane] = ublk_pack(ubacc);
					ubacc = (v2u){ 0, 0 };
				}
			}
		}
		if (step < nsteps) do_step(step);
		if constexpr (!DUMP) {
			// the last, partial block (nsteps - 1 is its last step unless the block was just closed)
			if (a.ublk && (nsteps & (SCAN_UBLK_STEPS - 1)) != 0) a.ublk[(((size_t)unit * a.ntiles + a.tile) * a.ublk_blocks + ((nsteps - 1) / SCAN_UBLK_STEPS)) * 64 + lane] = ublk_pack(ubacc);
		}
		if (a.unit_hz && __builtin_amdgcn_ballot_w64(to_int(hzacc) != 0) != 0ull && lane == 0) atomicOr(a.unit_hz + unit, 1);
		if (!DUMP && a.unit_first && first_enter != 0x7fffffff && lane == 0) atomicMin(a.unit_first + unit, first_enter);
	}
}

template <int RP>
static hipError_t launch_scan_dump_t(const ScanArgs& a, hipStream_t st)
{
	hipError_t err = hipMemsetAsync(a.counter, 0, sizeof(uint32_t), st);
	if (err != hipSuccess) return err;
	const long blocks = ((long)a.nwork + 3) / 4;              // one work item per wave
	hipLaunchKernelGGL((k_scan<RP, true>), dim3((unsigned)blocks), dim3(256), (size_t)5 * SCAN_CODE_STRIDE, st, a);
	return hipGetLastError();
}

template <int RP>
static hipError_t launch_scan_t(const ScanArgs& a, hipStream_t st)
{
	hipError_t err = hipMemsetAsync(a.counter, 0, sizeof(uint32_t), st);
	if (err != hipSuccess) return err;
	// Short-lived workgroups (each wave takes about two units from the queue) instead of a persistent grid: slots free up every
	// few milliseconds, so the latency-bound kernels of the other batches in flight get dispatched at once instead of waiting
	// for this kernel to drain.
	constexpr int WPB = 4, PER_WAVE = 2;                 // waves per workgroup, units per wave
	const long blocks = ((long)a.nwork + WPB * PER_WAVE - 1) / (WPB * PER_WAVE);
	hipLaunchKernelGGL((k_scan<RP>), dim3((unsigned)blocks), dim3(256), (size_t)5 * SCAN_CODE_STRIDE, st, a);
	return hipGetLastError();
}

// virtual lanes per reference stripe for a query of m rows: a multiple of 8 (so that a tile of 128 virtual lanes is
// always full) with at most 24 rows per virtual lane
int systolic_vs(int m) { const int seg = (m + 15) / 16; return 8 * ((seg + 191) / 192); }
int systolic_tiles(int m) { return systolic_vs(m) / 8; }
int systolic_snap_dwords(int m) { const int seg = (m + 15) / 16, vs = systolic_vs(m); return 2 * ((seg + vs - 1) / vs) + 6; }
bool systolic_fits(int m) { const int seg = (m + 15) / 16; return seg >= 8 && systolic_tiles(m) <= 16; }

hipError_t launch_scan(const ScanLaunch& L, hipStream_t st)
{
	if (L.nwork <= 0) return hipSuccess;
	if (!systolic_fits(L.m)) return hipErrorInvalidValue;      // tiny or huge queries: striped kernels
	ScanArgs a;
	a.tcodes = L.tcodes; a.unit_ids = L.unit_ids; a.unit_len = L.unit_len; a.nwork = L.nwork; a.tstride = L.tstride;
	a.counter = L.counter; a.qcodes = L.qcodes; a.m = L.m; a.m_pad = 16 * ((L.m + 15) / 16); a.seg_len16 = (L.m + 15) / 16;
	for (int i = 0; i < 25; i++) a.score[i] = L.score[i];
	a.colmax16 = L.colmax16; a.unit_hz = L.unit_hz; a.coarse = L.coarse;
	a.vs = systolic_vs(L.m); a.ntiles = a.vs / 8; a.boundary = L.boundary;
	a.unit_first = L.unit_first; a.snap = L.snap; a.snap_per_unit = L.snap_per_unit; a.dump_items = L.dump_items; a.dump_cols = L.dump_cols; a.dump_state = L.dump_state; a.rows_total = 16 * a.seg_len16;
	a.ublk = L.dump_items ? nullptr : L.ublk; a.ublk_blocks = L.ublk_blocks;
	if (a.ntiles > 1 && !a.boundary) return hipErrorInvalidValue;
	// RP must be exactly ceil(segLen/vs): every virtual lane then owns RP or RP-1 rows.  One launch per tile of 128
	// virtual lanes (long queries): tile t reads the bottom row tile t-1 left in `boundary` and overwrites it in place.
	const int rp = (a.seg_len16 + a.vs - 1) / a.vs;
	for (int t = 0; t < a.ntiles; t++) {
		a.tile = t;
		hipError_t err = hipErrorInvalidValue;
		switch (rp) {
#define FASIM_SCAN_CASE(N) case N: err = L.dump_items ? launch_scan_dump_t<N>(a, st) : launch_scan_t<N>(a, st); break;
		FASIM_SCAN_CASE(1) FASIM_SCAN_CASE(2) FASIM_SCAN_CASE(3) FASIM_SCAN_CASE(4) FASIM_SCAN_CASE(5) FASIM_SCAN_CASE(6)
		FASIM_SCAN_CASE(7) FASIM_SCAN_CASE(8) FASIM_SCAN_CASE(9) FASIM_SCAN_CASE(10) FASIM_SCAN_CASE(11) FASIM_SCAN_CASE(12)
		FASIM_SCAN_CASE(13) FASIM_SCAN_CASE(14) FASIM_SCAN_CASE(15) FASIM_SCAN_CASE(16) FASIM_SCAN_CASE(17) FASIM_SCAN_CASE(18)
		FASIM_SCAN_CASE(19) FASIM_SCAN_CASE(20) FASIM_SCAN_CASE(21) FASIM_SCAN_CASE(22) FASIM_SCAN_CASE(23) FASIM_SCAN_CASE(24)
#undef FASIM_SCAN_CASE
		default: break;
		}
		if (err != hipSuccess) return err;
	}
	return hipSuccess;
}

// ------------------------------------------------------------------------------------------------
// k_scan_post: one wave per unit.  Applies the reference's overflow rule (Q1, sswNew.cpp:384-395): the first
// column whose maximum reaches 251 and everything after it is treated as 0; derives the stage-1 score
// (own maximum unless a separate stage-1 pass supplied it), the threshold (int)(score*0.8), the hazard
// flag (a tainted column maximum that is a hit or decides the cut, or the kernel's own unit flag), and the
// ordered list of columns above the threshold.  Input values are 2*max + taint.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_scan_post(const uint16_t* __restrict__ colmax16, const int32_t* __restrict__ unit_ids,
	const int32_t* __restrict__ unit_len, int32_t tstride, const int32_t* __restrict__ stage1_in, uint32_t* __restrict__ hits,
	uint32_t hits_cap, uint32_t* __restrict__ hits_total, int32_t* __restrict__ hit_off, int32_t* __restrict__ hit_cnt,
	int32_t* __restrict__ thr_out, int32_t* __restrict__ stage1_out, int32_t* __restrict__ flags, const int32_t* __restrict__ unit_hz)
{
	const int unit = unit_ids[blockIdx.x];
	const int lane = threadIdx.x;
	const int n = unit_len[unit];
	const uint16_t* col = colmax16 + (int64_t)unit * tstride;
	int mx = 0, cut = n;
	for (int c0 = 0; c0 < n; c0 += 64) {
		const int c = c0 + lane;
		const int v = c < n ? (int)(col[c] >> 1) : 0;
		mx = max(mx, v);
		const unsigned long long over = __ballot(v >= 255 - BIAS);
		if (over && cut == n) cut = c0 + __ffsll((long long)over) - 1;
	}
	for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
	int s1 = mx;
	if (stage1_in && stage1_in[unit] >= 0) s1 = stage1_in[unit];
	const int thr = (int)((double)s1 * 0.8);          // Fasim-LongTarget.cpp:413
	int cnt = 0; bool hz = false;
	for (int c0 = 0; c0 < n; c0 += 64) {
		const int c = c0 + lane;
		const int raw = c < n ? (int)col[c] : 0;
		const int v = raw >> 1;
		if (c <= cut && (raw & 1) && (v > thr || v >= 255 - BIAS)) hz = true;
		const bool hit = c < cut && v > thr;
		cnt += __popcll(__ballot(hit));
	}
	hz = __ballot(hz) != 0ull || (unit_hz && unit_hz[unit] != 0);
	uint32_t off = 0;
	if (lane == 0) off = atomicAdd(hits_total, (uint32_t)cnt);
	off = __shfl(off, 0, 64);
	if (lane == 0) {
		hit_off[unit] = (int32_t)off; hit_cnt[unit] = cnt; thr_out[unit] = thr; stage1_out[unit] = s1;
		flags[unit] = (hz ? 1 : 0) | (cut < n ? 2 : 0) | (mx >= 16383 ? 4 : 0);     // 16383 = saturated 16-bit lanes
	}
	if ((uint64_t)off + (uint64_t)cnt > hits_cap) return;
	uint32_t wpos = off;
	for (int c0 = 0; c0 < n; c0 += 64) {
		const int c = c0 + lane;
		const int v = c < n ? (int)(col[c] >> 1) : 0;
		const bool hit = c < cut && v > thr;
		const unsigned long long b = __ballot(hit);
		if (hit) hits[wpos + __popcll(b & ((1ull << lane) - 1ull))] = ((uint32_t)c << 8) | (uint32_t)v;
		wpos += __popcll(b);
	}
}

hipError_t launch_scan_post(const uint16_t* colmax16, const int32_t* unit_ids, int32_t nwork, const int32_t* unit_len,
	int32_t tstride, const int32_t* stage1_in, uint32_t* hits, uint32_t hits_cap, uint32_t* hits_total, int32_t* hit_off,
	int32_t* hit_cnt, int32_t* thr_out, int32_t* stage1_out, int32_t* flags, const int32_t* unit_hz, hipStream_t st)
{
	if (nwork <= 0) return hipSuccess;
	hipError_t err = hipMemsetAsync(hits_total, 0, sizeof(uint32_t), st);
	if (err != hipSuccess) return err;
	hipLaunchKernelGGL(k_scan_post, dim3((unsigned)nwork), dim3(64), 0, st, colmax16, unit_ids, unit_len, tstride, stage1_in, hits,
		hits_cap, hits_total, hit_off, hit_cnt, thr_out, stage1_out, flags, unit_hz);
	return hipGetLastError();
}

// maximum of the column maxima (stored as 2*max + taint) of each listed unit (used for the separate stage-1 pass of units with N)
__global__ void __launch_bounds__(64) k_max16(const uint16_t* __restrict__ colmax16, const int32_t* __restrict__ unit_ids,
	const int32_t* __restrict__ unit_len, int32_t tstride, int32_t* __restrict__ out)
{
	const int unit = unit_ids[blockIdx.x];
	const int n = unit_len[unit];
	const uint16_t* col = colmax16 + (int64_t)unit * tstride;
	int mx = 0;
	for (int c = threadIdx.x; c < n; c += 64) mx = max(mx, (int)(col[c] >> 1));
	for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
	if (threadIdx.x == 0) out[unit] = mx;
}

hipError_t launch_max16(const uint16_t* colmax16, const int32_t* unit_ids, int32_t nwork, const int32_t* unit_len,
	int32_t tstride, int32_t* out, hipStream_t st)
{
	if (nwork <= 0) return hipSuccess;
	hipLaunchKernelGGL(k_max16, dim3((unsigned)nwork), dim3(64), 0, st, colmax16, unit_ids, unit_len, tstride, out);
	return hipGetLastError();
}

} // namespace fasim
