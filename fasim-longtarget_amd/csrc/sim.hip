// fasim-longtarget_amd/csrc/sim.hip -- row f3: classic SIM (the reference's -F path, sim.h:410-1143) on gfx950.
//
// Two kernels.  k_sim_forward replaces the first double loop of SIM() (sim.h:506-571): affine-gap local alignment scores of the
// whole (lncRNA x target) matrix where every cell also carries the START POINT of its best alignment, with the reference's
// tie-break ORDER (sim.h:481-493): larger score, then larger start row, then larger start column.  Cells whose score
// exceeds the threshold go to the K = 50 node list (addnode, sim.h:99-148), which is order dependent (row-major).
// k_sim_resweep (further down) replaces the re-sweeps of the rectangle an alignment may have influenced, which follow each of
// the K tracebacks (sim.h:884-1141); the tracebacks themselves are host code (host_sim.cpp, SimUnit::next_round).
//
// k_sim_forward, layout: one wave64 per unit.  The query rows are cut into strips of 64 (lane = row); a strip is swept over the
// target columns as a wavefront (lane l works on column step - l), so the left neighbour is the lane's own previous cell and
// the upper / diagonal neighbours arrive from lane l-1 by a wave shift.  Lane 63 leaves the strip's bottom row (C and D per
// column) in a per-unit row buffer in HBM, which lane 0 of the next strip reads back 64 columns at a time.
//
// The node list lives in the wave too: lane k holds node k (score, start, end, bounding box).  A lane appends the cells of
// ITS row that pass the threshold to the row's own segment of a per-unit scratch buffer (columns ascend with the steps), and
// after each strip the wave replays the 64 segments in row order through addnode, 64 events of one row at a time
// (nodes_add_batch below).  That is the row-major order of the reference without a sort, and nothing but the 50 nodes ever
// leaves the GPU.
//
// A DP state = one 64-bit key  (score + SIM_BIAS) << 32 | start_row << 16 | start_col,  so ORDER is an unsigned 64-bit max
// and "score - k" is a subtraction in the top field.  Scores are the reference's x10 values (match 50, mismatch -40, gap
// open 120, extension 40).  Limits: query and target at most 65534 long (16-bit start fields, and a re-sweep starts lines at
// row M + 1 / column N + 1; 8191 until round 3).  Four units per 256-thread workgroup (one wave each: the waves share nothing).
// Integer DP: no MFMA; 64-bit VALU arithmetic, DPP scans (see DESIGN.md section 9 for what a step costs).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace fasim {

constexpr int64_t SIM_BIAS = 1 << 23;           // scores stay within (-2^23, 2^23): |gap run| <= 120 + 40 * 65535 < 2^22, best score 50 * 65535 < 2^22
constexpr int SIM_SHIFT = 32;
constexpr int SIM_FIELD = 16;                   // bits of the start row / start column fields
constexpr int64_t SIM_MATCH = 50, SIM_MISMATCH = -40, SIM_Q = 120, SIM_R = 40;

__device__ __forceinline__ uint64_t sim_key(int64_t s, uint32_t i, uint32_t j) { return ((uint64_t)(s + SIM_BIAS) << SIM_SHIFT) | ((uint64_t)i << SIM_FIELD) | (uint64_t)j; }
__device__ __forceinline__ int64_t sim_score(uint64_t k) { return (int64_t)(k >> SIM_SHIFT) - SIM_BIAS; }
// Unsigned 64-bit maximum of two keys in ONE instruction: read as doubles, positive normal numbers order like their bit patterns
// (a key is (score + 2^23) << 32 | ..., with score + 2^23 in [2^22, 2^25): exponent field 4 ... 9, never a denormal, NaN or
// infinity; 0, the "no value" of the scans, is +0.0, below every key), and the maximum of two of them is a copy of one of them.
// v_cmp_gt_u64 + 2 x v_cndmask otherwise, fourteen times per 64-cell step.
__device__ __forceinline__ uint64_t umax64(uint64_t a, uint64_t b)
{
	double r;
	asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(__longlong_as_double((long long)a)), "v"(__longlong_as_double((long long)b)));
	return (uint64_t)__double_as_longlong(r);
}
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l)
{
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
	return ((uint64_t)hi << 32) | lo;
}

// DPP forms (no LDS crossbar round trip): value of lane - 1 (0 into lane 0), inclusive prefix maximum, wave maximum
#define SIM_DPP(old, v, ctrl, rows) __builtin_amdgcn_update_dpp((int)(old), (int)(v), (ctrl), (rows), 0xf, false)
__device__ __forceinline__ uint64_t dpp_up64(uint64_t v)
{
	const uint32_t lo = (uint32_t)SIM_DPP(0, (uint32_t)v, 0x138 /* wave_shr:1 */, 0xf), hi = (uint32_t)SIM_DPP(0, (uint32_t)(v >> 32), 0x138, 0xf);
	return ((uint64_t)hi << 32) | lo;
}
template <int CTRL, int ROWS> __device__ __forceinline__ uint64_t dpp_max64_step(uint64_t v)
{
	const uint32_t lo = (uint32_t)SIM_DPP(0, (uint32_t)v, CTRL, ROWS), hi = (uint32_t)SIM_DPP(0, (uint32_t)(v >> 32), CTRL, ROWS);
	return umax64(v, ((uint64_t)hi << 32) | lo);          // lanes without a source see 0: keys are positive
}
__device__ __forceinline__ uint64_t prefix_max64(uint64_t v)
{
	v = dpp_max64_step<0x111, 0xf>(v);      // row_shr:1, 2, 4, 8: inclusive scan of each row of 16
	v = dpp_max64_step<0x112, 0xf>(v);
	v = dpp_max64_step<0x114, 0xf>(v);
	v = dpp_max64_step<0x118, 0xf>(v);
	v = dpp_max64_step<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
	v = dpp_max64_step<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3
	return v;
}
__device__ __forceinline__ int wave_max_i32(int v)
{
	const int lowest = (int)0x80000000;
	v = max(v, SIM_DPP(lowest, v, 0x111, 0xf)); v = max(v, SIM_DPP(lowest, v, 0x112, 0xf));
	v = max(v, SIM_DPP(lowest, v, 0x114, 0xf)); v = max(v, SIM_DPP(lowest, v, 0x118, 0xf));
	v = max(v, SIM_DPP(lowest, v, 0x142, 0xa)); v = max(v, SIM_DPP(lowest, v, 0x143, 0xc));
	return __builtin_amdgcn_readlane(v, 63);
}

// ---- the node list (sim.h:99-148) in a wave: lane k < cnt holds node k.  addnode: a known start point is updated (a strictly
// larger score moves the end point; the bounding box grows), a new one is appended or, with K nodes present, overwrites the FIRST
// node of lowest score whatever its own score is.  In random sequence 8 % of the cells of the first sweep are such events (the
// x10 scores are compared with the unscaled threshold) and 80 % of the cells of a re-sweep (threshold 1), which makes this
// replay -- serial by definition -- the critical path of both kernels: the lowest score and the lanes that hold it are kept wave-uniform, so an
// eviction is a count-trailing-zeros, and the wave-wide minimum is only recomputed when that set runs empty.
struct NodeList {
	int score, start, endi, endj, top, bot, left, right;      // start = start row << 16 | start column
	int cnt;                                                  // wave-uniform
	int low; unsigned long long low_mask;                     // valid while cnt == SIM_K (nodes_add keeps them exact)
	int junk, l2;                                             // batch path: the first node of lowest score; a lower bound of all other scores
	int dbg_pass, dbg_bad;                                    // FASIM_SIM_DEBUG counters
	uint32_t* bits;                                           // LDS, SIM_BITS bits: hash set of the start points of all nodes but the junk slot
};
__device__ __forceinline__ void nodes_find_low(NodeList& L, int lane)
{
	unsigned v = lane < L.cnt ? (unsigned)L.score : 0xffffffffu;          // node scores are positive
	for (int o = 32; o; o >>= 1) { const unsigned w = (unsigned)__shfl_xor((int)v, o, 64); v = w < v ? w : v; }
	L.low = (int)v;
	L.low_mask = __ballot(lane < L.cnt && (unsigned)L.score == v);
}
__device__ __forceinline__ void nodes_add(NodeList& L, int lane, int c, int start, int ei, int ej)
{
	const unsigned long long live = (1ull << L.cnt) - 1;                  // cnt <= SIM_K = 50
	const unsigned long long same = __ballot(L.start == start) & live;
	if (same) {
		const int t = (int)__builtin_ctzll(same);
		const int old = __builtin_amdgcn_readlane(L.score, t);
		if (lane == t) {
			if (L.score < c) { L.score = c; L.endi = ei; L.endj = ej; }
			L.top = min(L.top, ei); L.bot = max(L.bot, ei); L.left = min(L.left, ej); L.right = max(L.right, ej);
		}
		if (old < c && L.cnt == SIM_K && ((L.low_mask >> t) & 1)) { L.low_mask &= ~(1ull << t); if (!L.low_mask) nodes_find_low(L, lane); }
		return;
	}
	int t;
	if (L.cnt < SIM_K) t = L.cnt++;
	else t = (int)__builtin_ctzll(L.low_mask);                            // the first node of lowest score gives way (sim.h:129-136)
	if (lane == t) { L.score = c; L.start = start; L.endi = ei; L.endj = ej; L.top = L.bot = ei; L.left = L.right = ej; }
	if (L.cnt < SIM_K) return;
	if (!L.low_mask) { nodes_find_low(L, lane); return; }                 // the list has just become full
	if (c < L.low) { L.low = c; L.low_mask = 1ull << t; }
	else if (c > L.low) { L.low_mask &= ~(1ull << t); if (!L.low_mask) nodes_find_low(L, lane); }
}

// The batch path asks for every event whether its start point is one of the 49 other nodes': 49 compares per step, nearly always
// in vain (0.7 % of the events of a random record hit).  A hash set of those start points in LDS (16 K bits per wave, rebuilt
// whenever the roles change) answers "none of them" for the whole step with one LDS read per event in most steps.
constexpr int SIM_BITS_LOG = 14, SIM_BITS_WORDS = (1 << SIM_BITS_LOG) / 32;
__device__ __forceinline__ uint32_t start_hash(int start) { return ((uint32_t)start * 0x9E3779B1u) >> (32 - SIM_BITS_LOG); }

// (junk, l2) for the batch path, from an exact (low, low_mask)
__device__ __forceinline__ void nodes_steady(NodeList& L, int lane)
{
	L.junk = -1;
	if (L.cnt < SIM_K) return;
	const int t = (int)__builtin_ctzll(L.low_mask);
	unsigned v = (lane < L.cnt && lane != t) ? (unsigned)L.score : 0xffffffffu;
	for (int o = 32; o; o >>= 1) { const unsigned w = (unsigned)__shfl_xor((int)v, o, 64); v = w < v ? w : v; }
	L.junk = t; L.l2 = (int)v;
	// the hash set of the others' start points, from scratch (the roles change on 3 % of the steps)
	for (int w = lane; w < SIM_BITS_WORDS; w += 64) L.bits[w] = 0;
	__builtin_amdgcn_s_waitcnt(0);
	if (lane < L.cnt && lane != t) { const uint32_t h = start_hash(L.start); atomicOr(&L.bits[h >> 5], 1u << (h & 31)); }
	__builtin_amdgcn_s_waitcnt(0);
}

// Up to 64 events of ONE row in column order (lane order), has = this lane carries one.  With a full list whose first lowest node
// (`junk`) scores below everything else, an event either belongs to one of the other 49 nodes -- in-place updates, which commute --
// or it is "junk": it overwrites the junk slot (or continues it, if it has the slot's start point), and as long as it scores below
// all the others (l2) the slot stays the first lowest node.  Of a batch's junk events only the last run of equal start points
// leaves a trace, so the batch costs one pass over the 49 start points instead of 64 serial addnode() calls.  Anything else (list
// not full, a junk event that would outrank another node) takes the serial path for the whole batch.
__device__ __forceinline__ void nodes_add_batch(NodeList& L, int lane, bool has, int c, int start, int ei, int ej)
{
	unsigned long long todo = __ballot(has);
	while (todo) {
		if (L.junk < 0) {
			// list not full yet: one event at a time
			const int b = (int)__builtin_ctzll(todo);
			todo &= todo - 1;
			nodes_add(L, lane, __builtin_amdgcn_readlane(c, b), __builtin_amdgcn_readlane(start, b), ei, __builtin_amdgcn_readlane(ej, b));
			if (L.cnt == SIM_K) nodes_steady(L, lane);
			continue;
		}
		const int t = L.junk;
		L.dbg_pass++;
		const bool mine = (todo >> lane) & 1;
		int hit = -1;
		const uint32_t hs = start_hash(start);
		if (__ballot(mine && ((L.bits[hs >> 5] >> (hs & 31)) & 1))) {
			const int others = lane == t ? -1 : L.start;                // -1 is no start point (rows and columns stay below 65535)
#pragma unroll
			for (int k = 0; k < SIM_K; k++) { const int sk = __builtin_amdgcn_readlane(others, k); hit = start == sk ? k : hit; }
			hit = mine ? hit : -1;
		}
		// the batch path takes the events before the first junk event that would outrank another node; that one goes through
		// addnode() on its own, after which the roles (junk slot, l2) are worked out again for the rest
		const unsigned long long bad = __ballot(mine && hit < 0 && c >= L.l2);
		const unsigned long long part = bad ? todo & ((1ull << (int)__builtin_ctzll(bad)) - 1) : todo;
		const bool now = (part >> lane) & 1;
		const bool junk = now && hit < 0;
		const unsigned long long jm = __ballot(junk);
		if (jm) {
			const int z = 63 - (int)__builtin_clzll(jm);                       // the last junk event and the run of its start point
			const int sz = __builtin_amdgcn_readlane(start, z);
			const unsigned long long neq = __ballot(junk && start != sz);
			const unsigned long long run = neq ? jm & ~((2ull << (63 - (int)__builtin_clzll(neq))) - 1) : jm;
			const bool continuing = !neq && __builtin_amdgcn_readlane(L.start, t) == sz;
			int best = -1, bj = 0;
			for (unsigned long long r = run; r; r &= r - 1) { const int b = (int)__builtin_ctzll(r); const int cb = __builtin_amdgcn_readlane(c, b); if (cb > best) { best = cb; bj = __builtin_amdgcn_readlane(ej, b); } }
			const int first_j = __builtin_amdgcn_readlane(ej, (int)__builtin_ctzll(run)), last_j = __builtin_amdgcn_readlane(ej, z);
			if (lane == t) {
				if (!continuing) { L.score = best; L.start = sz; L.endi = ei; L.endj = bj; L.top = L.bot = ei; L.left = first_j; L.right = last_j; }
				else {
					if (L.score < best) { L.score = best; L.endi = ei; L.endj = bj; }
					L.top = min(L.top, ei); L.bot = max(L.bot, ei); L.left = min(L.left, first_j); L.right = max(L.right, last_j);
				}
			}
		}
		for (unsigned long long hm = __ballot(now && hit >= 0); hm;) {
			const int k = __builtin_amdgcn_readlane(hit, (int)__builtin_ctzll(hm));
			const unsigned long long mk = __ballot(now && hit == k);
			hm &= ~mk;
			const bool in = (mk >> lane) & 1;
			const int best = wave_max_i32(in ? c : -1);
			const int bj = __builtin_amdgcn_readlane(ej, (int)__builtin_ctzll(__ballot(in && c == best)));     // the first event that reaches it
			const int first_j = __builtin_amdgcn_readlane(ej, (int)__builtin_ctzll(mk)), last_j = __builtin_amdgcn_readlane(ej, 63 - (int)__builtin_clzll(mk));
			if (lane == k) {
				if (L.score < best) { L.score = best; L.endi = ei; L.endj = bj; }
				L.top = min(L.top, ei); L.bot = max(L.bot, ei); L.left = min(L.left, first_j); L.right = max(L.right, last_j);
			}
		}
		todo &= ~part;
		if (bad) {
			const int b = (int)__builtin_ctzll(bad);
			todo &= ~(1ull << b);
			L.dbg_bad++;
			nodes_find_low(L, lane);                  // the batch path does not keep (low, low_mask)
			nodes_add(L, lane, __builtin_amdgcn_readlane(c, b), __builtin_amdgcn_readlane(start, b), ei, __builtin_amdgcn_readlane(ej, b));
			nodes_steady(L, lane);
		}
	}
}
__device__ __forceinline__ void nodes_load(NodeList& L, int lane, const SimNodeDev* src, int cnt, uint32_t* bits)
{
	L.bits = bits;
	L.cnt = cnt; L.low = 0; L.low_mask = 0; L.junk = -1; L.l2 = 0; L.dbg_pass = L.dbg_bad = 0;
	L.score = L.start = L.endi = L.endj = L.top = L.bot = L.left = L.right = 0;
	if (lane < cnt) {
		const SimNodeDev nd = src[lane];
		L.score = (int)nd.score; L.start = (int)((nd.stari << 16) | nd.starj); L.endi = (int)nd.endi; L.endj = (int)nd.endj;
		L.top = (int)nd.top; L.bot = (int)nd.bot; L.left = (int)nd.left; L.right = (int)nd.right;
	}
	if (cnt == SIM_K) { nodes_find_low(L, lane); nodes_steady(L, lane); }
}
__device__ __forceinline__ void nodes_store(const NodeList& L, int lane, SimNodeDev* dst)
{
	if (lane < L.cnt) {
		SimNodeDev o;
		o.score = L.score; o.stari = (L.start >> 16) & 0xffff; o.starj = L.start & 0xffff; o.endi = L.endi; o.endj = L.endj;
		o.top = L.top; o.bot = L.bot; o.left = L.left; o.right = L.right;
		dst[lane] = o;
	}
}

__global__ void __launch_bounds__(256) k_sim_forward(SimFwdArgs a, int32_t nunit)
{
	const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (unit >= nunit) return;
	const int lane = threadIdx.x & 63;
	const int N = a.unit_len[unit];
	const int M = a.m;
	const uint8_t* tc_unit = a.tcodes + (int64_t)unit * a.tstride;
	uint64_t* rowC = a.rowbuf + (int64_t)unit * 2 * a.row_stride;      // [0 .. N]: C of the finished strip's bottom row
	uint64_t* rowD = rowC + a.row_stride;
	SimEvent* seg = a.events + ((int64_t)unit * 64 + lane) * a.event_cap;      // my row's segment
	__shared__ uint32_t start_bits[4][SIM_BITS_WORDS];
	NodeList L;
	nodes_load(L, lane, nullptr, 0, start_bits[threadIdx.x >> 6]);
	const int64_t thr = a.min_score[unit];
	const uint64_t Rk = (uint64_t)SIM_R << SIM_SHIFT, QRk = (uint64_t)(SIM_Q + SIM_R) << SIM_SHIFT;
	const int nstrips = (M + 63) / 64;

	for (int strip = 0; strip < nstrips; strip++) {
		const int i = strip * 64 + lane + 1;                 // my query row (1-based)
		const bool row_ok = i <= M;
		const int qc = row_ok ? (int)a.qcodes[i - 1] : 250;
		const int last_lane = min(63, M - strip * 64 - 1);   // lane of the strip's bottom row
		// sweep-line state of my row (sim.h:508-515): c = 0, f = -Q, both starting at (i, 0)
		uint64_t c = sim_key(0, (uint32_t)i, 0), f = sim_key(-SIM_Q, (uint32_t)i, 0);
		uint64_t myC = 0, myD = 0;                           // what I hand to the row below: C and D of the column just finished
		uint64_t diag = sim_key(0, (uint32_t)(i - 1), 0);    // P for column 1: p = 0, (pi, pj) = (i - 1, 0)
		uint64_t bufC = 0, bufD = 0;                         // row-buffer chunk (lane l holds column chunk0 + l + 1)
		int tchunk = 0, tcode = 0;
		int nev = 0;                                         // cells of my row above the threshold in this strip
		const int nsteps = N + last_lane + 1;
		for (int step = 0; step < nsteps; step++) {
			if ((step & 63) == 0) {
				const int col = step + lane + 1;                 // 1-based column this lane prefetches for lane 0
				tchunk = col <= N ? (int)tc_unit[col - 1] : 4;
				if (strip > 0) { bufC = col <= N ? rowC[col] : 0; bufD = col <= N ? rowD[col] : 0; }
			}
			const int j = step - lane + 1;                       // my column at this step
			// target letter: lane 0 takes the new column, every other lane the letter lane-1 had one step ago
			const int newcode = __builtin_amdgcn_readlane(tchunk, step & 63);
			const int shifted = __shfl_up(tcode, 1, 64);
			tcode = lane == 0 ? newcode : shifted;
			// C and D of the row above in my column: lane-1's result of the previous step; lane 0: row 0 or the previous strip
			uint64_t upC = dpp_up64(myC), upD = dpp_up64(myD);
			if (lane == 0) {
				if (strip == 0) { upC = sim_key(0, 0, (uint32_t)j); upD = sim_key(-SIM_Q, 0, (uint32_t)j); }       // sim.h:497-505
				else { upC = readlane64(bufC, step & 63); upD = readlane64(bufD, step & 63); }
			}
			const bool valid = row_ok && j >= 1 && j <= N;
			if (valid) {
				f = umax64(f - Rk, c - QRk);                                                             // horizontal gap (sim.h:524-526)
				const uint64_t d = umax64(upD - Rk, upC - QRk);                                          // vertical gap (sim.h:527-533)
				const int64_t sc = (qc == tcode && qc < 4) ? SIM_MATCH : SIM_MISMATCH;
				const int64_t v = sim_score(diag) + sc;                                                  // (no aligned pair is excluded in the first sweep)
				uint64_t t = v <= 0 ? sim_key(0, (uint32_t)i, (uint32_t)j)
				                    : (uint64_t)((int64_t)diag + (sc << SIM_SHIFT));                     // keeps the diagonal's start point
				c = umax64(umax64(t, d), f);
				myC = c; myD = d;
				if (lane == last_lane) { rowC[j] = c; rowD[j] = d; }
			}
			if (j >= 1) diag = upC;                                                                    // C[i-1][j] is the diagonal of column j + 1
			                                                                                           // (before column 1 the start value (i-1, 0) stays)
			// cells above the threshold: appended to my row's segment (columns ascend)
			if (valid && sim_score(c) > thr) { SimEvent e; e.j = (uint32_t)j; e.pad = 0; e.key = c; seg[nev] = e; nev++; }
		}
		// ---- replay of the strip's rows through addnode (sim.h:99-148), in row order; 64 events per load
		// (the segments are rewritten strip after strip and read by other lanes of the wave: device-scope fence, so that no
		//  load is served from a line cached before the store)
		__builtin_amdgcn_s_waitcnt(0);
		__threadfence();
		for (int r = 0; r <= last_lane; r++) {
			const int cnt = __builtin_amdgcn_readlane(nev, r);
			const SimEvent* rs = a.events + ((int64_t)unit * 64 + r) * a.event_cap;
			const int ei = strip * 64 + r + 1;                         // the row of these cells
			for (int e0 = 0; e0 < cnt; e0 += 64) {
				const int nb = min(64, cnt - e0);
				uint32_t ej = 0, klo = 0, khi = 0;
				if (lane < nb) { const SimEvent e = rs[e0 + lane]; ej = e.j; klo = (uint32_t)e.key; khi = (uint32_t)(e.key >> 32); }
				const uint64_t key = ((uint64_t)khi << 32) | klo;
				nodes_add_batch(L, lane, lane < nb, (int)sim_score(key), (int)klo, ei, (int)ej);
			}
		}
		// the next strip reads the row buffer this strip has just written (same wave: program order is enough once the
		// stores have left the wave)
		__builtin_amdgcn_s_waitcnt(0);
		__threadfence_block();
	}
	nodes_store(L, lane, a.nodes + (int64_t)unit * SIM_K);
	if (lane == 0) a.node_count[unit] = L.cnt;
}

// ---- the re-sweeps between the K rounds (sim.h:884-1141) ------------------------------------------------------------------
// After the host has traced a round's alignment back, the rectangle that alignment may have influenced is swept again:
// backwards from its lower right corner, growing up and to the left one row or one column at a time until no DP state that
// leaves the rectangle starts inside it and no remaining node crosses it (no_cross, sim.h:150-165), then forwards over the
// final rectangle, feeding addnode.  One wave per unit; the units of a slice advance in lock step (one launch per round).
//
// A sweep line (a query row over target columns, or a target column over query rows) is worked on 64 positions at a time,
// lane = position.  The states across the line (S = best cell, G = gap entering from beyond the line) are independent per
// position; the gap ALONG the line is a running maximum of "value at q minus the cost of stretching to t", which with keys is
// an unsigned 64-bit prefix maximum of (key + R*q) -- six shuffle steps per 64 cells instead of a serial chain.  A cell reached
// by a gap never re-opens a better gap than the one it came by (Q > 0), so the prefix maximum over the cells' gap-free values
// is exact, ties included: keys order by (score, start row, start column) like ORDER (sim.h:481-493).
// The lines' states live in HBM / L2 and are written and read by different lanes of the one wave that owns the unit.  Stores are
// written through to the XCD's L2 and are complete once the wave's memory counter has drained; loads of them are issued at
// agent scope, which bypasses the CU's vector L1, so no stale line is ever served.  (A device-scope fence per line would write
// the whole L2 back each time: measured 300 us per 64 cells with 5 000 waves doing so.)
__device__ __forceinline__ uint64_t ld_l2(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wave_publish() { __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); }

struct SweepCarry { uint64_t corner, pre; };
struct SweepCell { uint64_t c, across, g; bool valid; int pos; };

__device__ __forceinline__ uint64_t key_sub(uint64_t k, int64_t s) { return k - ((uint64_t)s << SIM_SHIFT); }
__device__ __forceinline__ int key_i(uint64_t k) { return (int)((k >> SIM_FIELD) & 0xffff); }
__device__ __forceinline__ int key_j(uint64_t k) { return (int)(k & 0xffff); }

// 64 cells of a sweep line (sim.h:522-566 and its mirror images): position t0 + lane of n, at index p0 + dir * t.
// ROW: the line is query row `fix` and positions are target columns; else target column `fix` over query rows.
// what a chunk reads from memory: the states across the line and the letters of the sequence that varies along it.  The loop
// over a line loads chunk k + 1 before it computes chunk k, so that the memory latency hides behind the arithmetic.
struct ChunkIn { uint64_t oldS, oldG; int code; };
__device__ __forceinline__ ChunkIn chunk_load(const uint64_t* S, const uint64_t* G, int p0, int dir, int t0, int n, const uint8_t* seq, int lane)
{
	ChunkIn in;
	const int t = t0 + lane, p = p0 + dir * t;
	const bool valid = t < n;
	in.oldS = valid ? ld_l2(S + p) : 0; in.oldG = valid ? ld_l2(G + p) : 0;
	in.code = valid ? (int)seq[p - 1] : 250;
	return in;
}

template <bool ROW>
__device__ __forceinline__ SweepCell sweep_chunk(uint64_t* S, uint64_t* G, const ChunkIn& in, int fixcode, int p0, int dir, int t0, int n, int fix,
	uint64_t run0, uint64_t gapf0, SweepCarry& cy, int mine, unsigned long long mine_nz, int lane)
{
	SweepCell o;
	const int t = t0 + lane;
	o.valid = t < n;
	const int p = p0 + dir * t;
	o.pos = p;
	const int i = ROW ? fix : p, j = ROW ? p : fix;
	const uint64_t oldS = in.oldS, oldG = in.oldG;
	uint64_t corner = dpp_up64(oldS);
	if (lane == 0) corner = cy.corner;
	cy.corner = readlane64(oldS, 63);
	o.across = umax64(key_sub(oldG, SIM_R), key_sub(oldS, SIM_Q + SIM_R));
	const int64_t sc = (in.code == fixcode && fixcode < 4) ? SIM_MATCH : SIM_MISMATCH;
	bool tk = false;
	// pairs of this line that earlier rounds have aligned (lane e of `mine` = round e's partner of row / column `fix`, 0 = none)
	for (unsigned long long z = mine_nz; z; z &= z - 1) tk |= __builtin_amdgcn_readlane(mine, (int)__builtin_ctzll(z)) == (ROW ? j : i);
	const int64_t v = tk ? 0 : sim_score(corner) + sc;
	const uint64_t c0 = v <= 0 ? sim_key(0, (uint32_t)i, (uint32_t)j) : (uint64_t)((int64_t)corner + (sc << SIM_SHIFT));
	const uint64_t c1 = umax64(c0, o.across);
	// gap along the line
	const uint64_t inc = prefix_max64(o.valid ? c1 + ((uint64_t)(SIM_R * t) << SIM_SHIFT) : 0);
	const uint64_t exc = umax64(dpp_up64(inc), cy.pre);
	cy.pre = umax64(cy.pre, readlane64(inc, 63));
	uint64_t g = umax64(key_sub(gapf0, SIM_R * (int64_t)(t + 1)), key_sub(run0, SIM_Q + SIM_R * (int64_t)(t + 1)));
	if (exc) g = umax64(g, key_sub(exc, SIM_Q + SIM_R * (int64_t)t));
	o.g = g;
	o.c = umax64(c1, g);
	if (o.valid) { S[p] = o.c; G[p] = o.across; }
	return o;
}

// A unit's re-sweep is usually a few thousand cells, but now and then (a node that crosses the rectangle) it is the whole matrix:
// with the units of a slice advancing launch by launch, one such unit would hold up all the others.  So a launch spends at most
// `budget` 64-cell steps per unit and the kernel is resumable at line boundaries: an unfinished unit saves its loop state
// (SimSweepState) and carries on in the next launch, while the finished ones go through their next round on the host.
template <bool LDS>
__global__ void __launch_bounds__(LDS ? 64 : 256) k_sim_resweep(SimResweepArgs a, int32_t nunit)
{
	extern __shared__ uint64_t sim_lds[];
	// the launch covers the ACTIVE units only, through an index list: consecutive workgroups go to consecutive XCDs, and the heavy
	// units of a batch tend to be the same few of the 48 encodings (unit = segment x 48 + encoding), i.e. the same XCDs
	const int slot = LDS ? (int)blockIdx.x : (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
	if (slot >= nunit) return;
	const SimRoundReq rq = a.req[slot];
	const int unit = rq.unit;
	const int lane = threadIdx.x & 63;
	if (!rq.active) return;
	const int M = a.m;
	const int ustride = M + 2, cstride = (int)a.col_stride;
	const uint8_t* tcu = a.tcodes + (int64_t)unit * a.tstride;
	uint16_t* used = a.used + (int64_t)unit * SIM_K * ustride;          // [query row][round] -> target column (a line's 50 entries share a cache line or two:
	uint16_t* usedc = a.usedc + (int64_t)unit * SIM_K * cstride;        // [target column][round] -> query row   [round][line] cost 49 line fetches per sweep line)
	// the lines' states: in LDS when (columns + rows) fit (an LDS access is ~100 cycles, an L2 round trip ~2 us), with a home
	// in HBM that only an unfinished unit uses, between two launches
	uint64_t* gCS = a.colS + (int64_t)unit * a.col_stride; uint64_t* gCG = a.colG + (int64_t)unit * a.col_stride;
	uint64_t* gRS = a.rowS + (int64_t)unit * a.row_stride; uint64_t* gRG = a.rowG + (int64_t)unit * a.row_stride;
	uint64_t* CS = LDS ? sim_lds : gCS; uint64_t* CG = LDS ? sim_lds + a.col_stride : gCG;
	uint64_t* RS = LDS ? sim_lds + 2 * a.col_stride : gRS; uint64_t* RG = LDS ? sim_lds + 2 * a.col_stride + a.row_stride : gRG;
	const int mm = rq.mm, nn = rq.nn;
	// node list: lane k < cnt holds node k
	NodeList L;
	__shared__ uint32_t start_bits[LDS ? 1 : 4][SIM_BITS_WORDS];
	uint32_t* const bits = start_bits[LDS ? 0 : (threadIdx.x >> 6)];
	if (rq.active == 1) nodes_load(L, lane, a.nodes_in + (int64_t)slot * SIM_K, rq.node_count, bits);
	else nodes_load(L, lane, a.nodes + (int64_t)unit * SIM_K, a.node_count[unit], bits);
	int phase, i, m1, n1, rl, cl, floor_score, nround;
	bool grow_rows, grow_cols, positive;
	if (rq.active == 1) {
		// a new round: its aligned pairs join the rows' lists (slot = round number; at most one pair per row and round)
		const int round = a.used_cnt[unit];
		for (int k = lane; k < rq.pairs_count; k += 64) {
			const uint32_t pr = a.pairs[rq.pairs_first + k];
			used[(int64_t)(pr >> 16) * SIM_K + round] = (uint16_t)(pr & 0xffff);
			usedc[(int64_t)(pr & 0xffff) * SIM_K + round] = (uint16_t)(pr >> 16);
		}
		nround = round + 1;
		phase = 0; i = mm; m1 = rq.m1; n1 = rq.n1; rl = cl = 0; floor_score = rq.floor_score;
		grow_rows = grow_cols = true; positive = false;
		// backwards over the node's rectangle first (sim.h:884-931)
		for (int j = n1 + lane; j <= nn; j += 64) { CS[j] = sim_key(0, (uint32_t)(mm + 1), (uint32_t)j); CG[j] = sim_key(-SIM_Q, (uint32_t)(mm + 1), (uint32_t)j); }
		__builtin_amdgcn_s_waitcnt(0);
		__threadfence();
	} else {
		const SimSweepState st = a.state[unit];
		phase = st.phase; i = st.i; m1 = st.m1; n1 = st.n1; rl = st.rl; cl = st.cl; floor_score = st.floor_score; nround = st.nround;
		grow_rows = st.grow_rows != 0; grow_cols = st.grow_cols != 0; positive = st.positive != 0;
		if (LDS) {
			for (int j = max(n1, 0) + lane; j <= nn; j += 64) { CS[j] = gCS[j]; CG[j] = gCG[j]; }
			for (int r = max(m1, 0) + lane; r <= mm; r += 64) { RS[r] = gRS[r]; RG[r] = gRG[r]; }
			wave_publish();
		}
	}
	int budget = a.budget;
	// the time slice of a launch: checked every eighth line, so that a unit with cheap steps is not held to the pace of the one with
	// the most expensive steps (per launch every unit gets the same time, not the same number of steps)
	const long long t_start = (long long)wall_clock64();
	int lines = 0, steps = 0;
	auto slice_over = [&]() { if ((++lines & 7) == 0 && (long long)wall_clock64() - t_start > a.slice_ticks) budget = 0; };
	long long dbg_t[3] = { 0, 0, 0 }, dbg_n[3] = { 0, 0, 0 };      // a.debug: 100 MHz ticks / counts of backward steps, forward steps, events

	auto outside = [&](uint64_t k) { return key_i(k) > rl && key_j(k) > cl; };
	// one line; returns the last cell's (c, across, gapf); any_out / positive collected over the line
	struct LineEnd { uint64_t c, d, g; bool any_out; };
	auto row_mine = [&](int r) { return lane < nround ? (int)used[(int64_t)r * SIM_K + lane] : 0; };
	auto col_mine = [&](int j) { return lane < nround ? (int)usedc[(int64_t)j * SIM_K + lane] : 0; };
	auto row_line = [&](int r, int mine) {
		LineEnd e; e.any_out = false; e.c = e.d = e.g = 0;
		const int n = nn - n1 + 1;
		SweepCarry cy; cy.corner = sim_key(0, (uint32_t)(r + 1), (uint32_t)(nn + 1)); cy.pre = 0;
		const uint64_t run0 = sim_key(0, (uint32_t)r, (uint32_t)(nn + 1)), gapf0 = sim_key(-SIM_Q, (uint32_t)r, (uint32_t)(nn + 1));
		bool out = false, pos = false;
		const int qc = a.qcodes[r - 1];
		ChunkIn nx = chunk_load(CS, CG, nn, -1, 0, n, tcu, lane);
		const unsigned long long mine_nz = __ballot(mine != 0);
		for (int t0 = 0; t0 < n; t0 += 64) {
			const ChunkIn in = nx;
			if (t0 + 64 < n) nx = chunk_load(CS, CG, nn, -1, t0 + 64, n, tcu, lane);
			const SweepCell o = sweep_chunk<true>(CS, CG, in, qc, nn, -1, t0, n, r, run0, gapf0, cy, mine, mine_nz, lane);
			if (o.valid) { out |= outside(o.c) || outside(o.across) || outside(o.g); pos |= sim_score(o.c) > floor_score; }
			if (t0 + 64 >= n) { const int l = n - 1 - t0; e.c = readlane64(o.c, l); e.d = readlane64(o.across, l); e.g = readlane64(o.g, l); }
			budget--; steps++;
		}
		e.any_out = __ballot(out) != 0;
		if (__ballot(pos)) positive = true;
		if (lane == 0) { RS[r] = e.c; RG[r] = e.g; }
		wave_publish();
		slice_over();
		return e;
	};
	auto col_line = [&](int j, int mine) {
		LineEnd e; e.any_out = false; e.c = e.d = e.g = 0;
		const int n = mm - m1 + 1;
		SweepCarry cy; cy.corner = sim_key(0, (uint32_t)(mm + 1), (uint32_t)(j + 1)); cy.pre = 0;
		const uint64_t run0 = sim_key(0, (uint32_t)(mm + 1), (uint32_t)j), gapf0 = sim_key(-SIM_Q, (uint32_t)(mm + 1), (uint32_t)j);
		bool out = false, pos = false;
		const int tc = tcu[j - 1];
		ChunkIn nx = chunk_load(RS, RG, mm, -1, 0, n, a.qcodes, lane);
		const unsigned long long mine_nz = __ballot(mine != 0);
		for (int t0 = 0; t0 < n; t0 += 64) {
			const ChunkIn in = nx;
			if (t0 + 64 < n) nx = chunk_load(RS, RG, mm, -1, t0 + 64, n, a.qcodes, lane);
			const SweepCell o = sweep_chunk<false>(RS, RG, in, tc, mm, -1, t0, n, j, run0, gapf0, cy, mine, mine_nz, lane);
			if (o.valid) { out |= outside(o.c) || outside(o.across) || outside(o.g); pos |= sim_score(o.c) > floor_score; }
			if (t0 + 64 >= n) { const int l = n - 1 - t0; e.c = readlane64(o.c, l); e.d = readlane64(o.across, l); e.g = readlane64(o.g, l); }
			budget--; steps++;
		}
		e.any_out = __ballot(out) != 0;
		if (__ballot(pos)) positive = true;
		if (lane == 0) { CS[j] = e.c; CG[j] = e.g; }
		wave_publish();
		slice_over();
		return e;
	};

	long long tick = a.debug ? (long long)wall_clock64() : 0;
	// ---- phase 0: backwards over the node's rectangle
	if (phase == 0) {
		int mn = row_mine(i);                                   // (the next line's list is fetched while this line is swept)
		while (i >= m1 && budget > 0) { const int cur = mn; mn = row_mine(i - 1); row_line(i, cur); i--; }
		if (i < m1) { phase = 1; rl = m1; cl = n1; grow_rows = grow_cols = true; }
	}
	// ---- phase 1: growth (sim.h:933-1084): rl / cl = the smallest start row / column a state may have without leaving
	while (phase == 1 && budget > 0) {
		if ((grow_rows && m1 > 1) || (grow_cols && n1 > 1)) {
			const int mr = m1 > 1 ? row_mine(m1 - 1) : 0, mc = n1 > 1 ? col_mine(n1 - 1) : 0;
			if (grow_rows && m1 > 1) {
				m1--;
				const LineEnd e = row_line(m1, mr);
				grow_rows = e.any_out;
				if (!grow_cols && (outside(e.c) || outside(e.d) || outside(e.g))) grow_cols = true;
			}
			if (grow_cols && n1 > 1) {
				n1--;
				const LineEnd e = col_line(n1, mc);
				grow_cols = e.any_out;
				if (!grow_rows && (outside(e.c) || outside(e.d) || outside(e.g))) grow_rows = true;
			}
			continue;
		}
		bool grown = m1 == 1 && n1 == 1;
		if (!grown) {
			// no_cross (sim.h:150-165): the first node whose alignments may reach into the rectangle from outside
			const int n_stari = (L.start >> 16) & 0xffff, n_starj = L.start & 0xffff;
			const unsigned long long hit = __ballot(lane < L.cnt && n_stari <= mm && n_starj <= nn && L.bot >= m1 - 1 && L.right >= n1 - 1 && (n_stari < rl || n_starj < cl));
			if (!hit) grown = true;
			else {
				const int f = (int)__builtin_ctzll(hit);
				const int si = __builtin_amdgcn_readlane(n_stari, f), sj = __builtin_amdgcn_readlane(n_starj, f);
				if (si < rl) rl = si;
				if (sj < cl) cl = sj;
				grow_rows = grow_cols = true;
			}
		}
		if (grown) {
			m1--; n1--;
			if (positive) {
				phase = 2; i = m1 + 1;
				for (int j = n1 + 1 + lane; j <= nn; j += 64) { CS[j] = sim_key(0, (uint32_t)m1, (uint32_t)j); CG[j] = sim_key(-SIM_Q, (uint32_t)m1, (uint32_t)j); }
				wave_publish();
			} else phase = 3;
		}
	}
	if (a.debug) { const long long now = (long long)wall_clock64(); dbg_t[0] = now - tick; dbg_n[0] = steps; tick = now; }
	const int steps1 = steps;
	// ---- phase 2: forwards over the final rectangle, new nodes in row-major order (sim.h:1086-1141)
	if (phase == 2) {
		const int n = nn - n1;
		int mn = row_mine(min(i, M + 1));
		for (; i <= mm && budget > 0; i++) {
			SweepCarry cy; cy.corner = sim_key(0, (uint32_t)(i - 1), (uint32_t)n1); cy.pre = 0;
			const uint64_t run0 = sim_key(0, (uint32_t)i, (uint32_t)n1), gapf0 = sim_key(-SIM_Q, (uint32_t)i, (uint32_t)n1);
			const int mine = mn;
			mn = row_mine(i + 1);
			const unsigned long long mine_nz = __ballot(mine != 0);
			const int qc = a.qcodes[i - 1];
			ChunkIn nx = chunk_load(CS, CG, n1 + 1, 1, 0, n, tcu, lane);
			for (int t0 = 0; t0 < n; t0 += 64) {
				const ChunkIn in = nx;
				if (t0 + 64 < n) nx = chunk_load(CS, CG, n1 + 1, 1, t0 + 64, n, tcu, lane);
				const SweepCell o = sweep_chunk<true>(CS, CG, in, qc, n1 + 1, 1, t0, n, i, run0, gapf0, cy, mine, mine_nz, lane);
				budget--; steps++;
				unsigned long long ev = __ballot(o.valid && sim_score(o.c) > floor_score);
				if (ev) floor_score = 1;                         // min = addnode() (sim.h:1131): 1 from the first new cell on
				const long long te = a.debug ? (long long)wall_clock64() : 0;
				dbg_n[2] += __builtin_popcountll(ev);
				nodes_add_batch(L, lane, (ev >> lane) & 1, (int)sim_score(o.c), (int)(uint32_t)o.c, i, n1 + 1 + t0 + lane);
				if (a.debug) dbg_t[2] += (long long)wall_clock64() - te;
			}
			wave_publish();
			slice_over();
		}
		if (i > mm) phase = 3;
	}
	if (a.debug && lane == 0) {
		dbg_t[1] = (long long)wall_clock64() - tick; dbg_n[1] = steps - steps1;
		atomicAdd((unsigned long long*)a.debug + 6, (unsigned long long)L.dbg_pass); atomicAdd((unsigned long long*)a.debug + 7, (unsigned long long)L.dbg_bad);
		for (int k = 0; k < 3; k++) { atomicAdd((unsigned long long*)a.debug + 2 * k, (unsigned long long)dbg_n[k]); atomicAdd((unsigned long long*)a.debug + 2 * k + 1, (unsigned long long)dbg_t[k]); }
	}
	nodes_store(L, lane, phase != 3 ? a.nodes + (int64_t)unit * SIM_K : a.nodes_out + (int64_t)slot * SIM_K);
	if (lane == 0) {
		SimRoundOut ro; ro.node_count = L.cnt; ro.floor_score = floor_score; ro.pending = phase != 3; ro.pad = 0;
		a.out[slot] = ro;
		a.node_count[unit] = L.cnt; a.used_cnt[unit] = nround;
		if (phase != 3) {
			SimSweepState st;
			st.phase = phase; st.i = i; st.m1 = m1; st.n1 = n1; st.rl = rl; st.cl = cl; st.floor_score = floor_score; st.nround = nround;
			st.grow_rows = grow_rows; st.grow_cols = grow_cols; st.positive = positive;
			a.state[unit] = st;
		}
	}
	if (LDS && phase != 3) {
		for (int j = max(n1, 0) + lane; j <= nn; j += 64) { gCS[j] = CS[j]; gCG[j] = CG[j]; }
		for (int r = max(m1, 0) + lane; r <= mm; r += 64) { gRS[r] = RS[r]; gRG[r] = RG[r]; }
	}
}

hipError_t launch_sim_forward(const SimFwdArgs& a, int32_t nunit, hipStream_t st)
{
	if (nunit <= 0) return hipSuccess;
	hipLaunchKernelGGL(k_sim_forward, dim3((unsigned)((nunit + 3) / 4)), dim3(256), 0, st, a, nunit);
	return hipGetLastError();
}

hipError_t launch_sim_resweep(const SimResweepArgs& a, int32_t nunit, bool few_units, hipStream_t st)
{
	if (nunit <= 0) return hipSuccess;
	// the lines' states of one unit: 16 B per target column and per query row; in LDS when that fits one CU's 160 KB (minus a margin)
	const size_t lds = (size_t)(2 * a.col_stride + 2 * a.row_stride) * sizeof(uint64_t);
	// LDS variant: one unit per CU at a time, each 3-5 x faster -- for the launches in which few units are left (the tail of the
	// heaviest units); with thousands of units active the 4-units-per-workgroup variant keeps every wave slot of the chip busy
	if (few_units && lds <= 150 * 1024) {
		static bool attr_set = false;
		if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_sim_resweep<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr_set = true; }
		hipLaunchKernelGGL(k_sim_resweep<true>, dim3((unsigned)nunit), dim3(64), lds, st, a, nunit);
	} else hipLaunchKernelGGL(k_sim_resweep<false>, dim3((unsigned)((nunit + 3) / 4)), dim3(256), 0, st, a, nunit);
	return hipGetLastError();
}

} // namespace fasim
