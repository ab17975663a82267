// fasim-longtarget_amd/csrc/band.hip -- banded forward pass of stage 3 for gfx950 (round 3).
//
// The reference runs every window try of fastSIM over the WHOLE query (sswNew.cpp:476-672 via fastsim.h:202-272): m x L
// cells for an alignment that occupies a few dozen rows.  k_scan (scan.hip) has already seen every cell of the unit: it
// leaves, per virtual lane (~22 query rows) and per block of 64 pipeline steps, the maximum H of the segment DP.  The
// window DP starts from zeros at the window's first column, so its H is cellwise <= the segment DP's H: those block maxima
// are UPPER BOUNDS ub[v] for the rows of lane v in the window.
//
//   k_band_select   one wave per try.  From ub[] it picks a row band [r0, r0 + 48 G) (G = 8, 16 or 32 lanes of the band
//                   kernel) around the lanes that can reach the try's target score, and the smallest score theta_min that
//                   the band PROVES: every cell of the window with H >= theta_min lies in a lane that is entirely inside
//                   the band (all other lanes have ub < theta_min) and every path that ends there with such a score fits
//                   below the band's first row (it spans at most L + (5L - theta - 12) / 4 rows: L diagonal steps and
//                   the vertical gap residues the score can pay for).  It also writes the try's column stream: one
//                   16-bit word per column = LDS address of the column's profile rows for the band | void | last.
//   k_align_band<G> the systolic pipeline of align.hip cut into 64 / G sub-pipelines per wave.  Each sub-pipeline sweeps
//                   its own windows over its own 48 G rows; uniform 24 rows per virtual lane; the per-column bookkeeping of
//                   the 64 / G pipe ends runs as vector code.  A result with score >= theta_min is the reference's
//                   (score, ref_end, read_end) bit for bit ("Banded stage 3", has the argument);
//                   anything else is reported unproven and goes to the full-height kernel.
// Windows in which any bound reaches 148 can meet the reference's Q2 / overflow behaviour and are not banded.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "kernels.h"

namespace fasim {

typedef short b2s __attribute__((ext_vector_type(2)));
typedef unsigned short b2u __attribute__((ext_vector_type(2)));
typedef int b4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ b2s bs(b2u x) { return __builtin_bit_cast(b2s, x); }
__device__ __forceinline__ b2u bu(b2s x) { return __builtin_bit_cast(b2u, x); }
__device__ __forceinline__ int bi(b2s x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ int bi(b2u x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ b2u bu_from(int x) { return __builtin_bit_cast(b2u, x); }
__device__ __forceinline__ b2s bs_from(int x) { return __builtin_bit_cast(b2s, x); }

constexpr int BAND_RPB = 24;                 // rows per virtual lane (uniform; two virtual lanes = 48 rows per lane)
constexpr int BAND_LANE_STRIDE = 112;        // bytes per lane in the LDS profile: 2 x 24 x int16 + 16 (bank-conflict-free b128)
constexpr int BAND_SC = 32;                  // value scale: the low 5 bits of every H carry (31 - row in lane)
constexpr int BAND_NEG = -32768;
// windows in flight per sub-pipeline: a window is at least 4 stream columns and the pipe is 2 G columns deep -> at most G / 2 + 2;
// the descriptor FIFO of a sub-pipeline holds G entries
constexpr int BAND_VOID_BIT = 0x4000, BAND_LAST_BIT = 0x8000;
constexpr int BAND_THREADS = 512;

// rows owned by k_scan's virtual lane v (scan.hip lane_rows)
__device__ __forceinline__ void scan_lane_rows(int v, int seg_len, int vs, int* row0, int* rows)
{
	const int s = v / vs, j = v - s * vs, q = seg_len / vs, rem = seg_len - q * vs;
	*rows = q + (j < rem ? 1 : 0);
	*row0 = s * seg_len + j * q + (j < rem ? j : rem);
}

// ------------------------------------------------------------------------------------------------
// k_band_decide / k_band_emit
// ------------------------------------------------------------------------------------------------
struct BandSelArgs {
	const FwdProb* probs; const int32_t* target; const int32_t* idx;   // idx: NULL = tries 0..n-1, else the tries to look at
	int32_t n, tstride;
	const uint16_t* ublk; int32_t ublk_blocks, ntiles;
	const uint16_t* prev_ub; const int32_t* prev;      // bounds from an earlier full-height pass of the candidate (see kernels.h)
	int32_t m, seg16, vs, nl;            // nl = lanes of the band profile = ceil(16 * seg16 / 48)
	int32_t zs;                          // zone stride in profile lanes: a band that starts in [z * zs, (z + 1) * zs) belongs to zone z
	const uint8_t* tcodes;
	BandTry* list[3]; uint16_t* slots[3]; uint32_t list_cap;
	int4* dec;                           // [n]: {class or -1, first lane of the band, theta_min, slot}
	uint32_t* counts;                    // BAND_COUNTS entries, see kernels.h
	uint32_t* cursors;                   // [3][BAND_MAX_ZONES]: next free slot of every (class, zone) segment of the lists
	FwdOut* out;
	int32_t class_mask;                  // bit c: class c (G = 8 << c) may be used
	int32_t debug;
};

constexpr int SEL_MAXV = 2048;           // 16 tiles x 128 virtual lanes
constexpr int SEL_TRIES = 64;            // tries per workgroup (16 per wave)

// adds 1 (and `extra`) per lane to counters[key] (key < 0: none) with ONE atomic per distinct key of the wave; returns the lane's
// rank among the lanes of its key plus the counter's old value (= its slot when the counter is a cursor)
__device__ __forceinline__ uint32_t wave_key_add(uint32_t* counters, int key, int lane)
{
	uint32_t res = 0;
	unsigned long long todo = __builtin_amdgcn_ballot_w64(key >= 0);
	while (todo) {
		const int leader = __ffsll((long long)todo) - 1;
		const int k = __shfl(key, leader, 64);
		const unsigned long long mk = __builtin_amdgcn_ballot_w64(key == k);
		uint32_t base = 0;
		if (lane == leader) base = atomicAdd(counters + k, (uint32_t)__popcll(mk));
		base = (uint32_t)__shfl((int)base, leader, 64);
		if (key == k) res = base + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
		todo &= ~mk;
	}
	return res;
}

// Dynamic LDS: per wave ub[nv] and row0[nv + 1] (u16).
__global__ void __launch_bounds__(256) k_band_decide(BandSelArgs a)
{
	extern __shared__ __align__(16) uint16_t sel_lds[];
	__shared__ int d_cls[SEL_TRIES], d_cols[SEL_TRIES];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int nv = 128 * a.ntiles;
	uint16_t* ub = sel_lds + (size_t)wv * (2 * nv + 2);
	uint16_t* r0s = ub + nv;
	// rows of k_scan's virtual lanes (the same for every try)
	for (int v = lane; v < nv; v += 64) { int row0, rows; scan_lane_rows(v, a.seg16, a.vs, &row0, &rows); r0s[v] = (uint16_t)row0; }
	if (lane == 0) r0s[nv] = (uint16_t)(16 * a.seg16);
	const int w0 = blockIdx.x * SEL_TRIES;
	for (int k = 0; k < SEL_TRIES / 4; k++) {
		const int slot_k = wv * (SEL_TRIES / 4) + k, w = w0 + slot_k;
		if (w >= a.n) { if (lane == 0) { d_cls[slot_k] = -1; d_cols[slot_k] = 0; } continue; }
		const int p = a.idx ? a.idx[w] : w;
		const FwdProb pb = a.probs[p];
		const int L = pb.len, unit = pb.stream_off;              // (the host passes the unit in the field the band path does not use)
		const int t0 = (int)(pb.tbase - (int64_t)unit * a.tstride), pe = t0 + L - 1;
		__builtin_amdgcn_wave_barrier();
		// ---- upper bounds per virtual lane of k_scan: the blocks of steps [t0 + v, pe + v] (v = lane index inside the tile)
		int umax = 0;
		const int pv = a.prev ? a.prev[w] : -1;
		// start-based bounds of the reverse pass: lane maxima of the REVERSED problem (lane v' holds reversed rows r0s[v'] ..)
		const uint16_t* PU = pv >= 0 ? a.prev_ub + (size_t)pv * nv : nullptr;
		if (PU) {
			for (int i = 0; i < a.ntiles; i++) {
				const uint32_t y = *reinterpret_cast<const uint32_t*>(PU + 128 * i + 2 * lane);
				const int p0 = (int)(y & 0xffffu), p1 = (int)(y >> 16);
				ub[128 * i + 2 * lane] = (uint16_t)p0; ub[128 * i + 2 * lane + 1] = (uint16_t)p1;
				umax = p0 > umax ? p0 : umax; umax = p1 > umax ? p1 : umax;
			}
		} else
		for (int i = 0; i < a.ntiles; i++) {
			const uint16_t* U = a.ublk + (((size_t)unit * a.ntiles + i) * a.ublk_blocks) * 64 + lane;
			const int b0 = (t0 + 2 * lane) / SCAN_UBLK_STEPS;
			uint32_t x[5];
#pragma unroll
			for (int j = 0; j < 5; j++) { const int b = b0 + j < a.ublk_blocks ? b0 + j : a.ublk_blocks - 1; x[j] = U[(size_t)b * 64]; }
			int u0 = 0, u1 = 0;
			const int e0 = (pe + 2 * lane) / SCAN_UBLK_STEPS, s1 = (t0 + 2 * lane + 1) / SCAN_UBLK_STEPS, e1 = (pe + 2 * lane + 1) / SCAN_UBLK_STEPS;
#pragma unroll
			for (int j = 0; j < 5; j++) {
				const int b = b0 + j;
				const int lo = (int)(x[j] & 0xffu), hi = (int)(x[j] >> 8);                   // one byte per virtual lane
				if (b <= e0) u0 = lo > u0 ? lo : u0;
				if (b >= s1 && b <= e1) u1 = hi > u1 ? hi : u1;
			}
			if (u0 > 5 * L) u0 = 5 * L;
			if (u1 > 5 * L) u1 = 5 * L;
			ub[128 * i + 2 * lane] = (uint16_t)u0; ub[128 * i + 2 * lane + 1] = (uint16_t)u1;
			umax = u0 > umax ? u0 : umax; umax = u1 > umax ? u1 : umax;
		}
		for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(umax, o, 64); umax = y > umax ? y : umax; }
		__builtin_amdgcn_wave_barrier();
		int chosen = -1, ch_q0 = 0, ch_theta = 0;
		int T = a.target[w];
		if (T > umax) T = umax;
		// a bound of 148 or more: the reference's signed lazy-F exit (Q2) or its 8-bit overflow could show -> full-height kernel
		if (PU && umax < 148 && umax >= 1) {
			// Start-based bounds: ub[v'] = the best score of an alignment of THIS window that starts in reversed lane v', so umax is
			// the window's exact score.  Every cell with H >= theta ends an alignment that starts in a lane with ub >= theta and
			// reaches at most rmax(theta) - 1 rows further down: the band has to hold those lanes and that many rows below them.
			T = umax;
			const int mpad = 16 * a.seg16;
			int tf = 1 << 30, bf = -1;
			for (int v = lane; v < nv; v += 64) if ((int)ub[v] >= T) {
				const int lo = r0s[v], hi = r0s[v + 1];
				const int ot = a.m - hi > 0 ? a.m - hi : 0, ob = a.m - 1 - lo > 0 ? a.m - 1 - lo : 0;      // original rows of the lane
				tf = ot < tf ? ot : tf; bf = ob > bf ? ob : bf;
			}
			for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(tf, o, 64), z = __shfl_xor(bf, o, 64); tf = y < tf ? y : tf; bf = z > bf ? z : bf; }
			const int gaps = 5 * L - T - 12;
			const int rmax = gaps >= 4 ? L + gaps / 4 : L;
			for (int c = 0; c < 3 && chosen < 0; c++) {
				const int G = 8 << c;
				if (!((a.class_mask >> c) & 1) || G > a.nl) continue;
				int q0 = tf / 48; if (q0 > a.nl - G) q0 = a.nl - G;
				const int r0 = 48 * q0, r1 = r0 + 48 * G;
				if (r1 < mpad && r1 < bf + rmax) continue;                 // does not hold the rows below the last start lane
				int theta = 1;
				for (int v = lane; v < nv; v += 64) {
					const int lo = r0s[v], hi = r0s[v + 1], u = ub[v];
					const int ot = a.m - hi > 0 ? a.m - hi : 0, ob = a.m - 1 - lo > 0 ? a.m - 1 - lo : 0;
					int tau;
					if (ot < r0) tau = u + 1;                             // starts above the band: must not reach theta
					else if (r1 >= mpad) tau = 1;
					else {
						const int D = r1 - ob;                             // rows from the lane's last row to the end of the band
						const int ts = D >= L ? 9 * L - 4 * D - 15 : 5 * L + 1;
						tau = u + 1 < ts ? u + 1 : ts;
					}
					theta = tau > theta ? tau : theta;
				}
				for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(theta, o, 64); theta = y > theta ? y : theta; }
				if (theta <= T) { chosen = c; ch_q0 = q0; ch_theta = theta; }
			}
		} else if (!PU && umax < 148 && T >= 1) {
			int vf = 1 << 30, vl = -1;
			for (int v = lane; v < nv; v += 64) if ((int)ub[v] >= T) { vf = v < vf ? v : vf; vl = v > vl ? v : vl; }
			for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(vf, o, 64), z = __shfl_xor(vl, o, 64); vf = y < vf ? y : vf; vl = z > vl ? z : vl; }
			// rows a path of score >= T can span inside L columns
			const int gaps = 5 * L - T - 12;
			const int rmax = gaps >= 4 ? L + gaps / 4 : L;
			int top = (int)r0s[vf] - (rmax - 1); if (top < 0) top = 0;
			const int bot = (int)r0s[vl + 1];
			for (int c = 0; c < 3 && chosen < 0; c++) {
				const int G = 8 << c;
				if (!((a.class_mask >> c) & 1) || G > a.nl) continue;
				int q0 = top / 48; if (q0 > a.nl - G) q0 = a.nl - G;
				const int r0 = 48 * q0, r1 = r0 + 48 * G;
				if (r1 < bot) continue;                                   // does not reach the last lane that matters
				int theta = 1;
				for (int v = lane; v < nv; v += 64) {
					const int lo = r0s[v], hi = r0s[v + 1], u = ub[v];
					int tau;
					if (lo < r0 || hi > r1) tau = u + 1;                 // not entirely inside: must not reach theta
					else if (r0 == 0) tau = 1;
					else {
						const int D = lo - r0 + 1;                         // rows available to a path that ends in this lane
						const int ts = D >= L ? 9 * L - 4 * D - 15 : 5 * L + 1;
						tau = u + 1 < ts ? u + 1 : ts;
					}
					theta = tau > theta ? tau : theta;
				}
				for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(theta, o, 64); theta = y > theta ? y : theta; }
				if (theta <= T) { chosen = c; ch_q0 = q0; ch_theta = theta; }
			}
		}
		if (lane == 0) {
			d_cls[slot_k] = chosen < 0 ? -1 : chosen * BAND_MAX_ZONES + ch_q0 / a.zs;
			d_cols[slot_k] = chosen < 0 ? (umax >= 148 ? -1 : -2) : 4 * ((L + 2 + 3) / 4);
			a.dec[w] = make_int4(chosen, ch_q0, ch_theta, 0);
		}
	}
	__syncthreads();
	// ---- counters: tries per (class, zone), stream columns per class, and (debug) the tries left unbanded by reason
	if (wv == 0) {
		const int key = d_cls[lane], cols = d_cols[lane];
		(void)wave_key_add(a.counts, key, lane);
		for (int c = 0; c < 3; c++) {
			int x = (key >= 0 && key / BAND_MAX_ZONES == c) ? cols : 0;
			for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
			if (lane == 0 && x) atomicAdd(a.counts + BAND_COUNT_COLS + c, (uint32_t)x);
		}
		if (a.debug) (void)wave_key_add(a.counts, key < 0 && w0 + lane < a.n ? (cols == -1 ? BAND_COUNT_HOT : BAND_COUNT_NOBAND) : -1, lane);
	}
}

// second half: slots inside the (class, zone) segments of the lists, list entries and column streams
__global__ void __launch_bounds__(256) k_band_emit(BandSelArgs a)
{
	__shared__ int d_slot[SEL_TRIES], d_key[SEL_TRIES];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int w0 = blockIdx.x * SEL_TRIES;
	if (wv == 0) {
		const int w = w0 + lane;
		int key = -1;
		if (w < a.n) { const int4 d = a.dec[w]; if (d.x >= 0) key = d.x * BAND_MAX_ZONES + d.y / a.zs; }
		d_key[lane] = key;
		d_slot[lane] = (int)wave_key_add(a.cursors, key, lane);
	}
	__syncthreads();
	for (int k = 0; k < SEL_TRIES / 4; k++) {
		const int slot_k = wv * (SEL_TRIES / 4) + k, w = w0 + slot_k;
		if (w >= a.n) continue;
		const int p = a.idx ? a.idx[w] : w;
		const int key = d_key[slot_k];
		if (key < 0 || (uint32_t)d_slot[slot_k] >= a.list_cap) { if (lane == 0) a.out[p].flags = 16; continue; }
		const int4 d = a.dec[w];
		const int cls = d.x, q0 = d.y;
		const FwdProb pb = a.probs[p];
		const int L = pb.len, nq = (L + 2 + 3) / 4;
		const uint32_t slot = (uint32_t)d_slot[slot_k];
		if (lane == 0) {
			BandTry bt; bt.prob = p; bt.r0 = 48 * q0; bt.theta_min = d.z; bt.nq = nq;
			a.list[cls][slot] = bt;
			a.out[p].flags = 8;
		}
		// LDS lanes per code of this class's launches, band start relative to the zone's first lane
		const int G = 8 << cls;
		const int lc = a.zs + G < a.nl ? a.zs + G : a.nl;
		const int qrel = q0 - (q0 / a.zs) * a.zs;
		const int voidw = BAND_VOID_BIT | (5 * lc * 7);
		// two columns per lane and store: 64 lanes cover 128 columns per round
		uint32_t* s2 = reinterpret_cast<uint32_t*>(a.slots[cls] + (size_t)slot * BAND_SLOT_COLS);
		const int lead = 4 * nq - L;
		for (int kk = lane; kk < 2 * nq; kk += 64) {
			uint32_t word = 0;
#pragma unroll
			for (int h = 0; h < 2; h++) {
				const int pos = 2 * kk + h;
				int x = voidw;
				if (pos >= lead) {
					const int col = pos - lead;
					int code = a.tcodes[pb.tbase + col]; if (code > 4) code = 4;
					x = (code * lc + qrel) * 7;
					if (col == L - 1) x |= BAND_LAST_BIT;
				}
				word |= (uint32_t)x << (16 * h);
			}
			s2[kk] = word;
		}
	}
}

static void band_sel_args(const BandSelLaunch& L, BandSelArgs& a)
{
	a.probs = L.probs; a.target = L.target; a.idx = L.idx; a.n = L.n; a.tstride = L.tstride; a.ublk = L.ublk; a.ublk_blocks = L.ublk_blocks;
	a.m = L.m; a.seg16 = (L.m + 15) / 16; a.vs = systolic_vs(L.m); a.ntiles = a.vs / 8; a.nl = band_profile_lanes(L.m); a.zs = band_zone_stride(L.m);
	a.prev_ub = L.prev_ub; a.prev = L.prev_ub ? L.prev : nullptr;
	a.tcodes = L.tcodes; a.list_cap = L.list_cap; a.dec = L.dec; a.counts = L.counts; a.cursors = L.cursors; a.out = L.out; a.class_mask = L.class_mask; a.debug = L.debug;
	for (int c = 0; c < 3; c++) { a.list[c] = L.list[c]; a.slots[c] = L.slots[c]; }
}

hipError_t launch_band_decide(const BandSelLaunch& L, hipStream_t st)
{
	if (L.n <= 0) return hipSuccess;
	BandSelArgs a; band_sel_args(L, a);
	if (128 * a.ntiles > SEL_MAXV) return hipErrorInvalidValue;
	hipError_t err = hipMemsetAsync(a.counts, 0, BAND_COUNTS * sizeof(uint32_t), st);
	if (err != hipSuccess) return err;
	const size_t lds = (size_t)4 * (2 * 128 * a.ntiles + 2) * sizeof(uint16_t);
	hipLaunchKernelGGL(k_band_decide, dim3((unsigned)((L.n + SEL_TRIES - 1) / SEL_TRIES)), dim3(256), lds, st, a);
	return hipGetLastError();
}

// L.cursors must hold the first slot of every (class, zone) segment (the host derives them from the counts of launch_band_decide)
hipError_t launch_band_emit(const BandSelLaunch& L, hipStream_t st)
{
	if (L.n <= 0) return hipSuccess;
	BandSelArgs a; band_sel_args(L, a);
	hipLaunchKernelGGL(k_band_emit, dim3((unsigned)((L.n + SEL_TRIES - 1) / SEL_TRIES)), dim3(256), 0, st, a);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// k_align_band
// ------------------------------------------------------------------------------------------------
struct BandArgs {
	const BandTry* list; const uint16_t* slots;
	const BandZoneTab* tab;              // per workgroup: its zone and its share of the zone's segment of the list
	const uint8_t* qcodes; int32_t m, nl;
	int32_t lc;                          // profile lanes per code held in LDS (the zone's lanes plus one band height)
	FwdOut* out;
};

template <int G>
__device__ __forceinline__ int gshift(int x, int inject, bool is_start)
{
	// out.lo = x.hi of lane - 1 (first lane of a sub-pipeline: `inject`'s hi half), out.hi = x.lo
	int up = __builtin_amdgcn_update_dpp(inject, x, G == 32 ? 0x138 /* wave_shr:1 */ : 0x111 /* row_shr:1 */, 0xf, 0xf, false);
	if (G != 16) up = is_start ? inject : up;
	return __builtin_amdgcn_alignbit(x, up, 16);
}
template <int G>
__device__ __forceinline__ int gshift0(int x, bool is_start)
{
	int up = __builtin_amdgcn_mov_dpp(x, G == 32 ? 0x138 : 0x111, 0xf, 0xf, true);
	if (G != 16) up = is_start ? 0 : up;
	return __builtin_amdgcn_alignbit(x, up, 16);
}
template <int G>
__device__ __forceinline__ uint32_t gprev(uint32_t x, bool is_start)
{
	int up = __builtin_amdgcn_mov_dpp((int)x, G == 32 ? 0x138 : 0x111, 0xf, 0xf, true);
	if (G != 16) up = is_start ? 0 : up;
	return (uint32_t)up;
}

__device__ __forceinline__ int band_cell_score(const BandArgs& a, int t, int row)
{
	if (row >= 16 * ((a.m + 15) / 16)) return BAND_NEG;      // below the padded query: dead rows (never the column maximum)
	if (row >= a.m) return 0;                                 // zero-score pad rows (Q3)
	const int q = a.qcodes[row];
	return ((q == t && t < 4) ? 5 : -4) * BAND_SC;
}

template <int G>
__global__ void __launch_bounds__(BAND_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) k_align_band(BandArgs a)
{
	constexpr int NG = 64 / G;                               // sub-pipelines per wave
	constexpr int SC = BAND_SC;
	extern __shared__ __align__(16) uint8_t lds[];
	uint8_t* prof = lds;                                     // [5 codes][lc lanes][112] | void block [G][112]
	const int code_stride = a.lc * BAND_LANE_STRIDE;
	const BandZoneTab zt = a.tab[blockIdx.x];
	uint4* fifo = reinterpret_cast<uint4*>(lds + 5 * code_stride + G * BAND_LANE_STRIDE);     // [wave][NG][G]
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	for (int idx = threadIdx.x; idx < 5 * a.lc * 48; idx += blockDim.x) {
		const int r = idx % 48, l = (idx / 48) % a.lc, t = idx / (48 * a.lc);
		*reinterpret_cast<int16_t*>(prof + t * code_stride + l * BAND_LANE_STRIDE + r * 2) = (int16_t)band_cell_score(a, t, 48 * (zt.zbase + l) + r);
	}
	for (int idx = threadIdx.x; idx < G * 56; idx += blockDim.x)
		*reinterpret_cast<int16_t*>(prof + 5 * code_stride + idx * 2) = (int16_t)BAND_NEG;
	__syncthreads();

	const int j = lane & (G - 1), grp = lane / G;
	const bool is_start = j == 0, is_end = j == G - 1;
	constexpr int FIFO = G;
	uint4* myfifo = fifo + ((size_t)wv * NG + grp) * FIFO;
	const uint8_t* pl = prof + j * BAND_LANE_STRIDE;
	// row keys: (column maximum << 16) | (0xFFFF - row relative to the band's first row)
	const int kbase_lo = 0xFFFF - (2 * j) * BAND_RPB - 31, kbase_hi = 0xFFFF - (2 * j + 1) * BAND_RPB - 31;
	const uint32_t voidw = (uint32_t)(BAND_VOID_BIT | (5 * a.lc * 7));
	const uint2 void4 = make_uint2(voidw * 0x10001u, voidw * 0x10001u);

	// ---- feeder (meaningful in the first lane of each sub-pipeline): the windows of the zone's list segment are dealt out round
	//      robin over the sub-pipelines of the workgroups that serve the zone
	const int gstride = zt.nwg * (BAND_THREADS / 64) * NG, f_end = zt.first + zt.count;
	int f_idx = zt.first + (zt.wg * (BAND_THREADS / 64) + wv) * NG + grp;      // next window of this sub-pipeline
	const uint2* f_slot = nullptr; int f_q = 0, f_nq = 0;                      // current window: stream, position, length (groups of 4 columns)
	int n_start = 0;
	uint2 feed_next = void4;
	BandTry nbt; nbt.prob = -1; nbt.r0 = 0; nbt.theta_min = 0; nbt.nq = 0;
	if (is_start && f_idx < f_end) nbt = a.list[f_idx];

	int H[BAND_RPB], E[BAND_RPB];
#pragma unroll
	for (int r = 0; r < BAND_RPB; r++) { H[r] = 0; E[r] = 0; }
	int tc = (int)(voidw * 0x10001u);
	int hbot = 0, fbot = 0, recv_h_last = 0;
	uint32_t klo = 0, khi = 0;
	int runmax = 0, end_ref = -1, end_key = 0xFFFF, cidx = 0, n_end = 0;
	int drain = -1;                                          // iterations left once every sub-pipeline of the wave has run dry

	for (;;) {
		const uint2 feed = feed_next;                        // the four columns of this iteration (loaded one iteration ago)
		if (is_start) {
			if (f_q >= f_nq) {
				// the current window is fully issued: start the next one of this sub-pipeline (its descriptor was fetched when
				// the current one started)
				if (f_idx < f_end) {
					f_slot = reinterpret_cast<const uint2*>(a.slots + (size_t)f_idx * BAND_SLOT_COLS);
					f_nq = nbt.nq; f_q = 0;
					myfifo[n_start & (FIFO - 1)] = make_uint4((uint32_t)nbt.prob, (uint32_t)nbt.theta_min, (uint32_t)nbt.r0, 0u);
					n_start++;
					f_idx += gstride;
					if (f_idx < f_end) nbt = a.list[f_idx];
				} else { f_slot = nullptr; f_nq = 0; f_q = 0; }
			}
			if (f_slot) { feed_next = f_slot[f_q]; f_q++; } else feed_next = void4;
		}
		// Exit condition every wave reaches: once no sub-pipeline of the wave has a window left (`feed` may still hold the last
		// columns), G/2 + 2 further iterations push the last column through the 2 G virtual lanes, then the wave leaves.
		if (drain < 0) { if (__builtin_amdgcn_ballot_w64(is_start && f_slot != nullptr) == 0ull) drain = G / 2 + 3; }
		else if (--drain == 0) break;

#pragma unroll
		for (int k = 0; k < 4; k++) {
			const int inj = (int)((k < 2 ? feed.x : feed.y) >> (16 * (k & 1))) << 16;
			tc = gshift<G>(tc, inj, is_start);
			const int recv_h = gshift0<G>(hbot, is_start), recv_f = gshift0<G>(fbot, is_start);
			const uint32_t kin_lo = gprev<G>(khi, is_start), kin_hi = klo;
			const b2u tcu = bu_from(tc);
			// void columns clear E and F (the saturating subtractions use 0xFFFF) and score -inf
			const b2u isvoid = bu(bs(tcu << (b2u){ 1, 1 }) >> (b2s){ 15, 15 });
			const b2u dec = isvoid | (b2u){ GAP_EXT * SC, GAP_EXT * SC };
			const b2u gapo = isvoid | (b2u){ GAP_OPEN * SC, GAP_OPEN * SC };
			const uint8_t* pa = pl + ((tc & 0x3fff) << 4);
			const uint8_t* pbp = pl + (((tc >> 16) & 0x3fff) << 4) + 48;
			b4i PA[3], PB[3];
#pragma unroll
			for (int g = 0; g < 3; g++) { PA[g] = *reinterpret_cast<const b4i*>(pa + 16 * g); PB[g] = *reinterpret_cast<const b4i*>(pbp + 16 * g); }
			auto score_of = [&](int r) -> int { const int g = r >> 3, kk = r & 7; return __builtin_amdgcn_perm(PB[g][kk >> 1], PA[g][kk >> 1], (kk & 1) ? 0x07060302 : 0x05040100); };
			auto diag_plus_score = [](int hold, int sc) -> b2s { asm("v_pk_add_i16 %0, %1, %0" : "+v"(sc) : "v"(hold)); return bs_from(sc); };
			const int hdiag0 = recv_h_last;
			recv_h_last = recv_h;
			b2u f = bu_from(recv_f);
			b2s lkx[4] = { (b2s){ 0, 0 }, (b2s){ 0, 0 }, (b2s){ 0, 0 }, (b2s){ 0, 0 } };
			b2s t = diag_plus_score(hdiag0, score_of(0));
#pragma unroll
			for (int r = 0; r < BAND_RPB; r++) {
				b2s tnext = t;
				if (r + 1 < BAND_RPB) tnext = diag_plus_score(H[r], score_of(r + 1));
				b2s h = __builtin_elementwise_max(t, bs_from(E[r]));
				{ int hn; asm("v_pk_max_i16 %0, %2, %3" : "=v"(hn) : "0"(H[r]), "v"(bi(h)), "v"(bi(f)), "v"(bi(tnext))); h = bs_from(hn); }
				H[r] = bi(h);
				const b2u ho = __builtin_elementwise_sub_sat(bu(h), gapo);
				E[r] = bi(__builtin_elementwise_max(__builtin_elementwise_sub_sat(bu_from(E[r]), dec), ho));
				f = __builtin_elementwise_max(__builtin_elementwise_sub_sat(f, dec), ho);
				const b2s key = h | (b2s){ (short)(31 - r), (short)(31 - r) };
				lkx[r & 3] = __builtin_elementwise_max(lkx[r & 3], key);
				t = tnext;
			}
			hbot = H[BAND_RPB - 1];
			fbot = bi(f);
			const b2s lkey = __builtin_elementwise_max(__builtin_elementwise_max(lkx[0], lkx[1]), __builtin_elementwise_max(lkx[2], lkx[3]));
			const uint32_t lk = (uint32_t)bi(lkey);
			const uint32_t loc_lo = (((lk & 0xFFFFu) >> 5) << 16) | (uint32_t)(kbase_lo + (int)(lk & 31u));
			const uint32_t loc_hi = ((lk >> 21) << 16) | (uint32_t)(kbase_hi + (int)((lk >> 16) & 31u));
			klo = kin_lo > loc_lo ? kin_lo : loc_lo;
			khi = kin_hi > loc_hi ? kin_hi : loc_hi;

			// ---- pipe end of every sub-pipeline: the hi half of its last lane has just finished one column
			const int th = tc >> 16;
			const bool real = !(th & BAND_VOID_BIT);
			const int colmax = (int)(khi >> 16);
			const bool better = real && colmax > runmax;
			runmax = better ? colmax : runmax;
			end_ref = better ? cidx : end_ref;
			end_key = better ? (int)(khi & 0xFFFFu) : end_key;
			cidx += real ? 1 : 0;
			const bool emit = is_end && (th & BAND_LAST_BIT) && real;
			if (__builtin_amdgcn_ballot_w64(emit) != 0ull) {
				if (emit) {
					const uint4 fe = myfifo[n_end & (FIFO - 1)];
					n_end++;
					FwdOut o;
					o.score = runmax; o.ref_end = end_ref;
					int rd = runmax > 0 ? (int)fe.z + (0xFFFF - end_key) : 0;
					o.read_end = rd < a.m - 1 ? rd : a.m - 1;
					o.flags = runmax >= (int)fe.y ? 0 : 8; o.ref_begin = 0; o.read_begin = 0;
					a.out[fe.x] = o;
					runmax = 0; end_ref = -1; end_key = 0xFFFF; cidx = 0;
				}
			}
		}
	}
}

int band_profile_lanes(int m) { return (16 * ((m + 15) / 16) + 47) / 48; }
// Zones: a query longer than 64 profile lanes (3 072 rows) does not stage its whole profile; a workgroup of the band kernel holds
// the lanes of ONE zone (zs lanes plus one band height) and works on the tries whose band starts in that zone.
int band_zone_stride(int m) { const int nl = band_profile_lanes(m); int zs = 64; while ((nl + zs - 1) / zs > BAND_MAX_ZONES) zs *= 2; return nl <= zs ? nl : zs; }
static int band_lc(int m, int G) { const int nl = band_profile_lanes(m), zs = band_zone_stride(m); return zs + G < nl ? zs + G : nl; }
size_t band_lds_bytes(int m, int G) { return (size_t)(5 * band_lc(m, G) + G) * BAND_LANE_STRIDE + (size_t)(BAND_THREADS / 64) * 64 * sizeof(uint4); }
// classes (bit c: G = 8 << c) the band kernel runs for a query of m rows: the profile lanes of a zone must fit the LDS budget and the
// band must be at most 3/8 of the query's height (a band of half the height costs more than it saves: measured on H19, G = 32)
int band_classes(int m)
{
	if (!systolic_fits(m) || systolic_tiles(m) > 16) return 0;
	int mask = 0;
	const int nl = band_profile_lanes(m);
	for (int c = 0; c < 3; c++) {
		const int G = 8 << c;
		if (8 * G <= 3 * nl && band_lds_bytes(m, G) <= 72 * 1024 && (5 * band_lc(m, G) + G) * 7 < BAND_VOID_BIT) mask |= 1 << c;
	}
	return mask;
}

// Workgroups of one class's launch: every zone with tries gets a share of about 512 workgroups (two per CU) in proportion to its
// tries, at least one and at most one per 3 windows of a sub-pipeline (a short list is spread thin: a launch lasts as long as its
// longest sub-pipeline).  Returns the table the kernel reads (one entry per workgroup).
std::vector<BandZoneTab> band_plan(int m, int cls, const uint32_t* zone_count, const uint32_t* zone_first)
{
	const int G = 8 << cls, gpb = (BAND_THREADS / 64) * (64 / G), zs = band_zone_stride(m);
	const int nz = (band_profile_lanes(m) + zs - 1) / zs;
	uint64_t total = 0;
	for (int z = 0; z < nz; z++) total += zone_count[z];
	std::vector<BandZoneTab> tab;
	if (!total) return tab;
	for (int z = 0; z < nz; z++) {
		const uint32_t n = zone_count[z];
		if (!n) continue;
		long nwg = (long)((512.0 * n) / (double)total + 0.5);
		const long cap = ((long)n + 3L * gpb - 1) / (3L * gpb);
		if (nwg > cap) nwg = cap;
		if (nwg < 1) nwg = 1;
		for (long w = 0; w < nwg; w++) { BandZoneTab t; t.zbase = z * zs; t.first = (int32_t)zone_first[z]; t.count = (int32_t)n; t.wg = (int32_t)w; t.nwg = (int32_t)nwg; tab.push_back(t); }
	}
	return tab;
}

template <int G>
static hipError_t launch_band_t(const BandArgs& a, int nwg, size_t lds, hipStream_t st)
{
	static bool attr_set = false;                   // (benign race: the call is idempotent)
	if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_align_band<G>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024); attr_set = true; }
	hipLaunchKernelGGL((k_align_band<G>), dim3((unsigned)nwg), dim3(BAND_THREADS), lds, st, a);
	return hipGetLastError();
}

hipError_t launch_align_band(const BandLaunch& L, hipStream_t st)
{
	if (L.nwg <= 0) return hipSuccess;
	BandArgs a;
	a.list = L.list; a.slots = L.slots; a.tab = L.tab; a.qcodes = L.qcodes; a.m = L.m; a.nl = band_profile_lanes(L.m); a.out = L.out;
	const int G = 8 << L.cls;
	a.lc = band_lc(L.m, G);
	const size_t lds = band_lds_bytes(L.m, G);
	switch (L.cls) {
	case 0: return launch_band_t<8>(a, L.nwg, lds, st);
	case 1: return launch_band_t<16>(a, L.nwg, lds, st);
	case 2: return launch_band_t<32>(a, L.nwg, lds, st);
	default: return hipErrorInvalidValue;
	}
}

} // namespace fasim
