// fasim-longtarget_amd/csrc/sim.hip -- row f3, first step: the forward sweep of classic SIM (the reference's -F path) on gfx950.
//
// What it replaces: the first double loop of SIM() (sim.h:506-571): affine-gap local alignment scores of the whole
// (lncRNA x target) matrix where every cell also carries the START POINT of its best alignment, with the reference's
// tie-break ORDER (sim.h:481-493): larger score, then larger start row, then larger start column.  Cells whose score
// exceeds the threshold go to the K = 50 node list (addnode, sim.h:99-148), which is order dependent (row-major).
//
// Layout: one wave64 per unit.  The query rows are cut into strips of 64 (lane = row); a strip is swept over the target
// columns as a wavefront (lane l works on column step - l), so the left neighbour is the lane's own previous cell and the
// upper / diagonal neighbours arrive from lane l-1 by a wave shift.  Lane 63 leaves the strip's bottom row (C and D per
// column) in a per-unit row buffer in HBM, which lane 0 of the next strip reads back 64 columns at a time.
//
// The node list lives in the wave too: lane k holds node k (score, start, end, bounding box).  A lane appends the cells of
// ITS row that pass the threshold to the row's own segment of a per-unit scratch buffer (columns ascend with the steps), and
// after each strip the wave replays the 64 segments in row order through addnode: the search for a node with the same start
// point is one compare + ballot, the eviction of the first lowest-score node a wave minimum over (score, lane).  That is
// the row-major order of the reference without a sort, and nothing but the 50 nodes ever leaves the GPU.
//
// A DP state = one 64-bit key  (score + SIM_BIAS) << 32 | start_row << 16 | start_col,  so ORDER is an unsigned 64-bit max
// and "score - k" is a subtraction in the top field.  Scores are the reference's x10 values (match 50, mismatch -40, gap
// open 120, extension 40).  Limits: query and target at most 65535 long (16-bit start fields; 8191 until round 3).
// Four units per 256-thread workgroup (one wave each: the waves share nothing).
// Integer DP: no MFMA.  Plain 64-bit VALU arithmetic, not yet tuned.
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
__device__ __forceinline__ uint64_t umax64(uint64_t a, uint64_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint64_t shfl_up64(uint64_t v)
{
	const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, 1, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), 1, 64);
	return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l)
{
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
	return ((uint64_t)hi << 32) | lo;
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
	// node list: lane k < nn holds node k
	int nn = 0;
	int n_score = 0, n_stari = 0, n_starj = 0, n_endi = 0, n_endj = 0, n_top = 0, n_bot = 0, n_left = 0, n_right = 0;
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
			uint64_t upC = shfl_up64(myC), upD = shfl_up64(myD);
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
				for (int x = 0; x < nb; x++) {
					const int j = __builtin_amdgcn_readlane((int)ej, x);
					const uint64_t key = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)khi, x) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)klo, x);
					const int c = (int)sim_score(key), ci = (int)((key >> SIM_FIELD) & 0xffff), cj = (int)(key & 0xffff);
					const unsigned long long hit = __ballot(lane < nn && n_stari == ci && n_starj == cj);
					int target;
					bool fresh;
					if (hit) { target = (int)__builtin_ctzll(hit); fresh = false; }
					else if (nn < SIM_K) { target = nn; nn++; fresh = true; }
					else {
						// the first node of lowest score gives way (sim.h:129-136)
						long long v = lane < SIM_K ? (((long long)n_score << 6) | lane) : 0x7fffffffffffffffll;
						for (int o = 32; o; o >>= 1) { const long long w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
						target = (int)(v & 63); fresh = true;
					}
					if (lane == target) {
						if (fresh) { n_score = c; n_stari = ci; n_starj = cj; n_endi = ei; n_endj = j; n_top = n_bot = ei; n_left = n_right = j; }
						else {
							if (n_score < c) { n_score = c; n_endi = ei; n_endj = j; }
							n_top = min(n_top, ei); n_bot = max(n_bot, ei); n_left = min(n_left, j); n_right = max(n_right, j);
						}
					}
				}
			}
		}
		// the next strip reads the row buffer this strip has just written (same wave: program order is enough once the
		// stores have left the wave)
		__builtin_amdgcn_s_waitcnt(0);
		__threadfence_block();
	}
	if (lane < nn) {
		SimNodeDev o;
		o.score = n_score; o.stari = n_stari; o.starj = n_starj; o.endi = n_endi; o.endj = n_endj; o.top = n_top; o.bot = n_bot; o.left = n_left; o.right = n_right;
		a.nodes[(int64_t)unit * SIM_K + lane] = o;
	}
	if (lane == 0) a.node_count[unit] = nn;
}

hipError_t launch_sim_forward(const SimFwdArgs& a, int32_t nunit, hipStream_t st)
{
	if (nunit <= 0) return hipSuccess;
	hipLaunchKernelGGL(k_sim_forward, dim3((unsigned)((nunit + 3) / 4)), dim3(256), 0, st, a, nunit);
	return hipGetLastError();
}

} // namespace fasim
