// fasim-longtarget_amd/csrc/sim.hip -- row f3, first step: the forward sweep of classic SIM (the reference's -F path) on gfx950.
//
// What it replaces: the first double loop of SIM() (sim.h:506-571): affine-gap local alignment scores of the whole
// (lncRNA x target) matrix where every cell also carries the START POINT of its best alignment, with the reference's
// tie-break ORDER (sim.h:481-493): larger score, then larger start row, then larger start column.  Cells whose score
// exceeds the threshold go to the K = 50 node list (addnode, sim.h:99-148), which is order dependent; the kernel
// therefore only EMITS those cells and the host replays the list (engine.cpp, sim_replay_nodes).
//
// Layout: one wave64 per unit.  The query rows are cut into strips of 64 (lane = row); a strip is swept over the target
// columns as a wavefront (lane l works on column step - l), so the left neighbour is the lane's own previous cell and the
// upper / diagonal neighbours arrive from lane l-1 by a wave shift.  Lane 63 leaves the strip's bottom row (C and D per
// column) in a per-unit row buffer in HBM, which lane 0 of the next strip reads back 64 columns at a time.
//
// A DP state = one 64-bit key  (score + SIM_BIAS) << 26 | start_row << 13 | start_col,  so ORDER is an unsigned 64-bit max
// and "score - k" is a subtraction in the top field.  Scores are the reference's x10 values (match 50, mismatch -40, gap
// open 120, extension 40).  Limits of this first version: query and target at most 8191 long (13-bit start fields).
// Integer DP: no MFMA.  Plain 64-bit VALU arithmetic, not yet tuned (this is the first kernel of the row).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace fasim {

constexpr int64_t SIM_BIAS = 1 << 20;           // scores stay within (-2^20, 2^20): |gap run| <= 120 + 40 * 8191 < 2^19
constexpr int SIM_SHIFT = 26;
constexpr int64_t SIM_MATCH = 50, SIM_MISMATCH = -40, SIM_Q = 120, SIM_R = 40;

__device__ __forceinline__ uint64_t sim_key(int64_t s, uint32_t i, uint32_t j) { return ((uint64_t)(s + SIM_BIAS) << SIM_SHIFT) | ((uint64_t)i << 13) | (uint64_t)j; }
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

__global__ void __launch_bounds__(64) k_sim_forward(SimFwdArgs a)
{
	const int unit = blockIdx.x;
	const int lane = threadIdx.x;
	const int N = a.unit_len[unit];
	const int M = a.m;
	const uint8_t* tc_unit = a.tcodes + (int64_t)unit * a.tstride;
	uint64_t* rowC = a.rowbuf + (int64_t)unit * 2 * a.row_stride;      // [0 .. N]: C of the finished strip's bottom row
	uint64_t* rowD = rowC + a.row_stride;
	SimEvent* ev = a.events + (int64_t)unit * a.event_cap;
	uint32_t* count = a.event_count + unit;
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
			// cells above the threshold: compacted per step, in (step, lane) order
			const bool hit = valid && sim_score(c) > a.min_score[unit];
			const unsigned long long b = __ballot(hit);
			if (b) {
				uint32_t base = 0;
				if (lane == 0) base = atomicAdd(count, (uint32_t)__popcll(b));
				base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
				if (hit) {
					const uint32_t slot = base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
					if (slot < a.event_cap) { SimEvent e; e.i = (uint32_t)i; e.j = (uint32_t)j; e.key = c; ev[slot] = e; }
				}
			}
		}
		// the next strip reads the row buffer this strip has just written (same wave: program order is enough once the
		// stores have left the wave)
		__builtin_amdgcn_s_waitcnt(0);
		__threadfence_block();
	}
}

hipError_t launch_sim_forward(const SimFwdArgs& a, int32_t nunit, hipStream_t st)
{
	if (nunit <= 0) return hipSuccess;
	hipLaunchKernelGGL(k_sim_forward, dim3((unsigned)nunit), dim3(64), 0, st, a);
	return hipGetLastError();
}

} // namespace fasim
