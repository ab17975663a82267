// fasim-longtarget_amd/csrc/align.hip -- stage 3 (window alignments) for gfx950.
//
//   k_build_stream  concatenates the target codes of all windows of a round into one column stream
//                   [void, void, c0 .. cL-1(last)] per window
//   k_align_fwd     forward pass of ssw_align (sswNew.cpp:1470-1495) as the same systolic wave pipeline as
//                   scan.hip, fed by the window stream WITHOUT draining between windows: two "void" columns
//                   (score -inf, gap registers = 0xFFFF so that the saturating subtractions clear E and F)
//                   reset the DP state in flight.  DP values are kept scaled by 32 so that the low 5 bits of
//                   every H can carry (31 - row-in-lane): one v_pk_max then yields, per column, the maximum
//                   AND the smallest row that holds it (read_end, sswNew.cpp:621-629).  The TAINT variant (first,
//                   8-bit-semantics pass) scales by 64 and uses bit 5 as the Q2 taint bit of scan.hip: a window
//                   needs the stripe-faithful replay only if the cell that WINS (first column of the maximum,
//                   smallest row) is tainted -- every other cell only has to be "not larger", which the
//                   reference's smaller values cannot break.
//   k_finish        per alignment (one thread): the reverse pass (sswNew.cpp:1508-1520) and banded_sw
//                   (sswNew.cpp:1071-1259).  Every alignment of the winning score inside the reverse
//                   rectangle ends in its corner (ref_end is the FIRST column reaching the score, read_end
//                   the SMALLEST such row), so the reverse pass only has to follow paths that start in that
//                   corner: a pruned origin-anchored DP over a few hundred cells instead of (read_end+1) x
//                   (ref_end+1).  See DESIGN.md "reverse pass" for the argument.
//
// Windows in which the reference's signed lazy-F exit (Q2) could fire are flagged and re-run by the
// stripe-faithful kernel (kernels.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kernels.h"

namespace fasim {

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v2s a_s(v2u x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ v2u a_u(v2s x) { return __builtin_bit_cast(v2u, x); }
__device__ __forceinline__ int a_i(v2s x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ int a_i(v2u x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ v2u u_from(int x) { return __builtin_bit_cast(v2u, x); }
__device__ __forceinline__ v2s s_fromi(int x) { return __builtin_bit_cast(v2s, x); }

// packed 16-bit helpers as inline asm for the (rare) hazard branch: keeps the compiler from rewriting them into
// per-half compares/selects and from hoisting them into every step (see scan.hip)
__device__ __forceinline__ v2u apk_subs(v2u a, v2u b) { v2u r; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2u apk_subs_k(v2u a, uint32_t k) { v2u r; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "s"(k)); return r; }
__device__ __forceinline__ v2u apk_ksubs(uint32_t k, v2u a) { v2u r; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "s"(k), "v"(a)); return r; }
__device__ __forceinline__ v2u apk_minu(v2u a, v2u b) { v2u r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2u apk_minu_k(v2u a, uint32_t k) { v2u r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(k)); return r; }
__device__ __forceinline__ v2u apk_maxu(v2u a, v2u b) { v2u r; asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// x |= bits with input and output tied to one register (see scan.hip)
__device__ __forceinline__ void a_or_in_place(int& x, int bits) { asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(bits)); }
// the pipeline shift with zeros entering virtual lane 0 (no register to preload)
__device__ __forceinline__ int vshift2_zero(int x)
{
	const int up = __builtin_amdgcn_mov_dpp(x, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
	return __builtin_amdgcn_alignbit(x, up, 16);
}

__device__ __forceinline__ int vshift2(int x, int inject_hi)
{
	const int up = __builtin_amdgcn_update_dpp(inject_hi, x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
	return __builtin_amdgcn_alignbit(x, up, 16);
}

constexpr int AL_RS = 24;
constexpr int AL_LANE_STRIDE = 112;
constexpr int AL_CODE_STRIDE = 64 * AL_LANE_STRIDE;
constexpr int AL_SCALE = 32;                 // DP values are multiples of 32
constexpr int AL_NEG = -32768;               // void / dead score: H <= 31360 so H + AL_NEG < 0 always
constexpr int CODE_VOID = 4;                 // stream codes: A0 C1 G2 T3, 4 = void column (between windows), 5 = N
constexpr int CODE_SN = 5;                   // (the unit codes of kernels.h use 4 for N: k_build_stream re-codes)
constexpr int TAG_LAST = 8;                  // bit 3 of a stream byte: last column of a window
constexpr int TAG_HZ = 16;                   // bit 4 (in flight only): hazard seen in this column
// bits 5-6 of a stream byte: zone = how many of the candidate's LATER window tries (suffixes of this window) hold the column

// ------------------------------------------------------------------------------------------------
__global__ void k_build_stream(const uint8_t* __restrict__ tcodes, const FwdProb* __restrict__ probs, int32_t nprob,
	uint8_t* __restrict__ stream, const uint32_t* __restrict__ zones)
{
	const int p = blockIdx.x;
	if (p >= nprob) return;
	const FwdProb pb = probs[p];
	uint8_t* s = stream + pb.stream_off;
	// zones != NULL: the REVERSED window (last column first) for the reverse pass; zones[p] = lengths of the candidate's next
	// three window tries (one byte each, 0 = none).  They end in the window's last column, so in the reversed stream they are
	// prefixes: the zone of a column = how many of them hold it
	const uint32_t z = zones ? zones[p] : 0u;
	const int z1 = (int)(z & 0xffu), z2 = (int)((z >> 8) & 0xffu), z3 = (int)((z >> 16) & 0xffu);
	for (int c = threadIdx.x; c < pb.len + 2; c += blockDim.x) {
		uint8_t v;
		if (c < 2) v = CODE_VOID;
		else {
			const int col = c - 2;
			v = tcodes[pb.tbase + (zones ? pb.len - 1 - col : col)]; if (v >= 4) v = CODE_SN; if (col == pb.len - 1) v |= TAG_LAST;
			if (zones) v |= (uint8_t)(((col < z1) + (col < z2) + (col < z3)) << 5);
		}
		s[c] = v;
	}
}

hipError_t launch_build_stream(const uint8_t* tcodes, const FwdProb* probs, int32_t nprob, uint8_t* stream, const uint32_t* zones, hipStream_t st)
{
	if (nprob <= 0) return hipSuccess;
	hipLaunchKernelGGL(k_build_stream, dim3((unsigned)nprob), dim3(64), 0, st, tcodes, probs, nprob, stream, zones);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
struct FwdArgs {
	const uint8_t* stream;
	const FwdProb* probs;
	const int32_t* task_first;     // ntask + 1 problem indices
	int32_t ntask;
	uint32_t* counter;
	const uint8_t* qcodes;
	int32_t m, m_pad, seg_len16;
	FwdOut* out;
	int32_t vs, tile, ntiles;      // query tiling as in scan.hip
	uint4* boundary;               // [stream position]: {hbot | fbot<<16, fpo | hazard<<16, key, 0} between tiles
	// reverse pass (plain variant only; lane_ub != NULL): the query is staged reversed and the stream holds reversed windows, so H
	// of a cell = the best alignment that STARTS there.  Per window and zone z = 0..3 the maximum of every virtual lane over the
	// columns of zone >= z goes to lane_ub[(ub_slot[prob] * 4 + z) * nv + virtual lane]; no FwdOut is written.
	uint16_t* lane_ub; const int32_t* ub_slot; int32_t nv;
};

__device__ __forceinline__ void lane_rows_a(int v, int seg_len, int vs, int* row0, int* rows)
{
	const int s = v / vs, j = v - s * vs, q = seg_len / vs, rem = seg_len - q * vs;
	*rows = q + (j < rem ? 1 : 0);
	*row0 = s * seg_len + j * q + (j < rem ? j : rem);
}

// score of stream code t (0..3 bases, 4 void, 5 N) against the r-th row of global virtual lane v, scaled by sc
__device__ __forceinline__ int fwd_cell_score(const FwdArgs& a, int t, int v, int r, int sc)
{
	int row0v, rows_v;
	lane_rows_a(v, a.seg_len16, a.vs, &row0v, &rows_v);
	if (t == CODE_VOID || r >= rows_v) return AL_NEG;
	const int row = row0v + r;
	if (row >= a.m) return 0;                 // zero-score pad rows (Q3)
	const int q = a.qcodes[a.lane_ub ? a.m - 1 - row : row];
	return ((q == t && t < 4) ? 5 : -4) * sc;
}

// per-code int16 profile (43 KB) + one v_perm_b32 per row
// workgroup size: 512 threads = 8 waves share one 43 KB profile, so two workgroups (86 KB) put 4 waves
// on every SIMD; with 256-thread workgroups LDS allowed only three (3 waves per SIMD), and the packed VALU ops issue ~15 %
// slower at 3 waves per SIMD than at 4 (profiles/r02_valu_issue_bench.txt)
constexpr int FWD_THREADS = 512;
// REV (plain variant only): the reverse pass.  It wants the lane maxima per zone and nothing else, so the row tags of the keys,
// the key hand-over between lanes and the whole pipe end are left out (about 35 of 380 instructions per step).
template <int RP, bool TAINT, bool REV = false>
__global__ void __launch_bounds__(FWD_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) k_align_fwd(FwdArgs a)
{
	static_assert(!(TAINT && REV), "the reverse pass is a plain pass");
	constexpr int SC = TAINT ? 2 * AL_SCALE : AL_SCALE;      // value scale; TAINT: bit 5 = taint, bits 0..4 = row tag
	extern __shared__ __align__(16) uint8_t prof[];
	const int lane = threadIdx.x & 63;

	{
		for (int idx = threadIdx.x; idx < 6 * 128 * AL_RS; idx += blockDim.x) {
			const int r = idx % AL_RS;
			const int v = (idx / AL_RS) % 128;
			const int t = idx / (AL_RS * 128);
			const int sc = fwd_cell_score(a, t, 128 * a.tile + v, r, SC);
			*reinterpret_cast<int16_t*>(prof + t * AL_CODE_STRIDE + (v >> 1) * AL_LANE_STRIDE + (v & 1) * 48 + r * 2) = (int16_t)sc;
		}
	}
	__syncthreads();

	// stripe-aligned layout (see scan.hip): virtual lane 8k starts the reference's stripe k
	uint32_t fthr = 0xFFFFFFFFu, act = 0, startbits = 0;
	int row0[2];
	for (int h = 0; h < 2; h++) {
		const int v = 128 * a.tile + 2 * lane + h;
		int rows_v;
		lane_rows_a(v, a.seg_len16, a.vs, &row0[h], &rows_v);
		if (v % a.vs == 0 && v > 0) { fthr = (fthr & ~(0xFFFFu << (16 * h))) | ((131u * SC + (SC - 1)) << (16 * h)); startbits |= 0xFFFFu << (16 * h); }
		if (rows_v == RP) act |= 0xFFFFu << (16 * h);
	}
	const v2u fthr2 = u_from((int)fthr);
	const v2u actm = u_from((int)act);
	const v2u startm = u_from((int)startbits);
	const bool lvl2 = a.seg_len16 >= 96;      // (the 16-bit pass, !TAINT, needs no Q2 test at all: its compare is unaffected)
	const uint8_t* pl = prof + lane * AL_LANE_STRIDE;
	// (31 - r) tags for the row keys, and the base of the global-row key of my two virtual lanes
	const int kbase_lo = 0xFFFF - row0[0] - 31, kbase_hi = 0xFFFF - row0[1] - 31;

	for (;;) {
		int w = 0;
		if (lane == 0) w = (int)atomicAdd(a.counter, 1u);
		w = __builtin_amdgcn_readfirstlane(w);
		if (w >= a.ntask) break;
		const int p0 = a.task_first[w], p1 = a.task_first[w + 1];
		if (p1 <= p0) continue;
		const int s0 = a.probs[p0].stream_off;
		const FwdProb lastp = a.probs[p1 - 1];
		const int slen = lastp.stream_off + lastp.len + 2 - s0;
		const uint8_t* str = a.stream + s0;
		uint4* bnd = a.boundary + s0;
		const bool first_tile = a.tile == 0, last_tile = a.tile == a.ntiles - 1;
		uint4 bchunk = make_uint4(0u, 0u, 0u, 0u);

		int H[RP], E[RP];                // packed u16 pairs, kept as plain 32-bit registers (see scan.hip)
#pragma unroll
		for (int r = 0; r < RP; r++) { H[r] = 0; E[r] = 0; }
		int tc = (CODE_VOID << 16) | CODE_VOID;
		int hbot = 0, fbot = 0, recv_h_last = 0, fpo = 0;
		uint32_t klo = 0, khi = 0;           // (colmax << 16) | (0xFFFF - row) of my two virtual lanes' columns
		int chunk = CODE_VOID;
		// pipe-end state (meaningful in lane 63)
		int pidx = p0, cidx = 0, runmax = 0, end_ref = -1, end_read = 0, hzflag = 0, over = 0, wtaint = 0;
		v2u zacc0 = (v2u){ 0, 0 }, zacc1 = (v2u){ 0, 0 }, zacc2 = (v2u){ 0, 0 }, zacc3 = (v2u){ 0, 0 }, wcnt = (v2u){ 0, 0 };      // (reverse pass) zone maxima of my two virtual lanes; windows they have finished
		const int nsteps = slen + 127;
		for (int step = 0; step < nsteps; step++) {
			if ((step & 63) == 0) {
				const int c = step + lane;
				chunk = c < slen ? (int)str[c] : CODE_VOID;
				if (!first_tile) bchunk = c < slen ? bnd[c] : make_uint4(0u, 0u, 0u, 0u);
			}
			int newcode = __builtin_amdgcn_readlane(chunk, step & 63);
			int recv_h, recv_f, recv_fp;
			uint32_t kup;
			if (first_tile) {
				// zeros enter virtual lane 0 (bound_ctrl): nothing to preload
				tc = vshift2(tc, newcode << 16);
				recv_h = vshift2_zero(hbot); recv_f = vshift2_zero(fbot); recv_fp = vshift2_zero(fpo);
				kup = (uint32_t)__builtin_amdgcn_mov_dpp((int)khi, 0x138, 0xf, 0xf, true);
			} else {
				const uint32_t bx = (uint32_t)__builtin_amdgcn_readlane((int)bchunk.x, step & 63);
				const uint32_t by = (uint32_t)__builtin_amdgcn_readlane((int)bchunk.y, step & 63);
				const int in_k = __builtin_amdgcn_readlane((int)bchunk.z, step & 63);
				newcode |= (int)((by >> 16) & TAG_HZ);              // hazard seen by the tiles above
				tc = vshift2(tc, newcode << 16);
				recv_h = vshift2(hbot, (int)(bx << 16)); recv_f = vshift2(fbot, (int)(bx & 0xffff0000u)); recv_fp = vshift2(fpo, (int)(by << 16));
				kup = (uint32_t)__builtin_amdgcn_update_dpp(in_k, (int)khi, 0x138, 0xf, 0xf, false);
			}
			const uint32_t kin_lo = kup, kin_hi = klo;       // from virtual lane v-1 (same column, one step ago)
			const v2u tt = u_from(tc) & (v2u){ 7, 7 };                       // stream codes of my two halves
			// void columns clear E and F: the saturating subtractions use 0xFFFF instead of 4*SC / 16*SC
			const v2u isvoid = (v2u){ 0, 0 } - __builtin_elementwise_sub_sat((v2u){ 1, 1 }, tt ^ (v2u){ CODE_VOID, CODE_VOID });
			const v2u dec = isvoid | (v2u){ GAP_EXT * SC, GAP_EXT * SC };
			const v2u gapo = isvoid | (v2u){ GAP_OPEN * SC, GAP_OPEN * SC };
			const int hdiag0 = recv_h_last;
			recv_h_last = recv_h;
			v2u f = u_from(recv_f);
			v2s lkx[4] = { (v2s){ 0, 0 }, (v2s){ 0, 0 }, (v2s){ 0, 0 }, (v2s){ 0, 0 } };      // independent max chains
			constexpr int ROWS_PER_LOAD = 8;
			constexpr int NLOAD = (RP + ROWS_PER_LOAD - 1) / ROWS_PER_LOAD;
			v4i PA[NLOAD], PB[NLOAD];
			{
				const int t_lo = tc & 7, t_hi = (tc >> 16) & 7;
				const uint8_t* pa = pl + t_lo * AL_CODE_STRIDE;
				const uint8_t* pb = pl + t_hi * AL_CODE_STRIDE + 48;
#pragma unroll
				for (int g = 0; g < NLOAD; g++) {
					PA[g] = *reinterpret_cast<const v4i*>(pa + 16 * g);
					PB[g] = *reinterpret_cast<const v4i*>(pb + 16 * g);
				}
			}
			auto score_of = [&](int r) -> int {
				const int g = r >> 3, k = r & 7; return __builtin_amdgcn_perm(PB[g][k >> 1], PA[g][k >> 1], (k & 1) ? 0x07060302 : 0x05040100);
			};
			// (see scan.hip: the diagonal sum goes into the register of the score, the new H into the register of the old
			//  H, so the H column needs no second copy and no moves at the end of the step)
			auto diag_plus_score = [](int hold, int sc) -> v2s {
				if (TAINT) asm("v_pk_add_i16 %0, %1, %0 clamp" : "+v"(sc) : "v"(hold));
				else asm("v_pk_add_i16 %0, %1, %0" : "+v"(sc) : "v"(hold));
				return s_fromi(sc);
			};
			v2s t = diag_plus_score(hdiag0, score_of(0));
#pragma unroll
			for (int r = 0; r < RP; r++) {
				v2s tnext = t;
				if (r + 1 < RP) tnext = diag_plus_score(H[r], score_of(r + 1));
				v2s h = __builtin_elementwise_max(t, s_fromi(E[r]));
				{ int hn; asm("v_pk_max_i16 %0, %2, %3" : "=v"(hn) : "0"(H[r]), "v"(a_i(h)), "v"(a_i(f)), "v"(a_i(tnext))); h = s_fromi(hn); }
				H[r] = a_i(h);
				const v2u ho = __builtin_elementwise_sub_sat(a_u(h), gapo);
				E[r] = a_i(__builtin_elementwise_max(__builtin_elementwise_sub_sat(u_from(E[r]), dec), ho));
				const v2u fnew = __builtin_elementwise_max(__builtin_elementwise_sub_sat(f, dec), ho);
				const v2s key = REV ? h : (h | (v2s){ (short)(31 - r), (short)(31 - r) });
				if (r == RP - 1) {
					f = (fnew & actm) | (f & ~actm);
					lkx[r & 3] = __builtin_elementwise_max(lkx[r & 3], a_s(a_u(key) & actm));
					if (RP > 1) hbot = (a_i(h) & (int)act) | (H[RP > 1 ? RP - 2 : 0] & ~(int)act);
					else hbot = a_i(h);
				} else {
					f = fnew;
					lkx[r & 3] = __builtin_elementwise_max(lkx[r & 3], key);
				}
				t = tnext;
			}
			const v2s lkey = __builtin_elementwise_max(__builtin_elementwise_max(lkx[0], lkx[1]), __builtin_elementwise_max(lkx[2], lkx[3]));
			asm volatile("" :: "v"(a_i(lkey)));      // pin the reduction before the hazard branch (see scan.hip)
			fbot = a_i(f);
			if constexpr (REV) {
				{
					// reverse pass: my two columns' maxima (value = key >> 5) go into the running maxima of the zones that hold the column
					const v2u tcu = u_from(tc);
					const v2u zz = (tcu >> (v2u){ 5, 5 }) & (v2u){ 3, 3 };
					// (not the void columns: in the first one H still shows the E values the previous window left behind)
					const v2u val = (a_u(lkey) >> (v2u){ 5, 5 }) & ~isvoid;
					zacc0 = __builtin_elementwise_max(zacc0, val);
					zacc1 = __builtin_elementwise_max(zacc1, val & ((v2u){ 0, 0 } - __builtin_elementwise_min(zz, (v2u){ 1, 1 })));
					zacc2 = __builtin_elementwise_max(zacc2, val & ((v2u){ 0, 0 } - __builtin_elementwise_min(__builtin_elementwise_sub_sat(zz, (v2u){ 1, 1 }), (v2u){ 1, 1 })));
					zacc3 = __builtin_elementwise_max(zacc3, val & ((v2u){ 0, 0 } - __builtin_elementwise_sub_sat(zz, (v2u){ 2, 2 })));
					const v2u lastb = (tcu >> (v2u){ 3, 3 }) & (v2u){ 1, 1 };
					if (__builtin_amdgcn_ballot_w64(a_i(lastb) != 0) != 0ull) {
#pragma unroll
						for (int h = 0; h < 2; h++) {
							if (lastb[h]) {
								const int slot = a.ub_slot[p0 + (int)wcnt[h]];
								if (slot >= 0) {
									uint16_t* o = a.lane_ub + (size_t)slot * 4 * a.nv + 128 * a.tile + 2 * lane + h;
									o[0] = zacc0[h]; o[a.nv] = zacc1[h]; o[2 * (size_t)a.nv] = zacc2[h]; o[3 * (size_t)a.nv] = zacc3[h];
								}
							}
						}
						const v2u keep = lastb - (v2u){ 1, 1 };
						zacc0 &= keep; zacc1 &= keep; zacc2 &= keep; zacc3 &= keep;
						wcnt += lastb;
					}
				}
			}
			// Q2 (8-bit pass only).  Short queries: any F[b] >= 132 entering a stripe flags the column (TAG_HZ).  Otherwise the
			// row analysis of scan.hip marks the cells the reference's early lazy-F exit would have left smaller (taint bit).
			fpo = 0;
			if (TAINT) {
				if (lvl2) {
					constexpr uint32_t K1 = 0x00010001u, KGE = (uint32_t)(132 * SC - 1) * 0x10001u, KLT = (uint32_t)(144 * SC) * 0x10001u,
						KE = (uint32_t)(GAP_EXT * SC) * 0x10001u, KO = (uint32_t)(GAP_OPEN * SC) * 0x10001u;
					// cheap superset first: a stripe-start lane receives F >= 132, or some lane receives a live chain
					const v2u cand = apk_subs_k(u_from(recv_f) & startm, KGE) | u_from(recv_fp);
					bool enter = false;
					v2u fp_in = (v2u){ 0, 0 }, arm_in = (v2u){ 0, 0 };
					if (__builtin_amdgcn_ballot_w64(a_i(cand) != 0) != 0ull) {
						const v2u fpraw = ((u_from(recv_f) & startm) | (u_from(recv_fp) & ~startm)) & ~isvoid;
						fp_in = fpraw & (v2u){ 0x7fff, 0x7fff };
						arm_in = (fpraw >> (v2u){ 15, 15 }) & ~startm;
						const v2u hot = apk_subs_k(fp_in, KGE) | apk_minu(arm_in, fp_in);
						enter = __builtin_amdgcn_ballot_w64(a_i(hot) != 0) != 0ull;
					}
					if (enter) {
						v2u fp = fp_in, arm = arm_in;
						const v2u one = (v2u){ 1, 1 };
						// pass 1 reads H / E and collects the decisions as bit r of (dh, de)[r / 16]; pass 2 ORs the taint bit in
						uint32_t dh[2] = { 0u, 0u }, de[2] = { 0u, 0u };
#pragma unroll
						for (int r = 0; r < RP; r++) {
							const v2u hr = u_from(H[r]);
							const v2u ge = apk_minu_k(apk_subs_k(fp, KGE), K1);                         // Fp >= 132
							v2u lt = apk_minu_k(apk_ksubs(KLT, hr), K1) | ((hr >> (v2u){ 5, 5 }) & one);      // H < 144, or tainted
							v2u eq = apk_ksubs(K1, apk_subs(hr, fp));                                   // H == Fp (H >= Fp always)
							v2u nfp = apk_subs_k(fp, KE);
							if (r == RP - 1) { lt &= actm; eq &= actm; nfp = (nfp & actm) | (fp & ~actm); }
							const v2u dev = apk_minu(apk_minu(eq, fp), arm);                            // 0 / 1
							const v2u ho = apk_subs_k(hr, KO);
							const v2u efrom = apk_minu(apk_ksubs(K1, apk_subs(u_from(E[r]), ho)), ho);
							dh[r >> 4] |= (uint32_t)a_i(dev) << (r & 15);
							de[r >> 4] |= (uint32_t)a_i(apk_minu(efrom, dev)) << (r & 15);
							arm = apk_maxu(arm, apk_minu(ge, lt));
							fp = nfp;
						}
#pragma unroll
						for (int r = 0; r < RP; r++) {
							a_or_in_place(H[r], (int)(((dh[r >> 4] >> (r & 15)) & K1) << 5));
							a_or_in_place(E[r], (int)(((de[r >> 4] >> (r & 15)) & K1) << 5));
						}
						if (RP > 1) hbot = (H[RP - 1] & (int)act) | (H[RP > 1 ? RP - 2 : 0] & ~(int)act);
						else hbot = H[0];
						fpo = a_i(fp | (arm << (v2u){ 15, 15 }));
					}
				} else {
					const v2u hz_b = __builtin_elementwise_sub_sat(u_from(recv_f), fthr2);
					tc |= a_i(__builtin_elementwise_min(hz_b, (v2u){ 1, 1 }) << (v2u){ 4, 4 });
				}
			}
			if constexpr (REV) {
				// (no result per window: only the hand-over of the tile's bottom row to the next tile)
				if (!last_tile) {
					const int pos = step - 127;
					if (lane == 63 && pos >= 0 && pos < slen)
						bnd[pos] = make_uint4(((uint32_t)hbot >> 16) | ((uint32_t)fbot & 0xffff0000u), ((uint32_t)tc & 0xffff0000u), 0u, 0u);
				}
				continue;
			}
			// per-column (max, smallest row) keys
			const uint32_t lk = (uint32_t)a_i(lkey);
			const uint32_t loc_lo = (((lk & 0xFFFFu) >> 5) << 16) | (uint32_t)(kbase_lo + (int)(lk & 31u));
			const uint32_t loc_hi = ((lk >> 21) << 16) | (uint32_t)(kbase_hi + (int)((lk >> 16) & 31u));
			klo = kin_lo > loc_lo ? kin_lo : loc_lo;
			khi = kin_hi > loc_hi ? kin_hi : loc_hi;

			// ---- pipe end: virtual lane 127 has just finished one column of the stream ---------------------
			if (!last_tile) {
				// hand the bottom row of this tile to the next one (in place: this stream position was read 127 steps ago)
				const int pos = step - 127;
				if (lane == 63 && pos >= 0 && pos < slen)
					bnd[pos] = make_uint4(((uint32_t)hbot >> 16) | ((uint32_t)fbot & 0xffff0000u),
						((uint32_t)fpo >> 16) | ((uint32_t)tc & 0xffff0000u), khi, 0u);
			} else {
				// everything below is wave-uniform: the two values of lane 63 are read into scalar registers and the
				// bookkeeping runs on the scalar unit (it used to cost ~35 vector instructions per step for one active lane)
				const int tag = (__builtin_amdgcn_readlane(tc, 63) >> 16) & 0xff;
				const uint32_t k63 = (uint32_t)__builtin_amdgcn_readlane((int)khi, 63);
				if ((tag & 7) != CODE_VOID) {
					// TAINT: the key's value field is 2 * maximum + taint of the winning cell
					const int cfield = (int)(k63 >> 16);
					const int colmax = TAINT ? (cfield >> 1) : cfield;
					// 8-bit pass: the reference stops at the first column whose maximum reaches 251 (overflow -> 16-bit pass), so
					// nothing after that column matters (in particular not the saturated, hence "tainted", values further on)
					if (!(TAINT && over)) {
						if (tag & TAG_HZ) hzflag = 1;
						if (colmax > runmax) { runmax = colmax; end_ref = cidx; end_read = 0xFFFF - (int)(k63 & 0xFFFFu); wtaint = TAINT ? (cfield & 1) : 0; }
						if (runmax >= 255 - BIAS) over = 1;
					}
					cidx++;
					if (tag & TAG_LAST) {
						if (lane == 0) {
							FwdOut o;
							o.score = runmax; o.ref_end = end_ref; o.read_end = end_read < a.m - 1 ? end_read : a.m - 1;
							o.flags = hzflag | wtaint; o.ref_begin = 0; o.read_begin = 0;
							a.out[pidx] = o;
						}
						pidx++; cidx = 0; runmax = 0; end_ref = -1; end_read = 0; hzflag = 0; over = 0; wtaint = 0;
					}
				}
			}
		}
	}
}

template <int RP, bool TAINT, bool REV = false>
static hipError_t launch_fwd_t(const FwdArgs& a, hipStream_t st)
{
	hipError_t err = hipMemsetAsync(a.counter, 0, sizeof(uint32_t), st);
	if (err != hipSuccess) return err;
	// (the host sizes the tasks so that there are about 3072 of them: one task per wave, workgroups are short-lived)
	constexpr int WPB = FWD_THREADS / 64;
	long blocks = ((long)a.ntask + WPB - 1) / WPB;
	if (blocks > 256 * 2) blocks = 256 * 2;
	hipLaunchKernelGGL((k_align_fwd<RP, TAINT, REV>), dim3((unsigned)blocks), dim3(FWD_THREADS), (size_t)6 * AL_CODE_STRIDE, st, a);
	return hipGetLastError();
}

hipError_t launch_align_fwd(const FwdLaunch& L, hipStream_t st)
{
	if (L.ntask <= 0) return hipSuccess;
	if (!systolic_fits(L.m)) return hipErrorInvalidValue;
	FwdArgs a;
	a.stream = L.stream; a.probs = L.probs; a.task_first = L.task_first; a.ntask = L.ntask; a.counter = L.counter;
	a.qcodes = L.qcodes; a.m = L.m; a.m_pad = 16 * ((L.m + 15) / 16); a.seg_len16 = (L.m + 15) / 16; a.out = L.out;
	a.vs = systolic_vs(L.m); a.ntiles = a.vs / 8; a.boundary = L.boundary;
	a.lane_ub = L.word ? L.lane_ub : nullptr; a.ub_slot = L.ub_slot; a.nv = 128 * a.ntiles;
	if (a.ntiles > 1 && !a.boundary) return hipErrorInvalidValue;
	const int rp = (a.seg_len16 + a.vs - 1) / a.vs;
	for (int t = 0; t < a.ntiles; t++) {
		a.tile = t;
		hipError_t err = hipErrorInvalidValue;
		switch (rp) {
#define FASIM_FWD_CASE(N) case N: err = a.lane_ub ? launch_fwd_t<N, false, true>(a, st) : (L.word ? launch_fwd_t<N, false>(a, st) : launch_fwd_t<N, true>(a, st)); break;
		FASIM_FWD_CASE(1) FASIM_FWD_CASE(2) FASIM_FWD_CASE(3) FASIM_FWD_CASE(4) FASIM_FWD_CASE(5) FASIM_FWD_CASE(6)
		FASIM_FWD_CASE(7) FASIM_FWD_CASE(8) FASIM_FWD_CASE(9) FASIM_FWD_CASE(10) FASIM_FWD_CASE(11) FASIM_FWD_CASE(12)
		FASIM_FWD_CASE(13) FASIM_FWD_CASE(14) FASIM_FWD_CASE(15) FASIM_FWD_CASE(16) FASIM_FWD_CASE(17) FASIM_FWD_CASE(18)
		FASIM_FWD_CASE(19) FASIM_FWD_CASE(20) FASIM_FWD_CASE(21) FASIM_FWD_CASE(22) FASIM_FWD_CASE(23) FASIM_FWD_CASE(24)
#undef FASIM_FWD_CASE
		default: break;
		}
		if (err != hipSuccess) return err;
	}
	return hipSuccess;
}

// ------------------------------------------------------------------------------------------------
// k_finish
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int swsc(int a, int b) { return (a == b && a < 4) ? 5 : -4; }
__device__ __forceinline__ uint32_t cig(int len, int op) { return ((uint32_t)len << 4) | (uint32_t)op; }

constexpr int G_ROWS = 640;          // rows the origin-anchored reverse DP may touch
constexpr int G_NEG = -1000000;

// Reverse pass.  Returns 0 and sets (*ref_begin, *read_begin), or 1 when it cannot decide (caller re-runs the
// window on the stripe-faithful kernel).
__device__ int reverse_pass(const uint8_t* __restrict__ tw /* window codes */, const uint8_t* __restrict__ q, int S, int ref_end,
	int read_end, int32_t* Gp, int32_t* Ep, int* ref_begin, int* read_begin)
{
	const int R = read_end + 1, C = ref_end + 1;
	// column 0
	const int g0 = swsc(q[read_end], tw[ref_end]);
	if (g0 <= 0) return 1;
	if (g0 == S) { *ref_begin = ref_end; *read_begin = read_end; return 0; }
	int lo = 0, hi = 0;
	Gp[0] = g0; Ep[0] = G_NEG;
	for (int i = 1; i < R && i < G_ROWS; i++) {
		const int g = g0 - GAP_OPEN - GAP_EXT * (i - 1);
		const int rest = min(R - 1 - i, C - 1);
		if (g <= 0 || g + 5 * rest < S) break;
		Gp[i] = g; Ep[i] = G_NEG; hi = i;
	}
	for (int j = 1; j < C; j++) {
		const int tcode = tw[ref_end - j];
		int F = G_NEG;
		int diag = G_NEG;                // old G[i-1]
		int nlo = -1, nhi = -1;
		const int colrest = C - 1 - j;
		for (int i = lo; i < R; i++) {
			if (i >= G_ROWS) return 1;
			const bool in = (i <= hi);
			const int gp = in ? Gp[i] : G_NEG;
			const int ep = in ? Ep[i] : G_NEG;
			int e = max(ep - GAP_EXT, gp - GAP_OPEN);
			int g = diag > G_NEG / 2 ? diag + swsc(q[read_end - i], tcode) : G_NEG;
			g = max(g, max(e, F));
			const int rest = min(R - 1 - i, colrest);
			if (g <= 0 || g + 5 * rest < S) g = G_NEG;
			if (e <= 0 || e + 5 * rest < S) e = G_NEG;
			if (g == S) { *ref_begin = ref_end - j; *read_begin = read_end - i; return 0; }
			diag = gp;
			Gp[i] = g; Ep[i] = e;
			if (g > G_NEG / 2 || e > G_NEG / 2) { if (nlo < 0) nlo = i; nhi = i; }
			F = max(F - GAP_EXT, g - GAP_OPEN);
			if (F <= 0) F = G_NEG;
			if (i > hi && F <= G_NEG / 2 && diag <= G_NEG / 2) break;   // nothing can reach the rows below
		}
		if (nlo < 0) return 1;           // no path left: cannot happen for a consistent forward result
		lo = nlo; hi = nhi;
	}
	return 1;                            // score never reached inside the rectangle: let the exact kernel decide
}

__global__ void __launch_bounds__(64) k_finish(const uint8_t* __restrict__ tcodes, const uint8_t* __restrict__ qcodes,
	const FwdProb* __restrict__ probs, const FwdOut* __restrict__ fwd, int32_t nprob, uint8_t* __restrict__ scratch,
	int32_t scratch_cap, AlignOutDev* __restrict__ out, uint32_t* __restrict__ cigar_pool, uint32_t pool_cap,
	uint32_t* __restrict__ pool_count, const int32_t* __restrict__ idx_list)
{
	const int gid = blockIdx.x * blockDim.x + threadIdx.x;
	if (gid >= nprob) return;
	const int pi = idx_list ? idx_list[gid] : gid;          // scratch slot = gid, result slot = pi
	const FwdProb pb = probs[pi];
	const FwdOut fo = fwd[pi];
	AlignOutDev* o = out + pi;
	o->sw_score = 0; o->ref_begin = 0; o->ref_end = fo.ref_end; o->query_begin = 0; o->query_end = fo.read_end; o->cigar_len = 0;
	o->cigar_off = 0;
	if (fo.flags & 1) { o->status = 10; return; }                  // possible Q2 in the forward pass: stripe-faithful re-run
	if (fo.score <= 0 || fo.ref_end < 0) { o->status = 0; return; }   // nothing aligned
	uint8_t* my = scratch + (int64_t)gid * scratch_cap;
	const uint8_t* tw = tcodes + pb.tbase;
	int ref_begin = fo.ref_begin, read_begin = fo.read_begin;
	if (!(fo.flags & 2)) {
		// below 148 no F of the reverse pass can reach 132, so the signed lazy-F exit (Q2) cannot fire there; from 251 on
		// the reference uses its 16-bit kernels, whose compare is not affected
		if (fo.score >= 148 && fo.score < 255 - BIAS) { o->status = 11; return; }
		int32_t* Gp = reinterpret_cast<int32_t*>(my);
		int32_t* Ep = Gp + G_ROWS;
		if (reverse_pass(tw, qcodes, fo.score, fo.ref_end, fo.read_end, Gp, Ep, &ref_begin, &read_begin)) { o->status = 11; return; }
	}
	o->ref_begin = ref_begin; o->query_begin = read_begin;
	// ---- banded_sw (sswNew.cpp:1071-1259)
	const uint8_t* ref = tw + ref_begin;
	const uint8_t* read = qcodes + read_begin;
	const int refLen = fo.ref_end - ref_begin + 1, readLen = fo.read_end - read_begin + 1, score = fo.score;
	// The reference keeps three band arrays of 2*band+3 ints and 3 direction bytes per band cell, width_d = 2*band+1 per
	// row.  Only columns [max(0,i-band), min(refLen-1,i+band)] of row i exist, so array indices never exceed refLen and a
	// row never holds more than refLen cells: the arrays are capped at refLen+3 entries and the direction rows at
	// min(width_d, refLen) cells of ONE byte (bit 0: E opened, bit 1: F opened, bits 2-3: source of H).  A band much wider
	// than the window (read_begin far from read_end, as after a reverse pass derailed by Q2) therefore costs
	// readLen * refLen bytes instead of readLen * width_d * 3.
	const int wcap_max = (scratch_cap >= 65536) ? 4099 : 256;
	int32_t* h_b = reinterpret_cast<int32_t*>(my);
	int band = (refLen > readLen ? refLen - readLen : readLen - refLen) + 1;
	int maxv = 0, width = 0, width_d = 0, stride = 0;
	const int wcap = refLen + 3 < wcap_max ? refLen + 3 : wcap_max;
	int32_t* e_b = h_b + wcap;
	int32_t* h_c = e_b + wcap;
	uint8_t* direction = reinterpret_cast<uint8_t*>(h_c + wcap);
	const long dir_cap = (long)scratch_cap - (long)3 * wcap * 4;
	if (dir_cap <= 0) { o->status = 2; return; }
	do {
		width = band * 2 + 3; width_d = band * 2 + 1;
		const bool dense = width_d > refLen;              // rows indexed by the column itself
		stride = dense ? refLen : width_d;
		const int wuse = width < wcap ? width : wcap;      // entries of the band arrays that can ever be touched
		if ((width > wcap && refLen + 3 > wcap) || (long)stride * readLen > dir_cap) { o->status = 2; return; }
		for (int j = 1; j < wuse - 1; j++) h_b[j] = 0;
		for (int i = 0; i < readLen; i++) {
			int beg = 0, end = refLen - 1, u = 0;
			if (i - band > beg) beg = i - band;
			if (i + band < end) end = i + band;
			const int edge = end + 1 < width - 1 ? end + 1 : width - 1;
			int f = 0;
			h_b[0] = 0; e_b[0] = 0; h_c[0] = 0;
			if (edge < wcap) { h_b[edge] = 0; e_b[edge] = 0; }
			uint8_t* line = direction + (long)stride * i;
			const int x = i - band > 0 ? i - band : 0;
			const int xp = i - 1 - band > 0 ? i - 1 - band : 0;
			const int rd = read[i];
			const int lx = dense ? 0 : x;
			int hleft = 0;                                // h_c[j - x]: the H just written for column j-1 (h_c[0] = 0 at j = beg = x)
			int hdiag = h_b[beg - xp];                    // h_b[d], d = j - xp
			for (int j = beg; j <= end; j++) {
				u = j - x + 1;
				const int e = j - xp + 1;
				const int hbe = h_b[e];
				int t1 = i == 0 ? -GAP_OPEN : hbe - GAP_OPEN;
				int t2 = i == 0 ? -GAP_EXT : e_b[e] - GAP_EXT;
				const int ev = t1 > t2 ? t1 : t2;
				e_b[u] = ev;
				const int de = t1 > t2 ? 1 : 0;
				t1 = hleft - GAP_OPEN;
				t2 = f - GAP_EXT;
				f = t1 > t2 ? t1 : t2;
				const int df = t1 > t2 ? 1 : 0;
				const int e1 = ev > 0 ? ev : 0;
				const int f1 = f > 0 ? f : 0;
				t1 = e1 > f1 ? e1 : f1;
				t2 = hdiag + swsc(ref[j], rd);
				const int hv = t1 > t2 ? t1 : t2;
				h_c[u] = hv;
				if (hv > maxv) maxv = hv;
				const int dh = (t1 <= t2) ? 0 : (e1 > f1 ? 1 : 2);
				line[j - lx] = (uint8_t)(de | (df << 1) | (dh << 2));
				hleft = hv; hdiag = hbe;
			}
			for (int j = 1; j <= u; j++) h_b[j] = h_c[j];
		}
		band *= 2;
		if (maxv < score && band > 4 * (refLen + readLen) + 16) { o->status = 3; return; }
	} while (maxv < score);
	band /= 2;
	width_d = band * 2 + 1;
	const bool dense = width_d > refLen;
	uint32_t rc[ALIGN_MAX_CIGAR];
	int l = 0;
	int i = readLen - 1, j = refLen - 1, e = 0, state = 2, op = 0, prev_op = 0, status = 0;
	while (i > 0) {
		const int x = i - band > 0 ? i - band : 0;
		int beg = 0, end = refLen - 1;
		if (i - band > beg) beg = i - band;
		if (i + band < end) end = i + band;
		if (j < beg || j > end) { status = 3; break; }
		const uint32_t nib = direction[(long)stride * i + (j - (dense ? 0 : x))];
		// direction byte of the reference: state 0 (E) -> 3/2, state 1 (F) -> 5/4, state 2 (H) -> 1 or the E/F byte
		int dv;
		const int dE = (nib & 1) ? 3 : 2, dF = (nib & 2) ? 5 : 4;
		if (state == 0) dv = dE; else if (state == 1) dv = dF; else { const int dh = (nib >> 2) & 3; dv = dh == 0 ? 1 : (dh == 1 ? dE : dF); }
		if (dv == 1) { --i; --j; state = 2; op = 0; }
		else if (dv == 2) { --i; state = 0; op = 1; }
		else if (dv == 3) { --i; state = 2; op = 1; }
		else if (dv == 4) { --j; state = 1; op = 2; }
		else { --j; state = 2; op = 2; }
		if (op == prev_op) ++e;
		else {
			if (l >= ALIGN_MAX_CIGAR) { status = 4; break; }
			rc[l++] = cig(e, prev_op);
			prev_op = op; e = 1;
		}
	}
	if (status == 0) {
		if (op == 0) { if (l >= ALIGN_MAX_CIGAR) status = 4; else rc[l++] = cig(e + 1, 0); }
		else { if (l + 2 > ALIGN_MAX_CIGAR) status = 4; else { rc[l++] = cig(e, op); rc[l++] = cig(1, 0); } }
	}
	if (status == 0) {
		const uint32_t off = atomicAdd(pool_count, (uint32_t)l);
		if ((uint64_t)off + (uint64_t)l > (uint64_t)pool_cap) status = 4;
		else { o->cigar_off = off; for (int k = 0; k < l; k++) cigar_pool[off + k] = rc[l - 1 - k]; }
	}
	o->status = status;
	if (status != 0) return;
	o->sw_score = score;
	o->cigar_len = l;
}


// ------------------------------------------------------------------------------------------------
// k_finish_lds: the same reverse pass + banded_sw with the small per-alignment arrays in LDS (256 B per thread,
// word-interleaved across the 64 threads so equal indices fall into different banks: 16 KB per workgroup, ~10 waves
// per CU) and the direction matrix (1 byte per cell, [cell][lane] inside a per-wave slab) in global memory: its writes
// are fire-and-forget and only the traceback reads it back.  Alignments that do not fit report status 2 and are
// finished by k_finish.
//   words  0..63 : reverse pass ring, entry (row & 63): lo16 = G, hi16 = E   (reused by the banded arrays afterwards)
//   words  0..31 : banded: h_b (lo16) / e_b (hi16)
//   words 32..63 : banded: h_c (lo16)
// ------------------------------------------------------------------------------------------------
// Two sizes: WORDS = 64 (16 KB per workgroup; ring of 64 rows, band arrays of 32, 2 048 direction cells: 97.7 % of the alignments of
// the bench) and WORDS = 256 (64 KB; ring 256, band arrays 128, 16 384 cells) for what comes back from the small one with status 2.

struct LdsArena {
	uint32_t* base;                   // &lds[threadIdx.x]; word k lives at base[k * 64]
	__device__ __forceinline__ int lo(int k) const { return (int)(int16_t)(base[k * 64] & 0xffffu); }
	__device__ __forceinline__ int hi(int k) const { return (int)(int16_t)(base[k * 64] >> 16); }
	__device__ __forceinline__ void set(int k, int l, int h) const { base[k * 64] = ((uint32_t)(uint16_t)(int16_t)l) | ((uint32_t)(uint16_t)(int16_t)h << 16); }
	__device__ __forceinline__ void set_lo(int k, int l) const { base[k * 64] = (base[k * 64] & 0xffff0000u) | (uint32_t)(uint16_t)(int16_t)l; }
	__device__ __forceinline__ void set_hi(int k, int h) const { base[k * 64] = (base[k * 64] & 0x0000ffffu) | ((uint32_t)(uint16_t)(int16_t)h << 16); }
};

constexpr int GL_NEG = -30000;        // int16-safe "minus infinity" of the reverse pass

template <int FL_RING>
__device__ int reverse_pass_lds(const uint8_t* __restrict__ tw, const uint8_t* __restrict__ q, int S, int ref_end, int read_end,
	const LdsArena& A, int* ref_begin, int* read_begin)
{
	const int R = read_end + 1, C = ref_end + 1;
	const int g0 = swsc(q[read_end], tw[ref_end]);
	if (g0 <= 0) return 1;
	if (g0 == S) { *ref_begin = ref_end; *read_begin = read_end; return 0; }
	int lo = 0, hi = 0;
	A.set(0, g0, GL_NEG);
	// (a vertical gap straight out of the corner cell scores 5 - 16 < 0: column 0 holds row 0 only)
	for (int j = 1; j < C; j++) {
		const int tcode = tw[ref_end - j];
		int F = GL_NEG, diag = GL_NEG, nlo = -1, nhi = -1;
		const int colrest = C - 1 - j;
		if (hi - lo + 2 >= FL_RING) return 2;             // band wider than the ring: finish in global memory
		for (int i = lo; i < R; i++) {
			const bool in = (i <= hi);
			const int k = i & (FL_RING - 1);
			const int gp = in ? A.lo(k) : GL_NEG;
			const int ep = in ? A.hi(k) : GL_NEG;
			int e = max(ep - GAP_EXT, gp - GAP_OPEN);
			int g = diag > GL_NEG / 2 ? diag + swsc(q[read_end - i], tcode) : GL_NEG;
			g = max(g, max(e, F));
			const int rest = min(R - 1 - i, colrest);
			if (g <= 0 || g + 5 * rest < S) g = GL_NEG;
			if (e <= 0 || e + 5 * rest < S) e = GL_NEG;
			if (g == S) { *ref_begin = ref_end - j; *read_begin = read_end - i; return 0; }
			diag = gp;
			A.set(k, g, e);
			if (g > GL_NEG / 2 || e > GL_NEG / 2) { if (nlo < 0) nlo = i; nhi = i; }
			F = max(F - GAP_EXT, g - GAP_OPEN);
			if (F <= 0) F = GL_NEG;
			if (i > hi && F <= GL_NEG / 2 && diag <= GL_NEG / 2) break;
			if (i - lo + 2 >= FL_RING) return 2;
		}
		if (nlo < 0) return 1;
		lo = nlo; hi = nhi;
	}
	return 1;
}

template <int FL_WORDS, int FL_CELLS>
__global__ void __launch_bounds__(64) k_finish_lds(const uint8_t* __restrict__ tcodes, const uint8_t* __restrict__ qcodes,
	const FwdProb* __restrict__ probs, const FwdOut* __restrict__ fwd, const int32_t* __restrict__ order, int32_t nprob,
	uint8_t* __restrict__ dirs, AlignOutDev* __restrict__ out, uint32_t* __restrict__ cigar_pool, uint32_t pool_cap,
	uint32_t* __restrict__ pool_count)
{
	constexpr int FL_RING = FL_WORDS, FL_W = FL_WORDS / 2;      // ring of the reverse pass; max band array width (2*band+3), h_c behind h_b/e_b
	__shared__ uint32_t lds[FL_WORDS * 64];
	const int gid = blockIdx.x * blockDim.x + threadIdx.x;
	if (gid >= nprob) return;
	const int pi = order ? order[gid] : gid;              // alignments are processed in an order that groups similar sizes
	LdsArena A; A.base = lds + threadIdx.x;
	uint8_t* dir = dirs + (int64_t)blockIdx.x * (64 * FL_CELLS) + threadIdx.x;    // cell c at dir[c * 64]
	const FwdProb pb = probs[pi];
	const FwdOut fo = fwd[pi];
	AlignOutDev* o = out + pi;
	o->sw_score = 0; o->ref_begin = 0; o->ref_end = fo.ref_end; o->query_begin = 0; o->query_end = fo.read_end; o->cigar_len = 0;
	o->cigar_off = 0;
	if (fo.flags & 1) { o->status = 10; return; }
	if (fo.score <= 0 || fo.ref_end < 0) { o->status = 0; return; }
	const uint8_t* tw = tcodes + pb.tbase;
	int ref_begin = fo.ref_begin, read_begin = fo.read_begin;
	if (!(fo.flags & 2)) {
		// 148..250: an F >= 132 is possible in the reverse pass of the 8-bit kernel; >= 251 runs on the reference's
		// 16-bit kernels, whose compare is not affected
		if (fo.score >= 148 && fo.score < 255 - BIAS) { o->status = 11; return; }
		const int r = reverse_pass_lds<FL_RING>(tw, qcodes, fo.score, fo.ref_end, fo.read_end, A, &ref_begin, &read_begin);
		if (r) { o->status = r == 2 ? 2 : 11; return; }
	}
	o->ref_begin = ref_begin; o->query_begin = read_begin;
	const uint8_t* ref = tw + ref_begin;
	const uint8_t* read = qcodes + read_begin;
	const int refLen = fo.ref_end - ref_begin + 1, readLen = fo.read_end - read_begin + 1, score = fo.score;
	int band = (refLen > readLen ? refLen - readLen : readLen - refLen) + 1;
	int maxv = 0, width = 0, width_d = 0;
	do {
		width = band * 2 + 3; width_d = band * 2 + 1;
		if (width + 1 > FL_W || width_d * readLen > FL_CELLS) { o->status = 2; return; }
		for (int j = 1; j < width - 1; j++) A.set_lo(j, 0);                 // h_b[j] = 0
		for (int i = 0; i < readLen; i++) {
			int beg = 0, end = refLen - 1, u = 0;
			if (i - band > beg) beg = i - band;
			if (i + band < end) end = i + band;
			const int edge = end + 1 < width - 1 ? end + 1 : width - 1;
			int f = 0;
			A.set(0, 0, 0); A.set(edge, 0, 0); A.set_lo(FL_W, 0);               // h_b[0]=e_b[0]=h_b[edge]=e_b[edge]=h_c[0]=0
			const int x = i - band > 0 ? i - band : 0;
			const int xp = i - 1 - band > 0 ? i - 1 - band : 0;
			const int rd = read[i];
			const int rowcell = width_d * i;
			for (int j = beg; j <= end; j++) {
				u = j - x + 1;
				const int e = j - xp + 1, b = j - x, d = j - xp;
				int t1 = i == 0 ? -GAP_OPEN : A.lo(e) - GAP_OPEN;
				int t2 = i == 0 ? -GAP_EXT : A.hi(e) - GAP_EXT;
				const int ev = t1 > t2 ? t1 : t2;
				const int de = t1 > t2 ? 1 : 0;
				t1 = A.lo(FL_W + b) - GAP_OPEN;
				t2 = f - GAP_EXT;
				f = t1 > t2 ? t1 : t2;
				const int df = t1 > t2 ? 1 : 0;
				const int e1 = ev > 0 ? ev : 0;
				const int f1 = f > 0 ? f : 0;
				t1 = e1 > f1 ? e1 : f1;
				t2 = A.lo(d) + swsc(ref[j], rd);                                  // h_b[d] (still the previous row's)
				const int hv = t1 > t2 ? t1 : t2;
				A.set_hi(u, ev);                                                  // e_b[u]
				A.set_lo(FL_W + u, hv);                                             // h_c[u]
				if (hv > maxv) maxv = hv;
				const int dh = (t1 <= t2) ? 0 : (e1 > f1 ? 1 : 2);
				// bit0: E opened (3) vs extended (2); bit1: F opened (5) vs extended (4); bits 2-3: H from diag / E / F
				dir[(int64_t)(rowcell + (j - x)) * 64] = (uint8_t)(de | (df << 1) | (dh << 2));
			}
			for (int j = 1; j <= u; j++) A.set_lo(j, A.lo(FL_W + j));               // h_b[j] = h_c[j]
		}
		band *= 2;
		if (maxv < score && band > 4 * (refLen + readLen) + 16) { o->status = 3; return; }
	} while (maxv < score);
	band /= 2;
	uint32_t rc[ALIGN_MAX_CIGAR];
	int l = 0;
	int i = readLen - 1, j = refLen - 1, e = 0, state = 2, op = 0, prev_op = 0, status = 0;
	while (i > 0) {
		const int x = i - band > 0 ? i - band : 0;
		int beg = 0, end = refLen - 1;
		if (i - band > beg) beg = i - band;
		if (i + band < end) end = i + band;
		if (j < beg || j > end) { status = 3; break; }
		const uint32_t nib = dir[(int64_t)(width_d * i + (j - x)) * 64];
		// direction byte of the reference: state 0 (E) -> 3/2, state 1 (F) -> 5/4, state 2 (H) -> 1 or the E/F byte
		int dv;
		const int dE = (nib & 1) ? 3 : 2, dF = (nib & 2) ? 5 : 4;
		if (state == 0) dv = dE; else if (state == 1) dv = dF; else { const int dh = (nib >> 2) & 3; dv = dh == 0 ? 1 : (dh == 1 ? dE : dF); }
		if (dv == 1) { --i; --j; state = 2; op = 0; }
		else if (dv == 2) { --i; state = 0; op = 1; }
		else if (dv == 3) { --i; state = 2; op = 1; }
		else if (dv == 4) { --j; state = 1; op = 2; }
		else { --j; state = 2; op = 2; }
		if (op == prev_op) ++e;
		else {
			if (l >= ALIGN_MAX_CIGAR) { status = 4; break; }
			rc[l++] = cig(e, prev_op);
			prev_op = op; e = 1;
		}
	}
	if (status == 0) {
		if (op == 0) { if (l >= ALIGN_MAX_CIGAR) status = 4; else rc[l++] = cig(e + 1, 0); }
		else { if (l + 2 > ALIGN_MAX_CIGAR) status = 4; else { rc[l++] = cig(e, op); rc[l++] = cig(1, 0); } }
	}
	if (status == 0) {
		const uint32_t off = atomicAdd(pool_count, (uint32_t)l);
		if ((uint64_t)off + (uint64_t)l > (uint64_t)pool_cap) status = 4;
		else { o->cigar_off = off; for (int k = 0; k < l; k++) cigar_pool[off + k] = rc[l - 1 - k]; }
	}
	o->status = status;
	if (status != 0) return;
	o->sw_score = score;
	o->cigar_len = l;
}

hipError_t launch_finish(const uint8_t* tcodes, const uint8_t* qcodes, const FwdProb* probs, const FwdOut* fwd, const int32_t* order,
	int32_t nprob, uint8_t* dirs, AlignOutDev* out, uint32_t* cigar_pool, uint32_t pool_cap, uint32_t* pool_count, hipStream_t st)
{
	// LDS-resident kernel for every alignment; the few that do not fit come back with status 2 (see launch_finish_big).
	// `dirs` must hold ceil(nprob/64) * 64 * 2048 bytes.
	if (nprob <= 0) return hipSuccess;
	hipError_t err = hipMemsetAsync(pool_count, 0, sizeof(uint32_t), st);
	if (err != hipSuccess) return err;
	hipLaunchKernelGGL((k_finish_lds<64, 2048>), dim3((nprob + 63) / 64), dim3(64), 0, st, tcodes, qcodes, probs, fwd, order, nprob, dirs, out,
		cigar_pool, pool_cap, pool_count);
	return hipGetLastError();
}

// the same kernel with four times the LDS per alignment for the listed ones (results to out[idx_list[k]], cigars appended to the
// same pool); `dirs` must hold ceil(nlist/64) * 64 * 16384 bytes
hipError_t launch_finish_mid(const uint8_t* tcodes, const uint8_t* qcodes, const FwdProb* probs, const FwdOut* fwd, const int32_t* idx_list,
	int32_t nlist, uint8_t* dirs, AlignOutDev* out, uint32_t* cigar_pool, uint32_t pool_cap, uint32_t* pool_count, hipStream_t st)
{
	if (nlist <= 0) return hipSuccess;
	static bool attr_set = false;                   // (benign race: the call is idempotent)
	if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_finish_lds<256, 16384>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); attr_set = true; }
	hipLaunchKernelGGL((k_finish_lds<256, 16384>), dim3((nlist + 63) / 64), dim3(64), 0, st, tcodes, qcodes, probs, fwd, idx_list, nlist, dirs, out,
		cigar_pool, pool_cap, pool_count);
	return hipGetLastError();
}

// global-scratch variant for the listed alignments (results go to out[idx_list[k]], cigars are appended to the same pool)
hipError_t launch_finish_big(const uint8_t* tcodes, const uint8_t* qcodes, const FwdProb* probs, const FwdOut* fwd,
	const int32_t* idx_list, int32_t nlist, uint8_t* scratch, int32_t scratch_cap, AlignOutDev* out, uint32_t* cigar_pool,
	uint32_t pool_cap, uint32_t* pool_count, hipStream_t st)
{
	if (nlist <= 0) return hipSuccess;
	hipLaunchKernelGGL(k_finish, dim3((nlist + 63) / 64), dim3(64), 0, st, tcodes, qcodes, probs, fwd, nlist, scratch, scratch_cap, out,
		cigar_pool, pool_cap, pool_count, idx_list);
	return hipGetLastError();
}

} // namespace fasim
