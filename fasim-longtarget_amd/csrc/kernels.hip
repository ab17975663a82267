// fasim-longtarget_amd/csrc/kernels.hip -- gfx950 (MI355X) kernels of the triplex scan hot path.
//
//   k_encode    rule encodings (rules.h:94-318 transferString / reverseSeq) -> target codes per unit
//   k_striped   stripe-faithful Smith-Waterman: one 16-lane DPP row per problem, 4 problems per wave64,
//               problems pulled from a device work queue.  Three modes:
//                 MODE_PRE    sw_sse2_byte_once (sswNew.cpp:255-464): column maxima incl. Q1 break, Q2, Q3
//                 MODE_MAX1   calc_score_once (stats.h:879-956): exact max (8-bit, re-run in 16-bit on overflow)
//                 MODE_ALIGN  forward + reverse pass of ssw_align (sswNew.cpp:1446-1525)
//   k_hits      columns above the unit's threshold, ordered (first half of Aligner::preAlign, ssw_cpp.cpp:446-457)
//   k_banded    banded_sw traceback (sswNew.cpp:1071-1259), one thread per alignment
//
// Integer DP: no MFMA.  H/E columns and the striped query live in LDS; stripe-to-stripe hand-over of the
// diagonal H and of the lazy-F value uses 16-wide wave shuffles; target codes are read 16 columns at a
// time (one coalesced byte per lane) and broadcast by shuffle.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <type_traits>
#include "kernels.h"

namespace fasim {

// ------------------------------------------------------------------------------------------------
// k_encode
// ------------------------------------------------------------------------------------------------
__global__ void k_encode(const uint8_t* __restrict__ dna, const int32_t* __restrict__ seg_start,
	const int32_t* __restrict__ seg_len, const int32_t* __restrict__ enc_ids, int32_t nenc,
	const uint8_t* __restrict__ enc_lut, uint8_t* __restrict__ tcodes, int32_t tstride)
{
	const int unit = blockIdx.x;                 // seg * nenc + k
	const int seg = unit / nenc, k = unit - seg * nenc;
	const int enc = enc_ids[k];
	const int n = seg_len[seg];
	const int64_t s0 = seg_start[seg];
	const bool rev = (enc & 1) != 0;             // odd encodings are the reversed ones (Fasim-LongTarget.cpp:428,520)
	const uint8_t* lut = enc_lut + enc * 256;
	for (int c = threadIdx.x; c < tstride; c += blockDim.x) {
		const int src = rev ? (n - 1 - c) : c;
		tcodes[(int64_t)unit * tstride + c] = c < n ? lut[dna[s0 + src]] : (uint8_t)CODE_N;
	}
}

hipError_t launch_encode(const uint8_t* dna_dev, const int32_t* seg_start, const int32_t* seg_len, int32_t nseg,
	const int32_t* enc_ids, int32_t nenc, const uint8_t* enc_lut, uint8_t* tcodes, int32_t tstride, hipStream_t st)
{
	if (nseg <= 0 || nenc <= 0) return hipSuccess;
	hipLaunchKernelGGL(k_encode, dim3((unsigned)(nseg * nenc)), dim3(256), 0, st, dna_dev, seg_start, seg_len, enc_ids, nenc,
		enc_lut, tcodes, tstride);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// k_striped
// ------------------------------------------------------------------------------------------------
struct StripedArgs {
	const uint8_t* tcodes;
	const uint8_t* qcodes;
	const StripedProb* probs;
	int32_t nprob;
	uint32_t* counter;
	ScoreLut lut;
	int32_t s4;                 // bytes-of-rows stride of one stripe in LDS (multiple of 4, (s4/4) odd)
	uint8_t* colmax;
	uint16_t* colmax_w;         // MODE_PRE in word mode: 16-bit column maxima (same indexing as colmax)
	int32_t* max_out;
	AlignEnds* ends;
	// chunked hazard re-run (MODE_PRE, byte).  A problem = chunk `aux` of hazard unit `unit` (index into the hazard list):
	// columns [c[aux], c[aux + 1]) of the unit, with c = chunk_cols[unit][0 .. HAZARD_MAX_CHUNKS] (-1 after the last entry).
	// It starts from checkpoint state[chunk_base[unit] + aux] (the zero state if c[aux] == 0) and writes its column
	// maxima into its OWN row chunk_rows[chunk_base[unit] + aux][column].  At the end of a chunk the DP state is
	// compared with the next checkpoint: equal -> done (the next chunk's own run starts from exactly this state); different
	// (a Q2 deviation is still alive) -> the same group keeps going through the next chunk's columns, and so on.
	const uint16_t* state;      // [idx][2][state_rows]: H then E of every row, 2 * value + taint as k_scan carries them
	int32_t state_rows;
	const int32_t* chunk_cols;  // [hazard unit][HAZARD_MAX_CHUNKS + 1]
	const int32_t* chunk_base;  // [hazard unit]: index of its chunk 0 in state / chunk_rows
	uint8_t* chunk_rows;        // [chunk_base[unit] + chunk][row_stride]
	int32_t row_stride;
	int32_t* chunk_out;         // [problem][4]: {last chunk covered, overflow column or -1, ticks of 10 ns spent, start tick}
};

__device__ __forceinline__ int group_max16(int v)
{
	v = max(v, __shfl_xor(v, 8, 16));
	v = max(v, __shfl_xor(v, 4, 16));
	v = max(v, __shfl_xor(v, 2, 16));
	v = max(v, __shfl_xor(v, 1, 16));
	return v;
}
__device__ __forceinline__ int group_min16(int v)
{
	v = min(v, __shfl_xor(v, 8, 16));
	v = min(v, __shfl_xor(v, 4, 16));
	v = min(v, __shfl_xor(v, 2, 16));
	v = min(v, __shfl_xor(v, 1, 16));
	return v;
}
// true if any lane of this 16-lane group has c set
__device__ __forceinline__ bool group_any16(bool c)
{
	const unsigned long long b = __ballot(c);
	const int sh = (threadIdx.x & 63) & 48;
	return ((b >> sh) & 0xffffull) != 0;
}

// One column of the 8-bit kernel for this lane's stripe.  Returns the lane's column maximum.
// (main loop sswNew.cpp:332-357 / stats.h:673-689, lazy-F loop sswNew.cpp:360-371)
template <bool QUIRK>
__device__ __forceinline__ int column_byte(uint8_t* Hs, uint8_t* Es, const uint8_t* Qs, int segLen, uint32_t lut, int s)
{
	int hd = __shfl_up((int)Hs[segLen - 1], 1, 16);
	if (s == 0) hd = 0;
	int f = 0, cmax = 0;
	const int nfull = segLen & ~3;
	for (int j = 0; j < nfull; j += 4) {
		const uint32_t hw = *reinterpret_cast<const uint32_t*>(Hs + j);
		const uint32_t ew = *reinterpret_cast<const uint32_t*>(Es + j);
		const uint32_t qw = *reinterpret_cast<const uint32_t*>(Qs + j);
		uint32_t hn = 0, en = 0;
#pragma unroll
		for (int b = 0; b < 4; b++) {
			const int qc4 = (qw >> (8 * b)) & 0xff;
			const int p = (lut >> qc4) & 0xf;
			int e = (ew >> (8 * b)) & 0xff;
			const int hold = (hw >> (8 * b)) & 0xff;
			int h = hd + p - BIAS;
			h = min(h, 255 - BIAS);              // adds_epu8 saturation, then subs_epu8(bias)
			h = max(h, e);
			h = max(h, f);                       // e,f >= 0 => h >= 0
			cmax = max(cmax, h);
			hn |= (uint32_t)h << (8 * b);
			const int ho = h - GAP_OPEN;
			e = max(max(e - GAP_EXT, ho), 0);
			f = max(max(f - GAP_EXT, ho), 0);
			en |= (uint32_t)e << (8 * b);
			hd = hold;
		}
		*reinterpret_cast<uint32_t*>(Hs + j) = hn;
		*reinterpret_cast<uint32_t*>(Es + j) = en;
	}
	for (int j = nfull; j < segLen; j++) {
		const int p = (lut >> Qs[j]) & 0xf;
		int e = Es[j];
		const int hold = Hs[j];
		int h = hd + p - BIAS;
		h = min(h, 255 - BIAS);
		h = max(h, e);
		h = max(h, f);
		cmax = max(cmax, h);
		Hs[j] = (uint8_t)h;
		const int ho = h - GAP_OPEN;
		e = max(max(e - GAP_EXT, ho), 0);
		f = max(max(f - GAP_EXT, ho), 0);
		Es[j] = (uint8_t)e;
		hd = hold;
	}
	bool go = true;
	for (int k = 0; k < 16 && go; k++) {
		const int fs = __shfl_up(f, 1, 16);
		f = (s == 0) ? 0 : fs;
		for (int j = 0; j < segLen; j++) {
			int h = Hs[j];
			h = max(h, f);
			cmax = max(cmax, h);
			Hs[j] = (uint8_t)h;
			const int ho = max(h - GAP_OPEN, 0);
			f = max(f - GAP_EXT, 0);
			const bool c = QUIRK ? ((int)(int8_t)f > (int)(int8_t)ho) : (f > ho);
			// The reference leaves the loop when no lane has vF > vH - gapO (signed compare: Q2).  With H - gapO >= 128 that
			// compare stays true even for F = 0, and the reference then grinds through all 16 * segLen iterations -- which change
			// nothing once every lane's F has decayed to 0 (h = max(h, 0), F stays 0 through the shifts).  Leaving at that point
			// gives the same H column and the same maximum.
			if (!group_any16(c) || !group_any16(f > 0)) { go = false; break; }
		}
	}
	return cmax;
}

// One column of the 16-bit kernel (8 stripes; lanes 8..15 of the group idle with part == false).
// (sswNew.cpp:963-1001 / stats.h:555-598)
__device__ __forceinline__ int column_word(uint16_t* Hs, uint16_t* Es, const uint8_t* Qs, int segLen, uint32_t lut, int s, bool part)
{
	int hd = __shfl_up((int)(int16_t)Hs[segLen - 1], 1, 16);
	if (s == 0) hd = 0;
	int f = 0, cmax = 0;
	for (int j = 0; j < segLen; j++) {
		const int p = (int)((lut >> Qs[j]) & 0xf) - BIAS;
		int e = (int16_t)Es[j];
		const int hold = (int16_t)Hs[j];
		int h = min(hd + p, 32767);               // adds_epi16
		h = max(h, e);
		h = max(h, f);
		cmax = max(cmax, h);
		Hs[j] = (uint16_t)h;
		const int ho = max(h - GAP_OPEN, 0);       // subs_epu16
		e = max(max(e - GAP_EXT, 0), ho);
		f = max(max(f - GAP_EXT, 0), ho);
		Es[j] = (uint16_t)e;
		hd = hold;
	}
	bool go = true;
	for (int k = 0; k < 8 && go; k++) {
		const int fs = __shfl_up(f, 1, 16);
		f = (s == 0) ? 0 : fs;
		for (int j = 0; j < segLen; j++) {
			int h = (int16_t)Hs[j];
			h = max(h, f);
			cmax = max(cmax, h);
			Hs[j] = (uint16_t)h;
			const int ho = max(h - GAP_OPEN, 0);
			f = max(f - GAP_EXT, 0);
			if (!group_any16(part && f > ho)) { go = false; break; }
		}
	}
	return part ? cmax : 0;
}

template <int MODE, bool WORD, bool QUIRK>
__global__ void __launch_bounds__(256) k_striped(StripedArgs a)
{
	extern __shared__ __align__(16) uint8_t lds[];
	using HT = typename std::conditional<WORD, uint16_t, uint8_t>::type;
	constexpr int P = WORD ? 8 : 16;
	const int g = threadIdx.x >> 4;
	const int s = threadIdx.x & 15;
	const int S4 = a.s4;
	// MODE_PRE / MODE_MAX1: every problem of the launch aligns the same whole query, so its striped copy is shared by
	// the groups of the workgroup (placed after the per-group H/E regions); the alignment modes need one per group
	constexpr bool SHARED_Q = (MODE == MODE_PRE || MODE == MODE_MAX1);
	const size_t gbytes = (size_t)(SHARED_Q ? (WORD ? 64 : 32) : (WORD ? 80 : 48)) * S4;
	uint8_t* base = lds + g * gbytes;
	HT* Hs = reinterpret_cast<HT*>(base) + (size_t)s * S4;
	HT* Es = reinterpret_cast<HT*>(base) + (size_t)16 * S4 + (size_t)s * S4;
	uint8_t* Qs = (SHARED_Q ? lds + (blockDim.x >> 4) * gbytes : base + (size_t)(WORD ? 64 : 32) * S4) + (size_t)s * S4;
	const bool part = s < P;
	const uint32_t l0 = a.lut.row[0], l1 = a.lut.row[1], l2 = a.lut.row[2], l3 = a.lut.row[3], l4 = a.lut.row[4];

	int phase = -1;          // -1: fetch a problem; 0: forward pass; 1: reverse pass (MODE_ALIGN)
	int pi = 0, unit = 0, t0 = 0, refLen = 0, qlen = 0, dir = 0, terminate = 0, segLen = 1;
	int ci = 0, maxv = 0, end_ref = 0, end_read = 0, tchunk = CODE_N;
	int hz_unit = 0, hz_chunk = 0, hz_base = 0, pb_chunk0 = 0;   // chunked hazard re-run: hazard-list index, the chunk being worked on, the first one
	bool st_init = false;
	uint64_t t_start = 0;
	int64_t tbase = 0;
	bool overflow = false, setup = false, qrev = false, shared_q_built = false;
	AlignEnds res;
	res.score_fwd = res.ref_end = res.read_end = res.score_rev = res.ref_begin = res.read_begin = 0;

	for (;;) {
		if (phase < 0) {
			int idx = 0;
			if (s == 0) idx = (int)atomicAdd(a.counter, 1u);
			idx = __shfl(idx, 0, 16);
			if (idx >= a.nprob) break;
			pi = idx;
			const StripedProb pb = a.probs[idx];
			tbase = pb.tbase; t0 = pb.t0; refLen = pb.ref_len; qlen = pb.q_len; unit = pb.unit;
			dir = 0; qrev = false; terminate = WORD ? 65535 : 255;
			phase = 0; setup = true;
			if constexpr (MODE == MODE_PRE && !WORD) {
				st_init = false;
				if (a.chunk_out) { hz_unit = pb.unit; hz_chunk = pb_chunk0 = pb.aux; hz_base = a.chunk_base[hz_unit]; st_init = t0 > 0; t_start = wall_clock64(); }
			}
			res.score_fwd = res.ref_end = res.read_end = res.score_rev = res.ref_begin = res.read_begin = 0;
			if constexpr (MODE == MODE_REV) {
				// reverse pass only (sswNew.cpp:1508-1516): the forward result is known and exact
				dir = 1; qrev = true; terminate = pb.aux; phase = 1;
				res.score_fwd = pb.aux; res.ref_end = refLen - 1; res.read_end = qlen - 1;
			}
		}
		if (setup) {
			// stripe geometry of this pass: segLen = ceil(qlen / P) (sswNew.cpp:279, 911)
			segLen = (qlen + P - 1) / P;
			for (int j = 0; j < segLen; j++) {
				const int row = s * segLen + j;
				int code = CODE_PAD;                 // pad rows score 0 (sswNew.cpp:195, 690): Q3
				if (part && row < qlen) code = a.qcodes[qrev ? (qlen - 1 - row) : row];
				if (!SHARED_Q || g == 0 || !shared_q_built) Qs[j] = (uint8_t)(code * 4);
				Hs[j] = 0;
				Es[j] = 0;
				if constexpr (MODE == MODE_PRE && !WORD) {
					if (st_init) {                       // resume from a checkpoint: the state after column t0 - 1
						const uint16_t* sp = a.state + (size_t)(hz_base + hz_chunk) * 2 * a.state_rows;
						Hs[j] = (HT)(sp[row] >> 1); Es[j] = (HT)(sp[a.state_rows + row] >> 1);
					}
				}
			}
			shared_q_built = true;               // (all groups write identical bytes the first time: benign)
			ci = 0; maxv = 0; overflow = false;
			end_ref = WORD ? 0 : -1;                 // sswNew.cpp:278 vs :910
			end_read = qlen - 1;
			setup = false;
		}
		bool stop = (refLen <= 0);
		if (!stop) {
			const int i = dir ? (refLen - 1 - ci) : ci;          // window coordinate of this column
			if ((ci & 15) == 0) {
				const int cc = ci + s;
				const int ii = dir ? (refLen - 1 - cc) : cc;
				tchunk = (cc < refLen) ? (int)a.tcodes[tbase + t0 + ii] : CODE_N;
			}
			const int t = __shfl(tchunk, ci & 15, 16);
			const uint32_t lut = t == 0 ? l0 : t == 1 ? l1 : t == 2 ? l2 : t == 3 ? l3 : l4;
			int cmax;
			if constexpr (WORD) cmax = column_word(Hs, Es, Qs, segLen, lut, s, part);
			else cmax = column_byte<QUIRK>(Hs, Es, Qs, segLen, lut, s);
			const int colmax = group_max16(cmax);
			if (colmax > maxv) {
				maxv = colmax;
				if (!WORD && maxv + BIAS >= 255) { overflow = true; stop = true; }   // sswNew.cpp:386 / stats.h:729
				else {
					end_ref = i;
					if constexpr (MODE == MODE_ALIGN || MODE == MODE_REV) {
						// smallest row whose stored H equals the new maximum (sswNew.cpp:621-629)
						int best = 0x7fffffff;
						if (part) {
							for (int j = 0; j < segLen; j++) {
								if ((int)Hs[j] == colmax) { best = s * segLen + j; break; }
							}
						}
						best = group_min16(best);
						end_read = best < qlen - 1 ? best : qlen - 1;
					}
				}
			}
			if (!stop) {
				if constexpr (MODE == MODE_PRE) {
					if (s == 0) {
						if constexpr (WORD) a.colmax_w[tbase + t0 + i] = (uint16_t)colmax;
						else if (a.chunk_out) a.chunk_rows[(size_t)(hz_base + pb_chunk0) * a.row_stride + t0 + i] = (uint8_t)colmax;
						else a.colmax[tbase + t0 + i] = (uint8_t)colmax;
					}
				}
				if (colmax == terminate) stop = true;
			}
			ci++;
			if (ci >= refLen) stop = true;
		}
		if (stop) {
			const int score = (!WORD && overflow) ? 255 : maxv;
			if constexpr (MODE == MODE_PRE) {
				if constexpr (!WORD) {
					if (a.chunk_out) {
						// end of chunk hz_chunk: is the state the checkpoint the next chunk's own run started from?
						const int32_t* cc = a.chunk_cols + (size_t)hz_unit * (HAZARD_MAX_CHUNKS + 1);
						const int nxt = hz_chunk + 1;
						const int c_after = (nxt < HAZARD_MAX_CHUNKS) ? cc[nxt + 1] : -1;      // end of the next chunk (-1: there is none)
						if (!overflow && c_after >= 0) {
							const uint16_t* sc = a.state + (size_t)(hz_base + nxt) * 2 * a.state_rows;
							bool differs = false;
							for (int j = 0; j < segLen; j++) {
								const int row = s * segLen + j;
								if ((sc[row] >> 1) != (uint16_t)Hs[j] || (sc[a.state_rows + row] >> 1) != (uint16_t)Es[j]) { differs = true; break; }
							}
							if (group_any16(differs)) {
								// keep going: the deviation is alive.  (the target letters of the 16-column block in flight were
								// fetched against the old end: again)
								hz_chunk = nxt; refLen = c_after - t0;
								const int cc16 = (ci & ~15) + s;
								tchunk = (cc16 < refLen) ? (int)a.tcodes[tbase + t0 + cc16] : CODE_N;
								continue;
							}
						}
						if (s == 0) {
							int32_t* co = a.chunk_out + (size_t)4 * pi;
							co[0] = hz_chunk; co[1] = overflow ? t0 + ci - 1 : -1; co[2] = (int)(wall_clock64() - t_start); co[3] = (int)(uint32_t)t_start;
						}
					} else if (overflow) {
						// Q1: the overflowing column and everything after it stay 0 (calloc'd array, sswNew.cpp:282)
						for (int c = (ci - 1) + s; c < refLen; c += 16) a.colmax[tbase + t0 + c] = 0;
					}
				}
				if (s == 0 && !a.chunk_out) a.max_out[unit] = score;
				phase = -1;
			} else if constexpr (MODE == MODE_MAX1) {
				if (s == 0) a.max_out[unit] = score;
				phase = -1;
			} else {
				if (MODE == MODE_ALIGN && phase == 0) {
					res.score_fwd = score;
					res.ref_end = end_ref;
					res.read_end = (maxv == 0) ? 0 : end_read;
					if (score == 0 || (!WORD && score == 255) || end_ref < 0) {
						if (s == 0) a.ends[pi] = res;          // nothing aligned, or needs the 16-bit re-run
						phase = -1;
					} else {
						// reverse pass over the reversed query prefix, columns ref_end..0 (sswNew.cpp:1508-1516)
						qlen = res.read_end + 1; refLen = res.ref_end + 1;
						dir = 1; qrev = true; terminate = score;
						phase = 1; setup = true;
					}
				} else {
					res.score_rev = score;
					res.ref_begin = end_ref;
					res.read_begin = res.read_end - ((maxv == 0) ? 0 : end_read);
					if (s == 0) a.ends[pi] = res;
					phase = -1;
				}
			}
		}
	}
}

template <int MODE, bool WORD, bool QUIRK>
static hipError_t launch_striped_t(const StripedLaunch& L, hipStream_t st)
{
	if (L.nprob <= 0) return hipSuccess;
	constexpr int P = WORD ? 8 : 16;
	int segLen = (L.max_qlen + P - 1) / P;
	int s4 = (segLen + 3) & ~3;
	if (((s4 / 4) & 1) == 0) s4 += 4;          // odd dword stride between stripes: conflict-free LDS rows
	constexpr bool SHARED_Q = (MODE == MODE_PRE || MODE == MODE_MAX1);
	const size_t qbytes = (size_t)16 * s4;
	const size_t gbytes = (size_t)(SHARED_Q ? (WORD ? 64 : 32) : (WORD ? 80 : 48)) * s4;
	// up to four 16-lane groups (problems) per one-wave workgroup; long queries get fewer groups so that one group's
	// H/E/query stripes still fit the 160 KB of LDS
	int groups = (int)(((size_t)160 * 1024 - (SHARED_Q ? qbytes : 0)) / gbytes);
	// at most 4 problems per workgroup (a cap of two would keep a workgroup's LDS below the ~20 KB that
	// four k_scan workgroups leave free on a CU, so these latency-bound kernels can start while a scan is running
	const int max_groups = 4;
	if (L.spread) { if (groups > 16) groups = 16; }        // four waves of four problems: the whole CU (see below)
	else if (groups > max_groups) groups = max_groups;
	if (groups < 1) return hipErrorInvalidValue;           // query too long for the LDS-resident kernel
	size_t shmem = (size_t)groups * gbytes + (SHARED_Q ? qbytes : 0);
	// spread: a request of more than half the CU's LDS keeps the dispatcher from packing several of these workgroups onto one
	// CU.  A wave of this kernel alone on its SIMD issues back to back; four of them on one SIMD take four times as long.
	if (L.spread && shmem < (size_t)84 * 1024) shmem = (size_t)84 * 1024;
	auto kern = k_striped<MODE, WORD, QUIRK>;
	hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
	if (err != hipSuccess) return err;
	err = hipMemsetAsync(L.counter, 0, sizeof(uint32_t), st);
	if (err != hipSuccess) return err;
	StripedArgs a;
	a.tcodes = L.tcodes; a.qcodes = L.qcodes; a.probs = L.probs; a.nprob = L.nprob; a.counter = L.counter;
	a.lut = L.lut; a.s4 = s4; a.colmax = L.colmax; a.colmax_w = L.colmax_w; a.max_out = L.max_out; a.ends = L.ends;
	a.state = L.state; a.state_rows = L.state_rows; a.chunk_cols = L.chunk_cols; a.chunk_base = L.chunk_base; a.chunk_rows = L.chunk_rows; a.row_stride = L.row_stride; a.chunk_out = L.chunk_out;
	// enough one-wave workgroups to fill the chip at the LDS-limited occupancy; the queue balances the rest
	int per_cu = (int)((160 * 1024) / shmem);
	if (per_cu < 1) per_cu = 1;
	if (per_cu > 16) per_cu = 16;
	long blocks = (long)256 * per_cu;
	const long need = ((long)L.nprob + groups - 1) / groups;
	if (blocks > need) blocks = need;
	hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3((unsigned)(16 * groups)), shmem, st, a);
	return hipGetLastError();
}

hipError_t launch_striped(StripedMode mode, bool word, bool quirk, const StripedLaunch& a, hipStream_t st)
{
	switch (mode) {
	case MODE_PRE:
		// (the reference's 16-bit pre-align kernel is dead code (Q1); the word variant serves the ssw.h shim, which needs the
		//  column maxima of sw_sse2_word for the sub-optimal score of alignments that overflow 8 bits)
		return word ? launch_striped_t<MODE_PRE, true, false>(a, st) : launch_striped_t<MODE_PRE, false, true>(a, st);
	case MODE_MAX1:
		return word ? launch_striped_t<MODE_MAX1, true, false>(a, st) : launch_striped_t<MODE_MAX1, false, false>(a, st);
	case MODE_ALIGN:
		return word ? launch_striped_t<MODE_ALIGN, true, false>(a, st) : launch_striped_t<MODE_ALIGN, false, true>(a, st);
	case MODE_REV:
		return launch_striped_t<MODE_REV, false, true>(a, st);      // byte mode only (forward score < 251)
	}
	(void)quirk;
	return hipErrorInvalidValue;
}

// ---- chunked hazard re-run: planning and merging (one wave per hazard unit) ------------------------------------------
// k_hazard_plan: where the stripe-faithful re-run of a unit starts and how its columns are cut into chunks.
//   * start f = first pipeline step at which k_scan saw a possible Q2 taint, minus the 127 columns the pipeline holds: the
//     column maxima before f are the reference's already (k_hazard_merge copies them).
//   * [f, n) is cut into K <= HAZARD_MAX_CHUNKS chunks of about equal COST.  A column whose maximum reaches `hot_thr` is
//     priced `hot_w` times a plain one: there the reference's lazy-F loop keeps running (H - gapO >= 128 reads as negative
//     in its signed compare, sswNew.cpp:369), and so does the emulation.
// chunk_cols[k][0 .. K] = chunk boundaries (c[0] = f, c[K] = n, -1 beyond).
__global__ void __launch_bounds__(64) k_hazard_plan(const int32_t* __restrict__ unit_ids, const int32_t* __restrict__ unit_len,
	const int32_t* __restrict__ unit_first, const uint16_t* __restrict__ colmax16, int32_t tstride, int32_t target, int32_t hot_thr, int32_t hot_w,
	int32_t* __restrict__ chunk_cols)
{
	constexpr int KC = HAZARD_MAX_CHUNKS;
	const int k = blockIdx.x, lane = threadIdx.x, unit = unit_ids[k], n = unit_len[unit];
	const int fs = unit_first[unit];
	const int f = fs >= 0x7f000000 ? 0 : max(0, min(n - 1, fs - 127));
	const uint16_t* src = colmax16 + (int64_t)unit * tstride;
	auto weight = [&](int c) { return ((int)(src[c] >> 1) >= hot_thr) ? hot_w : 1; };
	const int B = (n - f + 63) / 64;
	const int c0 = min(n, f + lane * B), c1 = min(n, c0 + B);
	int wsum = 0;
	for (int c = c0; c < c1; c++) wsum += weight(c);
	int incl = wsum;
	for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
	const int W = max(1, __shfl(incl, 63, 64));
	const int K = max(1, min(KC, (W + target - 1) / target));
	int32_t* cc = chunk_cols + (size_t)k * (KC + 1);
	if (lane == 0) cc[0] = f;
	for (int j = K + lane; j <= KC; j += 64) cc[j] = j == K ? n : -1;
	// boundary q = the first column whose exclusive cost prefix p satisfies floor(p * K / W) >= q  (W / K > hot_w: the quotient
	// moves by at most one per column)
	int p = incl - wsum;
	int qlast = (c0 > f && c0 < n) ? (int)(((int64_t)(p - weight(c0 - 1)) * K) / W) : 0;
	for (int c = c0; c < c1; c++) {
		const int q = (int)(((int64_t)p * K) / W);
		if (q > qlast && q < K) cc[q] = c;
		qlast = q;
		p += weight(c);
	}
}

hipError_t launch_hazard_plan(const int32_t* unit_ids, int32_t nlist, const int32_t* unit_len, const int32_t* unit_first, const uint16_t* colmax16,
	int32_t tstride, int32_t target, int32_t hot_thr, int32_t hot_w, int32_t* chunk_cols, hipStream_t st)
{
	if (nlist <= 0) return hipSuccess;
	hipLaunchKernelGGL(k_hazard_plan, dim3((unsigned)nlist), dim3(64), 0, st, unit_ids, unit_len, unit_first, colmax16, tstride, target, hot_thr, hot_w,
		chunk_cols);
	return hipGetLastError();
}

// k_hazard_merge: the u8 column maxima of a hazard unit, put together from
//   * columns [0, c[0]): the systolic result (u16: 2 * max + taint -> u8) -- no taint could arise before c[0];
//   * chunk r: the private row of the group that covered it (src_chunk[k][r]: the chunk that group STARTED with; a group whose
//     end state differed from the next checkpoint has kept going through the following chunks);
//   * the overflow rule (Q1, sswNew.cpp:386): the first column whose maximum reaches 255 - bias and everything after it are 0.
//     zero_from[k] = the overflow column a valid group reported (-1: none); an overflow inside the prefix is found here.
__global__ void __launch_bounds__(64) k_hazard_merge(const uint16_t* __restrict__ colmax16, uint8_t* __restrict__ colmax, const int32_t* __restrict__ unit_ids,
	const int32_t* __restrict__ unit_len, const int32_t* __restrict__ chunk_cols, const int32_t* __restrict__ chunk_base, const int32_t* __restrict__ src_chunk, const int32_t* __restrict__ zero_from,
	const uint8_t* __restrict__ chunk_rows, int32_t row_stride, int32_t tstride)
{
	constexpr int KC = HAZARD_MAX_CHUNKS;
	const int k = blockIdx.x, unit = unit_ids[k], n = unit_len[unit];
	const uint16_t* src = colmax16 + (int64_t)unit * tstride;
	uint8_t* dst = colmax + (int64_t)unit * tstride;
	const int32_t* cc = chunk_cols + (size_t)k * (KC + 1);
	const int up = cc[0];
	int zf = zero_from[k];
	int pf = 0x7fffffff;
	for (int c = threadIdx.x; c < up; c += 64) if ((int)(src[c] >> 1) + BIAS >= 255) { pf = c; break; }
	for (int o = 32; o; o >>= 1) pf = min(pf, __shfl_xor(pf, o, 64));
	if (pf != 0x7fffffff) zf = pf;
	for (int c = threadIdx.x; c < n; c += 64) {
		int v;
		if (zf >= 0 && c >= zf) v = 0;
		else if (c < up) v = (int)(src[c] >> 1);
		else {
			// the chunk that holds column c: the last r with cc[r] <= c (boundaries ascend, -1 after the last one)
			int lo = 0, hi = KC - 1;
			while (lo < hi) { const int mid = (lo + hi + 1) >> 1; const int b = cc[mid]; if (b >= 0 && b <= c) lo = mid; else hi = mid - 1; }
			const int r = lo;
			v = chunk_rows[(size_t)(chunk_base[k] + src_chunk[(size_t)k * KC + r]) * row_stride + c];
		}
		dst[c] = (uint8_t)v;
	}
}

hipError_t launch_hazard_merge(const uint16_t* colmax16, uint8_t* colmax, const int32_t* unit_ids, int32_t nlist, const int32_t* unit_len,
	const int32_t* chunk_cols, const int32_t* chunk_base, const int32_t* src_chunk, const int32_t* zero_from, const uint8_t* chunk_rows, int32_t row_stride, int32_t tstride, hipStream_t st)
{
	if (nlist <= 0) return hipSuccess;
	hipLaunchKernelGGL(k_hazard_merge, dim3((unsigned)nlist), dim3(64), 0, st, colmax16, colmax, unit_ids, unit_len, chunk_cols, chunk_base, src_chunk, zero_from,
		chunk_rows, row_stride, tstride);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// k_hits: one wave per unit
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_hits(const uint8_t* __restrict__ colmax, const int32_t* __restrict__ unit_ids, const int32_t* __restrict__ unit_len,
	const int32_t* __restrict__ stage1, int32_t tstride, uint32_t* __restrict__ hits, uint32_t hits_cap,
	uint32_t* __restrict__ hits_total, int32_t* __restrict__ hit_off, int32_t* __restrict__ hit_cnt, int32_t* __restrict__ thr_out)
{
	const int unit = unit_ids ? unit_ids[blockIdx.x] : (int)blockIdx.x;
	const int lane = threadIdx.x;
	const int n = unit_len[unit];
	// minScore = (int)(calc_score_once(...) * 0.8)  -- double multiply, truncation (Fasim-LongTarget.cpp:413)
	const int thr = (int)((double)stage1[unit] * 0.8);
	const uint8_t* col = colmax + (int64_t)unit * tstride;
	int cnt = 0;
	for (int c0 = 0; c0 < n; c0 += 64) {
		const int c = c0 + lane;
		const bool hit = c < n && (int)col[c] > thr;
		cnt += __popcll(__ballot(hit));
	}
	uint32_t off = 0;
	if (lane == 0) off = atomicAdd(hits_total, (uint32_t)cnt);
	off = __shfl(off, 0, 64);
	if (lane == 0) { hit_off[unit] = (int32_t)off; hit_cnt[unit] = cnt; thr_out[unit] = thr; }
	if ((uint64_t)off + (uint64_t)cnt > hits_cap) return;      // host sees hits_total > cap and retries
	uint32_t w = off;
	for (int c0 = 0; c0 < n; c0 += 64) {
		const int c = c0 + lane;
		const bool hit = c < n && (int)col[c] > thr;
		const unsigned long long b = __ballot(hit);
		if (hit) {
			const int before = __popcll(b & ((1ull << lane) - 1ull));
			hits[w + before] = ((uint32_t)c << 8) | (uint32_t)col[c];
		}
		w += __popcll(b);
	}
}

hipError_t launch_hits(const uint8_t* colmax, const int32_t* unit_ids, const int32_t* unit_len, const int32_t* stage1, int32_t nunit,
	int32_t tstride, uint32_t* hits, uint32_t hits_cap, uint32_t* hits_total, int32_t* hit_off, int32_t* hit_cnt,
	int32_t* thr_out, hipStream_t st)
{
	if (nunit <= 0) return hipSuccess;
	hipError_t err = hipMemsetAsync(hits_total, 0, sizeof(uint32_t), st);
	if (err != hipSuccess) return err;
	hipLaunchKernelGGL(k_hits, dim3(nunit), dim3(64), 0, st, colmax, unit_ids, unit_len, stage1, tstride, hits, hits_cap, hits_total,
		hit_off, hit_cnt, thr_out);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// k_banded: banded_sw (sswNew.cpp:1071-1259), one thread per alignment, state in a private slice of
// global scratch: [h_b | e_b | h_c : 3 * wmax int32][direction bytes ...]
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int sw_score(int a, int b) { return (a == b && a < 4) ? 5 : -4; }
__device__ __forceinline__ uint32_t cigar_pack(int len, int op) { return ((uint32_t)len << 4) | (uint32_t)op; }   // op 0=M 1=I 2=D

__global__ void __launch_bounds__(64) k_banded(const uint8_t* __restrict__ tcodes, const uint8_t* __restrict__ qcodes,
	const BandProb* __restrict__ probs, int32_t nprob, uint8_t* __restrict__ scratch, BandOut* __restrict__ out)
{
	const int pi = blockIdx.x * blockDim.x + threadIdx.x;
	if (pi >= nprob) return;
	const BandProb pb = probs[pi];
	const uint8_t* ref = tcodes + pb.tbase;
	const uint8_t* read = qcodes + pb.q_begin;
	const int refLen = pb.ref_len, readLen = pb.read_len, score = pb.score;
	uint8_t* my = scratch + pb.scratch_off;
	const int cap = pb.scratch_cap;
	// array slots sized for the widest band this scratch slice can hold
	int wmax = cap / 64;
	if (wmax > 4099) wmax = 4099;
	if (wmax < 8) wmax = 8;
	int32_t* h_b = reinterpret_cast<int32_t*>(my);
	int32_t* e_b = h_b + wmax;
	int32_t* h_c = e_b + wmax;
	int8_t* direction = reinterpret_cast<int8_t*>(h_c + wmax);
	const long dir_cap = (long)cap - (long)3 * wmax * 4;
	BandOut* o = out + pi;

	int band = (refLen > readLen ? refLen - readLen : readLen - refLen) + 1;
	int maxv = 0, width = 0, width_d = 0;
	do {
		width = band * 2 + 3; width_d = band * 2 + 1;
		if (width + 1 > wmax || (long)width_d * readLen * 3 + 3 > dir_cap) { o->status = 2; o->cigar_len = 0; return; }
		for (int j = 1; j < width - 1; j++) h_b[j] = 0;
		for (int i = 0; i < readLen; i++) {
			int beg = 0, end = refLen - 1, u = 0;
			if (i - band > beg) beg = i - band;
			if (i + band < end) end = i + band;
			const int edge = end + 1 < width - 1 ? end + 1 : width - 1;
			int f = 0;
			h_b[0] = 0; e_b[0] = 0; h_b[edge] = 0; e_b[edge] = 0; h_c[0] = 0;
			int8_t* line = direction + (long)width_d * i * 3;
			const int x = i - band > 0 ? i - band : 0;
			const int xp = i - 1 - band > 0 ? i - 1 - band : 0;
			const int rd = read[i];
			for (int j = beg; j <= end; j++) {
				u = j - x + 1;
				const int e = j - xp + 1, b = j - x, d = j - xp;
				int8_t* cell = line + (j - x) * 3;
				int t1 = i == 0 ? -GAP_OPEN : h_b[e] - GAP_OPEN;
				int t2 = i == 0 ? -GAP_EXT : e_b[e] - GAP_EXT;
				const int ev = t1 > t2 ? t1 : t2;
				e_b[u] = ev;
				const int8_t de = t1 > t2 ? 3 : 2;
				cell[0] = de;
				t1 = h_c[b] - GAP_OPEN;
				t2 = f - GAP_EXT;
				f = t1 > t2 ? t1 : t2;
				const int8_t df = t1 > t2 ? 5 : 4;
				cell[1] = df;
				const int e1 = ev > 0 ? ev : 0;
				const int f1 = f > 0 ? f : 0;
				t1 = e1 > f1 ? e1 : f1;
				t2 = h_b[d] + sw_score(ref[j], rd);
				const int hv = t1 > t2 ? t1 : t2;
				h_c[u] = hv;
				if (hv > maxv) maxv = hv;
				cell[2] = (t1 <= t2) ? (int8_t)1 : (e1 > f1 ? de : df);
			}
			for (int j = 1; j <= u; j++) h_b[j] = h_c[j];
		}
		band *= 2;
		// the reference would loop forever here; we stop and report undefined behaviour
		if (maxv < score && band > 4 * (refLen + readLen) + 16) { o->status = 3; o->cigar_len = 0; return; }
	} while (maxv < score);
	band /= 2;

	// trace back; the cigar is produced end-to-start, so fill the output from the back
	uint32_t rc[MAX_CIGAR_DEV];
	int l = 0;
	int i = readLen - 1, j = refLen - 1, e = 0, state = 2;
	int op = 0, prev_op = 0;     // 0=M 1=I 2=D
	int status = 0;
	while (i > 0) {
		const int x = i - band > 0 ? i - band : 0;
		int beg = 0, end = refLen - 1;
		if (i - band > beg) beg = i - band;
		if (i + band < end) end = i + band;
		if (j < beg || j > end) { status = 3; break; }   // the reference reads memory it never wrote
		const int8_t dv = direction[(long)width_d * i * 3 + (j - x) * 3 + state];
		if (dv == 1) { --i; --j; state = 2; op = 0; }
		else if (dv == 2) { --i; state = 0; op = 1; }
		else if (dv == 3) { --i; state = 2; op = 1; }
		else if (dv == 4) { --j; state = 1; op = 2; }
		else if (dv == 5) { --j; state = 2; op = 2; }
		else { status = 1; break; }
		if (op == prev_op) ++e;
		else {
			if (l >= MAX_CIGAR_DEV) { status = 2; break; }
			rc[l++] = cigar_pack(e, prev_op);
			prev_op = op; e = 1;
		}
	}
	if (status == 0) {
		if (op == 0) {
			if (l >= MAX_CIGAR_DEV) status = 2; else rc[l++] = cigar_pack(e + 1, 0);
		} else {
			if (l + 2 > MAX_CIGAR_DEV) status = 2; else { rc[l++] = cigar_pack(e, op); rc[l++] = cigar_pack(1, 0); }
		}
	}
	o->status = status;
	if (status != 0) { o->cigar_len = 0; return; }
	o->cigar_len = l;
	for (int k = 0; k < l; k++) o->cigar[k] = rc[l - 1 - k];
}

hipError_t launch_banded(const uint8_t* tcodes, const uint8_t* qcodes, const BandProb* probs, int32_t nprob,
	uint8_t* scratch, BandOut* out, hipStream_t st)
{
	if (nprob <= 0) return hipSuccess;
	hipLaunchKernelGGL(k_banded, dim3((nprob + 63) / 64), dim3(64), 0, st, tcodes, qcodes, probs, nprob, scratch, out);
	return hipGetLastError();
}

} // namespace fasim
