// oracle/fasim_oracle.cpp -- TEST INFRASTRUCTURE, not product code.  See fasim_oracle.h.
//
// Scalar, stripe-faithful CPU restatement of the Fasim-LongTarget hot path.  It is written
// from the algorithm's description and is kept deliberately
// simple: small fixed-width lane arrays stand in for the 128-bit SSE registers so that the
// layout-dependent quirks of the reference (Q1 byte-overflow break, Q2 signed lazy-F exit,
// Q3 zero-score pad rows) come out identically by construction.
//
// Parity: PINNED against the compiled reference (oracle/_ref) -- see tests/test_oracle_*.py.

#include "fasim_oracle.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <thread>
#include <atomic>

namespace fo {

// =====================================================================================
// a3 -- rule encodings.  Table rows are the outputs for DNA letters A,T,G,C (N -> N, any other
// letter -> N), read off rules.h:6-53 (characters 5..8 of each 10-character rule string).
// Order = canonical execution order of LongTarget() (Fasim-LongTarget.cpp:404-585).
// =====================================================================================
static const char* const RULE_OUT[48] = {
	// parallel rules 1..6: (strand 0: PARAj) , (strand 1: PARAjREV, reversed)
	"TGGT", "GTTG", "TGCT", "GTTC", "TGTT", "GTTT", "TGGC", "GTCG", "TGCC", "GTCC", "TGTC", "GTCT",
	// antiparallel rules 1..18: (strand 1: ANTIj) , (strand 0: ANTIjREV, reversed)
	"GTTG", "TGGT", "GTTC", "TGCT", "GTTA", "TGAT", "GTCG", "TGGC", "GTCC", "TGCC", "GTCA", "TGAC",
	"GATG", "AGGT", "GATC", "AGCT", "GATA", "AGAT", "GACG", "AGGC", "GACC", "AGCC", "GACA", "AGAC",
	"GCTG", "CGGT", "GCTC", "CGCT", "GCTA", "CGAT", "GCCG", "CGGC", "GCCC", "CGCC", "GCCA", "CGAC",
};

EncInfo enc_info(int enc)
{
	EncInfo e;
	if (enc < 12) { e.para = 1; e.rule = enc / 2 + 1; e.strand = enc & 1; e.reversed = (enc & 1) != 0; }
	else { int k = enc - 12; e.para = -1; e.rule = k / 2 + 1; e.strand = (k & 1) ? 0 : 1; e.reversed = (k & 1) != 0; }
	return e;
}

static char complement_base(char c, bool& keep)
{
	keep = true;
	switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; case 'N': return 'N'; }
	keep = false; // rules.h:82-83: unknown letters are dropped by complement()
	return 0;
}

void encode_unit(const std::string& seg, int enc, std::string& target, std::string& src)
{
	const EncInfo e = enc_info(enc);
	const char* o = RULE_OUT[enc];
	target.resize(seg.size());
	for (size_t i = 0; i < seg.size(); i++) {
		char c = seg[i], t;
		if (c == 'A') t = o[0]; else if (c == 'T') t = o[1]; else if (c == 'G') t = o[2]; else if (c == 'C') t = o[3];
		else t = 'N';   // 'N' -> 'N' and any other letter -> 'N' (rules.h:286-312)
		target[i] = t;
	}
	if (e.reversed) std::reverse(target.begin(), target.end());
	// display strand (Fasim-LongTarget.cpp:415,429-431,500-501,521-522)
	bool need_comp = (e.strand == 1);
	src.clear();
	if (need_comp) { for (char c : seg) { bool keep; char t = complement_base(c, keep); if (keep) src.push_back(t); } }
	else src = seg;
	if (e.reversed) std::reverse(src.begin(), src.end());
}

// =====================================================================================
// a1/a2
// =====================================================================================
void cut_sequence(const std::string& dna, int cut, int overlap, std::vector<int>& starts)
{
	starts.clear();
	unsigned int pos = 0;   // fastsim.h:74
	while (pos < dna.size()) { starts.push_back((int)pos); pos += cut; pos -= overlap; }
}

bool same_seq(const std::string& seg)
{
	size_t cnt[6] = { 0, 0, 0, 0, 0, 0 };
	for (char c : seg) {
		switch (c) { case 'A': cnt[0]++; break; case 'C': cnt[1]++; break; case 'G': cnt[2]++; break;
		case 'T': cnt[3]++; break; case 'U': cnt[4]++; break; case 'N': cnt[5]++; break; default: break; }
	}
	for (int k = 0; k < 6; k++) if (cnt[k] == seg.size()) return true;
	return false;
}

// =====================================================================================
// a4 -- stage 1: exact maximum local-alignment score under the stage-1 scoring
// (stats.h:201-228: A,C,G,T identity +5 / -4, U == T, anything else is N scoring -1 against
// everything; gap 16 for the first residue, 4 per further residue: stats.h:947).  The reference
// computes it with Farrar's 8-bit kernel and re-runs in 16 bit on overflow (stats.h:948-951),
// i.e. it returns the true maximum, which a plain Gotoh recurrence gives directly.
// =====================================================================================
static inline int stage1_code(char c)
{
	switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
	case 'T': case 't': return 3; case 'U': case 'u': return 3; default: return 4; }
}

int stage1_max(const std::string& rna, const std::string& target)
{
	const int m = (int)rna.size(), n = (int)target.size();
	std::vector<int> q(m), H(m + 1, 0), E(m + 1, 0);
	for (int i = 0; i < m; i++) q[i] = stage1_code(rna[i]);
	int best = 0;
	for (int c = 0; c < n; c++) {
		const int t = stage1_code(target[c]);
		int diag = 0, F = 0;
		for (int i = 1; i <= m; i++) {
			const int qi = q[i - 1];
			const int s = (qi == 4 || t == 4) ? -1 : (qi == t ? 5 : -4);
			int h = diag + s;
			if (h < 0) h = 0;
			if (E[i] > h) h = E[i];
			if (F > h) h = F;
			diag = H[i];
			H[i] = h;
			if (h > best) best = h;
			const int open = h - 16;
			E[i] = std::max(E[i] - 4, open); if (E[i] < 0) E[i] = 0;
			F = std::max(F - 4, open); if (F < 0) F = 0;
		}
	}
	return best;
}

// =====================================================================================
// stages 2/3 -- the SSW kernels.  Base codes: A,a,U,u -> 0 ; C,c -> 1 ; G,g -> 2 ; T,t -> 3 ;
// everything else 4 (ssw_cpp.cpp:13-26).  Matrix +5 on the ACGT diagonal, -4 elsewhere
// (ssw_cpp.cpp:28-53, 238-250).  gapO 16, gapE 4.
// =====================================================================================
static inline int8_t ssw_code(char c)
{
	switch (c) { case 'A': case 'a': case 'U': case 'u': return 0; case 'C': case 'c': return 1;
	case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
static inline int ssw_score(int a, int b) { return (a == b && a < 4) ? 5 : -4; }
static const int GAPO = 16, GAPE = 4, BIAS = 4;

struct EndInfo { int score, ref, read; };

// test hook: when false the lazy-F exit uses the unsigned comparison (what the reference would compute without Q2)
static thread_local bool g_signed_lazy_f = true;

static inline uint8_t adds8(uint8_t a, uint8_t b) { int v = a + b; return (uint8_t)(v > 255 ? 255 : v); }
static inline uint8_t subs8(uint8_t a, uint8_t b) { return (uint8_t)(a > b ? a - b : 0); }

// 8-bit striped kernel: 16 byte lanes, stripe s covers rows [s*segLen, (s+1)*segLen).
//   once == true : sw_sse2_byte_once (sswNew.cpp:255-464) -> fills maxcol (size refLen, zeros after a Q1 break)
//   once == false: sw_sse2_byte      (sswNew.cpp:476-672) -> EndInfo
static EndInfo sw_byte(const int8_t* ref, int dir, int refLen, const int8_t* read, int readLen,
	int terminate, std::vector<int>* maxcol)
{
	const int P = 16;
	const int segLen = (readLen + P - 1) / P;
	// query profile, pads score 0 i.e. store the bias (sswNew.cpp:195)
	std::vector<uint8_t> prof((size_t)5 * segLen * P);
	for (int t = 0; t < 5; t++)
		for (int j = 0; j < segLen; j++)
			for (int s = 0; s < P; s++) {
				const int row = j + s * segLen;
				prof[((size_t)t * segLen + j) * P + s] = (uint8_t)(row >= readLen ? BIAS : ssw_score(t, read[row]) + BIAS);
			}
	std::vector<uint8_t> Hs((size_t)segLen * P, 0), Hl((size_t)segLen * P, 0), E((size_t)segLen * P, 0), Hmax((size_t)segLen * P, 0);
	uint8_t* pvHStore = Hs.data(); uint8_t* pvHLoad = Hl.data();
	if (maxcol) maxcol->assign(refLen, 0);
	int max = 0, end_read = readLen - 1, end_ref = -1;
	int begin = 0, end = refLen, step = 1;
	if (dir == 1) { begin = refLen - 1; end = -1; step = -1; }
	for (int i = begin; i != end; i += step) {
		uint8_t vF[P], vH[P], vMaxColumn[P];
		for (int s = 0; s < P; s++) { vF[s] = 0; vMaxColumn[s] = 0; }
		// vH = last stored vector shifted up by one lane
		vH[0] = 0;
		for (int s = 1; s < P; s++) vH[s] = pvHStore[(size_t)(segLen - 1) * P + s - 1];
		const uint8_t* vP = &prof[(size_t)ref[i] * segLen * P];
		std::swap(pvHLoad, pvHStore);
		for (int j = 0; j < segLen; j++) {
			uint8_t* hs = pvHStore + (size_t)j * P; const uint8_t* hl = pvHLoad + (size_t)j * P;
			uint8_t* e = E.data() + (size_t)j * P; const uint8_t* p = vP + (size_t)j * P;
			for (int s = 0; s < P; s++) {
				uint8_t h = subs8(adds8(vH[s], p[s]), BIAS);
				uint8_t ev = e[s];
				if (ev > h) h = ev;
				if (vF[s] > h) h = vF[s];
				if (h > vMaxColumn[s]) vMaxColumn[s] = h;
				hs[s] = h;
				h = subs8(h, GAPO);
				ev = subs8(ev, GAPE);
				if (h > ev) ev = h;
				e[s] = ev;
				uint8_t f = subs8(vF[s], GAPE);
				vF[s] = f > h ? f : h;
				vH[s] = hl[s];
			}
		}
		// lazy-F loop with the reference's *signed* exit test (sswNew.cpp:360-371, 581-592): Q2
		for (int k = 0; k < P; k++) {
			for (int s = P - 1; s > 0; s--) vF[s] = vF[s - 1];
			vF[0] = 0;
			for (int j = 0; j < segLen; j++) {
				uint8_t* hs = pvHStore + (size_t)j * P;
				bool any = false;
				for (int s = 0; s < P; s++) {
					uint8_t h = hs[s];
					if (vF[s] > h) h = vF[s];
					if (h > vMaxColumn[s]) vMaxColumn[s] = h;
					hs[s] = h;
					h = subs8(h, GAPO);
					vF[s] = subs8(vF[s], GAPE);
					if (g_signed_lazy_f ? ((int8_t)vF[s] > (int8_t)h) : (vF[s] > h)) any = true;
				}
				if (!any) goto lazy_done;
			}
		}
	lazy_done:;
		int colmax = 0;
		for (int s = 0; s < P; s++) if (vMaxColumn[s] > colmax) colmax = vMaxColumn[s];
		if (colmax > max) {
			max = colmax;
			if (max + BIAS >= 255) break;      // Q1: leaves this column (and all later ones) unrecorded
			end_ref = i;
			memcpy(Hmax.data(), pvHStore, (size_t)segLen * P);
		}
		if (maxcol) (*maxcol)[i] = colmax;
		if (colmax == terminate) break;
	}
	for (int idx = 0; idx < segLen * P; idx++) {
		if (Hmax[idx] == max) {
			const int row = idx / P + (idx % P) * segLen;
			if (row < end_read) end_read = row;
		}
	}
	EndInfo r;
	r.score = (max + BIAS >= 255) ? 255 : max;
	r.ref = end_ref; r.read = end_read;
	return r;
}

static inline int16_t adds16(int16_t a, int16_t b) { int v = (int)a + (int)b; return (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v)); }
static inline int16_t subsu16(int16_t a, int16_t b) { uint16_t x = (uint16_t)a, y = (uint16_t)b; return (int16_t)(uint16_t)(x > y ? x - y : 0); }

// 16-bit striped kernel: 8 word lanes (sswNew.cpp:893-1069)
static EndInfo sw_word(const int8_t* ref, int dir, int refLen, const int8_t* read, int readLen, int terminate)
{
	const int P = 8;
	const int segLen = (readLen + P - 1) / P;
	std::vector<int16_t> prof((size_t)5 * segLen * P);
	for (int t = 0; t < 5; t++)
		for (int j = 0; j < segLen; j++)
			for (int s = 0; s < P; s++) {
				const int row = j + s * segLen;
				prof[((size_t)t * segLen + j) * P + s] = (int16_t)(row >= readLen ? 0 : ssw_score(t, read[row]));
			}
	std::vector<int16_t> Hs((size_t)segLen * P, 0), Hl((size_t)segLen * P, 0), E((size_t)segLen * P, 0), Hmax((size_t)segLen * P, 0);
	int16_t* pvHStore = Hs.data(); int16_t* pvHLoad = Hl.data();
	int max = 0, end_read = readLen - 1, end_ref = 0;
	int begin = 0, end = refLen, step = 1;
	if (dir == 1) { begin = refLen - 1; end = -1; step = -1; }
	for (int i = begin; i != end; i += step) {
		int16_t vF[P], vH[P], vMaxColumn[P];
		for (int s = 0; s < P; s++) { vF[s] = 0; vMaxColumn[s] = 0; }
		vH[0] = 0;
		for (int s = 1; s < P; s++) vH[s] = pvHStore[(size_t)(segLen - 1) * P + s - 1];
		const int16_t* vP = &prof[(size_t)ref[i] * segLen * P];
		std::swap(pvHLoad, pvHStore);
		for (int j = 0; j < segLen; j++) {
			int16_t* hs = pvHStore + (size_t)j * P; const int16_t* hl = pvHLoad + (size_t)j * P;
			int16_t* e = E.data() + (size_t)j * P; const int16_t* p = vP + (size_t)j * P;
			for (int s = 0; s < P; s++) {
				int16_t h = adds16(vH[s], p[s]);
				int16_t ev = e[s];
				if (ev > h) h = ev;
				if (vF[s] > h) h = vF[s];
				if (h > vMaxColumn[s]) vMaxColumn[s] = h;
				hs[s] = h;
				h = subsu16(h, GAPO);
				ev = subsu16(ev, GAPE);
				if (h > ev) ev = h;
				e[s] = ev;
				int16_t f = subsu16(vF[s], GAPE);
				vF[s] = f > h ? f : h;
				vH[s] = hl[s];
			}
		}
		for (int k = 0; k < P; k++) {
			for (int s = P - 1; s > 0; s--) vF[s] = vF[s - 1];
			vF[0] = 0;
			for (int j = 0; j < segLen; j++) {
				int16_t* hs = pvHStore + (size_t)j * P;
				bool any = false;
				for (int s = 0; s < P; s++) {
					int16_t h = hs[s];
					if (vF[s] > h) h = vF[s];
					if (h > vMaxColumn[s]) vMaxColumn[s] = h;
					hs[s] = h;
					h = subsu16(h, GAPO);
					vF[s] = subsu16(vF[s], GAPE);
					if (vF[s] > h) any = true;
				}
				if (!any) goto lazy_done;
			}
		}
	lazy_done:;
		int colmax = 0;
		for (int s = 0; s < P; s++) if (vMaxColumn[s] > colmax) colmax = vMaxColumn[s];
		if (colmax > max) {
			max = colmax;
			end_ref = i;
			memcpy(Hmax.data(), pvHStore, (size_t)segLen * P * sizeof(int16_t));
		}
		if (colmax == terminate) break;
	}
	for (int idx = 0; idx < segLen * P; idx++) {
		if (Hmax[idx] == max) {
			const int row = idx / P + (idx % P) * segLen;
			if (row < end_read) end_read = row;
		}
	}
	EndInfo r; r.score = max; r.ref = end_ref; r.read = end_read;
	return r;
}

std::vector<int> pre_align(const std::string& rna, const std::string& target)
{
	std::vector<int8_t> q(rna.size()), t(target.size());
	for (size_t i = 0; i < rna.size(); i++) q[i] = ssw_code(rna[i]);
	for (size_t i = 0; i < target.size(); i++) t[i] = ssw_code(target[i]);
	std::vector<int> col;
	// the 16-bit re-run of ssw_pre_align is gated on a recorded column >= 255, which the 8-bit
	// kernel can never produce (sswNew.cpp:386 breaks first) -- so only the byte kernel runs.
	sw_byte(t.data(), 0, (int)t.size(), q.data(), (int)q.size(), 255, &col);
	return col;
}

// ---- a7 ------------------------------------------------------------------------------------
std::vector<Cand> pick_candidates(const std::vector<int>& cols, int threshold)
{
	std::vector<Cand> hits, out;
	for (int i = 0; i < (int)cols.size(); i++) if (cols[i] > threshold) hits.push_back({ cols[i], i });
	const int nh = (int)hits.size();
	int num = 0;
	while (num < nh) {
		if (num == nh - 1) { out.push_back(hits[num]); break; }
		int gap = hits[num + 1].pos - hits[num].pos;
		if (gap < 5 && gap > 0) {
			// a chain of hits whose consecutive gaps are < 5 -> its first maximum
			int start = num, best = start;
			while (num + 1 <= nh - 1) {
				gap = hits[num + 1].pos - hits[num].pos;
				if (!(gap < 5 && gap > 0)) break;
				num++;
			}
			for (int k = start; k <= num; k++) if (hits[k].score > hits[best].score) best = k;
			num++;
			out.push_back(hits[best]);
		} else {
			out.push_back(hits[num]);
			num++;
		}
	}
	return out;
}

// ---- a10: banded traceback (sswNew.cpp:1071-1259) ----------------------------------------------
static inline uint32_t cigar_int(uint32_t len, char op) { return (len << 4) | (op == 'M' ? 0u : op == 'I' ? 1u : 2u); }

// Returns false where the reference returns NULL ("Trace back error") or where it would read
// memory it never wrote in a way that cannot be reproduced (then `tainted` is set).
static bool banded_cigar(const int8_t* ref, const int8_t* read, int refLen, int readLen, int score,
	int band_width, std::vector<uint32_t>& cigar, bool& tainted)
{
	std::vector<int32_t> h_b(16, 0), e_b(16, 0), h_c(16, 0);
	std::vector<int8_t> direction;       // flat, persists across band passes like the realloc'd buffer
	std::vector<uint8_t> stamp;          // pass number that last wrote a byte (0 = never)
	int max = 0, width = 0, width_d = 0, pass = 0;
	do {
		pass++;
		width = band_width * 2 + 3; width_d = band_width * 2 + 1;
		if ((int)h_b.size() < width + 1) { h_b.resize(width + 1, 0); e_b.resize(width + 1, 0); h_c.resize(width + 1, 0); }
		const size_t need = (size_t)width_d * readLen * 3 + 3;
		if (direction.size() < need) { direction.resize(need, 0); stamp.resize(need, 0); }
		for (int j = 1; j < width - 1; j++) h_b[j] = 0;
		for (int i = 0; i < readLen; i++) {
			int beg = 0, end = refLen - 1, u = 0;
			if (i - band_width > beg) beg = i - band_width;
			if (i + band_width < end) end = i + band_width;
			const int edge = end + 1 < width - 1 ? end + 1 : width - 1;
			int f = 0;
			h_b[0] = e_b[0] = h_b[edge] = e_b[edge] = h_c[0] = 0;
			const size_t line = (size_t)width_d * i * 3;
			const int x = i - band_width > 0 ? i - band_width : 0;
			const int xp = i - 1 - band_width > 0 ? i - 1 - band_width : 0;
			for (int j = beg; j <= end; j++) {
				u = j - x + 1;
				const int e = j - xp + 1, b = j - x, d = j - xp;
				const size_t de = line + (size_t)(j - x) * 3, df = de + 1, dh = de + 2;
				int t1 = i == 0 ? -GAPO : h_b[e] - GAPO;
				int t2 = i == 0 ? -GAPE : e_b[e] - GAPE;
				e_b[u] = t1 > t2 ? t1 : t2;
				direction[de] = t1 > t2 ? 3 : 2; stamp[de] = (uint8_t)pass;
				t1 = h_c[b] - GAPO;
				t2 = f - GAPE;
				f = t1 > t2 ? t1 : t2;
				direction[df] = t1 > t2 ? 5 : 4; stamp[df] = (uint8_t)pass;
				const int e1 = e_b[u] > 0 ? e_b[u] : 0;
				const int f1 = f > 0 ? f : 0;
				t1 = e1 > f1 ? e1 : f1;
				t2 = h_b[d] + ssw_score(ref[j], read[i]);
				h_c[u] = t1 > t2 ? t1 : t2;
				if (h_c[u] > max) max = h_c[u];
				if (t1 <= t2) direction[dh] = 1;
				else direction[dh] = e1 > f1 ? direction[de] : direction[df];
				stamp[dh] = (uint8_t)pass;
			}
			for (int j = 1; j <= u; j++) h_b[j] = h_c[j];
		}
		band_width *= 2;
		if (max < score && band_width > 4 * (refLen + readLen) + 16) { tainted = true; return false; } // reference would never terminate
	} while (max < score);
	band_width /= 2;

	// trace back
	std::vector<uint32_t> c;
	int i = readLen - 1, j = refLen - 1, e = 0;
	char op = 'M', prev_op = 'M';
	int state = 2;
	long line = (long)width_d * (readLen - 1) * 3;
	while (i > 0) {
		const int x = i - band_width > 0 ? i - band_width : 0;
		const long idx = line + (long)(j - x) * 3 + state;
		if (idx < 0 || idx >= (long)direction.size()) { tainted = true; return false; }
		if (stamp[idx] != pass) tainted = true;   // the reference reads stale / uninitialised memory here
		switch (direction[idx]) {
		case 1: --i; --j; state = 2; line -= (long)width_d * 3; op = 'M'; break;
		case 2: --i; state = 0; line -= (long)width_d * 3; op = 'I'; break;
		case 3: --i; state = 2; line -= (long)width_d * 3; op = 'I'; break;
		case 4: --j; state = 1; op = 'D'; break;
		case 5: --j; state = 2; op = 'D'; break;
		default: return false;   // "Trace back error" -> NULL
		}
		if (op == prev_op) ++e;
		else { c.push_back(cigar_int(e, prev_op)); prev_op = op; e = 1; }
	}
	if (op == 'M') c.push_back(cigar_int(e + 1, op));
	else { c.push_back(cigar_int(e, op)); c.push_back(cigar_int(1, 'M')); }
	cigar.assign(c.rbegin(), c.rend());
	return true;
}

std::string Alignment::cigar_string() const
{
	std::ostringstream os;
	for (uint32_t c : cigar) os << (c >> 4) << ((c & 0xf) > 8 ? 'M' : "MIDNSHP=X"[c & 0xf]);
	return os.str();
}

// ---- a9/a11: ssw_align (sswNew.cpp:1446-1547) + Aligner::Align (ssw_cpp.cpp:599-643) -----------
static Alignment align_codes(const int8_t* read, int readLen, const int8_t* ref, int refLen)
{
	Alignment al;
	bool word = false;
	EndInfo best = sw_byte(ref, 0, refLen, read, readLen, 255, nullptr);
	if (best.score == 255) { best = sw_word(ref, 0, refLen, read, readLen, 65535); word = true; }
	if (best.score == 0) { al.sw_score = 0; return al; }   // nothing aligned; callers drop score 0
	const int read_end = best.read, ref_end = best.ref;
	std::vector<int8_t> rr(read_end + 1);
	for (int k = 0; k <= read_end; k++) rr[k] = read[read_end - k];
	EndInfo rev = word ? sw_word(ref, 1, ref_end + 1, rr.data(), read_end + 1, best.score)
	                   : sw_byte(ref, 1, ref_end + 1, rr.data(), read_end + 1, best.score, nullptr);
	const int score1 = rev.score < best.score ? rev.score : best.score;
	const int ref_begin = rev.ref, read_begin = read_end - rev.read;
	const int rl = ref_end - ref_begin + 1, ql = read_end - read_begin + 1;
	if (ref_begin < 0 || rl <= 0 || ql <= 0) { al.tainted = true; al.sw_score = 0; return al; }
	const int band = std::abs(rl - ql) + 1;
	std::vector<uint32_t> cig;
	bool tainted = false;
	if (!banded_cigar(ref + ref_begin, read + read_begin, rl, ql, score1, band, cig, tainted)) {
		al.sw_score = 0; al.tainted = tainted;    // NULL -> sw_score 0 (ssw_cpp.cpp:631-633)
		return al;
	}
	al.sw_score = score1; al.ref_begin = ref_begin; al.ref_end = ref_end;
	al.query_begin = read_begin; al.query_end = read_end; al.cigar = cig; al.tainted = tainted;
	return al;
}

Alignment align(const std::string& rna, const std::string& window)
{
	std::vector<int8_t> q(rna.size()), t(window.size());
	for (size_t i = 0; i < rna.size(); i++) q[i] = ssw_code(rna[i]);
	for (size_t i = 0; i < window.size(); i++) t[i] = ssw_code(window[i]);
	return align_codes(q.data(), (int)q.size(), t.data(), (int)t.size());
}

// ---- a12/a13: triplex construction (fastsim.h:291-414, 416-560; sim.h:72-97) ---------------------
float triplex_score(char c1, char c2, int para)
{
	if (para > 0) {
		if (c1 == 'A' && c2 == 'T') return 3.7; else if (c1 == 'T' && c2 == 'G') return 2.8;
		else if (c1 == 'G' && c2 == 'G') return 2.2; else if (c1 == 'G' && c2 == 'T') return 2.4;
		else if (c1 == 'G' && c2 == 'C') return 4.5; else if (c1 == 'C' && c2 == 'T') return 2.6;
		else if (c1 == 'C' && c2 == 'C') return 2.4;
	} else {
		if (c1 == 'A' && c2 == 'A') return 3.0; else if (c1 == 'A' && c2 == 'T') return 3.5;
		else if (c1 == 'A' && c2 == 'C') return 1.0; else if (c1 == 'T' && c2 == 'G') return 1.0;
		else if (c1 == 'G' && c2 == 'A') return 1.0; else if (c1 == 'G' && c2 == 'G') return 3.0;
		else if (c1 == 'G' && c2 == 'C') return 3.0; else if (c1 == 'C' && c2 == 'T') return 2.0;
		else if (c1 == 'C' && c2 == 'C') return 1.0;
	}
	return 0;
}

static void convert_triplex(const Alignment& al, std::vector<Triplex>& list, const std::string& rna,
	const std::string& target, const std::string& src, long dna_start, int rule, int strand, int para,
	int penaltyT, int penaltyC, int ntMin, int ntMax)
{
	std::string ref_align, read_align, src_align;
	int q = al.ref_begin, p = al.query_begin;
	for (uint32_t c : al.cigar) {
		const uint32_t len = c >> 4; const uint32_t op = c & 0xf;
		for (uint32_t k = 0; k < len; k++) {
			if (op == 1) { ref_align += '-'; src_align += '-'; read_align += rna[p++]; }
			else if (op == 2) { ref_align += target[q]; src_align += (q < (int)src.size() ? src[q] : '\0'); q++; read_align += '-'; }
			else { ref_align += target[q]; src_align += (q < (int)src.size() ? src[q] : '\0'); q++; read_align += rna[p++]; }
		}
	}
	const int nt = (int)ref_align.length();
	int match = 0, mis = 0;
	for (int i = 0; i < nt; i++) { if (ref_align[i] == read_align[i]) match++; else mis++; }
	const float identity = (float)(100 * match) / (float)(match + mis);
	float tri_score = 0.0f, hashvalue = 0.0f, prescore = 0.0f;
	char prechar = 0, curchar = 0;
	if (nt >= ntMin && nt <= ntMax) {
		for (int i = 0; i < nt; i++) {
			curchar = (ref_align[i] == '-') ? '-' : src_align[i];
			hashvalue = triplex_score(curchar, read_align[i], para);
			if (curchar == prechar && curchar == 'T') { tri_score = tri_score - prescore + penaltyT; hashvalue = penaltyT; }
			if (curchar == prechar && curchar == 'C') { tri_score = tri_score - prescore + penaltyC; hashvalue = penaltyC; }
			prescore = hashvalue;
			if (ref_align[i] != '-') prechar = curchar;
			tri_score += hashvalue;
		}
		tri_score = tri_score / nt;
	}
	const float score = (float)al.sw_score;
	int refStart, refEnd;
	const int n = (int)target.size();
	if ((para > 0 && strand == 1) || (para < 0 && strand == 0)) { refStart = n - al.ref_end - 1; refEnd = n - al.ref_begin - 1; }
	else { refStart = al.ref_begin + 1; refEnd = al.ref_end + 1; }
	if (nt >= ntMin) {
		Triplex t;
		t.stari = al.query_begin + 1; t.endi = al.query_end + 1;
		t.starj = (int)(refStart + dna_start); t.endj = (int)(refEnd + dna_start);
		t.strand = strand; t.reverse = para; t.rule = rule; t.nt = nt;
		t.score = score; t.identity = identity; t.tri_score = tri_score;
		t.stri_align = read_align; t.strj_align = src_align;
		list.push_back(t);
	}
}

// ---- a14: comparators (fastsim.h:92-156) --------------------------------------------------------
static bool cmp_multiple(const Triplex& a, const Triplex& b)
{
	if (a.stari == b.stari) { if (a.starj == b.starj) return a.score > b.score; else return a.starj > b.starj; }
	else return a.starj > b.starj;
}
static bool cmp_multiple2(const Triplex& a, const Triplex& b)
{
	if (a.endi == b.endi) { if (a.starj == b.starj) return a.score > b.score; else return a.starj < b.starj; }
	else return a.starj < b.starj;
}
static bool cmp_single(const Triplex& a, const Triplex& b) { return a.score > b.score; }
static bool same_triplex(const Triplex& a, const Triplex& b)
{
	if (a.stari == b.stari && a.starj == b.starj && a.endi == b.endi && a.endj == b.endj && a.score == b.score) return true;
	else if (b.stari >= a.stari && b.starj >= a.starj && b.endi <= a.endi && b.endj <= a.endj && b.score < a.score) return true;
	return false;
}

// ---- a8 + a14: fastSIM (fastsim.h:158-289) -------------------------------------------------------
void fast_sim_unit(const std::string& rna, const std::string& target, const std::string& src,
	long dna_start, int min_score, int strand, int para, int rule, const Params& p,
	std::vector<Triplex>& out, UnitTrace* trace)
{
	std::vector<int> cols = pre_align(rna, target);
	std::vector<Cand> cands = pick_candidates(cols, min_score);
	if (trace) {
		trace->colhash = fnv1a_ints(cols.data(), (int)cols.size());
		trace->nhits = 0;
		for (int i = 0; i < (int)cols.size(); i++) if (cols[i] > min_score) { trace->nhits++; trace->hits.push_back({ i, cols[i] }); }
		trace->cands = cands;
		trace->tries.resize(cands.size());
	}
	std::vector<Triplex> mine;
	std::vector<int8_t> q(rna.size()), t(target.size());
	for (size_t i = 0; i < rna.size(); i++) q[i] = ssw_code(rna[i]);
	for (size_t i = 0; i < target.size(); i++) t[i] = ssw_code(target[i]);
	for (size_t ci = 0; ci < cands.size(); ci++) {
		float Iden = 0.6;
		int cutlength = 0, bestcut = 0, flag = 0, it = 0;
		Alignment al, best;
		best.sw_score = 0;
		while (Iden <= 1) {
			cutlength = (int)(cands[ci].score + 24) / (9 * Iden - 4) + 1;
			cutlength = cands[ci].pos - cutlength + 1 > 0 ? cutlength : cands[ci].pos + 1;
			al = align_codes(q.data(), (int)q.size(), t.data() + (cands[ci].pos - cutlength + 1), cutlength);
			if (trace) trace->tries[ci].push_back({ it, cutlength, al });
			if (al.sw_score >= cands[ci].score) { flag = 1; break; }
			if (al.sw_score > best.sw_score && al.ref_end == cutlength - 1) { best = al; bestcut = cutlength; flag = 2; }
			Iden += 0.1;
			it++;
		}
		if (flag == 2) { al = best; cutlength = bestcut; }
		if (al.sw_score != 0) {
			al.ref_begin += cands[ci].pos - cutlength + 1;
			al.ref_end += cands[ci].pos - cutlength + 1;
			convert_triplex(al, mine, rna, target, src, dna_start, rule, strand, para, p.penaltyT, p.penaltyC, p.ntMin, p.ntMax);
		}
	}
	std::sort(mine.begin(), mine.end(), cmp_multiple);
	mine.erase(std::unique(mine.begin(), mine.end(), same_triplex), mine.end());
	std::sort(mine.begin(), mine.end(), cmp_multiple2);
	mine.erase(std::unique(mine.begin(), mine.end(), same_triplex), mine.end());
	std::sort(mine.begin(), mine.end(), cmp_single);
	const size_t lim = mine.size() > 50 ? 50 : mine.size();
	for (size_t i = 0; i < lim; i++) {
		const Triplex& a = mine[i];
		if (a.identity >= p.minIdentity && a.tri_score >= p.minStability && a.nt >= p.ntMin) {
			out.push_back(a);
			if (trace) trace->triplexes.push_back(a);
		}
	}
}

// ---- LongTarget() (Fasim-LongTarget.cpp:379-598) ------------------------------------------------
static void run_one_unit(const Params& p, const std::string& rna, const std::string& seg, int segi, int enc,
	long dna_start, std::vector<Triplex>& out, std::vector<UnitTrace>* traces)
{
	const EncInfo e = enc_info(enc);
	std::string target, src;
	encode_unit(seg, enc, target, src);
	const int s1 = stage1_max(rna, target);
	const int minscore = (int)(s1 * 0.8);
	UnitTrace tr; UnitTrace* trp = nullptr;
	if (traces) { tr.seg = segi; tr.enc = enc; tr.dna_start = dna_start; tr.n = (int)target.size(); tr.stage1 = s1; tr.thr = minscore; trp = &tr; }
	if (p.classicSim) {
		const size_t before = out.size();
		sim_unit(rna, target, src, dna_start, minscore, e.strand, e.para, e.rule, p, out);
		if (trp) trp->triplexes.assign(out.begin() + (long)before, out.end());
	} else fast_sim_unit(rna, target, src, dna_start, minscore, e.strand, e.para, e.rule, p, out, trp);
	if (traces) traces->push_back(std::move(tr));
}

static std::vector<int> enabled_encodings(const Params& p)
{
	std::vector<int> v;
	if (p.strand >= 0) {
		if (p.rule == 0) for (int e = 0; e < 12; e++) v.push_back(e);
		else if (p.rule > 0 && p.rule < 7) { v.push_back((p.rule - 1) * 2); v.push_back((p.rule - 1) * 2 + 1); }
	}
	if (p.strand <= 0) {
		if (p.rule == 0) for (int e = 12; e < 48; e++) v.push_back(e);
		else if (p.rule >= 1 && p.rule <= 18) { v.push_back(12 + (p.rule - 1) * 2); v.push_back(12 + (p.rule - 1) * 2 + 1); }
	}
	return v;
}

void long_target(const Params& p, const std::string& rna, const std::string& dna,
	std::vector<Triplex>& out, std::vector<UnitTrace>* traces, std::vector<int>* skipped,
	int seg_first, int seg_count, int threads)
{
	std::vector<int> starts;
	cut_sequence(dna, p.cutLength, p.overlapLength, starts);
	const std::vector<int> encs = enabled_encodings(p);
	const int nseg = (int)starts.size();
	const int lo = std::max(0, seg_first), hi = (int)std::min<long>(nseg, (long)seg_first + seg_count);
	// one work item per (segment, encoding) unit; results are put back into canonical order afterwards
	const int nenc = (int)encs.size();
	std::vector<std::vector<Triplex>> per_unit((size_t)nseg * (size_t)std::max(1, nenc));
	std::vector<std::vector<UnitTrace>> per_unit_tr(per_unit.size());
	std::vector<char> skip(nseg, 0);
	for (int s = lo; s < hi; s++) skip[s] = same_seq(dna.substr(starts[s], p.cutLength)) ? 1 : 0;
	std::atomic<long> next((long)lo * nenc);
	auto worker = [&]() {
		for (;;) {
			const long u = next.fetch_add(1);
			if (u >= (long)hi * nenc) break;
			const int s = (int)(u / nenc), k = (int)(u % nenc);
			if (skip[s]) continue;
			const std::string seg = dna.substr(starts[s], p.cutLength);
			run_one_unit(p, rna, seg, s, encs[k], starts[s], per_unit[(size_t)u], traces ? &per_unit_tr[(size_t)u] : nullptr);
		}
	};
	if (threads <= 1) worker();
	else { std::vector<std::thread> th; for (int k = 0; k < threads; k++) th.emplace_back(worker); for (auto& t : th) t.join(); }
	std::vector<Triplex> all;
	for (int s = lo; s < hi; s++) {
		if (skip[s] && skipped) skipped->push_back(s);
		for (int k = 0; k < nenc; k++) {
			const size_t u = (size_t)s * nenc + k;
			for (auto& t : per_unit[u]) all.push_back(t);
			if (traces) for (auto& t : per_unit_tr[u]) traces->push_back(std::move(t));
		}
	}
	for (const Triplex& a : all)
		if (a.score >= p.scoreMin && a.identity >= p.minIdentity && a.tri_score >= p.minStability && a.nt >= p.cLength) out.push_back(a);
}

// ---- a15/a16 -------------------------------------------------------------------------------------
void assign_genome(std::vector<Triplex>& list, const std::string& chr, long start_genome)
{
	for (Triplex& t : list) if (t.genomestart == 0) { t.chr = chr; t.genomestart = t.starj + start_genome - 1; t.genomeend = t.endj + start_genome - 1; }
}

struct Axis { int triplexnum = 0, neartriplex = 0; };

void cluster_triplex(int dd, int length, std::vector<Triplex>& list)
{
	// Fasim-LongTarget.cpp:600-691, same sequential semantics (std::map with default-inserting [])
	std::map<size_t, Axis> axis;
	int find = 0, max_near = 0, max_pos = 0;
	for (Triplex& t : list) {
		if (t.nt > length) {
			const int middle = (int)((t.stari + t.endi) / 2);
			t.middle = middle; t.motif = 0;
			axis[middle].triplexnum++;
			for (int i = -dd; i <= dd; i++) {
				if (i > 0) axis[middle + i].neartriplex += (dd - i);
				else if (i < 0) axis[middle + i].neartriplex += (dd + i);
				if (axis[middle].triplexnum > 0) {
					if (axis[middle + i].neartriplex > max_near) { max_near = axis[middle + i].neartriplex; max_pos = middle + i; find = 1; }
				}
			}
			t.neartriplex = axis[middle].neartriplex;
		}
	}
	int theclass = 1;
	while (find) {
		for (int i = max_pos - dd; i <= max_pos + dd; i++) {
			for (Triplex& t : list) if (t.middle == i && t.motif == 0) { t.motif = theclass; t.center = max_pos; }
			axis.erase(i);
		}
		max_near = 0; find = 0;
		for (int i = 0; (size_t)i < axis.size(); i++) {
			if (axis[i].neartriplex > max_near) { max_near = axis[i].neartriplex; max_pos = i; find = 1; }
		}
		++theclass;
	}
}

static const char* strand_name(int reverse, int strand)
{
	if (reverse == 1 && strand == 0) return "ParaPlus";
	if (reverse == 1 && strand == 1) return "ParaMinus";
	if (reverse == -1 && strand == 1) return "AntiMinus";
	if (reverse == -1 && strand == 0) return "AntiPlus";
	return "";
}
static bool cmp_motif(const Triplex& a, const Triplex& b) { return a.motif < b.motif; }

std::string tfosorted_text(std::vector<Triplex>& list, const Params& p)
{
	std::ostringstream o;
	o << "QueryStart\t" << "QueryEnd\t" << "StartInSeq\t" << "EndInSeq\t" << "Direction\t" << "Chr\t" << "StartInGenome\t"
	  << "EndInGenome\t" << "MeanStability\t" << "MeanIdentity(%)\t" << "Strand\t" << "Rule\t" << "Score\t" << "Nt(bp)\t"
	  << "Class\t" << "MidPoint\t" << "Center\t" << "TFO sequence\t" << "TTS sequence" << std::endl;
	cluster_triplex(p.cDistance, p.cLength, list);
	std::sort(list.begin(), list.end(), cmp_motif);
	for (const Triplex& a : list) {
		if (a.motif == 0) continue;
		o << a.stari << "\t" << a.endi << "\t" << a.starj << "\t" << a.endj << "\t" << (a.starj < a.endj ? "R\t" : "L\t") << a.chr << "\t"
		  << a.genomestart << "\t" << a.genomeend << "\t" << a.tri_score << "\t" << a.identity << "\t" << strand_name(a.reverse, a.strand)
		  << "\t" << a.rule << "\t" << a.score << "\t" << a.nt << "\t" << a.motif << "\t" << a.middle << "\t" << a.center << "\t"
		  << a.stri_align << "\t" << a.strj_align << std::endl;
	}
	return o.str();
}

// print_cluster() (Fasim-LongTarget.cpp:694-795) for one class level, as text.  `list` must already carry the class of
// every triplex (cluster_triplex / tfosorted_text ran).  The per-position coverage map of :661-672 is rebuilt here from
// the records of that class; the run-length walk below keeps the reference's quirks: the first run starts one position
// early (:751), the LAST covered position is always emitted as a run of its own (:732-736, :741), and a zero-signal
// line bridges every gap (:760-764).  The stdout chatter of :698 (an uninitialised buffer) is not reproduced.
std::string tfoclass_text(const std::vector<Triplex>& list, int level, const std::string& chr, long start_genome_m1,
	long dna_size, const std::string& rna_name, const Params& p)
{
	std::map<size_t, size_t> cover;
	for (const Triplex& t : list) {
		if (t.motif != level) continue;
		if (t.endj > t.starj) { for (int j = t.starj; j < t.endj; j++) cover[(size_t)j]++; }
		else { for (int j = t.endj; j < t.starj; j++) cover[(size_t)j]++; }
	}
	std::ostringstream o;
	const long sg = start_genome_m1;
	o << "browser position " << chr << ":" << sg << "-" << sg + dna_size << std::endl;
	o << "browser hide all" << std::endl;
	o << "browser pack refGene encodeRegions" << std::endl;
	o << "browser full altGraph" << std::endl;
	o << "# 300 base wide bar graph, ausoScale is on by default == graphing" << std::endl;
	o << "# limits will dynamically change to always show full range of data" << std::endl;
	o << "# in viewing window, priority = 20 position this as the second graph" << std::endl;
	o << "# Note, zero-relative, half-open coordinate system in use for bedGraph format" << std::endl;
	o << "track type=bedGraph name='" << rna_name << " TTS (" << level << ")' description='" << p.cDistance << "-" << p.cLength
	  << "' visibility=full color=200,100,0 altColor=0,100,200 priority=20" << std::endl;
	if (cover.empty()) return o.str();
	struct Row { int a, b, v; };
	std::vector<Row> rows;
	const int sgi = (int)sg;
	const int final_genome = (int)(cover.rbegin()->first + sg);
	bool first_run = true;
	for (auto it = cover.begin(); it != cover.end();) {
		const int first0 = (int)it->first;
		int last = (int)it->first, val = (int)it->second;
		if ((int)(it->first + sg) == final_genome) { rows.push_back({ first0 + sgi - 1, last + sgi, val }); break; }
		++it;
		while (std::labs((long)(it->first - (size_t)last)) == 1 && (int)it->second == val) {
			if ((int)(it->first + sg) == final_genome) break;
			last = (int)it->first; val = (int)it->second;
			++it;
		}
		rows.push_back({ first0 + sgi - (first_run ? 2 : 1), last + sgi, val });
		first_run = false;
		if (std::labs((long)(it->first - (size_t)last)) != 1) rows.push_back({ last + sgi, (int)it->first + sgi - 1, 0 });
	}
	for (const Row& r : rows) o << chr << "\t" << r.a << "\t" << r.b << "\t" << r.v << std::endl;
	return o.str();
}

uint64_t fnv1a_ints(const int* v, int n)
{
	uint64_t h = 1469598103934665603ULL;
	for (int i = 0; i < n; i++) {
		const uint32_t x = (uint32_t)v[i];
		for (int b = 0; b < 4; b++) { h ^= (x >> (8 * b)) & 0xff; h *= 1099511628211ULL; }
	}
	return h;
}

bool read_fasta(const char* path, std::string& header, std::string& seq)
{
	std::ifstream in(path);
	if (!in) return false;
	std::string line; header.clear(); seq.clear();
	bool first = true;
	while (std::getline(in, line)) {
		while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
		if (first && !line.empty() && line[0] == '>') { header = line.substr(1); first = false; continue; }
		first = false;
		if (!line.empty() && line[0] == '>') break;
		seq += line;
	}
	return true;
}

void parse_dna_header(const std::string& header, std::string& species, std::string& chr, long& start)
{
	// '>species|chr|start-end' (Fasim-LongTarget.cpp:226-255); start parsed with atoi
	species.clear(); chr.clear(); start = 0;
	std::string tmp, startstr; int j = 0;
	for (char c : header) {
		if (c == '|' && j == 0) { species = tmp; j++; tmp.clear(); continue; }
		if (c == '|' && j == 1) { chr = tmp; j++; tmp.clear(); continue; }
		if (c == '-' && j == 2) { startstr = tmp; tmp.clear(); continue; }
		tmp += c;
	}
	start = atoi(startstr.c_str());
}

} // namespace fo

extern "C" {
// column maxima with the lazy-F exit made unsigned (no Q2): used by tests to locate units where Q2 matters
void fo_pre_align_noq2(const char* rna, int m, const char* target, int n, int* out_cols)
{
	fo::g_signed_lazy_f = false;
	std::vector<int> c = fo::pre_align(std::string(rna, m), std::string(target, n));
	fo::g_signed_lazy_f = true;
	memcpy(out_cols, c.data(), sizeof(int) * n);
}
int fo_stage1_max(const char* rna, int m, const char* target, int n) { return fo::stage1_max(std::string(rna, m), std::string(target, n)); }
void fo_pre_align(const char* rna, int m, const char* target, int n, int* out_cols)
{
	std::vector<int> c = fo::pre_align(std::string(rna, m), std::string(target, n));
	memcpy(out_cols, c.data(), sizeof(int) * n);
}
int fo_pick_candidates(const int* cols, int n, int thr, int* out_score, int* out_pos, int cap)
{
	std::vector<int> v(cols, cols + n);
	std::vector<fo::Cand> c = fo::pick_candidates(v, thr);
	for (int i = 0; i < (int)c.size() && i < cap; i++) { out_score[i] = c[i].score; out_pos[i] = c[i].pos; }
	return (int)c.size();
}
int fo_align(const char* rna, int m, const char* window, int n, int* out5, uint32_t* cigar, int cap)
{
	fo::Alignment a = fo::align(std::string(rna, m), std::string(window, n));
	out5[0] = a.sw_score; out5[1] = a.ref_begin; out5[2] = a.ref_end; out5[3] = a.query_begin; out5[4] = a.query_end;
	if ((int)a.cigar.size() > cap) return -1;
	for (size_t i = 0; i < a.cigar.size(); i++) cigar[i] = a.cigar[i];
	return (int)a.cigar.size();
}
int fo_sim_forward_nodes(const char* rna, int m, const char* target, int n, long min_score, long* out, int cap)
{
	std::vector<fo::SimNode> nodes;
	fo::sim_forward_nodes(std::string(rna, m), std::string(target, n), min_score, nodes);
	for (int k = 0; k < (int)nodes.size() && k < cap; k++) {
		const fo::SimNode& v = nodes[k];
		const long f[9] = { v.score, v.stari, v.starj, v.endi, v.endj, v.top, v.bot, v.left, v.right };
		for (int x = 0; x < 9; x++) out[9 * k + x] = f[x];
	}
	return (int)nodes.size();
}
void fo_encode_unit(const char* seg, int n, int enc, char* target, char* src)
{
	std::string t, s;
	fo::encode_unit(std::string(seg, n), enc, t, s);
	memcpy(target, t.data(), t.size());
	memset(src, 0, n); memcpy(src, s.data(), s.size());
}
}
