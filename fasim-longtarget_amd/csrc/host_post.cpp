// Host-side half of the hot path; see host_post.h.  Product code: never includes anything from oracle/.
#include "host_post.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <sstream>
#include <thread>

namespace fasim {

// ---- a3: the 48 encodings in LongTarget()'s execution order --------------------------------------
// Outputs for the DNA letters A,T,G,C (characters 5..8 of the rule strings, rules.h:6-53); N and every
// other letter map to N (rules.h:286-312).  Even index = forward table, odd index = the "REV" table whose
// result is reversed (Fasim-LongTarget.cpp:410-431, 499-522).
static const char* const kRuleOut[48] = {
	"TGGT", "GTTG", "TGCT", "GTTC", "TGTT", "GTTT", "TGGC", "GTCG", "TGCC", "GTCC", "TGTC", "GTCT",
	"GTTG", "TGGT", "GTTC", "TGCT", "GTTA", "TGAT", "GTCG", "TGGC", "GTCC", "TGCC", "GTCA", "TGAC",
	"GATG", "AGGT", "GATC", "AGCT", "GATA", "AGAT", "GACG", "AGGC", "GACC", "AGCC", "GACA", "AGAC",
	"GCTG", "CGGT", "GCTC", "CGCT", "GCTA", "CGAT", "GCCG", "CGGC", "GCCC", "CGCC", "GCCA", "CGAC",
};

const char* rule_out(int enc) { return kRuleOut[enc]; }

EncInfo enc_info(int enc)
{
	EncInfo e;
	e.reversed = (enc & 1) != 0;
	if (enc < 12) { e.para = 1; e.rule = enc / 2 + 1; e.strand = enc & 1; }
	else { e.para = -1; e.rule = (enc - 12) / 2 + 1; e.strand = ((enc - 12) & 1) ? 0 : 1; }
	return e;
}

static inline char map_base(const char* o, char c)
{
	switch (c) { case 'A': return o[0]; case 'T': return o[1]; case 'G': return o[2]; case 'C': return o[3]; default: return 'N'; }
}
static inline uint8_t letter_code(char c)
{
	switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; }
}

void build_enc_lut(uint8_t* lut)
{
	for (int enc = 0; enc < 48; enc++)
		for (int b = 0; b < 256; b++) lut[enc * 256 + b] = letter_code(map_base(kRuleOut[enc], (char)b));
}

std::vector<int> enabled_encodings(const fasim_params& p)
{
	// Fasim-LongTarget.cpp:404-585: -t >= 0 runs the parallel block (rules 1..6 only), -t <= 0 the antiparallel one
	std::vector<int> v;
	if (p.strand >= 0) {
		if (p.rule == 0) for (int e = 0; e < 12; e++) v.push_back(e);
		else if (p.rule > 0 && p.rule < 7) { v.push_back(2 * (p.rule - 1)); v.push_back(2 * (p.rule - 1) + 1); }
	}
	if (p.strand <= 0) {
		if (p.rule == 0) for (int e = 12; e < 48; e++) v.push_back(e);
		else if (p.rule >= 1 && p.rule <= 18) { v.push_back(12 + 2 * (p.rule - 1)); v.push_back(12 + 2 * (p.rule - 1) + 1); }
	}
	return v;
}

// src strand shown in the TTS column: seq / complement(seq), reversed for the REV encodings.
// complement() drops letters outside ACGTN (rules.h:82-83).
static void make_src(const char* seg, int n, bool comp, bool rev, std::string& src)
{
	src.clear();
	src.reserve(n);
	if (!comp) src.assign(seg, seg + n);
	else {
		for (int i = 0; i < n; i++) {
			switch (seg[i]) { case 'A': src.push_back('T'); break; case 'C': src.push_back('G'); break;
			case 'G': src.push_back('C'); break; case 'T': src.push_back('A'); break; case 'N': src.push_back('N'); break; default: break; }
		}
	}
	if (rev) std::reverse(src.begin(), src.end());
}

void encode_unit_host(const char* seg, int n, int enc, std::string& target, std::string& src)
{
	const EncInfo e = enc_info(enc);
	target.resize(n);
	for (int c = 0; c < n; c++) target[c] = map_base(kRuleOut[enc], seg[e.reversed ? n - 1 - c : c]);
	make_src(seg, n, e.strand == 1, e.reversed, src);
}

bool same_seq(const char* seg, int n)
{
	// Fasim-LongTarget.cpp:873-933: a segment made of one repeated letter of ACGTUN is skipped
	if (n == 0) return true;   // every counter equals size() == 0
	const char c0 = seg[0];
	if (!(c0 == 'A' || c0 == 'C' || c0 == 'G' || c0 == 'T' || c0 == 'U' || c0 == 'N')) return false;
	for (int i = 1; i < n; i++) if (seg[i] != c0) return false;
	return true;
}

// ---- a7: peak picking (ssw_cpp.cpp:470-572) ---------------------------------------------------------
void pick_candidates(const uint32_t* hits, int nhits, std::vector<Cand>& out)
{
	out.clear();
	int k = 0;
	while (k < nhits) {
		const int pos = (int)(hits[k] >> 8), sc = (int)(hits[k] & 0xff);
		if (k == nhits - 1) { out.push_back({ sc, pos }); break; }
		int gap = (int)(hits[k + 1] >> 8) - pos;
		if (gap > 0 && gap < 5) {
			// run of hits whose consecutive distances are < 5 -> first maximum of the run
			int bi = k, bs = sc;
			while (k + 1 < nhits) {
				gap = (int)(hits[k + 1] >> 8) - (int)(hits[k] >> 8);
				if (!(gap > 0 && gap < 5)) break;
				k++;
				const int s2 = (int)(hits[k] & 0xff);
				if (s2 > bs) { bs = s2; bi = k; }
			}
			out.push_back({ bs, (int)(hits[bi] >> 8) });
			k++;
		} else {
			out.push_back({ sc, pos });
			k++;
		}
	}
}

// ---- a8: window policy (fastsim.h:204-211) ----------------------------------------------------------
bool window_for_try(int it, int cand_score, int cand_pos, int* cutlength)
{
	float Iden = 0.6;
	for (int k = 0; k < it; k++) Iden += 0.1;          // float += double, as in the reference loop
	if (!(Iden <= 1)) return false;
	int cl = (int)(cand_score + 24) / (9 * Iden - 4) + 1;
	cl = cand_pos - cl + 1 > 0 ? cl : cand_pos + 1;
	*cutlength = cl;
	return true;
}

// ---- a12/a13: triplex record from an alignment (fastsim.h:291-414, 416-560; sim.h:72-97) ------------
float triplex_stability(char c1, char c2, int para)
{
	if (para > 0) {
		if (c1 == 'A' && c2 == 'T') return 3.7; if (c1 == 'T' && c2 == 'G') return 2.8; if (c1 == 'G' && c2 == 'G') return 2.2;
		if (c1 == 'G' && c2 == 'T') return 2.4; if (c1 == 'G' && c2 == 'C') return 4.5; if (c1 == 'C' && c2 == 'T') return 2.6;
		if (c1 == 'C' && c2 == 'C') return 2.4;
	} else {
		if (c1 == 'A' && c2 == 'A') return 3.0; if (c1 == 'A' && c2 == 'T') return 3.5; if (c1 == 'A' && c2 == 'C') return 1.0;
		if (c1 == 'T' && c2 == 'G') return 1.0; if (c1 == 'G' && c2 == 'A') return 1.0; if (c1 == 'G' && c2 == 'G') return 3.0;
		if (c1 == 'G' && c2 == 'C') return 3.0; if (c1 == 'C' && c2 == 'T') return 2.0; if (c1 == 'C' && c2 == 'C') return 1.0;
	}
	return 0;
}

// letter of the display strand at unit column q (strand==1: complement(seg), reversed for the REV encodings)
static inline char comp_letter(char c)
{
	switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return c; }
}

bool only_acgtn(const char* seg, int n)
{
	for (int i = 0; i < n; i++) { const char c = seg[i]; if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N')) return false; }
	return true;
}

void convert_triplex(const AlignResult& al, const uint32_t* cigar, const std::string& rna, const char* seg, int n, int enc,
	long dna_start, const fasim_params& p, std::vector<HostTriplex>& list, bool seg_acgtn, bool with_strings)
{
	// the alignment's length is the sum of its CIGAR runs: one shorter than -ni never becomes a record (the check further
	// down, fastsim.h:385), so its strings, identity and stability need not be worked out
	{
		int64_t total = 0;
		for (int k = 0; k < al.cigar_len; k++) total += cigar[k] >> 4;
		if (total < p.ntMin) return;
	}
	const EncInfo e = enc_info(enc);
	const char* o = kRuleOut[enc];
	// complement() drops letters outside ACGTN (rules.h:82-83); only then the display strand has to be
	// materialised, otherwise its letters are computed on the fly
	const bool clean = seg_acgtn || e.strand != 1;
	std::string src;
	if (!clean) make_src(seg, n, true, e.reversed, src);
	auto src_at = [&](int q) -> char {
		if (!clean) return (q >= 0 && q < (int)src.size()) ? src[q] : '\0';
		if (q < 0 || q >= n) return '\0';
		const char c = seg[e.reversed ? n - 1 - q : q];
		return e.strand == 1 ? comp_letter(c) : c;
	};
	if (!with_strings) {
		std::vector<TriplexNum> one;
		convert_triplex_num(al, cigar, rna, seg, n, enc, dna_start, p, one, seg_acgtn);
		if (one.empty()) return;
		HostTriplex t;
		t.stari = one[0].stari; t.endi = one[0].endi; t.starj = one[0].starj; t.endj = one[0].endj;
		t.strand = e.strand; t.reverse = e.para; t.rule = e.rule; t.nt = one[0].nt;
		t.score = one[0].score; t.identity = one[0].identity; t.tri_score = one[0].tri_score;
		list.push_back(std::move(t));
		return;
	}
	// expand the CIGAR from (ref_begin, query_begin): M -> (target, src, rna); I -> ('-','-',rna); D -> (target, src,'-')
	// (per-thread buffers)
	thread_local std::string tgt_al, tts, tfo;
	tgt_al.clear(); tts.clear(); tfo.clear();
	int q = al.ref_begin, r = al.query_begin;
	for (int k = 0; k < al.cigar_len; k++) {
		const uint32_t len = cigar[k] >> 4, op = cigar[k] & 0xf;
		for (uint32_t t = 0; t < len; t++) {
			if (op == 1) { tgt_al.push_back('-'); tts.push_back('-'); tfo.push_back(rna[r++]); }
			else {
				const char tl = (q >= 0 && q < n) ? map_base(o, seg[e.reversed ? n - 1 - q : q]) : '\0';
				tgt_al.push_back(tl);
				tts.push_back(src_at(q));
				q++;
				if (op == 2) tfo.push_back('-'); else tfo.push_back(rna[r++]);
			}
		}
	}
	const int nt = (int)tgt_al.size();
	int match = 0, mis = 0;
	for (int i = 0; i < nt; i++) { if (tgt_al[i] == tfo[i]) match++; else mis++; }
	const float identity = (float)(100 * match) / (float)(match + mis);     // fastsim.h:335
	float tri = 0.0f;
	if (nt >= p.ntMin && nt <= p.ntMax) {
		// left-to-right float accumulation with the TT / CC run substitution (fastsim.h:344-383)
		float prev_v = 0.0f, v = 0.0f;
		char prev_c = 0, cur = 0;
		for (int i = 0; i < nt; i++) {
			cur = (tgt_al[i] == '-') ? '-' : tts[i];
			v = triplex_stability(cur, tfo[i], e.para);
			if (cur == prev_c && cur == 'T') { tri = tri - prev_v + p.penaltyT; v = p.penaltyT; }
			if (cur == prev_c && cur == 'C') { tri = tri - prev_v + p.penaltyC; v = p.penaltyC; }
			prev_v = v;
			if (tgt_al[i] != '-') prev_c = cur;
			tri += v;
		}
		tri = tri / nt;
	}
	if (nt < p.ntMin) return;
	HostTriplex t;
	int rs, re;
	if ((e.para > 0 && e.strand == 1) || (e.para < 0 && e.strand == 0)) { rs = n - al.ref_end - 1; re = n - al.ref_begin - 1; }   // fastsim.h:389-396
	else { rs = al.ref_begin + 1; re = al.ref_end + 1; }
	t.stari = al.query_begin + 1; t.endi = al.query_end + 1;
	t.starj = (int)(rs + dna_start); t.endj = (int)(re + dna_start);
	t.strand = e.strand; t.reverse = e.para; t.rule = e.rule; t.nt = nt;
	t.score = (float)al.sw_score; t.identity = identity; t.tri_score = tri;
	if (with_strings) { t.tfo = tfo; t.tts = tts; }
	list.push_back(std::move(t));
}

// The numbers only (every candidate alignment of a scan goes through here; the strings are built for the ~1 % that survive
// all filters): one walk over the CIGAR, the same comparisons and the same left-to-right float accumulation as above.
void convert_triplex_num(const AlignResult& al, const uint32_t* cigar, const std::string& rna, const char* seg, int n, int enc,
	long dna_start, const fasim_params& p, std::vector<TriplexNum>& list, bool seg_acgtn)
{
	int64_t total = 0;
	for (int k = 0; k < al.cigar_len; k++) total += cigar[k] >> 4;
	if (total < p.ntMin) return;                                       // fastsim.h:385
	const EncInfo e = enc_info(enc);
	const char* o = kRuleOut[enc];
	const bool clean = seg_acgtn || e.strand != 1;
	std::string src;
	if (!clean) make_src(seg, n, true, e.reversed, src);
	auto src_at = [&](int q) -> char {
		if (!clean) return (q >= 0 && q < (int)src.size()) ? src[q] : '\0';
		if (q < 0 || q >= n) return '\0';
		const char c = seg[e.reversed ? n - 1 - q : q];
		return e.strand == 1 ? comp_letter(c) : c;
	};
	const int nt = (int)total;
	const bool want_tri = nt >= p.ntMin && nt <= p.ntMax;
	int match = 0, mis = 0;
	float tri = 0.0f, prev_v = 0.0f, v = 0.0f;
	char prev_c = 0, cur = 0;
	int q = al.ref_begin, r = al.query_begin;
	for (int k = 0; k < al.cigar_len; k++) {
		const uint32_t len = cigar[k] >> 4, op = cigar[k] & 0xf;
		for (uint32_t t = 0; t < len; t++) {
			char tgt, tt, tf;
			if (op == 1) { tgt = '-'; tt = '-'; tf = rna[r++]; }
			else {
				tgt = (q >= 0 && q < n) ? map_base(o, seg[e.reversed ? n - 1 - q : q]) : '\0';
				tt = src_at(q);
				q++;
				tf = (op == 2) ? '-' : rna[r++];
			}
			if (tgt == tf) match++; else mis++;
			if (want_tri) {
				cur = (tgt == '-') ? '-' : tt;
				v = triplex_stability(cur, tf, e.para);
				if (cur == prev_c && cur == 'T') { tri = tri - prev_v + p.penaltyT; v = p.penaltyT; }
				if (cur == prev_c && cur == 'C') { tri = tri - prev_v + p.penaltyC; v = p.penaltyC; }
				prev_v = v;
				if (tgt != '-') prev_c = cur;
				tri += v;
			}
		}
	}
	TriplexNum t;
	t.identity = (float)(100 * match) / (float)(match + mis);            // fastsim.h:335
	if (want_tri) tri = tri / nt;
	int rs, re;
	if ((e.para > 0 && e.strand == 1) || (e.para < 0 && e.strand == 0)) { rs = n - al.ref_end - 1; re = n - al.ref_begin - 1; }   // fastsim.h:389-396
	else { rs = al.ref_begin + 1; re = al.ref_end + 1; }
	t.stari = al.query_begin + 1; t.endi = al.query_end + 1;
	t.starj = (int)(rs + dna_start); t.endj = (int)(re + dna_start);
	t.nt = nt; t.cand = -1;
	t.score = (float)al.sw_score; t.tri_score = tri;
	list.push_back(t);
}

// ---- a14 (fastsim.h:92-156, 273-288).  The comparators are not strict weak orderings; the order that
// comes out is whatever libstdc++'s std::sort/std::unique produce for this comparison sequence, so we
// call exactly those algorithms on the same input order.
template <class T> static bool by_start(const T& a, const T& b)
{
	if (a.stari == b.stari) return a.starj == b.starj ? a.score > b.score : a.starj > b.starj;
	return a.starj > b.starj;
}
template <class T> static bool by_end(const T& a, const T& b)
{
	if (a.endi == b.endi) return a.starj == b.starj ? a.score > b.score : a.starj < b.starj;
	return a.starj < b.starj;
}
template <class T> static bool by_score(const T& a, const T& b) { return a.score > b.score; }
template <class T> static bool redundant(const T& a, const T& b)
{
	if (a.stari == b.stari && a.starj == b.starj && a.endi == b.endi && a.endj == b.endj && a.score == b.score) return true;
	return b.stari >= a.stari && b.starj >= a.starj && b.endi <= a.endi && b.endj <= a.endj && b.score < a.score;
}

template <class T> static void dedup_top_t(std::vector<T>& mine, const fasim_params& p, std::vector<T>& out)
{
	std::sort(mine.begin(), mine.end(), by_start<T>);
	mine.erase(std::unique(mine.begin(), mine.end(), redundant<T>), mine.end());
	std::sort(mine.begin(), mine.end(), by_end<T>);
	mine.erase(std::unique(mine.begin(), mine.end(), redundant<T>), mine.end());
	std::sort(mine.begin(), mine.end(), by_score<T>);
	const size_t lim = mine.size() > 50 ? 50 : mine.size();
	for (size_t i = 0; i < lim; i++)
		if (mine[i].identity >= p.minIdentity && mine[i].tri_score >= p.minStability && mine[i].nt >= p.ntMin) out.push_back(mine[i]);
}
void dedup_top(std::vector<HostTriplex>& mine, const fasim_params& p, std::vector<HostTriplex>& out) { dedup_top_t(mine, p, out); }
void dedup_top_num(std::vector<TriplexNum>& mine, const fasim_params& p, std::vector<TriplexNum>& out) { dedup_top_t(mine, p, out); }

// ---- a16: cluster_triplex (Fasim-LongTarget.cpp:600-691) --------------------------------------------
// Same sequential semantics with flat arrays instead of std::map<size_t, axis>.  The reference's class
// search `for (i = 0; i < axis_map.size(); i++) axis_map[i]...` visits exactly the keys 0..max(key) (it
// inserts the missing ones while it runs), so only the largest live key has to be tracked.
void cluster_triplex(int dd, int length, std::vector<HostTriplex>& list)
{
	int top = -1;
	for (const HostTriplex& t : list) if (t.nt > length) top = std::max(top, (t.stari + t.endi) / 2 + dd);
	if (top < 0) return;
	std::vector<int> near(top + 2, 0);
	std::vector<char> live(top + 2, 0);
	std::vector<std::vector<int>> by_mid(top + 2);
	int max_near = 0, max_pos = 0, find = 0;
	for (size_t k = 0; k < list.size(); k++) {
		HostTriplex& t = list[k];
		if (t.nt <= length) { if (t.middle >= 0 && t.middle <= top) by_mid[t.middle].push_back((int)k); continue; }
		const int middle = (t.stari + t.endi) / 2;
		t.middle = middle; t.motif = 0;
		by_mid[middle].push_back((int)k);
		for (int i = -dd; i <= dd; i++) {
			const int pos = middle + i;          // middle - dd >= 0 unless the caller asked for FASIM_TAIL_CLAMP_CLUSTER (see
			if (pos < 0) continue;               // fasim_tfosorted_ex): positions before the query start then do not exist
			live[pos] = 1;
			if (i > 0) near[pos] += dd - i; else if (i < 0) near[pos] += dd + i;
			if (near[pos] > max_near) { max_near = near[pos]; max_pos = pos; find = 1; }
		}
		t.neartriplex = near[middle];
	}
	int max_key = top;
	while (max_key >= 0 && !live[max_key]) max_key--;
	int cls = 1;
	while (find) {
		for (int i = max_pos - dd; i <= max_pos + dd; i++) {
			if (i >= 0 && i <= top) {
				for (int k : by_mid[i]) if (list[k].motif == 0) { list[k].motif = cls; list[k].center = max_pos; }
				live[i] = 0; near[i] = 0;
			}
		}
		while (max_key >= 0 && !live[max_key]) max_key--;
		max_near = 0; find = 0;
		for (int i = 0; i <= max_key; i++) {
			live[i] = 1;                          // operator[] inserts the key
			if (near[i] > max_near) { max_near = near[i]; max_pos = i; find = 1; }
		}
		++cls;
	}
}

static const char* strand_name(int reverse, int strand)
{
	if (reverse == 1) return strand == 0 ? "ParaPlus" : (strand == 1 ? "ParaMinus" : "");
	if (reverse == -1) return strand == 1 ? "AntiMinus" : (strand == 0 ? "AntiPlus" : "");
	return "";
}
static bool by_motif(const HostTriplex& a, const HostTriplex& b) { return a.motif < b.motif; }

// default operator<<(float): "%g" with precision 6 (libstdc++'s num_put formats through the same printf conversion)
static inline void put_float(std::string& o, float v) { char b[40]; const int n = snprintf(b, sizeof b, "%g", (double)v); o.append(b, (size_t)n); }
static inline void put_int(std::string& o, long v) { char b[24]; const int n = snprintf(b, sizeof b, "%ld", v); o.append(b, (size_t)n); }

std::string tfosorted_text(std::vector<HostTriplex>& list, const std::string& chr, long start_genome, const fasim_params& p)
{
	for (HostTriplex& t : list) {                 // main(): Fasim-LongTarget.cpp:141-149
		if (!t.genome_set) { t.genomestart = t.starj + start_genome - 1; t.genomeend = t.endj + start_genome - 1; t.genome_set = true; }
	}
	cluster_triplex(p.cDistance, p.cLength, list);
	std::sort(list.begin(), list.end(), by_motif); // Fasim-LongTarget.cpp:813 (unstable, same algorithm)
	// The rows are independent once the order is fixed: format them in parallel slices (same bytes as the reference's
	// ostream inserters: integers in decimal, floats as "%g", tabs, '\n' from std::endl) and concatenate.
	const size_t n = list.size();
	const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
	const size_t nt = n < 20000 ? 1 : std::min<size_t>(std::min<unsigned>(hw, 32u), n / 10000);
	std::vector<std::string> part(nt);
	auto work = [&](size_t ti) {
		std::string& o = part[ti];
		const size_t i0 = n * ti / nt, i1 = n * (ti + 1) / nt;
		o.reserve((i1 - i0) * 230);
		for (size_t i = i0; i < i1; i++) {
			const HostTriplex& a = list[i];
			if (a.motif == 0) continue;               // rows never clustered (nt == lg) are dropped (:819)
			put_int(o, a.stari); o += '\t'; put_int(o, a.endi); o += '\t'; put_int(o, a.starj); o += '\t'; put_int(o, a.endj); o += '\t';
			o += (a.starj < a.endj ? 'R' : 'L'); o += '\t'; o += chr; o += '\t';
			put_int(o, a.genomestart); o += '\t'; put_int(o, a.genomeend); o += '\t';
			put_float(o, a.tri_score); o += '\t'; put_float(o, a.identity); o += '\t';
			o += strand_name(a.reverse, a.strand); o += '\t'; put_int(o, a.rule); o += '\t'; put_float(o, a.score); o += '\t';
			put_int(o, a.nt); o += '\t'; put_int(o, a.motif); o += '\t'; put_int(o, a.middle); o += '\t'; put_int(o, a.center); o += '\t';
			o += a.tfo; o += '\t'; o += a.tts; o += '\n';
		}
	};
	if (nt == 1) work(0);
	else { std::vector<std::thread> th; for (size_t k = 0; k < nt; k++) th.emplace_back(work, k); for (auto& t : th) t.join(); }
	std::string out = "QueryStart\tQueryEnd\tStartInSeq\tEndInSeq\tDirection\tChr\tStartInGenome\tEndInGenome\tMeanStability\t"
	                  "MeanIdentity(%)\tStrand\tRule\tScore\tNt(bp)\tClass\tMidPoint\tCenter\tTFO sequence\tTTS sequence\n";
	size_t total = out.size();
	for (const std::string& s : part) total += s.size();
	out.reserve(total);
	for (const std::string& s : part) out += s;
	return out;
}

// print_cluster() (Fasim-LongTarget.cpp:694-795): bedGraph of the TTS coverage of one class.  The reference walks a
// std::map with one entry per covered DNA position; here the coverage is swept from interval end points (it scales
// with the number of triplexes, not with the covered base pairs) and the walk's quirks are applied to whole runs:
// the first run starts one position early (:751), the last covered position is always a run of its own (:732, :741)
// and every gap gets a zero line (:760-764).
std::string tfoclass_text(const std::vector<HostTriplex>& list, int level, const std::string& chr, long start_genome,
	long dna_size, const std::string& rna_name, const fasim_params& p)
{
	const long sg = start_genome - 1;                    // printResult passes start_genome - 1 (:834)
	std::ostringstream o;
	o << "browser position " << chr << ":" << sg << "-" << sg + dna_size << std::endl;
	o << "browser hide all" << std::endl << "browser pack refGene encodeRegions" << std::endl << "browser full altGraph" << std::endl;
	o << "# 300 base wide bar graph, ausoScale is on by default == graphing" << std::endl;
	o << "# limits will dynamically change to always show full range of data" << std::endl;
	o << "# in viewing window, priority = 20 position this as the second graph" << std::endl;
	o << "# Note, zero-relative, half-open coordinate system in use for bedGraph format" << std::endl;
	o << "track type=bedGraph name='" << rna_name << " TTS (" << level << ")' description='" << p.cDistance << "-" << p.cLength
	  << "' visibility=full color=200,100,0 altColor=0,100,200 priority=20" << std::endl;

	std::vector<std::pair<int, int>> ev;                 // (position, +1 / -1), half-open [lo, hi)
	for (const HostTriplex& t : list) {
		if (t.motif != level || t.starj == t.endj) continue;
		ev.push_back({ std::min(t.starj, t.endj), 1 });
		ev.push_back({ std::max(t.starj, t.endj), -1 });
	}
	if (ev.empty()) return o.str();
	std::sort(ev.begin(), ev.end());
	struct Run { int a, b, v; };                         // inclusive positions a..b with coverage v > 0
	std::vector<Run> runs;
	int depth = 0;
	for (size_t i = 0; i < ev.size();) {
		const int pos = ev[i].first;
		while (i < ev.size() && ev[i].first == pos) depth += ev[i++].second;
		if (depth <= 0 || i >= ev.size()) continue;
		if (!runs.empty() && runs.back().b + 1 == pos && runs.back().v == depth) runs.back().b = ev[i].first - 1;   // same level goes on
		else runs.push_back({ pos, ev[i].first - 1, depth });
	}
	const int last = runs.back().b;                      // final_genome - start_genome (:723-726)
	if (runs.back().a < last) { runs.back().b = last - 1; runs.push_back({ last, last, runs[runs.size() - 1].v }); }
	const int sgi = (int)sg;
	for (size_t k = 0; k < runs.size(); k++) {
		const Run& r = runs[k];
		if (k + 1 == runs.size()) { o << chr << "\t" << r.a + sgi - 1 << "\t" << r.b + sgi << "\t" << r.v << std::endl; break; }
		o << chr << "\t" << r.a + sgi - (k == 0 ? 2 : 1) << "\t" << r.b + sgi << "\t" << r.v << std::endl;
		if (runs[k + 1].a - r.b != 1) o << chr << "\t" << r.b + sgi << "\t" << runs[k + 1].a + sgi - 1 << "\t" << 0 << std::endl;
	}
	return o.str();
}

} // namespace fasim
