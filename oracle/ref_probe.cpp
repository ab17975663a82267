// oracle/ref_probe.cpp -- TEST INFRASTRUCTURE, not product code.
//
// A probe (our own code) that is compiled against the reference implementation's
// sources *where they lie* (-I/root/reference, linking ssw_cpp.cpp / sswNew.cpp from
// there; see oracle/Makefile).  Nothing from the reference is copied into this repo;
// the resulting binary lives in oracle/_ref/ (git-ignored) and is used to
//   (1) generate the golden fixtures under tests/golden/ (tests/golden/make_golden.py), and
//   (2) validate oracle/fasim_oracle.cpp and the HIP path on arbitrary seeded inputs.
//
// It calls the reference's own entry points for every stage of the hot path:
//   calc_score_once  (stats.h:879)        -> stage-1 max score
//   ssw_init/ssw_pre_align (ssw.h:78,128) -> stage-2 column maxima
//   Aligner::preAlign (ssw_cpp.cpp:388)   -> candidates
//   Aligner::Align    (ssw_cpp.cpp:599)   -> window alignments
//   fastSIM           (fastsim.h:158)     -> per-unit triplex list
//   SIM               (sim.h:410)         -> per-unit triplex list of the -F (classic SIM) path  [simscan]
// and re-states only the unit enumeration of LongTarget() (Fasim-LongTarget.cpp:395-586),
// which cannot be linked because it lives in the file that defines main().
//
// Output is a line protocol (see tools/probe_format.md is NOT needed: the format is
// documented next to each printf below and parsed by tests/refprobe.py).

#include "fastsim.h"   // reference header (pulls ssw_cpp.h, ssw.h, sim.h, stats.h, rules.h)

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <iostream>
#include <fstream>

// NB: the reference headers leak the macros N, K, MM, AA, QQ, RW, EL, ES, NA, MAXSQ,
// BIGNUM and `using namespace std`; none of those identifiers are used below.

static uint64_t fnv1a_ints(const int* v, int n)
{
	uint64_t h = 1469598103934665603ULL;
	for (int i = 0; i < n; i++) {
		uint32_t x = (uint32_t)v[i];
		for (int b = 0; b < 4; b++) {
			h ^= (x >> (8 * b)) & 0xff;
			h *= 1099511628211ULL;
		}
	}
	return h;
}

static uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// single-record FASTA (header '>species|chr|start-end'), all sequence lines joined
static bool read_fasta(const char* path, std::string& header, std::string& seq)
{
	std::ifstream in(path);
	if (!in) return false;
	std::string line;
	header.clear(); seq.clear();
	bool first = true;
	while (std::getline(in, line)) {
		while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
		if (first && !line.empty() && line[0] == '>') { header = line.substr(1); first = false; continue; }
		first = false;
		if (!line.empty() && line[0] == '>') break; // only the first record
		seq += line;
	}
	return true;
}

static const int8_t* nt_table()
{
	static int8_t t[128];
	static bool init = false;
	if (!init) {
		for (int i = 0; i < 128; i++) t[i] = 4;
		t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3;
		t['U'] = t['u'] = 0; // sic: the reference maps U to A (ssw_cpp.cpp:21-25)
		init = true;
	}
	return t;
}

// raw ssw_pre_align through the C ABI exactly like Aligner::preAlign does (ssw_cpp.cpp:394-415)
static std::vector<int> ref_pre_align(const std::string& q, const std::string& r)
{
	std::vector<int8_t> tq(q.size()), tr(r.size());
	const int8_t* tab = nt_table();
	for (size_t i = 0; i < q.size(); i++) tq[i] = tab[(int)q[i] & 127];
	for (size_t i = 0; i < r.size(); i++) tr[i] = tab[(int)r[i] & 127];
	int8_t mat[25];
	int id = 0;
	for (int i = 0; i < 4; i++) { for (int j = 0; j < 4; j++) mat[id++] = (i == j) ? 5 : -4; mat[id++] = -4; }
	for (int j = 0; j < 5; j++) mat[id++] = -4;
	s_profile* p = ssw_init(tq.data(), (int)q.size(), mat, 5, 2);
	int* col = ssw_pre_align(p, tr.data(), (int)r.size(), 16, 4, 0x0f, 0, 32767, 15, 0);
	std::vector<int> out(col, col + r.size());
	free(col);
	init_destroy(p);
	return out;
}

static void print_alignment(const char* tag, int it, int L, const StripedSmithWaterman::Alignment& a)
{
	// tag it L sw_score ref_begin ref_end query_begin query_end cigar
	printf("%s %d %d %d %d %d %d %d %s\n", tag, it, L, (int)a.sw_score, a.ref_begin, a.ref_end,
		a.query_begin, a.query_end, a.cigar_string.empty() ? "*" : a.cigar_string.c_str());
}

struct unit_desc { int strand, Para, rule; };

static void run_unit(const std::string& rna, const std::string& seq1, int seg, int enc,
	long dnaStartPos, const unit_desc& u, struct para& pl, bool detail)
{
	// Fasim-LongTarget.cpp:410-431 / 499-522
	std::string seq2, src;
	if (u.Para > 0 && u.strand == 0) { seq2 = transferString(seq1, 0, 1, u.rule); src = seq1; }
	else if (u.Para > 0 && u.strand == 1) { seq2 = transferString(seq1, 1, 1, u.rule); reverseSeq(seq2); src = seq1; complement(src); reverseSeq(src); }
	else if (u.Para < 0 && u.strand == 1) { seq2 = transferString(seq1, 1, -1, u.rule); src = seq1; complement(src); }
	else { seq2 = transferString(seq1, 0, -1, u.rule); reverseSeq(seq2); src = seq1; reverseSeq(src); }

	std::string rnac = rna;
	int s1 = calc_score_once(rnac, seq2, (int)dnaStartPos, pl.rule);
	int minscore = s1 * 0.8; // Fasim-LongTarget.cpp:413 (double multiply, truncation)

	std::vector<int> col = ref_pre_align(rna, seq2);
	int nhits = 0;
	for (size_t i = 0; i < col.size(); i++) if (col[i] > minscore) nhits++;

	StripedSmithWaterman::Aligner aligner;
	StripedSmithWaterman::Filter filter;
	StripedSmithWaterman::Alignment alignment;
	std::vector<StripedSmithWaterman::scoreInfo> cand;
	aligner.preAlign(rna.c_str(), seq2.c_str(), (int)seq2.size(), filter, &alignment, 15, minscore, cand, 5, -4);

	// U seg enc dnaStartPos strand Para rule n stage1 thr colhash nhits ncand
	printf("U %d %d %ld %d %d %d %d %d %d %016llx %d %d\n", seg, enc, dnaStartPos, u.strand, u.Para, u.rule,
		(int)seq2.size(), s1, minscore, (unsigned long long)fnv1a_ints(col.data(), (int)col.size()), nhits, (int)cand.size());
	if (detail) {
		// H pos score  (every column above the threshold)
		for (size_t i = 0; i < col.size(); i++) if (col[i] > minscore) printf("H %d %d\n", (int)i, col[i]);
	}
	for (size_t i = 0; i < cand.size(); i++) {
		printf("C %d %d\n", cand[i].score, cand[i].position);
		if (!detail) continue;
		// the window policy of fastSIM (fastsim.h:202-237), restated only to expose each try
		float Iden = 0.6;
		int it = 0;
		while (Iden <= 1) {
			int cutlength = (int)(cand[i].score + 24) / (9 * Iden - 4) + 1;
			cutlength = cand[i].position - cutlength + 1 > 0 ? cutlength : cand[i].position + 1;
			std::string small = seq2.substr(cand[i].position - cutlength + 1, cutlength);
			aligner.Align(rna.c_str(), small.c_str(), (int)small.size(), filter, &alignment, 15);
			print_alignment("T", it, cutlength, alignment);
			if (alignment.sw_score >= cand[i].score) break;
			Iden += 0.1;
			it++;
		}
	}
	// the reference's own fastSIM for the unit's triplex list
	std::vector<struct triplex> tl;
	std::string a = rna, b = seq2, c = src;
	fastSIM(a, b, c, dnaStartPos, minscore, 5, -4, -12, -4, tl, u.strand, u.Para, u.rule,
		pl.ntMin, pl.ntMax, pl.penaltyT, pl.penaltyC, pl);
	for (size_t i = 0; i < tl.size(); i++) {
		const struct triplex& t = tl[i];
		// X stari endi starj endj strand reverse rule nt score identity(bits) tri_score(bits) stri strj
		printf("X %d %d %d %d %d %d %d %d %d %08x %08x %s %s\n", t.stari, t.endi, t.starj, t.endj, t.strand,
			t.reverse, t.rule, t.nt, (int)t.score, fbits(t.identity), fbits(t.tri_score),
			t.stri_align.c_str(), t.strj_align.c_str());
	}
}

// -F path: the reference's SIM() for one unit (Fasim-LongTarget.cpp:420-426 and the other seven call sites)
static void run_unit_sim(const std::string& rna, const std::string& seq1, int seg, int enc, long dnaStartPos, const unit_desc& u, struct para& pl)
{
	std::string seq2, src;
	if (u.Para > 0 && u.strand == 0) { seq2 = transferString(seq1, 0, 1, u.rule); src = seq1; }
	else if (u.Para > 0 && u.strand == 1) { seq2 = transferString(seq1, 1, 1, u.rule); reverseSeq(seq2); src = seq1; complement(src); reverseSeq(src); }
	else if (u.Para < 0 && u.strand == 1) { seq2 = transferString(seq1, 1, -1, u.rule); src = seq1; complement(src); }
	else { seq2 = transferString(seq1, 0, -1, u.rule); reverseSeq(seq2); src = seq1; reverseSeq(src); }
	std::string rnac = rna;
	int s1 = calc_score_once(rnac, seq2, (int)dnaStartPos, pl.rule);
	int minscore = s1 * 0.8;
	std::vector<struct triplex> tl;
	std::string a = rna, b = seq2, c = src;
	SIM(a, b, c, dnaStartPos, minscore, 5, -4, -12, -4, tl, u.strand, u.Para, u.rule, pl.ntMin, pl.ntMax, pl.penaltyT, pl.penaltyC);
	// V seg enc dnaStartPos strand Para rule n stage1 thr ntriplex
	printf("V %d %d %ld %d %d %d %d %d %d %d\n", seg, enc, dnaStartPos, u.strand, u.Para, u.rule, (int)seq2.size(), s1, minscore, (int)tl.size());
	for (size_t i = 0; i < tl.size(); i++) {
		const struct triplex& t = tl[i];
		printf("X %d %d %d %d %d %d %d %d %d %08x %08x %s %s\n", t.stari, t.endi, t.starj, t.endj, t.strand,
			t.reverse, t.rule, t.nt, (int)t.score, fbits(t.identity), fbits(t.tri_score),
			t.stri_align.c_str(), t.strj_align.c_str());
	}
	fflush(stdout);
}

static int cmd_scan(int argc, char** argv)
{
	if (argc < 4) { fprintf(stderr, "usage: scan|simscan rna.fa dna.fa [-r R] [-t T] [-detail 0|1] [-segfirst a] [-segcount n] [flags as fasim]\n"); return 2; }
	const bool classic = !strcmp(argv[1], "simscan");      // -F: SIM() instead of fastSIM()
	std::string rh, rna, dh, dna;
	if (!read_fasta(argv[2], rh, rna) || !read_fasta(argv[3], dh, dna)) { fprintf(stderr, "cannot read input\n"); return 2; }
	struct para pl;
	pl.rule = 0; pl.cutLength = 5000; pl.strand = 0; pl.overlapLength = 100; pl.minScore = 0;
	pl.detailOutput = false; pl.ntMin = 20; pl.ntMax = 100000; pl.scoreMin = 0.0; pl.minIdentity = 60.0;
	pl.minStability = 1; pl.penaltyT = -1000; pl.penaltyC = 0; pl.cDistance = 15; pl.cLength = 50;
	pl.doFastSim = true; pl.corenum = 1;
	bool detail = true;
	int segfirst = 0, segcount = 1 << 30;
	for (int i = 4; i + 1 < argc; i += 2) {
		std::string k = argv[i]; const char* v = argv[i + 1];
		if (k == "-r") pl.rule = atoi(v); else if (k == "-t") pl.strand = atoi(v);
		else if (k == "-c") pl.cutLength = atoi(v); else if (k == "-o") pl.overlapLength = atoi(v);
		else if (k == "-i") pl.minIdentity = atoi(v); else if (k == "-S") pl.minStability = atoi(v);
		else if (k == "-ni") pl.ntMin = atoi(v); else if (k == "-na") pl.ntMax = atoi(v);
		else if (k == "-pc") pl.penaltyC = atoi(v); else if (k == "-pt") pl.penaltyT = atoi(v);
		else if (k == "-detail") detail = atoi(v) != 0;
		else if (k == "-segfirst") segfirst = atoi(v); else if (k == "-segcount") segcount = atoi(v);
		else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
	}
	std::vector<std::string> segs; std::vector<int> starts; int cutn = 0;
	cutSequence(dna, segs, starts, pl.cutLength, pl.overlapLength, cutn);
	printf("Q %d %d %d\n", (int)rna.size(), (int)dna.size(), (int)segs.size());
	// silence the reference's stdout chatter ("unknown letter", ...) is not needed: it only
	// prints from the driver file which is not linked here.
	for (int s = 0; s < (int)segs.size(); s++) {
		if (s < segfirst || s >= segfirst + segcount) continue;
		std::string seq1 = segs[s];
		// same_seq (Fasim-LongTarget.cpp:873) restated: skip segments made of one repeated letter
		bool same = true;
		for (size_t i = 1; i < seq1.size(); i++) if (seq1[i] != seq1[0]) { same = false; break; }
		if (same && !seq1.empty() && strchr("ACGTUN", seq1[0])) { printf("K %d\n", s); continue; }
		int enc = 0;
		if (pl.strand >= 0) {
			for (int j = 1; j <= 6; j++) {
				unit_desc a = { 0, 1, j }, b = { 1, 1, j };
				bool on = (pl.rule == 0) || (pl.rule == j);
				if (on) { if (classic) run_unit_sim(rna, seq1, s, enc, starts[s], a, pl); else run_unit(rna, seq1, s, enc, starts[s], a, pl, detail); } enc++;
				if (on) { if (classic) run_unit_sim(rna, seq1, s, enc, starts[s], b, pl); else run_unit(rna, seq1, s, enc, starts[s], b, pl, detail); } enc++;
			}
		} else enc = 12;
		if (pl.strand <= 0) {
			for (int j = 1; j <= 18; j++) {
				unit_desc a = { 1, -1, j }, b = { 0, -1, j };
				bool on = (pl.rule == 0) || (pl.rule == j);
				if (on) { if (classic) run_unit_sim(rna, seq1, s, enc, starts[s], a, pl); else run_unit(rna, seq1, s, enc, starts[s], a, pl, detail); } enc++;
				if (on) { if (classic) run_unit_sim(rna, seq1, s, enc, starts[s], b, pl); else run_unit(rna, seq1, s, enc, starts[s], b, pl, detail); } enc++;
			}
		}
	}
	return 0;
}

// batch mode on stdin: one request per line, fields separated by blanks
//   S query ref            -> "S score"                      calc_score_once
//   P query ref            -> "P n v0 v1 ..."                ssw_pre_align column maxima
//   K query ref thr        -> "K ncand s0 p0 s1 p1 ..."      Aligner::preAlign candidates
//   A query ref            -> "A score rb re qb qe cigar"    Aligner::Align
static int cmd_batch()
{
	std::string line;
	StripedSmithWaterman::Aligner aligner;
	StripedSmithWaterman::Filter filter;
	while (std::getline(std::cin, line)) {
		if (line.empty()) continue;
		char op = line[0];
		std::istringstream is(line.substr(1));
		std::string q, r; int thr = 0;
		is >> q >> r;
		if (op == 'S') {
			printf("S %d\n", calc_score_once(q, r, 0, 0));
		} else if (op == 'P') {
			std::vector<int> col = ref_pre_align(q, r);
			printf("P %d", (int)col.size());
			for (size_t i = 0; i < col.size(); i++) printf(" %d", col[i]);
			printf("\n");
		} else if (op == 'K') {
			is >> thr;
			StripedSmithWaterman::Alignment al;
			std::vector<StripedSmithWaterman::scoreInfo> cand;
			aligner.preAlign(q.c_str(), r.c_str(), (int)r.size(), filter, &al, 15, thr, cand, 5, -4);
			printf("K %d", (int)cand.size());
			for (size_t i = 0; i < cand.size(); i++) printf(" %d %d", cand[i].score, cand[i].position);
			printf("\n");
		} else if (op == 'A') {
			StripedSmithWaterman::Alignment al;
			aligner.Align(q.c_str(), r.c_str(), (int)r.size(), filter, &al, 15);
			printf("A %d %d %d %d %d %s\n", (int)al.sw_score, al.ref_begin, al.ref_end, al.query_begin, al.query_end,
				al.cigar_string.empty() ? "*" : al.cigar_string.c_str());
		} else {
			printf("? unknown request\n");
		}
		fflush(stdout);
	}
	return 0;
}

int main(int argc, char** argv)
{
	if (argc >= 2 && (!strcmp(argv[1], "scan") || !strcmp(argv[1], "simscan"))) return cmd_scan(argc, argv);
	if (argc >= 2 && !strcmp(argv[1], "batch")) return cmd_batch();
	fprintf(stderr, "usage: ref_probe scan rna.fa dna.fa [opts] | ref_probe batch < requests\n");
	return 2;
}
