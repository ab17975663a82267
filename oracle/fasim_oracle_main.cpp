// oracle/fasim_oracle_main.cpp -- TEST INFRASTRUCTURE, not product code.
// CLI around the CPU restatement:
//   fasim_oracle scan rna.fa dna.fa [opts]       same line protocol as oracle/ref_probe.cpp `scan`
//   fasim_oracle tfosorted rna.fa dna.fa [opts]   prints the -TFOsorted text to stdout
//   fasim_oracle tfoclass rna.fa dna.fa -level L [opts]   prints the -TFOclass<L> bedGraph text to stdout
//   fasim_oracle simscan rna.fa dna.fa [opts]     the -F path per unit, same lines as ref_probe `simscan` (V + X)
//   -F 1 with tfosorted / tfoclass: classic SIM instead of fastSIM (the reference's -F)
// Options: -r -t -c -o -i -S -ni -na -pc -pt -ds -lg (as the reference CLI, Fasim-LongTarget.cpp:271-283)
//          -detail 0|1 -segfirst a -segcount n -threads T
#include "fasim_oracle.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>

static uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main(int argc, char** argv)
{
	if (argc < 4 || (strcmp(argv[1], "scan") && strcmp(argv[1], "tfosorted") && strcmp(argv[1], "tfoclass") && strcmp(argv[1], "simscan"))) {
		fprintf(stderr, "usage: fasim_oracle scan|simscan|tfosorted|tfoclass rna.fa dna.fa [opts]\n");
		return 2;
	}
	const bool scan = !strcmp(argv[1], "scan");
	const bool simscan = !strcmp(argv[1], "simscan");
	std::string rh, rna, dh, dna;
	if (!fo::read_fasta(argv[2], rh, rna) || !fo::read_fasta(argv[3], dh, dna)) { fprintf(stderr, "cannot read input\n"); return 2; }
	fo::Params p;
	bool detail = true; int segfirst = 0, segcount = 1 << 30, threads = 1, level = 0;
	for (int i = 4; i + 1 < argc; i += 2) {
		std::string k = argv[i]; const char* v = argv[i + 1];
		if (k == "-r") p.rule = atoi(v); else if (k == "-t") p.strand = atoi(v);
		else if (k == "-c") p.cutLength = atoi(v); else if (k == "-o") p.overlapLength = atoi(v);
		else if (k == "-i") p.minIdentity = atoi(v); else if (k == "-S") p.minStability = atoi(v);
		else if (k == "-ni") p.ntMin = atoi(v); else if (k == "-na") p.ntMax = atoi(v);
		else if (k == "-pc") p.penaltyC = atoi(v); else if (k == "-pt") p.penaltyT = atoi(v);
		else if (k == "-ds") p.cDistance = atoi(v); else if (k == "-lg") p.cLength = atoi(v);
		else if (k == "-detail") detail = atoi(v) != 0;
		else if (k == "-segfirst") segfirst = atoi(v); else if (k == "-segcount") segcount = atoi(v);
		else if (k == "-threads") threads = atoi(v);
		else if (k == "-level") level = atoi(v);
		else if (k == "-F") p.classicSim = atoi(v) != 0;
		else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
	}
	std::vector<fo::Triplex> list;
	if (simscan) {
		p.classicSim = true;
		std::vector<fo::UnitTrace> traces; std::vector<int> skipped, starts;
		fo::cut_sequence(dna, p.cutLength, p.overlapLength, starts);
		fo::long_target(p, rna, dna, list, &traces, &skipped, segfirst, segcount, threads);
		printf("Q %d %d %d\n", (int)rna.size(), (int)dna.size(), (int)starts.size());
		size_t ti = 0, ki = 0;
		for (int s = 0; s < (int)starts.size(); s++) {
			if (ki < skipped.size() && skipped[ki] == s) { printf("K %d\n", s); ki++; continue; }
			for (; ti < traces.size() && traces[ti].seg == s; ti++) {
				const fo::UnitTrace& u = traces[ti];
				const fo::EncInfo e = fo::enc_info(u.enc);
				printf("V %d %d %ld %d %d %d %d %d %d %d\n", u.seg, u.enc, u.dna_start, e.strand, e.para, e.rule, u.n, u.stage1, u.thr, (int)u.triplexes.size());
				for (const fo::Triplex& t : u.triplexes)
					printf("X %d %d %d %d %d %d %d %d %d %08x %08x %s %s\n", t.stari, t.endi, t.starj, t.endj, t.strand, t.reverse,
						t.rule, t.nt, (int)t.score, fbits(t.identity), fbits(t.tri_score), t.stri_align.c_str(), t.strj_align.c_str());
			}
		}
		return 0;
	}
	if (scan) {
		std::vector<fo::UnitTrace> traces; std::vector<int> skipped, starts;
		fo::cut_sequence(dna, p.cutLength, p.overlapLength, starts);
		fo::long_target(p, rna, dna, list, &traces, &skipped, segfirst, segcount, threads);
		printf("Q %d %d %d\n", (int)rna.size(), (int)dna.size(), (int)starts.size());
		size_t ti = 0, ki = 0;
		for (int s = 0; s < (int)starts.size(); s++) {
			if (ki < skipped.size() && skipped[ki] == s) { printf("K %d\n", s); ki++; continue; }
			for (; ti < traces.size() && traces[ti].seg == s; ti++) {
				const fo::UnitTrace& u = traces[ti];
				const fo::EncInfo e = fo::enc_info(u.enc);
				printf("U %d %d %ld %d %d %d %d %d %d %016llx %d %d\n", u.seg, u.enc, u.dna_start, e.strand, e.para, e.rule,
					u.n, u.stage1, u.thr, (unsigned long long)u.colhash, u.nhits, (int)u.cands.size());
				if (detail) for (auto& h : u.hits) printf("H %d %d\n", h.first, h.second);
				for (size_t c = 0; c < u.cands.size(); c++) {
					printf("C %d %d\n", u.cands[c].score, u.cands[c].pos);
					if (!detail) continue;
					for (const fo::TryRecord& t : u.tries[c]) {
						std::string cg = t.a.cigar_string();
						printf("T %d %d %d %d %d %d %d %s\n", t.it, t.L, t.a.sw_score, t.a.ref_begin, t.a.ref_end, t.a.query_begin,
							t.a.query_end, cg.empty() ? "*" : cg.c_str());
					}
				}
				for (const fo::Triplex& t : u.triplexes)
					printf("X %d %d %d %d %d %d %d %d %d %08x %08x %s %s\n", t.stari, t.endi, t.starj, t.endj, t.strand, t.reverse,
						t.rule, t.nt, (int)t.score, fbits(t.identity), fbits(t.tri_score), t.stri_align.c_str(), t.strj_align.c_str());
			}
		}
		return 0;
	}
	fo::long_target(p, rna, dna, list, nullptr, nullptr, segfirst, segcount, threads);
	std::string species, chr; long start;
	fo::parse_dna_header(dh, species, chr, start);
	fo::assign_genome(list, chr, start);
	std::string txt = fo::tfosorted_text(list, p);
	if (level > 0) {
		// lncName = the whole RNA header line with every '>' removed, as readRna() keeps it (Fasim-LongTarget.cpp:180-191)
		std::string name;
		for (char c : rh) if (c != '>') name += c;
		txt = fo::tfoclass_text(list, level, chr, start - 1, (long)dna.size(), name, p);
	}
	fwrite(txt.data(), 1, txt.size(), stdout);
	return 0;
}
