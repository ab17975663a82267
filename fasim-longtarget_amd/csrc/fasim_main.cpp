// fasim -- CLI driver with the reference's flags (initEnv(), Fasim-LongTarget.cpp:269-377) and output files
// (printResult(), :797-829), calling the HIP path through the C-ABI of libfasim_hip.so.
//
//   fasim -f1 DNA.fa -f2 RNA.fa [-r R] [-O outdir] [-c cut] [-o overlap] [-t strand] [-i identity]
//         [-S stability] [-ni ntmin] [-na ntmax] [-pc C] [-pt T] [-ds dist] [-lg len] [-cn n]
//   extras: --device N, --stats (timing/statistics on stderr),
//           --all-records: scan EVERY record of a multi-record DNA file (a genome), one record in memory at a time,
//           and write one set of output files per record: <species>-<lnc>-<f1 stem>.<chr>-TFOsorted / -TFOclass...
//
// Differences, all documented in DESIGN.md: without --all-records only the first record of a multi-record
// FASTA file is scanned (the reference accumulates the records, defect B1); -F (classic SIM) and -d are
// not supported.
#include <getopt.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/fasim_hip.h"

struct DnaRecord { std::string species, chr, seq; long start = 0; };

static void strip_eol(std::string& s) { s.erase(std::remove(s.begin(), s.end(), '\r'), s.end()); s.erase(std::remove(s.begin(), s.end(), '\n'), s.end()); }

// header '>species|chr|start-end' (readDna(), Fasim-LongTarget.cpp:226-255); start parsed with atoi
static void parse_header(const std::string& line, DnaRecord& r)
{
	std::string tmp, start; int j = 0;
	for (char c : line) {
		if (c == '>') { tmp.clear(); continue; }
		if (c == '|' && j == 0) { r.species = tmp; j++; tmp.clear(); continue; }
		if (c == '|' && j == 1) { r.chr = tmp; j++; tmp.clear(); continue; }
		if (c == '-' && j == 2) { start = tmp; tmp.clear(); continue; }
		tmp += c;
	}
	r.start = atoi(start.c_str());
}

// streaming FASTA reader: one record at a time (a genome never sits in memory as a whole)
struct DnaReader {
	std::ifstream in; std::string pending; bool have_pending = false;
	bool open(const std::string& path) { in.open(path); return (bool)in; }
	bool next(DnaRecord& r)
	{
		std::string line;
		if (!have_pending) {
			while (std::getline(in, line)) if (!line.empty() && line[0] == '>') { pending = line; have_pending = true; break; }
			if (!have_pending) return false;
		}
		r = DnaRecord();
		parse_header(pending, r);
		have_pending = false;
		while (std::getline(in, line)) {
			if (!line.empty() && line[0] == '>') { pending = line; have_pending = true; break; }
			strip_eol(line); r.seq += line;
		}
		return true;
	}
};

static bool read_rna(const std::string& path, std::string& name, std::string& seq)
{
	std::ifstream in(path);
	if (!in) return false;
	std::string line;
	std::getline(in, line);
	for (char c : line) if (c != '>') name += c;      // readRna(), Fasim-LongTarget.cpp:180-191
	strip_eol(name);
	while (std::getline(in, line)) { strip_eol(line); seq += line; }
	return true;
}

int main(int argc, char* const* argv)
{
	fasim_params p; fasim_params_default(&p);
	std::string f1 = "./", f2 = "./", outdir = "./";
	int device = 0; bool stats = false, all_records = false;
	const char* optstring = "f:s:r:O:c:m:t:i:S:z:Y:Z:h:C:D:E:o:y:Fd";
	struct option lo[] = {
		{ "f1", required_argument, NULL, 'f' }, { "f2", required_argument, NULL, 's' }, { "ni", required_argument, NULL, 'y' },
		{ "na", required_argument, NULL, 'z' }, { "pc", required_argument, NULL, 'Y' }, { "pt", required_argument, NULL, 'Z' },
		{ "cn", required_argument, NULL, 'C' }, { "ds", required_argument, NULL, 'D' }, { "lg", required_argument, NULL, 'E' },
		{ "device", required_argument, NULL, 1001 }, { "stats", no_argument, NULL, 1002 }, { "all-records", no_argument, NULL, 1003 }, { 0, 0, 0, 0 } };
	int opt;
	while ((opt = getopt_long_only(argc, argv, optstring, lo, NULL)) != -1) {
		switch (opt) {
		case 'f': f1 = optarg; break;
		case 's': f2 = optarg; break;
		case 'r': p.rule = atoi(optarg); break;
		case 'O': outdir = optarg; break;
		case 'c': p.cutLength = atoi(optarg); break;
		case 'm': break;                                  // minScore: parsed and unused by the reference
		case 't': p.strand = atoi(optarg); break;
		case 'i': p.minIdentity = atoi(optarg); break;    // sic: atoi (B10)
		case 'S': p.minStability = atoi(optarg); break;   // sic: atoi (B10)
		case 'y': p.ntMin = atoi(optarg); break;
		case 'z': p.ntMax = atoi(optarg); break;
		case 'Y': p.penaltyC = atoi(optarg); break;
		case 'Z': p.penaltyT = atoi(optarg); break;
		case 'o': p.overlapLength = atoi(optarg); break;
		case 'D': p.cDistance = atoi(optarg); break;
		case 'E': p.cLength = atoi(optarg); break;
		case 'C': break;                                  // -cn only picked a result vector in the reference
		case 'F': fprintf(stderr, "fasim: -F (classic SIM) is outside the accelerated path\n"); return 2;
		case 'd': break;
		case 1001: device = atoi(optarg); break;
		case 1002: stats = true; break;
		case 1003: all_records = true; break;
		default: fprintf(stderr, "usage: fasim -f1 DNA.fa -f2 RNA.fa [-O outdir] [-r R] [-t T] [-lg L] ...\n"); return 2;
		}
	}
	std::string lnc_name, rna;
	DnaReader reader;
	if (!reader.open(f1)) { fprintf(stderr, "fasim: cannot read DNA file %s\n", f1.c_str()); return 1; }
	if (!read_rna(f2, lnc_name, rna) || rna.empty()) { fprintf(stderr, "fasim: cannot read RNA file %s\n", f2.c_str()); return 1; }
	std::cout << "Searching triplexes using Fasim" << std::endl << lnc_name << std::endl;

	fasim_engine* eng = nullptr;
	if (fasim_engine_create(device, &eng) != FASIM_OK) { fprintf(stderr, "fasim: %s\n", fasim_last_error(nullptr)); return 1; }
	if (fasim_set_query(eng, rna.data(), (int)rna.size()) != FASIM_OK) { fprintf(stderr, "fasim: %s\n", fasim_last_error(eng)); return 1; }

	// file name: <O>/<species>-<lncName>-<f1 minus 3 chars>-TFOsorted (:123, 800-802); with --all-records the record's
	// chr is appended to the stem so that the records of a genome do not overwrite each other
	const std::string base = f1.substr(0, f1.size() >= 3 ? f1.size() - 3 : 0);
	DnaRecord rec;
	size_t nrec = 0;
	while (reader.next(rec)) {
		if (nrec > 0 && !all_records) {
			fprintf(stderr, "fasim: %s holds more than one record: only the first one was scanned (use --all-records; see DESIGN.md, B1)\n", f1.c_str());
			break;
		}
		fasim_result* res = nullptr;
		if (fasim_scan(eng, rec.seq.data(), (int64_t)rec.seq.size(), 0, -1, &p, &res) != FASIM_OK) { fprintf(stderr, "fasim: %s\n", fasim_last_error(eng)); return 1; }
		if (stats) {
			const fasim_scan_stats& s = res->stats;
			fprintf(stderr, "[fasim] record %zu (%s): %lld segments (%lld skipped), %lld units, %lld candidates, %lld align calls, "
				"%.3f s total (stage1 %.3f, stage2 %.3f, stage3 %.3f, host %.3f), %.2f logical Gcells/s\n", nrec, rec.chr.c_str(), (long long)s.segments,
				(long long)s.segments_skipped, (long long)s.units, (long long)s.candidates, (long long)s.align_calls, s.t_total_s,
				s.t_stage1_s, s.t_stage2_s, s.t_stage3_s, s.t_host_s, s.logical_cells / s.t_total_s / 1e9);
		}
		const std::string stem = outdir + "/" + rec.species + "-" + lnc_name + "-" + base + (all_records ? "." + rec.chr : std::string());
		char* text = nullptr; int64_t len = 0;
		if (fasim_tfosorted(res->recs, res->count, res->pool, res->pool_len, rec.chr.c_str(), rec.start, &p, &text, &len) != FASIM_OK) { fprintf(stderr, "fasim: %s\n", fasim_last_error(nullptr)); return 1; }
		{ std::ofstream of((stem + "-TFOsorted").c_str(), std::ios::trunc); of.write(text, (std::streamsize)len); }
		fasim_free(text);
		for (int level = 1; level <= 2; level++) {   // print_cluster x2 (:832-836): <prefix>-TFOclass<level>-<ds>-<lg> (:706)
			if (fasim_tfoclass(res->recs, res->count, level, rec.chr.c_str(), rec.start, (int64_t)rec.seq.size(),
				lnc_name.c_str(), &p, &text, &len) != FASIM_OK) { fprintf(stderr, "fasim: %s\n", fasim_last_error(nullptr)); return 1; }
			const std::string cpath = stem + "-TFOclass" + std::to_string(level) + "-" + std::to_string(p.cDistance) + "-" + std::to_string(p.cLength);
			std::ofstream cf(cpath.c_str(), std::ios::trunc);
			cf.write(text, (std::streamsize)len);
			fasim_free(text);
		}
		fasim_result_free(res);
		nrec++;
	}
	if (nrec == 0) { fprintf(stderr, "fasim: no record in DNA file %s\n", f1.c_str()); return 1; }
	fasim_engine_destroy(eng);
	std::cout << "finished normally" << std::endl;
	return 0;
}
