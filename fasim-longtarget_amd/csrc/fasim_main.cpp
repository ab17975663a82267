// fasim -- CLI driver with the reference's flags (initEnv(), Fasim-LongTarget.cpp:269-377) and output files
// (printResult(), :797-829), calling the HIP path through the C-ABI of libfasim_hip.so.
//
//   fasim -f1 DNA.fa -f2 RNA.fa [-r R] [-O outdir] [-c cut] [-o overlap] [-t strand] [-i identity]
//         [-S stability] [-ni ntmin] [-na ntmax] [-pc C] [-pt T] [-ds dist] [-lg len] [-cn n]
//   extras: --device N, --stats (timing/statistics on stderr)
//
// Differences, all documented in DESIGN.md: multi-record FASTA files are read record by record (the
// reference accumulates them, defect B1); -F (classic SIM) and -d are not supported; the -TFOclass
// bedGraph files are not written.
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

static bool read_dna(const std::string& path, std::vector<DnaRecord>& recs)
{
	std::ifstream in(path);
	if (!in) return false;
	std::string line;
	while (std::getline(in, line)) {
		if (!line.empty() && line[0] == '>') { recs.emplace_back(); parse_header(line, recs.back()); }
		else if (!recs.empty()) { strip_eol(line); recs.back().seq += line; }
	}
	return true;
}

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
	int device = 0; bool stats = false;
	const char* optstring = "f:s:r:O:c:m:t:i:S:z:Y:Z:h:C:D:E:o:y:Fd";
	struct option lo[] = {
		{ "f1", required_argument, NULL, 'f' }, { "f2", required_argument, NULL, 's' }, { "ni", required_argument, NULL, 'y' },
		{ "na", required_argument, NULL, 'z' }, { "pc", required_argument, NULL, 'Y' }, { "pt", required_argument, NULL, 'Z' },
		{ "cn", required_argument, NULL, 'C' }, { "ds", required_argument, NULL, 'D' }, { "lg", required_argument, NULL, 'E' },
		{ "device", required_argument, NULL, 1001 }, { "stats", no_argument, NULL, 1002 }, { 0, 0, 0, 0 } };
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
		default: fprintf(stderr, "usage: fasim -f1 DNA.fa -f2 RNA.fa [-O outdir] [-r R] [-t T] [-lg L] ...\n"); return 2;
		}
	}
	std::vector<DnaRecord> recs; std::string lnc_name, rna;
	if (!read_dna(f1, recs) || recs.empty()) { fprintf(stderr, "fasim: cannot read DNA file %s\n", f1.c_str()); return 1; }
	if (!read_rna(f2, lnc_name, rna) || rna.empty()) { fprintf(stderr, "fasim: cannot read RNA file %s\n", f2.c_str()); return 1; }
	std::cout << "Searching triplexes using Fasim" << std::endl << lnc_name << std::endl;

	fasim_engine* eng = nullptr;
	if (fasim_engine_create(device, &eng) != FASIM_OK) { fprintf(stderr, "fasim: %s\n", fasim_last_error(nullptr)); return 1; }
	if (fasim_set_query(eng, rna.data(), (int)rna.size()) != FASIM_OK) { fprintf(stderr, "fasim: %s\n", fasim_last_error(eng)); return 1; }

	// records of all FASTA records, with genome coordinates fixed per record (main(), :141-149)
	std::vector<fasim_triplex> all; std::string pool;
	// the reference prints ONE file named after the first record (:164-166); later records reuse its chr only
	// through the per-row chr field, which we keep per record by writing one row block per record.
	std::string text_all, text_class[2];
	for (size_t r = 0; r < recs.size(); r++) {
		fasim_result* res = nullptr;
		if (fasim_scan(eng, recs[r].seq.data(), (int64_t)recs[r].seq.size(), 0, -1, &p, &res) != FASIM_OK) { fprintf(stderr, "fasim: %s\n", fasim_last_error(eng)); return 1; }
		if (stats) {
			const fasim_scan_stats& s = res->stats;
			fprintf(stderr, "[fasim] record %zu: %lld segments (%lld skipped), %lld units, %lld candidates, %lld align calls, "
				"%.3f s total (stage1 %.3f, stage2 %.3f, stage3 %.3f, host %.3f), %.2f logical Gcells/s\n", r, (long long)s.segments,
				(long long)s.segments_skipped, (long long)s.units, (long long)s.candidates, (long long)s.align_calls, s.t_total_s,
				s.t_stage1_s, s.t_stage2_s, s.t_stage3_s, s.t_host_s, s.logical_cells / s.t_total_s / 1e9);
		}
		if (recs.size() == 1) {
			char* text = nullptr; int64_t len = 0;
			if (fasim_tfosorted(res->recs, res->count, res->pool, res->pool_len, recs[r].chr.c_str(), recs[r].start, &p, &text, &len) != FASIM_OK) { fprintf(stderr, "fasim: %s\n", fasim_last_error(nullptr)); return 1; }
			text_all.assign(text, (size_t)len);
			fasim_free(text);
			for (int level = 1; level <= 2; level++) {   // print_cluster x2 (:832-836)
				if (fasim_tfoclass(res->recs, res->count, level, recs[r].chr.c_str(), recs[r].start, (int64_t)recs[r].seq.size(),
					lnc_name.c_str(), &p, &text, &len) != FASIM_OK) { fprintf(stderr, "fasim: %s\n", fasim_last_error(nullptr)); return 1; }
				text_class[level - 1].assign(text, (size_t)len);
				fasim_free(text);
			}
		} else {
			fprintf(stderr, "fasim: multi-record DNA files: only the first record is written (see DESIGN.md, B1)\n");
			if (r == 0) {
				char* text = nullptr; int64_t len = 0;
				if (fasim_tfosorted(res->recs, res->count, res->pool, res->pool_len, recs[r].chr.c_str(), recs[r].start, &p, &text, &len) != FASIM_OK) { fprintf(stderr, "fasim: %s\n", fasim_last_error(nullptr)); return 1; }
				text_all.assign(text, (size_t)len);
				fasim_free(text);
			}
		}
		fasim_result_free(res);
	}
	// file name: <O>/<species>-<lncName>-<f1 minus 3 chars>-TFOsorted (:123, 800-802)
	const std::string base = f1.substr(0, f1.size() >= 3 ? f1.size() - 3 : 0);
	const std::string path = outdir + "/" + recs[0].species + "-" + lnc_name + "-" + base + "-TFOsorted";
	std::ofstream of(path.c_str(), std::ios::trunc);
	of << text_all;
	of.close();
	for (int level = 1; level <= 2 && recs.size() == 1; level++) {   // <prefix>-TFOclass<level>-<ds>-<lg> (:706)
		const std::string cpath = path.substr(0, path.size() - 10) + "-TFOclass" + std::to_string(level) + "-" +
			std::to_string(p.cDistance) + "-" + std::to_string(p.cLength);
		std::ofstream cf(cpath.c_str(), std::ios::trunc);
		cf << text_class[level - 1];
	}
	fasim_engine_destroy(eng);
	std::cout << "finished normally" << std::endl;
	return 0;
}
