// fasim -- CLI driver with the reference's flags (initEnv(), Fasim-LongTarget.cpp:269-377) and output files
// (printResult(), :797-829), calling the HIP path through the C-ABI of libfasim_hip.so.
//
//   fasim -f1 DNA.fa -f2 RNA.fa [-r R] [-O outdir] [-c cut] [-o overlap] [-t strand] [-i identity]
//         [-S stability] [-ni ntmin] [-na ntmax] [-pc C] [-pt T] [-ds dist] [-lg len] [-cn n]
//   extras (none of them changes what is computed for an input the reference handles):
//     --device N            HIP device (default 0)
//     --devices LIST        several devices of one node, e.g. 0-7 or 0,1,2: the segments of every DNA record are cut
//                           into contiguous shards, one engine (host thread) per device, records merged in shard order
//     --all-records         scan EVERY record of a multi-record DNA file (a genome), one record in memory at a time, and
//                           write one set of output files per record: <species>-<lnc>-<f1 stem>.<chr>-TFOsorted / -TFOclass...
//     --accumulate-records  bug-compatible with the reference's reader (defect B1, Fasim-LongTarget.cpp:219-262): record
//                           k is scanned as the concatenation of records 1..k, later headers are parsed with the stale
//                           field counter, everything is written into ONE output set named after the first record
//     --upper               upper-case the DNA while reading (soft-masked genomes; the reference treats lower case as N)
//     --clamp-cluster       defined behaviour where the reference's clustering does not terminate (see fasim_hip.h)
//     --stats               timing/statistics on stderr (parse, scan, tail, write)
//   -f2 may hold several lncRNAs (one '>' record each): every lncRNA is scanned against the DNA record while it is
//   resident (fasim_scan_queries) and gets its own output set.  A single-record -f2 behaves exactly like the reference.
//
// Differences, all documented in DESIGN.md: without --all-records / --accumulate-records only the first record of a
// multi-record DNA file is scanned; -d is parsed and ignored as in the reference.  -F (classic SIM): the forward sweep runs
// on the GPU, the rest of SIM() on host threads (first version of that path).
#include <getopt.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <deque>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fasim_hip.h"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct DnaRecord { std::string species, chr, seq; long start = 0; };

static void strip_eol(std::string& s) { s.erase(std::remove(s.begin(), s.end(), '\r'), s.end()); s.erase(std::remove(s.begin(), s.end(), '\n'), s.end()); }

// header '>species|chr|start-end' as readDna() parses it (Fasim-LongTarget.cpp:226-255).  `field` is the reference's
// counter j: it is never reset there, so in --accumulate-records mode it is carried from header to header and the
// second and later headers keep the first record's species / chr and get atoi("species|chr|start") as their start.
struct HeaderParser {
	int field = 0;
	std::string species, chr, start;
	void parse(const std::string& line)
	{
		std::string tmp;
		for (char c : line) {
			if (c == '>') { tmp.clear(); continue; }
			if (c == '|' && field == 0) { species = tmp; field++; tmp.clear(); continue; }
			if (c == '|' && field == 1) { chr = tmp; field++; tmp.clear(); continue; }
			if (c == '-' && field == 2) { start = tmp; tmp.clear(); continue; }
			tmp += c;
		}
	}
};

// streaming FASTA reader: one record at a time (a genome never sits in memory as a whole)
struct DnaReader {
	std::ifstream in; std::string pending; bool have_pending = false; bool upper = false; bool sticky_fields = false;
	HeaderParser hp;
	bool open(const std::string& path) { in.open(path); return (bool)in; }
	bool next(DnaRecord& r)
	{
		std::string line;
		if (!have_pending) {
			while (std::getline(in, line)) if (!line.empty() && line[0] == '>') { pending = line; have_pending = true; break; }
			if (!have_pending) return false;
		}
		r = DnaRecord();
		if (!sticky_fields) hp = HeaderParser();
		hp.parse(pending);
		r.species = hp.species; r.chr = hp.chr; r.start = atoi(hp.start.c_str());
		have_pending = false;
		while (std::getline(in, line)) {
			if (!line.empty() && line[0] == '>') { pending = line; have_pending = true; break; }
			strip_eol(line); r.seq += line;
		}
		if (upper) fasim_upper_case(&r.seq[0], (int64_t)r.seq.size());
		return true;
	}
};

struct Rna { std::string name, seq; };

// One '>' record: exactly readRna() (Fasim-LongTarget.cpp:174-200): the name is the first line without '>' characters.
// Several records: one lncRNA each (the reference would glue the later header lines into the sequence).
static bool read_rnas(const std::string& path, std::vector<Rna>& out)
{
	std::ifstream in(path);
	if (!in) return false;
	std::string line;
	bool first = true;
	while (std::getline(in, line)) {
		if (first || (!line.empty() && line[0] == '>')) {
			Rna r;
			for (char c : line) if (c != '>') r.name += c;
			strip_eol(r.name);
			out.push_back(r);
			first = false;
			continue;
		}
		strip_eol(line); out.back().seq += line;
	}
	return !out.empty();
}

// "0-7", "0,1,2", "0,0,0"
static std::vector<int> parse_devices(const char* s)
{
	std::vector<int> v;
	const char* p = s;
	while (*p) {
		char* e = nullptr;
		const long a = strtol(p, &e, 10);
		if (e == p) break;
		long b = a;
		p = e;
		if (*p == '-') { b = strtol(p + 1, &e, 10); p = e; }
		for (long k = a; k <= b; k++) v.push_back((int)k);
		if (*p == ',') p++;
	}
	return v;
}

// returns 0 when every byte reached the file (a missing -O directory or a full disk must not end in "finished normally")
static int write_file(const std::string& path, const char* text, int64_t len)
{
	std::ofstream of(path.c_str(), std::ios::trunc);
	if (of) of.write(text, (std::streamsize)len);
	of.close();
	if (!of) { fprintf(stderr, "fasim: cannot write %s\n", path.c_str()); return 1; }
	return 0;
}

struct Timers { double parse = 0, scan = 0, tail = 0, write = 0, tail_wait = 0; };

// -TFOsorted + the two -TFOclass files of one lncRNA (printResult(), Fasim-LongTarget.cpp:797-836): one clustering, three texts
static int write_outputs(const fasim_result* res, const std::string& stem, const std::string& chr, long start, int64_t dna_len,
	const std::string& lnc_name, const fasim_params& p, int flags, Timers& tm)
{
	char* text[3] = { nullptr, nullptr, nullptr }; int64_t len[3] = { 0, 0, 0 };
	double t0 = now_s();
	if (fasim_tail_outputs(res->recs, res->count, res->pool, res->pool_len, chr.c_str(), start, dna_len, lnc_name.c_str(), &p, flags,
		&text[0], &len[0], &text[1], &len[1], &text[2], &len[2]) != FASIM_OK) { fprintf(stderr, "fasim: %s\n", fasim_last_error(nullptr)); return 1; }
	tm.tail += now_s() - t0; t0 = now_s();
	int bad = write_file(stem + "-TFOsorted", text[0], len[0]);
	for (int level = 1; level <= 2; level++)     // print_cluster x2 (:832-836): <prefix>-TFOclass<level>-<ds>-<lg> (:706)
		bad |= write_file(stem + "-TFOclass" + std::to_string(level) + "-" + std::to_string(p.cDistance) + "-" + std::to_string(p.cLength), text[level], len[level]);
	for (char* t : text) fasim_free(t);
	tm.write += now_s() - t0;
	return bad;
}

// The host tail of record i (clustering, text, file writes) runs on its own thread while the devices already scan record
// i+1; at most `kMaxPending` results wait to be written.
static std::mutex g_out_mu;
static int g_out_failed = 0;

// Scans one DNA record with every lncRNA on every device: device d takes the d-th contiguous block of segments
// (SURVEY 8(e)); per lncRNA the shard results are merged in shard order, which is the reference's canonical order.
static int scan_record(const std::vector<fasim_engine*>& engines, const std::vector<Rna>& rnas, const std::string& dna, const fasim_params& p,
	std::vector<fasim_result*>& out)
{
	const int nd = (int)engines.size(), nq = (int)rnas.size();
	std::vector<const char*> qp((size_t)nq); std::vector<int32_t> ql((size_t)nq);
	for (int q = 0; q < nq; q++) { qp[(size_t)q] = rnas[(size_t)q].seq.data(); ql[(size_t)q] = (int32_t)rnas[(size_t)q].seq.size(); }
	out.assign((size_t)nq, nullptr);
	if (nd == 1) {
		if (fasim_scan_queries(engines[0], qp.data(), ql.data(), nq, dna.data(), (int64_t)dna.size(), 0, -1, &p, out.data()) != FASIM_OK) { fprintf(stderr, "fasim: %s\n", fasim_last_error(engines[0])); return 1; }
		return 0;
	}
	const int64_t nseg = fasim_segment_count((int64_t)dna.size(), &p);
	std::vector<std::vector<fasim_result*>> part((size_t)nd, std::vector<fasim_result*>((size_t)nq, nullptr));
	std::vector<int> rc((size_t)nd, 0);
	std::vector<std::thread> th;
	for (int d = 0; d < nd; d++) {
		th.emplace_back([&, d] {
			const int64_t base = nseg / nd, rem = nseg % nd;
			const int64_t first = d * base + std::min<int64_t>(d, rem), count = base + (d < rem ? 1 : 0);
			rc[(size_t)d] = fasim_scan_queries(engines[(size_t)d], qp.data(), ql.data(), nq, dna.data(), (int64_t)dna.size(), first, count, &p, part[(size_t)d].data());
		});
	}
	for (auto& t : th) t.join();
	int bad = 0;
	for (int d = 0; d < nd; d++) if (rc[(size_t)d] != FASIM_OK) { fprintf(stderr, "fasim: device shard %d: %s\n", d, fasim_last_error(engines[(size_t)d])); bad = 1; }
	for (int q = 0; q < nq && !bad; q++) {
		std::vector<const fasim_triplex*> recs((size_t)nd); std::vector<int64_t> counts((size_t)nd), plens((size_t)nd); std::vector<const char*> pools((size_t)nd);
		for (int d = 0; d < nd; d++) { const fasim_result* r = part[(size_t)d][(size_t)q]; recs[(size_t)d] = r->recs; counts[(size_t)d] = r->count; pools[(size_t)d] = r->pool; plens[(size_t)d] = r->pool_len; }
		if (fasim_merge_results(recs.data(), counts.data(), pools.data(), plens.data(), nd, &out[(size_t)q]) != FASIM_OK) { fprintf(stderr, "fasim: %s\n", fasim_last_error(nullptr)); bad = 1; break; }
		// statistics of the merged result: sums over the shards (times: the slowest shard)
		fasim_scan_stats& st = out[(size_t)q]->stats;
		for (int d = 0; d < nd; d++) {
			const fasim_scan_stats& x = part[(size_t)d][(size_t)q]->stats;
			st.segments += x.segments; st.segments_skipped += x.segments_skipped; st.units += x.units; st.candidates += x.candidates;
			st.align_calls += x.align_calls; st.logical_cells += x.logical_cells; st.hazard_units += x.hazard_units;
			st.t_total_s = std::max(st.t_total_s, x.t_total_s);
		}
	}
	for (auto& v : part) for (fasim_result* r : v) fasim_result_free(r);
	return bad;
}

int main(int argc, char* const* argv)
{
	fasim_params p; fasim_params_default(&p);
	std::string f1 = "./", f2 = "./", outdir = "./";
	std::vector<int> devices(1, 0);
	bool stats = false, all_records = false, accumulate = false, upper = false;
	int tail_flags = 0;
	const char* optstring = "f:s:r:O:c:m:t:i:S:z:Y:Z:h:C:D:E:o:y:Fd";
	struct option lo[] = {
		{ "f1", required_argument, NULL, 'f' }, { "f2", required_argument, NULL, 's' }, { "ni", required_argument, NULL, 'y' },
		{ "na", required_argument, NULL, 'z' }, { "pc", required_argument, NULL, 'Y' }, { "pt", required_argument, NULL, 'Z' },
		{ "cn", required_argument, NULL, 'C' }, { "ds", required_argument, NULL, 'D' }, { "lg", required_argument, NULL, 'E' },
		{ "device", required_argument, NULL, 1001 }, { "stats", no_argument, NULL, 1002 }, { "all-records", no_argument, NULL, 1003 },
		{ "devices", required_argument, NULL, 1004 }, { "accumulate-records", no_argument, NULL, 1005 }, { "upper", no_argument, NULL, 1006 },
		{ "clamp-cluster", no_argument, NULL, 1007 }, { 0, 0, 0, 0 } };
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
		case 'C': break;                                  // -cn only picked a result vector in the reference (:129-163); see --devices
		case 'F': p.classicSim = 1; break;                // doFastSim = false (Fasim-LongTarget.cpp:360-362): SIM() instead of fastSIM()
		case 'd': break;
		case 1001: devices.assign(1, atoi(optarg)); break;
		case 1002: stats = true; break;
		case 1003: all_records = true; break;
		case 1004: devices = parse_devices(optarg); if (devices.empty()) { fprintf(stderr, "fasim: bad --devices list\n"); return 2; } break;
		case 1005: accumulate = true; break;
		case 1006: upper = true; break;
		case 1007: tail_flags |= FASIM_TAIL_CLAMP_CLUSTER; break;
		default: fprintf(stderr, "usage: fasim -f1 DNA.fa -f2 RNA.fa [-O outdir] [-r R] [-t T] [-lg L] ... [--devices 0-7] [--all-records] [--upper]\n"); return 2;
		}
	}
	if (all_records && accumulate) { fprintf(stderr, "fasim: --all-records and --accumulate-records exclude each other\n"); return 2; }
	Timers tm;
	const double t_start = now_s();
	std::vector<Rna> rnas;
	DnaReader reader;
	reader.upper = upper; reader.sticky_fields = accumulate;
	if (!reader.open(f1)) { fprintf(stderr, "fasim: cannot read DNA file %s\n", f1.c_str()); return 1; }
	if (!read_rnas(f2, rnas)) { fprintf(stderr, "fasim: cannot read RNA file %s\n", f2.c_str()); return 1; }
	for (const Rna& r : rnas) if (r.seq.empty()) { fprintf(stderr, "fasim: empty RNA record '%s' in %s\n", r.name.c_str(), f2.c_str()); return 1; }
	std::cout << "Searching triplexes using Fasim" << std::endl;
	for (const Rna& r : rnas) std::cout << r.name << std::endl;

	std::vector<fasim_engine*> engines;
	for (int d : devices) {
		fasim_engine* e = nullptr;
		if (fasim_engine_create(d, &e) != FASIM_OK) { fprintf(stderr, "fasim: %s\n", fasim_last_error(nullptr)); return 1; }
		engines.push_back(e);
	}
	if (engines.size() > 1 && !getenv("FASIM_HOST_THREADS")) {
		// one process, several engines: every engine keeps six host threads per core of its share of the machine (at most the
		// single-engine default of 96; the host side of a batch is a burst that wants ~9 threads per batch in flight)
		const unsigned hc = std::max(1u, std::thread::hardware_concurrency());
		const int per = std::max(16, (int)std::min(96u, 6 * (hc / (unsigned)engines.size())));
		for (fasim_engine* e : engines) fasim_set_option(e, "host_threads", per);
	}

	// file name: <O>/<species>-<lncName>-<f1 minus 3 chars>-TFOsorted (:123, 800-802); with --all-records the record's
	// chr is appended to the stem so that the records of a genome do not overwrite each other
	const std::string base = f1.substr(0, f1.size() >= 3 ? f1.size() - 3 : 0);
	DnaRecord rec;
	size_t nrec = 0;
	int64_t total_nt = 0;
	std::deque<std::thread> pending;
	std::set<std::string> stems_seen;
	if (accumulate) {
		// B1: tmpDNA is never cleared, so record k holds records 1..k; all triplexes go into ONE list that is printed with
		// the first record's species / chr / start / length (main(), Fasim-LongTarget.cpp:133-166)
		std::vector<std::vector<fasim_result*>> per_rec;
		std::vector<long> starts;
		std::string first_species, first_chr, cumulative; long first_start = 0; int64_t first_len = 0;
		for (;;) {
			double t0 = now_s();
			if (!reader.next(rec)) break;
			tm.parse += now_s() - t0;
			cumulative += rec.seq;
			if (nrec == 0) { first_species = rec.species; first_chr = rec.chr; first_start = rec.start; first_len = (int64_t)cumulative.size(); }
			starts.push_back(rec.start);
			t0 = now_s();
			std::vector<fasim_result*> res;
			if (scan_record(engines, rnas, cumulative, p, res)) return 1;
			tm.scan += now_s() - t0;
			total_nt += (int64_t)cumulative.size();
			per_rec.push_back(res);
			nrec++;
		}
		for (size_t q = 0; q < rnas.size() && nrec; q++) {
			std::vector<const fasim_triplex*> recs(nrec); std::vector<int64_t> counts(nrec), plens(nrec); std::vector<const char*> pools(nrec);
			for (size_t k = 0; k < nrec; k++) {
				fasim_result* r = per_rec[k][q];
				for (int64_t i = 0; i < r->count; i++) r->recs[i].genome_shift = (int32_t)(starts[k] - first_start);
				recs[k] = r->recs; counts[k] = r->count; pools[k] = r->pool; plens[k] = r->pool_len;
			}
			fasim_result* merged = nullptr;
			if (fasim_merge_results(recs.data(), counts.data(), pools.data(), plens.data(), (int32_t)nrec, &merged) != FASIM_OK) { fprintf(stderr, "fasim: %s\n", fasim_last_error(nullptr)); return 1; }
			const std::string stem = outdir + "/" + first_species + "-" + rnas[q].name + "-" + base;
			if (write_outputs(merged, stem, first_chr, first_start, first_len, rnas[q].name, p, tail_flags, tm)) return 1;
			fasim_result_free(merged);
		}
		for (auto& v : per_rec) for (fasim_result* r : v) fasim_result_free(r);
	} else {
		for (;;) {
			double t0 = now_s();
			if (!reader.next(rec)) break;
			tm.parse += now_s() - t0;
			if (nrec > 0 && !all_records) {
				fprintf(stderr, "fasim: %s holds more than one record: only the first one was scanned (use --all-records; see DESIGN.md, B1)\n", f1.c_str());
				break;
			}
			t0 = now_s();
			std::vector<fasim_result*> res;
			if (scan_record(engines, rnas, rec.seq, p, res)) {
				// the tails of earlier records still run on their threads: a joinable std::thread must not be destroyed
				for (std::thread& t : pending) t.join();
				return 1;
			}
			tm.scan += now_s() - t0;
			total_nt += (int64_t)rec.seq.size();
			for (size_t q = 0; q < rnas.size(); q++) {
				if (stats) {
					const fasim_scan_stats& s = res[q]->stats;
					fprintf(stderr, "[fasim] record %zu (%s) x %s: %lld segments (%lld skipped), %lld units, %lld candidates, %lld align calls, %lld records\n",
						nrec, rec.chr.c_str(), rnas[q].name.c_str(), (long long)s.segments, (long long)s.segments_skipped, (long long)s.units,
						(long long)s.candidates, (long long)s.align_calls, (long long)res[q]->count);
				}
				const std::string stem = outdir + "/" + rec.species + "-" + rnas[q].name + "-" + base + (all_records ? "." + rec.chr : std::string());
				if (!stems_seen.insert(stem).second) fprintf(stderr, "fasim: warning: %s-TFOsorted is written twice (two lncRNAs or records of the same name): the later one wins\n", stem.c_str());
				// tail + write on a background thread: the next record is parsed and scanned meanwhile
				while (pending.size() >= 4) { pending.front().join(); pending.pop_front(); }
				fasim_result* r = res[q];
				const std::string chr = rec.chr, lname = rnas[q].name; const long start = rec.start; const int64_t dlen = (int64_t)rec.seq.size();
				pending.emplace_back([=, &tm, &p]() {
					Timers mine;
					const int bad = write_outputs(r, stem, chr, start, dlen, lname, p, tail_flags, mine);
					fasim_result_free(r);
					std::lock_guard<std::mutex> lk(g_out_mu);
					tm.tail += mine.tail; tm.write += mine.write; if (bad) g_out_failed = 1;
				});
			}
			nrec++;
		}
		const double t_wait = now_s();
		for (std::thread& t : pending) t.join();
		pending.clear();
		tm.tail_wait = now_s() - t_wait;
		if (g_out_failed) return 1;
	}
	if (nrec == 0) { fprintf(stderr, "fasim: no record in DNA file %s\n", f1.c_str()); return 1; }
	for (fasim_engine* e : engines) fasim_engine_destroy(e);
	if (stats) {
		const double total = now_s() - t_start;
		fprintf(stderr, "[fasim] end to end %.3f s: parse %.3f, scan %.3f, tail %.3f + write %.3f on background threads (%.3f s not hidden behind the next scan) (%zu DNA record(s), %lld nt, %zu lncRNA(s), %zu device shard(s)) = %.3f Mbp/s per lncRNA\n",
			total, tm.parse, tm.scan, tm.tail, tm.write, tm.tail_wait, nrec, (long long)total_nt, rnas.size(), engines.size(),
			(double)total_nt * (double)rnas.size() / total / 1e6);
	}
	std::cout << "finished normally" << std::endl;
	return 0;
}
