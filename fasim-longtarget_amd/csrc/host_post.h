// Host-side half of the path: peak picking, the window policy, triplex
// construction, de-duplication, clustering and the -TFOsorted writer.  Integer/float order follows the
// reference exactly (file:line cited at each function in host_post.cpp).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/fasim_hip.h"

namespace fasim {

struct EncInfo { int strand, para, rule; bool reversed; };
EncInfo enc_info(int enc);
const char* rule_out(int enc);                       // outputs for DNA letters A,T,G,C
void build_enc_lut(uint8_t* lut /*[48][256]*/);      // DNA byte -> target code (A0 C1 G2 T3 N4)
std::vector<int> enabled_encodings(const fasim_params& p);
void encode_unit_host(const char* seg, int n, int enc, std::string& target, std::string& src);
bool same_seq(const char* seg, int n);

struct Cand { int score, pos; };
void pick_candidates(const uint32_t* hits, int nhits, std::vector<Cand>& out);   // hits = (pos<<8)|score, ascending pos

struct AlignResult {            // CIGAR ops live in a pool owned by the caller: pool[cigar_off .. cigar_off+cigar_len)
	int sw_score = 0, ref_begin = 0, ref_end = 0, query_begin = 0, query_end = 0;
	int cigar_len = 0;
	uint32_t cigar_off = 0;
	int failed = 0;             // 1: the reference's ssw_align returns NULL here (banded_sw found no path): sw_score is 0
};

// window length tried at iteration `it` (0..3) for a candidate (fastsim.h:204-211); returns false when the
// loop `while (Iden <= 1)` has ended
bool window_for_try(int it, int cand_score, int cand_pos, int* cutlength);

struct HostTriplex {
	int stari, endi, starj, endj, strand, reverse, rule, nt;
	float score, identity, tri_score;
	std::string tfo, tts;
	int seg = 0, enc = 0;
	int cand = -1;                 // caller's index of the alignment this record came from (strings are filled in later)
	int middle = 0, center = 0, motif = 0, neartriplex = 0;
	long genomestart = 0, genomeend = 0;
	bool genome_set = false;          // genomestart / genomeend were filled in by the caller (accumulate mode: shifted coordinates, which may be 0)
};

// convertMyTriplex (fastsim.h:291-414): appends to `list` when nt >= ntMin
// with_strings = false leaves tfo/tts empty (everything the dedup and the filters look at is numeric): the scan builds
// the two strings only for the few records that survive them, by calling again with with_strings = true
void convert_triplex(const AlignResult& al, const uint32_t* cigar, const std::string& rna, const char* seg, int n, int enc,
	long dna_start, const fasim_params& p, std::vector<HostTriplex>& list, bool seg_acgtn, bool with_strings = true);
bool only_acgtn(const char* seg, int n);
float triplex_stability(char c1, char c2, int para);                    // triplex_score (sim.h:72-97)
// row f3: everything of SIM() after its first sweep (sim.h:572-1141) for one unit; `nodes` = the node list the forward sweep
// left (consumed).  Appends the unit's triplexes (nt within [ntMin, ntMax]) in the reference's order.  host_sim.cpp
void sim_finish_unit(const std::string& rna, const std::string& target, const std::string& src, long dna_start, long min_score,
	int enc, const fasim_params& p, std::vector<fasim_sim_node>& nodes, std::vector<HostTriplex>& out);
// The same as a state machine, one round at a time, so that the re-sweeps of all units of a slice can run on the device between
// the rounds (k_sim_resweep, sim.hip): next_round() = sim.h:572-883 (best node, linear-space traceback, triplex record); it returns
// false when the unit is finished.  *sweep: a re-sweep of the node's rectangle box = {m1, mm, n1, nn} follows (not after the last
// node); new_pairs (optional) receives the (query row, target column) pairs the traceback has just marked as used.  The caller
// then updates `nodes` / `floor_score` by the device kernel or by resweep_host().
struct SimUnit {
	std::vector<fasim_sim_node> nodes;
	std::vector<HostTriplex> out;
	long floor_score = 0;            // the reference's `min`: 0, then 1 (addnode's return value)
	long rounds_left = 0;
	SimUnit(const std::string& rna, const std::string& target, const std::string& src, long dna_start, long min_score, int enc,
		const fasim_params& p, std::vector<fasim_sim_node>&& nodes);
	~SimUnit();
	SimUnit(const SimUnit&) = delete;
	SimUnit& operator=(const SimUnit&) = delete;
	bool next_round(bool* sweep, int box[4], std::vector<std::pair<int, int>>* new_pairs);
	void resweep_host();
private:
	struct Impl; Impl* im;
};
// tail of fastSIM (fastsim.h:273-288): sort/unique/sort/unique/sort, top 50, identity/stability/nt filter
void dedup_top(std::vector<HostTriplex>& mine, const fasim_params& p, std::vector<HostTriplex>& out);

// The numbers of a record only (what the dedup and the filters look at), trivially copyable: the scan runs every candidate
// alignment (~3 * 10^7 per 50 Mb) through convert_triplex_num / dedup_top_num and builds a HostTriplex with its strings only
// for the survivors.  std::sort / std::unique make the same sequence of comparisons whatever the element type, so the
// (comparator-dependent) order is the one the HostTriplex version produces.
struct TriplexNum { int stari, endi, starj, endj, nt, cand; float score, identity, tri_score; };
void convert_triplex_num(const AlignResult& al, const uint32_t* cigar, const std::string& rna, const char* seg, int n, int enc,
	long dna_start, const fasim_params& p, std::vector<TriplexNum>& list, bool seg_acgtn);
void dedup_top_num(std::vector<TriplexNum>& mine, const fasim_params& p, std::vector<TriplexNum>& out);

void cluster_triplex(int dd, int length, std::vector<HostTriplex>& list);
std::string tfosorted_text(std::vector<HostTriplex>& list, const std::string& chr, long start_genome, const fasim_params& p);
// bedGraph of one class (print_cluster); `list` must have been clustered (tfosorted_text or cluster_triplex)
std::string tfoclass_text(const std::vector<HostTriplex>& list, int level, const std::string& chr, long start_genome,
	long dna_size, const std::string& rna_name, const fasim_params& p);

} // namespace fasim
