// oracle/fasim_oracle.h -- TEST INFRASTRUCTURE, not product code.
//
// CPU restatement (scalar, stripe-faithful) of Fasim-LongTarget's hot path, used ONLY as a
// checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
// (fasim-longtarget_amd/) never includes, links or calls anything in this directory.
//
// Parity status: PINNED.  Every function below is checked against the reference
// implementation compiled from /root/reference (oracle/_ref/ref_probe, oracle/_ref/fasim_ref)
// and against the golden fixtures those produced (tests/golden/, tests/golden/make_golden.py).
#ifndef FASIM_ORACLE_H
#define FASIM_ORACLE_H

#include <stdint.h>
#include <string>
#include <vector>

namespace fo {

// ---- a3: rule encodings (rules.h:6-53, 59-93, 94-318; Fasim-LongTarget.cpp:404-585) -------
struct EncInfo { int strand; int para; int rule; bool reversed; };
EncInfo enc_info(int enc);                       // enc in [0,48), canonical execution order
// target = what the lncRNA is aligned to; src = strand shown as TTS / used for stability
void encode_unit(const std::string& seg, int enc, std::string& target, std::string& src);

// ---- a1/a2: segmenter (fastsim.h:71-90) and same_seq (Fasim-LongTarget.cpp:873-933) --------
void cut_sequence(const std::string& dna, int cut, int overlap, std::vector<int>& starts);
bool same_seq(const std::string& seg);

// ---- a4: stage-1 exact max local score (stats.h:879-956) -----------------------------------
int stage1_max(const std::string& rna, const std::string& target);

// ---- a5/a6: stage-2 column maxima (sswNew.cpp:176-201, 255-464, 1309-1437) ----------------
std::vector<int> pre_align(const std::string& rna, const std::string& target);

// ---- a7: peak picking (ssw_cpp.cpp:427-572) -------------------------------------------------
struct Cand { int score; int pos; };
std::vector<Cand> pick_candidates(const std::vector<int>& cols, int threshold);

// ---- a9-a11: window alignment (sswNew.cpp:476-672, 893-1069, 1071-1259, 1446-1547) --------
struct Alignment {
	int sw_score = 0, ref_begin = 0, ref_end = 0, query_begin = 0, query_end = 0;
	std::vector<uint32_t> cigar;   // BAM encoding (len<<4 | op), op 0=M 1=I 2=D
	bool tainted = false;          // traceback touched memory the reference leaves undefined
	std::string cigar_string() const;
};
Alignment align(const std::string& rna, const std::string& window);

// ---- a8, a12-a14: fastSIM for one unit (fastsim.h:158-289, 291-414, 416-560) ---------------
struct Triplex {
	int stari, endi, starj, endj, reverse, strand, rule, nt;
	float score, identity, tri_score;
	std::string stri_align, strj_align;
	int middle = 0, center = 0, motif = 0, neartriplex = 0;
	long genomestart = 0, genomeend = 0;
	std::string chr;
};
struct Params {   // Fasim-LongTarget.cpp:284-303 defaults
	int rule = 0, cutLength = 5000, strand = 0, overlapLength = 100;
	int ntMin = 20, ntMax = 100000;
	float scoreMin = 0.0f, minIdentity = 60.0f, minStability = 1.0f;
	int penaltyT = -1000, penaltyC = 0, cDistance = 15, cLength = 50;
	bool classicSim = false;   // -F: doFastSim = false (Fasim-LongTarget.cpp:360-362): SIM() instead of fastSIM()
};
struct TryRecord { int it, L; Alignment a; };
struct UnitTrace {      // everything the probe prints for one unit
	int seg, enc; long dna_start; int n, stage1, thr; uint64_t colhash; int nhits;
	std::vector<std::pair<int, int>> hits;      // (pos, score) above threshold
	std::vector<Cand> cands;
	std::vector<std::vector<TryRecord>> tries;   // per candidate
	std::vector<Triplex> triplexes;              // fastSIM output for the unit
};
void fast_sim_unit(const std::string& rna, const std::string& target, const std::string& src,
	long dna_start, int min_score, int strand, int para, int rule, const Params& p,
	std::vector<Triplex>& out, UnitTrace* trace);

// ---- row f3: the -F path, classic SIM (sim.h:99-1143; oracle/fasim_sim_oracle.cpp) -----------------
float triplex_score(char c1, char c2, int para);                       // sim.h:72-97
struct SimNode { long score, stari, starj, endi, endj, top, bot, left, right; };   // vertex (sim.h:47-58)
// SIM() for one unit: appends the unit's triplexes (nt within [ntMin, ntMax]) in the reference's order
void sim_unit(const std::string& rna, const std::string& target, const std::string& src, long dna_start, long min_score,
	int strand, int para, int rule, const Params& p, std::vector<Triplex>& out);
// only the first sweep of SIM() (sim.h:506-571): the node list it leaves (checker of the HIP forward pass)
void sim_forward_nodes(const std::string& rna, const std::string& target, long min_score, std::vector<SimNode>& nodes);

// ---- whole scan: LongTarget() (Fasim-LongTarget.cpp:379-598) --------------------------------
// Units are enumerated in canonical (segment, encoding) order; `traces` (optional) gets one entry
// per executed unit; `skipped` (optional) the indices of same_seq segments.
void long_target(const Params& p, const std::string& rna, const std::string& dna,
	std::vector<Triplex>& out, std::vector<UnitTrace>* traces, std::vector<int>* skipped,
	int seg_first = 0, int seg_count = 1 << 30, int threads = 1);

// ---- a15/a16: genome coordinates, clustering, -TFOsorted text -------------------------------
void assign_genome(std::vector<Triplex>& list, const std::string& chr, long start_genome);
void cluster_triplex(int dd, int length, std::vector<Triplex>& list);
std::string tfosorted_text(std::vector<Triplex>& list, const Params& p);   // clusters + sorts `list`
// bedGraph text of print_cluster() for class `level`; call after tfosorted_text().  start_genome_m1 = start_genome - 1.
std::string tfoclass_text(const std::vector<Triplex>& list, int level, const std::string& chr, long start_genome_m1,
	long dna_size, const std::string& rna_name, const Params& p);

uint64_t fnv1a_ints(const int* v, int n);
bool read_fasta(const char* path, std::string& header, std::string& seq);
void parse_dna_header(const std::string& header, std::string& species, std::string& chr, long& start);

} // namespace fo

// plain C entry points for ctypes (tests/ only)
extern "C" {
int  fo_stage1_max(const char* rna, int m, const char* target, int n);
void fo_pre_align(const char* rna, int m, const char* target, int n, int* out_cols);
int  fo_pick_candidates(const int* cols, int n, int thr, int* out_score, int* out_pos, int cap);
// out[0..4] = score, ref_begin, ref_end, query_begin, query_end; returns cigar length (<= cap) or -1
int  fo_align(const char* rna, int m, const char* window, int n, int* out5, uint32_t* cigar, int cap);
void fo_encode_unit(const char* seg, int n, int enc, char* target, char* src);
// first sweep of SIM(): out[9 * k ..] = score, stari, starj, endi, endj, top, bot, left, right of node k; returns the node count
int  fo_sim_forward_nodes(const char* rna, int m, const char* target, int n, long min_score, long* out, int cap);
}

#endif
