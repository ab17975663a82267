// oracle/fasim_sim_oracle.cpp -- TEST INFRASTRUCTURE, not product code.
//
// CPU restatement of the reference's "-F" path: the classic SIM algorithm (Huang & Miller: the K best non-intersecting
// local alignments in linear space) as LongTarget uses it, /root/reference/sim.h:99-1143, called from LongTarget() at
// Fasim-LongTarget.cpp:420-426 (and the seven sibling call sites) with match 5, mismatch -4, gap open 12, gap extension 4.
//
// Parity status: PINNED by tests/golden/demoF.simscan.gz, simF12k.simscan.gz (the reference's own SIM() per unit, printed
// by oracle/ref_probe.cpp `simscan`) and demoF_lg40.TFOsorted / .TFOclass{1,2} (`fasim_ref -F`), see
// tests/test_oracle_golden.py.
//
// Behaviour of the reference that is restated on purpose (each marked [ref] below):
//   * scores are carried times 10 (V = 50 / -40, Q = 120, R = 40), but the forward sweep compares them with the UNSCALED
//     threshold min_score (sim.h:567), so almost every cell with two adjacent matches reaches the node list; the real
//     threshold is applied after the traceback (sim.h:594, score / 10.0 <= min_score)
//   * a new start point always replaces the lowest-scoring of the K = 50 nodes, whatever its own score (sim.h:129-137)
//   * addnode() returns 1, so the later sweeps use "c > 1" as their threshold after the first insertion (sim.h:147, 1137)
//   * nt is the lncRNA span endi - stari + 1, not the alignment length (sim.h:592)
//   * ties are broken towards the larger start point (ORDER, sim.h:481-493)
// Letters outside ACGT index uninitialised entries of the reference's score table (sim.h:464-468): inputs of the fixtures
// hold ACGT only.
#include "fasim_oracle.h"

#include <algorithm>
#include <cstring>
#include <list>

namespace fo {

namespace {

const int KNODES = 50;                          // sim.h:17

struct Tri { long s, i, j; };                   // score and the start point it comes from
// ORDER (sim.h:481-493): a takes b when b is lexicographically larger in (score, start row, start column)
inline void take_better(Tri& a, const Tri& b)
{
	if (a.s < b.s || (a.s == b.s && (a.i < b.i || (a.i == b.i && a.j < b.j)))) a = b;
}

struct Sim {
	const char* A; const char* B;               // 1-based: A[1..M] lncRNA, B[1..N] target
	long M, N;
	long sub[128][128];                         // V (sim.h:464-468): 50 on the ACGT diagonal, -40 otherwise
	long Q, R;
	std::vector<SimNode> nodes;                 // LIST
	std::vector<long> CC, DD, RR, SS, EE, FF;   // indexed by column
	std::vector<long> HH, WW, II, JJ, XX, YY;   // indexed by row
	std::vector<std::list<long>> used;          // row[i]: columns j already aligned to row i (sim.h:462, DIAG)
	std::vector<long> script; long* sapp = nullptr; long last = 0; long I = 0, J = 0;

	bool is_used(long i, long j) const
	{
		for (long x : used[(size_t)i]) if (x == j) return true;
		return false;
	}

	// addnode (sim.h:99-148)
	void add_node(long c, long ci, long cj, long i, long j)
	{
		for (SimNode& n : nodes) {
			if (n.stari == ci && n.starj == cj) {
				if (n.score < c) { n.score = c; n.endi = i; n.endj = j; }
				n.top = std::min(n.top, i); n.bot = std::max(n.bot, i); n.left = std::min(n.left, j); n.right = std::max(n.right, j);
				return;
			}
		}
		SimNode fresh{ c, ci, cj, i, j, i, i, j, j };
		if ((int)nodes.size() == KNODES) {
			size_t low = 0;                         // [ref] the first node with the smallest score is overwritten
			for (size_t d = 1; d < nodes.size(); d++) if (nodes[d].score < nodes[low].score) low = d;
			nodes[low] = fresh;
		} else nodes.push_back(fresh);
	}

	// The cell recurrence shared by all five sweeps of SIM() (sim.h:522-566 and its four repetitions).  `c`, `f`, `p` are the
	// running values of the sweep line; (Cv, Cs_i, Cs_j) / (Dv, Ds_i, Ds_j) the stored state of the perpendicular direction.
	struct Line { Tri c, f, p; };
	inline Tri cell(Line& ln, long& Cv, long& Ci, long& Cj, long& Dv, long& Di, long& Dj, long i, long j, long match, Tri* d_out)
	{
		ln.f.s -= R;
		Tri c = ln.c; c.s = c.s - Q - R;
		take_better(ln.f, c);
		c = Tri{ Cv - Q - R, Ci, Cj };
		Tri d{ Dv - R, Di, Dj };
		take_better(d, c);
		long v = 0;
		if (!is_used(i, j)) v = ln.p.s + match;     // DIAG: an aligned pair cannot be used twice
		if (v <= 0) c = Tri{ 0, i, j }; else c = Tri{ v, ln.p.i, ln.p.j };
		take_better(c, d);
		take_better(c, ln.f);
		ln.p = Tri{ Cv, Ci, Cj };
		Cv = c.s; Ci = c.i; Cj = c.j;
		Dv = d.s; Di = d.i; Dj = d.j;
		ln.c = c;
		if (d_out) *d_out = d;
		return c;
	}

	// ---- linear-space traceback: diff() (sim.h:167-346) ---------------------------------------------------------------
	long gap(long k) const { return k <= 0 ? 0 : Q + R * k; }
	void op_del(long k) { I += k; if (last < 0) last = sapp[-1] -= k; else last = *sapp++ = -k; }
	void op_ins(long k) { J += k; if (last < 0) { sapp[-1] = k; *sapp++ = last; } else last = *sapp++ = k; }
	void op_rep() { last = *sapp++ = 0; }

	long diff(const char* a, const char* b, long m, long n, long tb, long te)
	{
		if (n <= 0) { if (m > 0) op_del(m); return -gap(m); }
		if (m <= 1) {
			if (m <= 0) { op_ins(n); return -gap(n); }
			if (tb > te) tb = te;
			long midc = -(tb + R + gap(n)), midj = 0;
			const long* va = sub[(int)a[1]];
			for (long j = 1; j <= n; j++) {
				if (is_used(I + 1, j + J)) continue;
				const long c = va[(int)b[j]] - (gap(j - 1) + gap(n - j));
				if (c > midc) { midc = c; midj = j; }
			}
			if (midj == 0) { op_ins(n); op_del(1); }
			else {
				if (midj > 1) op_ins(midj - 1);
				op_rep();
				I++; J++;
				used[(size_t)I].push_back(J);
				if (midj < n) op_ins(n - midj);
			}
			return midc;
		}
		const long midi = m / 2;
		// forward half: CC / DD over rows 1..midi
		CC[0] = 0;
		long t = -Q;
		for (long j = 1; j <= n; j++) { CC[(size_t)j] = t = t - R; DD[(size_t)j] = t - Q; }
		t = -tb;
		for (long i = 1; i <= midi; i++) {
			long s = CC[0], c, e, d;
			CC[0] = c = t = t - R;
			e = t - Q;
			const long* va = sub[(int)a[i]];
			for (long j = 1; j <= n; j++) {
				c = c - Q - R; e = e - R; if (c > e) e = c;
				c = CC[(size_t)j] - Q - R; d = DD[(size_t)j] - R; if (c > d) d = c;
				if (!is_used(i + I, j + J)) c = s + va[(int)b[j]];      // [ref] a used pair leaves c at CC[j] - Q - R
				if (c < d) c = d;
				if (c < e) c = e;
				s = CC[(size_t)j]; CC[(size_t)j] = c; DD[(size_t)j] = d;
			}
		}
		DD[0] = CC[0];
		// backward half: RR / SS over rows m-1..midi
		RR[(size_t)n] = 0;
		t = -Q;
		for (long j = n - 1; j >= 0; j--) { RR[(size_t)j] = t = t - R; SS[(size_t)j] = t - Q; }
		t = -te;
		for (long i = m - 1; i >= midi; i--) {
			long s = RR[(size_t)n], c, e, d;
			RR[(size_t)n] = c = t = t - R;
			e = t - Q;
			const long* va = sub[(int)a[i + 1]];
			for (long j = n - 1; j >= 0; j--) {
				c = c - Q - R; e = e - R; if (c > e) e = c;
				c = RR[(size_t)j] - Q - R; d = SS[(size_t)j] - R; if (c > d) d = c;
				if (!is_used(i + 1 + I, j + 1 + J)) c = s + va[(int)b[j + 1]];
				if (c < d) c = d;
				if (c < e) c = e;
				s = RR[(size_t)j]; RR[(size_t)j] = c; SS[(size_t)j] = d;
			}
		}
		SS[(size_t)n] = RR[(size_t)n];
		long midc = CC[0] + RR[0], midj = 0; int type = 1;
		for (long j = 0; j <= n; j++) {
			const long c = CC[(size_t)j] + RR[(size_t)j];
			if (c >= midc && (c > midc || (CC[(size_t)j] != DD[(size_t)j] && RR[(size_t)j] == SS[(size_t)j]))) { midc = c; midj = j; }
		}
		for (long j = n; j >= 0; j--) {
			const long c = DD[(size_t)j] + SS[(size_t)j] + Q;
			if (c > midc) { midc = c; midj = j; type = 2; }
		}
		if (type == 1) {
			diff(a, b, midi, midj, tb, Q);
			diff(a + midi, b + midj, m - midi, n - midj, Q, te);
		} else {
			diff(a, b, midi - 1, midj, tb, 0);
			op_del(2);
			diff(a + midi + 1, b + midj, m - midi - 1, n - midj, 0, te);
		}
		return midc;
	}

	// display() (sim.h:348-389): the two aligned strings and the identity
	static float expand(const char* a, const char* b, long m, long n, const long* S, std::string& sa, std::string& sb)
	{
		long i = 0, j = 0, match = 0, mis = 0;
		sa.clear(); sb.clear();
		while (i < m || j < n) {
			while (i < m && j < n && *S == 0) { ++i; ++j; if (a[i] == b[j]) ++match; else ++mis; sa += a[i]; sb += b[j]; S++; }
			if (i < m || j < n) {
				const long op = *S++;
				if (op > 0) for (long f = 0; f < op; f++) { sa += '-'; sb += b[++j]; ++mis; }
				else for (long f = 0; f < -op; f++) { sb += '-'; sa += a[++i]; ++mis; }
			}
		}
		return (float)(100 * match) / (float)(match + mis);
	}

	void init(const std::string& rnaA, const std::string& tgtB)
	{
		M = (long)rnaA.size(); N = (long)tgtB.size();
		for (auto& r : sub) for (long& x : r) x = -40;                                      // (only ACGT x ACGT is defined in the reference)
		for (char c : { 'A', 'C', 'G', 'T' }) sub[(int)c][(int)c] = 50;
		Q = 120; R = 40;
		CC.assign((size_t)N + 1, 0); DD = RR = SS = EE = FF = CC;
		HH.assign((size_t)M + 1, 0); WW = II = JJ = XX = YY = HH;
		script.assign((size_t)(N + M + 2), 0);
		used.assign((size_t)M + 2, std::list<long>());
		nodes.clear();
	}

	// first sweep (sim.h:506-571)
	void forward(long min_score)
	{
		for (long j = 1; j <= N; j++) { CC[(size_t)j] = 0; RR[(size_t)j] = 0; EE[(size_t)j] = j; DD[(size_t)j] = -Q; SS[(size_t)j] = 0; FF[(size_t)j] = j; }
		for (long i = 1; i <= M; i++) {
			Line ln{ Tri{ 0, i, 0 }, Tri{ -Q, i, 0 }, Tri{ 0, i - 1, 0 } };
			const long* va = sub[(int)A[i]];
			for (long j = 1; j <= N; j++) {
				const Tri c = cell(ln, CC[(size_t)j], RR[(size_t)j], EE[(size_t)j], DD[(size_t)j], SS[(size_t)j], FF[(size_t)j], i, j, va[(int)B[j]], nullptr);
				if (c.s > min_score) add_node(c.s, c.i, c.j, i, j);                          // [ref] x10 score against the unscaled threshold
			}
		}
	}
};

} // namespace

void sim_forward_nodes(const std::string& rna, const std::string& target, long min_score, std::vector<SimNode>& nodes)
{
	Sim s;
	const std::string a = ' ' + rna, b = ' ' + target;
	s.A = a.c_str(); s.B = b.c_str();
	s.init(rna, target);
	s.forward(min_score);
	nodes = s.nodes;
}

// SIM() (sim.h:410-1143) for one unit
void sim_unit(const std::string& rna, const std::string& target, const std::string& src, long dna_start, long min_score,
	int strand, int para, int rule, const Params& p, std::vector<Triplex>& out)
{
	Sim s;
	const std::string a = ' ' + rna, b = ' ' + target;
	s.A = a.c_str(); s.B = b.c_str();
	s.init(rna, target);
	const long Q = s.Q, N = s.N;
	s.forward(min_score);
	long minv = 0;                                     // `min` of the reference: 0, then addnode()'s return value 1
	auto& L = s.nodes;
	for (long count = (long)L.size() - 1; count >= 0; count--) {
		// the best remaining node goes to the end of the list and leaves it (sim.h:574-584)
		size_t best = 0;
		for (size_t i = 1; i < L.size(); i++) if (L[i].score > L[best].score) best = i;
		const SimNode cur = L[best];
		if (best != L.size() - 1) L[best] = L.back();
		L.pop_back();
		long score = cur.score;
		long stari = cur.stari + 1, starj = cur.starj + 1;
		const long endi = cur.endi, endj = cur.endj;
		long m1 = cur.top, mm = cur.bot, n1 = cur.left, nn = cur.right;
		long rl = endi - stari + 1, cl = endj - starj + 1;
		s.I = stari - 1; s.J = starj - 1;
		s.sapp = s.script.data(); s.last = 0;
		const int nt = (int)(endi - stari + 1);        // [ref] the lncRNA span
		s.diff(s.A + stari - 1, s.B + starj - 1, rl, cl, Q, Q);
		if ((double)score / 10.0 <= (double)min_score) break;                                   // sim.h:594
		std::string stri, strj;
		const float identity = Sim::expand(s.A + stari - 1, s.B + starj - 1, rl, cl, s.script.data(), stri, strj);
		if (nt >= p.ntMin && nt <= p.ntMax) {
			// stability with the TT / CC run penalties (sim.h:696-731), display strand read from `src`
			float tri = 0.0f, prescore = 0.0f;
			char prechar = 0, curchar = 0;
			std::string tts;
			long j = 0;
			for (size_t i = 0; i < strj.size(); i++) {
				float h;
				if (strj[i] == '-') { curchar = '-'; h = triplex_score(curchar, stri[i], para); tts += '-'; }
				else { curchar = src[(size_t)(starj + j - 1)]; h = triplex_score(curchar, stri[i], para); tts += curchar; j++; }
				if (curchar == prechar && curchar == 'T') { tri = tri - prescore + (float)p.penaltyT; h = (float)p.penaltyT; }
				if (curchar == prechar && curchar == 'C') { tri = tri - prescore + (float)p.penaltyC; h = (float)p.penaltyC; }
				prescore = h;
				if (strj[i] != '-') prechar = curchar;
				tri += h;
			}
			score /= 10;
			const float final_score = (float)score;
			tri /= nt;
			long ref_start, ref_end;
			if (para < 0 && strand == 0) { ref_start = N - endj + 1; ref_end = N - starj + 1; }
			else if (para > 0 && strand == 1) { ref_start = N - endj - 1; ref_end = N - starj - 1; }
			else { ref_start = starj; ref_end = endj; }
			Triplex t;
			t.stari = (int)stari; t.endi = (int)endi; t.starj = (int)(ref_start + dna_start); t.endj = (int)(ref_end + dna_start);
			t.strand = strand; t.reverse = para; t.rule = rule; t.nt = nt; t.score = final_score; t.identity = identity; t.tri_score = tri;
			t.stri_align = stri; t.strj_align = tts;
			out.push_back(t);
		}
		if (!count) continue;

		// ---- the region the removed alignment may have influenced is recomputed (sim.h:884-1141) ------------------------
		auto& CC = s.CC; auto& DD = s.DD; auto& RR = s.RR; auto& SS = s.SS; auto& EE = s.EE; auto& FF = s.FF;
		auto& HH = s.HH; auto& WW = s.WW; auto& II = s.II; auto& JJ = s.JJ; auto& XX = s.XX; auto& YY = s.YY;
		bool flag = false;
		for (long j = nn; j >= n1; j--) { CC[(size_t)j] = 0; EE[(size_t)j] = j; DD[(size_t)j] = -Q; FF[(size_t)j] = j; RR[(size_t)j] = SS[(size_t)j] = mm + 1; }
		// backward sweep of one row over columns nn..n1; returns the last (c, d) and leaves f in ln
		auto row_back = [&](long i, Sim::Line& ln, Tri& c_last, Tri& d_last, bool watch, bool& rflag) {
			const long* va = s.sub[(int)s.A[i]];
			for (long j = nn; j >= n1; j--) {
				Tri d;
				const Tri c = s.cell(ln, CC[(size_t)j], RR[(size_t)j], EE[(size_t)j], DD[(size_t)j], SS[(size_t)j], FF[(size_t)j], i, j, va[(int)s.B[j]], &d);
				if (c.s > minv) flag = true;
				if (watch && !rflag && ((c.i > rl && c.j > cl) || (d.i > rl && d.j > cl) || (ln.f.i > rl && ln.f.j > cl))) rflag = true;
				c_last = c; d_last = d;
			}
			HH[(size_t)i] = CC[(size_t)n1]; II[(size_t)i] = RR[(size_t)n1]; JJ[(size_t)i] = EE[(size_t)n1];
			WW[(size_t)i] = ln.f.s; XX[(size_t)i] = ln.f.i; YY[(size_t)i] = ln.f.j;
		};
		{
			bool dummy = false; Tri c0{ 0, 0, 0 }, d0{ 0, 0, 0 };
			for (long i = mm; i >= m1; i--) {
				Sim::Line ln{ Tri{ 0, i, nn + 1 }, Tri{ -Q, i, nn + 1 }, Tri{ 0, i + 1, nn + 1 } };
				row_back(i, ln, c0, d0, false, dummy);
			}
		}
		for (rl = m1, cl = n1;;) {
			bool rflag = true, cflag = true;
			while ((rflag && m1 > 1) || (cflag && n1 > 1)) {
				if (rflag && m1 > 1) {
					rflag = false;
					m1--;
					Sim::Line ln{ Tri{ 0, m1, nn + 1 }, Tri{ -Q, m1, nn + 1 }, Tri{ 0, m1 + 1, nn + 1 } };
					Tri c{ 0, m1, nn + 1 }, d{ 0, 0, 0 };
					bool any = false;
					{
						// (when the column range is empty the reference tests its variables as initialised: c start (m1, nn+1), d unset)
						const long* va = s.sub[(int)s.A[m1]];
						for (long j = nn; j >= n1; j--) {
							c = s.cell(ln, CC[(size_t)j], RR[(size_t)j], EE[(size_t)j], DD[(size_t)j], SS[(size_t)j], FF[(size_t)j], m1, j, va[(int)s.B[j]], &d);
							any = true;
							if (c.s > minv) flag = true;
							if (!rflag && ((c.i > rl && c.j > cl) || (d.i > rl && d.j > cl) || (ln.f.i > rl && ln.f.j > cl))) rflag = true;
						}
					}
					(void)any;
					HH[(size_t)m1] = CC[(size_t)n1]; II[(size_t)m1] = RR[(size_t)n1]; JJ[(size_t)m1] = EE[(size_t)n1];
					WW[(size_t)m1] = ln.f.s; XX[(size_t)m1] = ln.f.i; YY[(size_t)m1] = ln.f.j;
					if (!cflag && ((c.i > rl && c.j > cl) || (d.i > rl && d.j > cl) || (ln.f.i > rl && ln.f.j > cl))) cflag = true;
				}
				if (cflag && n1 > 1) {
					cflag = false;
					n1--;
					// one column, rows mm..m1, with the row-indexed state (HH, II, JJ / WW, XX, YY)
					Sim::Line ln{ Tri{ 0, mm + 1, n1 }, Tri{ -Q, mm + 1, n1 }, Tri{ 0, mm + 1, n1 + 1 } };
					Tri c{ 0, mm + 1, n1 }, d{ 0, 0, 0 };
					const long* vb = s.sub[(int)s.B[n1]];
					for (long i = mm; i >= m1; i--) {
						c = s.cell(ln, HH[(size_t)i], II[(size_t)i], JJ[(size_t)i], WW[(size_t)i], XX[(size_t)i], YY[(size_t)i], i, n1, vb[(int)s.A[i]], &d);
						if (c.s > minv) flag = true;
						if (!cflag && ((c.i > rl && c.j > cl) || (d.i > rl && d.j > cl) || (ln.f.i > rl && ln.f.j > cl))) cflag = true;
					}
					CC[(size_t)n1] = HH[(size_t)m1]; RR[(size_t)n1] = II[(size_t)m1]; EE[(size_t)n1] = JJ[(size_t)m1];
					DD[(size_t)n1] = ln.f.s; SS[(size_t)n1] = ln.f.i; FF[(size_t)n1] = ln.f.j;
					if (!rflag && ((c.i > rl && c.j > cl) || (d.i > rl && d.j > cl) || (ln.f.i > rl && ln.f.j > cl))) rflag = true;
				}
			}
			if (m1 == 1 && n1 == 1) break;
			// no_cross (sim.h:150-165): a remaining node whose region reaches into the recomputed one and starts further out
			bool crossed = false;
			for (const SimNode& nd : L) {
				if (nd.stari <= mm && nd.starj <= nn && nd.bot >= m1 - 1 && nd.right >= n1 - 1 && (nd.stari < rl || nd.starj < cl)) {
					if (nd.stari < rl) rl = nd.stari;
					if (nd.starj < cl) cl = nd.starj;
					crossed = true;
					break;
				}
			}
			if (!crossed) break;
		}
		m1--; n1--;
		if (flag) {
			for (long j = n1 + 1; j <= nn; j++) { CC[(size_t)j] = 0; RR[(size_t)j] = m1; EE[(size_t)j] = j; DD[(size_t)j] = -Q; SS[(size_t)j] = m1; FF[(size_t)j] = j; }
			for (long i = m1 + 1; i <= mm; i++) {
				Sim::Line ln{ Tri{ 0, i, n1 }, Tri{ -Q, i, n1 }, Tri{ 0, i - 1, n1 } };
				const long* va = s.sub[(int)s.A[i]];
				for (long j = n1 + 1; j <= nn; j++) {
					const Tri c = s.cell(ln, CC[(size_t)j], RR[(size_t)j], EE[(size_t)j], DD[(size_t)j], SS[(size_t)j], FF[(size_t)j], i, j, va[(int)s.B[j]], nullptr);
					if (c.s > minv) { s.add_node(c.s, c.i, c.j, i, j); minv = 1; }       // [ref] min = addnode(...) = 1
				}
			}
		}
	}
}

} // namespace fo
