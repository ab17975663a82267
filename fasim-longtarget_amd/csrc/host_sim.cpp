// fasim-longtarget_amd/csrc/host_sim.cpp -- row f3: the host half of the -F path (classic SIM, sim.h:410-1143).
//
// The GPU does SIM()'s first sweep over the whole (lncRNA x target) matrix (sim.hip + the node-list replay in engine.cpp)
// and hands over the K = 50 node list.  What follows in the reference (sim.h:572-1141) is K times: take the best node, trace
// its alignment back in linear space (diff, sim.h:167-346), build the triplex record (sim.h:596-745), and re-sweep the
// rectangle that alignment may have influenced -- backwards until no remaining node crosses it (no_cross, sim.h:150-165),
// then forwards again, feeding new nodes.  Those rectangles are a few hundred cells wide and the K steps depend on each
// other, so this part runs on the host, one unit per host thread.  Every tie rule, the x10 scores against the unscaled
// threshold, "min = addnode() = 1" and nt = the lncRNA span are the reference's.
#include "host_post.h"

#include <algorithm>
#include <cstring>

namespace fasim {

namespace {

struct P3 { long s; long i, j; };                 // score + the start point it belongs to
// ORDER (sim.h:481-493): lexicographic maximum of (score, start row, start column)
inline void keep_larger(P3& a, const P3& b)
{
	if (b.s > a.s || (b.s == a.s && (b.i > a.i || (b.i == a.i && b.j > a.j)))) a = b;
}

constexpr long kQ = 120, kR = 40, kMatch = 50, kMismatch = -40;     // 10 x (gap open 12, extension 4, match 5, mismatch -4)
inline long pair_score(char a, char b)
{
	// V (sim.h:464-468) is defined for ACGT x ACGT only; any other letter counts as a mismatch here
	return (a == b && (a == 'A' || a == 'C' || a == 'G' || a == 'T')) ? kMatch : kMismatch;
}
inline long gap_cost(long k) { return k <= 0 ? 0 : kQ + kR * k; }

struct Work {
	const char* A; const char* B; long M, N;        // 1-based views
	std::vector<fasim_sim_node>* list;
	std::vector<long> CC, DD, RR, SS, EE, FF, HH, WW, II, JJ, XX, YY;
	std::vector<std::vector<int>> used;             // per query row: target columns already aligned to it (DIAG)
	std::vector<long> script; long* sp = nullptr; long last = 0, I = 0, J = 0;

	bool taken(long i, long j) const { for (int x : used[(size_t)i]) if (x == (int)j) return true; return false; }

	void add_node(long c, long ci, long cj, long i, long j)
	{
		for (fasim_sim_node& n : *list) {
			if (n.stari != ci || n.starj != cj) continue;
			if (n.score < c) { n.score = c; n.endi = i; n.endj = j; }
			if (n.top > i) n.top = i;
			if (n.bot < i) n.bot = i;
			if (n.left > j) n.left = j;
			if (n.right < j) n.right = j;
			return;
		}
		const fasim_sim_node fresh = { c, ci, cj, i, j, i, i, j, j };
		if ((int)list->size() == FASIM_SIM_K) {
			size_t low = 0;
			for (size_t d = 1; d < list->size(); d++) if ((*list)[d].score < (*list)[low].score) low = d;
			(*list)[low] = fresh;
		} else list->push_back(fresh);
	}

	// one cell of a sweep (sim.h:522-566): `run` = the cell before it on the sweep line, `gapf` the gap state along the line,
	// `corner` the diagonal neighbour; (sv, si, sj) / (gv, gi, gj) the stored states across the line
	struct Line { P3 run, gapf, corner; };
	P3 step(Line& ln, long& sv, long& si, long& sj, long& gv, long& gi, long& gj, long i, long j, long sc, P3& across)
	{
		ln.gapf.s -= kR;
		keep_larger(ln.gapf, P3{ ln.run.s - kQ - kR, ln.run.i, ln.run.j });
		across = P3{ gv - kR, gi, gj };
		keep_larger(across, P3{ sv - kQ - kR, si, sj });
		long v = taken(i, j) ? 0 : ln.corner.s + sc;
		P3 c = v <= 0 ? P3{ 0, i, j } : P3{ v, ln.corner.i, ln.corner.j };
		keep_larger(c, across);
		keep_larger(c, ln.gapf);
		ln.corner = P3{ sv, si, sj };
		sv = c.s; si = c.i; sj = c.j;
		gv = across.s; gi = across.i; gj = across.j;
		ln.run = c;
		return c;
	}

	// ---- edit script (sim.h:170-196) ----
	void del(long k) { I += k; if (last < 0) last = sp[-1] -= k; else last = *sp++ = -k; }
	void ins(long k) { J += k; if (last < 0) { sp[-1] = k; *sp++ = last; } else last = *sp++ = k; }
	void rep() { last = *sp++ = 0; }

	// linear-space alignment of a[1..m] with b[1..n] (sim.h:167-346)
	long diff(const char* a, const char* b, long m, long n, long tb, long te)
	{
		if (n <= 0) { if (m > 0) del(m); return -gap_cost(m); }
		if (m <= 1) {
			if (m <= 0) { ins(n); return -gap_cost(n); }
			if (tb > te) tb = te;
			long best = -(tb + kR + gap_cost(n)), at = 0;
			for (long j = 1; j <= n; j++) {
				if (taken(I + 1, j + J)) continue;
				const long c = pair_score(a[1], b[j]) - (gap_cost(j - 1) + gap_cost(n - j));
				if (c > best) { best = c; at = j; }
			}
			if (at == 0) { ins(n); del(1); }
			else {
				if (at > 1) ins(at - 1);
				rep();
				I++; J++;
				used[(size_t)I].push_back((int)J);
				if (at < n) ins(n - at);
			}
			return best;
		}
		const long mid = m / 2;
		long t, s, c, e, d;
		CC[0] = 0; t = -kQ;
		for (long j = 1; j <= n; j++) { CC[(size_t)j] = t = t - kR; DD[(size_t)j] = t - kQ; }
		t = -tb;
		for (long i = 1; i <= mid; i++) {
			s = CC[0]; CC[0] = c = t = t - kR; e = t - kQ;
			for (long j = 1; j <= n; j++) {
				c = c - kQ - kR; e = e - kR; if (c > e) e = c;
				c = CC[(size_t)j] - kQ - kR; d = DD[(size_t)j] - kR; if (c > d) d = c;
				if (!taken(i + I, j + J)) c = s + pair_score(a[i], b[j]);
				if (c < d) c = d;
				if (c < e) c = e;
				s = CC[(size_t)j]; CC[(size_t)j] = c; DD[(size_t)j] = d;
			}
		}
		DD[0] = CC[0];
		RR[(size_t)n] = 0; t = -kQ;
		for (long j = n - 1; j >= 0; j--) { RR[(size_t)j] = t = t - kR; SS[(size_t)j] = t - kQ; }
		t = -te;
		for (long i = m - 1; i >= mid; i--) {
			s = RR[(size_t)n]; RR[(size_t)n] = c = t = t - kR; e = t - kQ;
			for (long j = n - 1; j >= 0; j--) {
				c = c - kQ - kR; e = e - kR; if (c > e) e = c;
				c = RR[(size_t)j] - kQ - kR; d = SS[(size_t)j] - kR; if (c > d) d = c;
				if (!taken(i + 1 + I, j + 1 + J)) c = s + pair_score(a[i + 1], b[j + 1]);
				if (c < d) c = d;
				if (c < e) c = e;
				s = RR[(size_t)j]; RR[(size_t)j] = c; SS[(size_t)j] = d;
			}
		}
		SS[(size_t)n] = RR[(size_t)n];
		long best = CC[0] + RR[0], at = 0; bool through_gap = false;
		for (long j = 0; j <= n; j++) {
			c = CC[(size_t)j] + RR[(size_t)j];
			if (c > best || (c == best && CC[(size_t)j] != DD[(size_t)j] && RR[(size_t)j] == SS[(size_t)j])) { best = c; at = j; }
		}
		for (long j = n; j >= 0; j--) {
			c = DD[(size_t)j] + SS[(size_t)j] + kQ;
			if (c > best) { best = c; at = j; through_gap = true; }
		}
		if (!through_gap) {
			diff(a, b, mid, at, tb, kQ);
			diff(a + mid, b + at, m - mid, n - at, kQ, te);
		} else {
			diff(a, b, mid - 1, at, tb, 0);
			del(2);
			diff(a + mid + 1, b + at, m - mid - 1, n - at, 0, te);
		}
		return best;
	}
};

} // namespace

// ---- one unit of the -F path as a state machine: next_round() = sim.h:572-883 (best node, traceback, record), then the re-sweep of
//      the influenced rectangle either on the device (k_sim_resweep, sim.hip) or here (resweep_host) --------------------------------
struct SimUnit::Impl {
	std::string a1, b1, src;
	Work w;
	EncInfo info;
	fasim_params p;
	long dna_start = 0, min_score = 0;
	long m1 = 0, mm = 0, n1 = 0, nn = 0;              // bounding box of the node of the current round
};

SimUnit::SimUnit(const std::string& rna, const std::string& target, const std::string& src_, long dna_start, long min_score, int enc,
	const fasim_params& p, std::vector<fasim_sim_node>&& nodes_) : nodes(std::move(nodes_)), im(new Impl())
{
	Impl& I = *im;
	I.a1 = ' ' + rna; I.b1 = ' ' + target; I.src = src_;
	I.info = enc_info(enc); I.p = p; I.dna_start = dna_start; I.min_score = min_score;
	Work& w = I.w;
	w.A = I.a1.c_str(); w.B = I.b1.c_str(); w.M = (long)rna.size(); w.N = (long)target.size();
	w.list = &nodes;
	w.CC.assign((size_t)w.N + 1, 0); w.DD = w.RR = w.SS = w.EE = w.FF = w.CC;
	w.HH.assign((size_t)w.M + 1, 0); w.WW = w.II = w.JJ = w.XX = w.YY = w.HH;
	w.used.assign((size_t)w.M + 2, std::vector<int>());
	w.script.assign((size_t)(w.M + w.N + 2), 0);
	rounds_left = (long)nodes.size();
}
SimUnit::~SimUnit() { delete im; }

bool SimUnit::next_round(bool* sweep, int box[4], std::vector<std::pair<int, int>>* new_pairs)
{
	*sweep = false;
	if (rounds_left <= 0 || nodes.empty()) return false;
	Impl& I = *im; Work& w = I.w;
	w.list = &nodes;
	const fasim_params& p = I.p; const EncInfo& info = I.info;
	const long N = w.N;
	const long round = --rounds_left;
	size_t best = 0;
	for (size_t k = 1; k < nodes.size(); k++) if (nodes[k].score > nodes[best].score) best = k;
	const fasim_sim_node cur = nodes[best];
	if (best != nodes.size() - 1) nodes[best] = nodes.back();
	nodes.pop_back();
	long score = cur.score;
	const long stari = cur.stari + 1, starj = cur.starj + 1, endi = cur.endi, endj = cur.endj;
	I.m1 = cur.top; I.mm = cur.bot; I.n1 = cur.left; I.nn = cur.right;
	const long rl = endi - stari + 1, cl = endj - starj + 1;
	w.I = stari - 1; w.J = starj - 1; w.sp = w.script.data(); w.last = 0;
	const int nt = (int)(endi - stari + 1);
	// the pairs this alignment uses (DIAG) are what the re-sweep must exclude: remember where the rows' lists stood
	std::vector<size_t> before;
	if (new_pairs) { before.resize((size_t)rl + 2); for (long i = 0; i <= rl + 1 && stari - 1 + i <= w.M + 1; i++) before[(size_t)i] = w.used[(size_t)(stari - 1 + i)].size(); }
	w.diff(w.A + stari - 1, w.B + starj - 1, rl, cl, kQ, kQ);
	if (new_pairs) {
		new_pairs->clear();
		for (long i = 0; i <= rl + 1 && stari - 1 + i <= w.M + 1; i++) {
			const std::vector<int>& u = w.used[(size_t)(stari - 1 + i)];
			for (size_t k = before[(size_t)i]; k < u.size(); k++) new_pairs->push_back({ (int)(stari - 1 + i), u[k] });
		}
	}
	if ((double)score / 10.0 <= (double)I.min_score) { rounds_left = 0; return false; }             // sim.h:594

	// aligned strings and identity (display, sim.h:348-389)
	std::string tfo, tgt;
	long matches = 0, others = 0;
	{
		const char* a = w.A + stari - 1; const char* b = w.B + starj - 1; const long* S = w.script.data();
		long i = 0, j = 0;
		while (i < rl || j < cl) {
			while (i < rl && j < cl && *S == 0) { ++i; ++j; if (a[i] == b[j]) ++matches; else ++others; tfo += a[i]; tgt += b[j]; S++; }
			if (i < rl || j < cl) {
				const long op = *S++;
				if (op > 0) for (long f = 0; f < op; f++) { tfo += '-'; tgt += b[++j]; ++others; }
				else for (long f = 0; f < -op; f++) { tgt += '-'; tfo += a[++i]; ++others; }
			}
		}
	}
	const float identity = (float)(100 * matches) / (float)(matches + others);
	if (nt >= p.ntMin && nt <= p.ntMax) {
		// mean stability with the TT / CC run penalties (sim.h:696-731); the TTS string is read from the display strand
		float tri = 0.0f, before_h = 0.0f;
		char prev = 0, curc = 0;
		std::string tts;
		long j = 0;
		for (size_t k = 0; k < tgt.size(); k++) {
			float h;
			if (tgt[k] == '-') { curc = '-'; h = triplex_stability(curc, tfo[k], info.para); tts += '-'; }
			else { curc = I.src[(size_t)(starj + j - 1)]; h = triplex_stability(curc, tfo[k], info.para); tts += curc; j++; }
			if (curc == prev && curc == 'T') { tri = tri - before_h + (float)p.penaltyT; h = (float)p.penaltyT; }
			if (curc == prev && curc == 'C') { tri = tri - before_h + (float)p.penaltyC; h = (float)p.penaltyC; }
			before_h = h;
			if (tgt[k] != '-') prev = curc;
			tri += h;
		}
		score /= 10;
		tri /= nt;
		long ref_start, ref_end;
		if (info.para < 0 && info.strand == 0) { ref_start = N - endj + 1; ref_end = N - starj + 1; }
		else if (info.para > 0 && info.strand == 1) { ref_start = N - endj - 1; ref_end = N - starj - 1; }
		else { ref_start = starj; ref_end = endj; }
		HostTriplex t;
		t.stari = (int)stari; t.endi = (int)endi; t.starj = (int)(ref_start + I.dna_start); t.endj = (int)(ref_end + I.dna_start);
		t.strand = info.strand; t.reverse = info.para; t.rule = info.rule; t.nt = nt;
		t.score = (float)score; t.identity = identity; t.tri_score = tri;
		t.tfo = tfo; t.tts = tts;
		out.push_back(std::move(t));
	}
	if (round == 0) return false;                        // the last node: no re-sweep (sim.h:884)
	*sweep = true;
	box[0] = (int)I.m1; box[1] = (int)I.mm; box[2] = (int)I.n1; box[3] = (int)I.nn;
	return true;
}

// ---- re-sweep of the influenced rectangle on the host (sim.h:884-1141): fasim_sim_finish_unit and the device path's self-check
void SimUnit::resweep_host()
{
	Impl& I = *im; Work& w = I.w;
	w.list = &nodes;
	auto& CC = w.CC; auto& DD = w.DD; auto& RR = w.RR; auto& SS = w.SS; auto& EE = w.EE; auto& FF = w.FF;
	auto& HH = w.HH; auto& WW = w.WW; auto& II = w.II; auto& JJ = w.JJ; auto& XX = w.XX; auto& YY = w.YY;
	long m1 = I.m1, mm = I.mm, n1 = I.n1, nn = I.nn, rl = 0, cl = 0;
	{
		bool positive = false;
		for (long j = nn; j >= n1; j--) { CC[(size_t)j] = 0; EE[(size_t)j] = j; DD[(size_t)j] = -kQ; FF[(size_t)j] = j; RR[(size_t)j] = SS[(size_t)j] = mm + 1; }
		auto outside = [&](const P3& x) { return x.i > rl && x.j > cl; };
		for (long i = mm; i >= m1; i--) {
			Work::Line ln{ P3{ 0, i, nn + 1 }, P3{ -kQ, i, nn + 1 }, P3{ 0, i + 1, nn + 1 } };
			P3 d{ 0, 0, 0 };
			for (long j = nn; j >= n1; j--) {
				const P3 c = w.step(ln, CC[(size_t)j], RR[(size_t)j], EE[(size_t)j], DD[(size_t)j], SS[(size_t)j], FF[(size_t)j], i, j, pair_score(w.A[i], w.B[j]), d);
				if (c.s > floor_score) positive = true;
			}
			HH[(size_t)i] = CC[(size_t)n1]; II[(size_t)i] = RR[(size_t)n1]; JJ[(size_t)i] = EE[(size_t)n1];
			WW[(size_t)i] = ln.gapf.s; XX[(size_t)i] = ln.gapf.i; YY[(size_t)i] = ln.gapf.j;
		}
		for (rl = m1, cl = n1;;) {
			bool grow_rows = true, grow_cols = true;
			while ((grow_rows && m1 > 1) || (grow_cols && n1 > 1)) {
				if (grow_rows && m1 > 1) {
					grow_rows = false;
					m1--;
					Work::Line ln{ P3{ 0, m1, nn + 1 }, P3{ -kQ, m1, nn + 1 }, P3{ 0, m1 + 1, nn + 1 } };
					P3 c{ 0, m1, nn + 1 }, d{ 0, 0, 0 };
					for (long j = nn; j >= n1; j--) {
						c = w.step(ln, CC[(size_t)j], RR[(size_t)j], EE[(size_t)j], DD[(size_t)j], SS[(size_t)j], FF[(size_t)j], m1, j, pair_score(w.A[m1], w.B[j]), d);
						if (c.s > floor_score) positive = true;
						if (!grow_rows && (outside(c) || outside(d) || outside(ln.gapf))) grow_rows = true;
					}
					HH[(size_t)m1] = CC[(size_t)n1]; II[(size_t)m1] = RR[(size_t)n1]; JJ[(size_t)m1] = EE[(size_t)n1];
					WW[(size_t)m1] = ln.gapf.s; XX[(size_t)m1] = ln.gapf.i; YY[(size_t)m1] = ln.gapf.j;
					if (!grow_cols && (outside(c) || outside(d) || outside(ln.gapf))) grow_cols = true;
				}
				if (grow_cols && n1 > 1) {
					grow_cols = false;
					n1--;
					Work::Line ln{ P3{ 0, mm + 1, n1 }, P3{ -kQ, mm + 1, n1 }, P3{ 0, mm + 1, n1 + 1 } };
					P3 c{ 0, mm + 1, n1 }, d{ 0, 0, 0 };
					for (long i = mm; i >= m1; i--) {
						c = w.step(ln, HH[(size_t)i], II[(size_t)i], JJ[(size_t)i], WW[(size_t)i], XX[(size_t)i], YY[(size_t)i], i, n1, pair_score(w.B[n1], w.A[i]), d);
						if (c.s > floor_score) positive = true;
						if (!grow_cols && (outside(c) || outside(d) || outside(ln.gapf))) grow_cols = true;
					}
					CC[(size_t)n1] = HH[(size_t)m1]; RR[(size_t)n1] = II[(size_t)m1]; EE[(size_t)n1] = JJ[(size_t)m1];
					DD[(size_t)n1] = ln.gapf.s; SS[(size_t)n1] = ln.gapf.i; FF[(size_t)n1] = ln.gapf.j;
					if (!grow_rows && (outside(c) || outside(d) || outside(ln.gapf))) grow_rows = true;
				}
			}
			if (m1 == 1 && n1 == 1) break;
			bool crossed = false;                          // no_cross (sim.h:150-165)
			for (const fasim_sim_node& nd : nodes) {
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
		if (positive) {
			for (long j = n1 + 1; j <= nn; j++) { CC[(size_t)j] = 0; RR[(size_t)j] = m1; EE[(size_t)j] = j; DD[(size_t)j] = -kQ; SS[(size_t)j] = m1; FF[(size_t)j] = j; }
			for (long i = m1 + 1; i <= mm; i++) {
				Work::Line ln{ P3{ 0, i, n1 }, P3{ -kQ, i, n1 }, P3{ 0, i - 1, n1 } };
				P3 d{ 0, 0, 0 };
				for (long j = n1 + 1; j <= nn; j++) {
					const P3 c = w.step(ln, CC[(size_t)j], RR[(size_t)j], EE[(size_t)j], DD[(size_t)j], SS[(size_t)j], FF[(size_t)j], i, j, pair_score(w.A[i], w.B[j]), d);
					if (c.s > floor_score) { w.add_node(c.s, c.i, c.j, i, j); floor_score = 1; }
				}
			}
		}
	}
}

void sim_finish_unit(const std::string& rna, const std::string& target, const std::string& src, long dna_start, long min_score,
	int enc, const fasim_params& p, std::vector<fasim_sim_node>& nodes, std::vector<HostTriplex>& out)
{
	SimUnit u(rna, target, src, dna_start, min_score, enc, p, std::move(nodes));
	bool sweep = false; int box[4];
	while (u.next_round(&sweep, box, nullptr)) if (sweep) u.resweep_host();
	for (HostTriplex& t : u.out) out.push_back(std::move(t));
	nodes = std::move(u.nodes);
}

} // namespace fasim
