// tests/parity/band_study.cpp -- ANALYSIS TOOL (test infrastructure; links the oracle for the rule encodings and the
// peak picking).  Not product code.
//
// Question: how many query rows does a stage-3 window try really need?  For a sample of units the
// tool computes the textbook segment DP (what k_scan computes), per (virtual lane, K-step block) maxima U as k_scan
// could emit them, the candidates and the window tries of fastSIM, and for every executed try
//   * the virtual lanes whose upper bound reaches the try's score (flagged lanes),
//   * a band DP over [first flagged - M, last flagged] with the unknown row above the band replaced by its upper
//     bound carried as TAINTED values, and whether the winning cell comes out clean,
// and prints the cost of the band scheme against the full-height sweep for a few block sizes K and margins M.
//
//   g++ -O2 -std=c++17 -I oracle tests/parity/band_study.cpp oracle/_build/libfasim_oracle.so -o /tmp/band_study
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "fasim_oracle.h"

static int code_of(char c) { switch (c) { case 'A': case 'a': case 'U': case 'u': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; } }
static inline int sc(int q, int t) { return (q == t && q < 4) ? 5 : -4; }

struct Layout {
	int m, seg, vs, nv, rp;
	std::vector<int> row0, rows, lane_of_row;
	void init(int m_) {
		m = m_; seg = (m + 15) / 16; vs = 8 * ((seg + 191) / 192); nv = 16 * vs; rp = (seg + vs - 1) / vs;
		row0.resize(nv); rows.resize(nv); lane_of_row.assign(16 * seg, 0);
		for (int v = 0; v < nv; v++) {
			const int s = v / vs, j = v - s * vs, q = seg / vs, rem = seg - q * vs;
			rows[v] = q + (j < rem ? 1 : 0); row0[v] = s * seg + j * q + (j < rem ? j : rem);
			for (int r = 0; r < rows[v]; r++) lane_of_row[row0[v] + r] = v;
		}
	}
};

struct WinRes { int S, ref_end, read_end; };

// full-height textbook window DP (C1/C6): rows 0..mp-1 (pads score 0)
static WinRes window_dp(const std::vector<int>& q, int m, int mp, const int* t, int L, std::vector<int>* colmax_out = nullptr)
{
	std::vector<int> H(mp + 1, 0), E(mp + 1, 0);
	WinRes r = { 0, -1, 0 };
	for (int c = 0; c < L; c++) {
		int diag = 0, F = 0, cm = 0, cmrow = mp;
		for (int i = 0; i < mp; i++) {
			int h = diag + (i < m ? sc(q[i], t[c]) : 0);
			h = std::max(std::max(h, 0), std::max(E[i], F));
			diag = H[i]; H[i] = h;
			if (h > cm) { cm = h; cmrow = i; }
			E[i] = std::max(std::max(E[i] - 4, h - 16), 0);
			F = std::max(std::max(F - 4, h - 16), 0);
		}
		if (colmax_out) (*colmax_out)[c] = cm;
		if (cm > r.S) { r.S = cm; r.ref_end = c; r.read_end = std::min(cmrow, m - 1); }
	}
	return r;
}

// band DP over rows [r0, r1) with the row above the band (r0 - 1) replaced by the upper bound `utop` carried as a
// tainted value (value*2+1).  Returns the winner and whether it is clean.
struct BandRes { int S, ref_end, read_end; bool clean; };
static BandRes band_dp(const std::vector<int>& q, int m, int r0, int r1, const int* t, int L, int utop)
{
	const int n = r1 - r0;
	std::vector<int> H(n, 0), E(n, 0);   // 2*value + taint
	BandRes r = { 0, -1, 0, true };
	const bool open = r0 > 0;
	for (int c = 0; c < L; c++) {
		// what may come from above: H of row r0-1 in columns c-1 (diagonal) and c (vertical F)
		const int capc = std::min(utop, 5 * c);          // H_win[.][c-1] <= 5 * c  (c columns so far)
		const int capv = std::min(utop, 5 * (c + 1));
		int diag = open ? (capc > 0 ? 2 * capc + 1 : 0) : 0;
		int F = open ? (capv - 16 > 0 ? 2 * (capv - 16) + 1 : 0) : 0;
		int cm = 0, cmrow = n;
		for (int i = 0; i < n; i++) {
			const int row = r0 + i;
			int h = diag + 2 * (row < m ? sc(q[row], t[c]) : 0);
			if (h < 0) h = 0;
			// max with ties to the tainted operand: compare (value, taint) as the integer 2*value+taint
			h = std::max(h, std::max(E[i], F));
			if (h < 2) h = h & ~1;                        // value 0 carries no information
			diag = H[i]; H[i] = h;
			if ((h >> 1) > (cm >> 1)) { cm = h; cmrow = row; }
			else if ((h >> 1) == (cm >> 1) && (h & 1)) cm |= 1;      // a tainted tie further down: the column maximum counts as tainted
			const int ho = h - 32 > 0 ? h - 32 : 0;
			int e = E[i] - 8 > 0 ? E[i] - 8 : 0;
			E[i] = std::max(e, ho);
			int f = F - 8 > 0 ? F - 8 : 0;
			F = std::max(f, ho);
		}
		if ((cm >> 1) > r.S) { r.S = cm >> 1; r.ref_end = c; r.read_end = std::min(cmrow, m - 1); r.clean = !(cm & 1); }
	}
	return r;
}

int main(int argc, char** argv)
{
	if (argc < 4) { fprintf(stderr, "usage: band_study rna.fa dna.fa nunits [seed]\n"); return 2; }
	std::string h, rna, dna;
	if (!fo::read_fasta(argv[1], h, rna) || !fo::read_fasta(argv[2], h, dna)) { fprintf(stderr, "read error\n"); return 1; }
	const int nunits = atoi(argv[3]);
	unsigned long long rs = argc > 4 ? strtoull(argv[4], 0, 10) : 1;
	auto rnd = [&]() { rs += 0x9E3779B97F4A7C15ull; unsigned long long z = rs; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
	const int m = (int)rna.size();
	Layout lay; lay.init(m);
	const int mp = 16 * lay.seg;
	std::vector<int> q(m);
	for (int i = 0; i < m; i++) q[i] = code_of(rna[i]);
	printf("m=%d seg=%d vs=%d nv=%d rp=%d\n", m, lay.seg, lay.vs, lay.nv, lay.rp);
	std::vector<int> starts;
	fo::cut_sequence(dna, 5000, 100, starts);

	const int KS[] = { 1, 8, 16, 32, 64 };
	const int NK = 5;
	const int MS[] = { 1, 2, 3, 4, 6 };
	const int NM = 5;
	// accumulators
	double full_cells = 0, ntries = 0, ncand = 0, naccept = 0;
	double flagged_sum[NK] = { 0 }, span_sum[NK] = { 0 };
	double band_cells[NK][NM] = { { 0 } }, band_fail[NK][NM] = { { 0 } }, band_cells_cls[NK][NM] = { { 0 } };
	double q2poss = 0;
	std::map<int, long> span_hist16;
	double ratio_sum = 0, ratio_n = 0;
	std::map<int, long> ratio_hist;
	long try_hist[5] = { 0 };
	const int NPOL = 6;
	const int pol_maxC[NPOL] = { 16, 32, 32, 64, 64, 32 };
	const double pol_rho[NPOL] = { 0.85, 0.85, 0.75, 0.85, 0.75, 0.85 };
	const int pol_second[NPOL] = { 0, 0, 0, 0, 0, 1 };
	double pol_cost[NPOL][3] = { { 0 } }, pol_full[3] = { 0 }, pol_proven[NPOL][3] = { { 0 } }, pol_fallback[NPOL][3] = { { 0 } }, pol_direct_full[NPOL][3] = { { 0 } }, pol_second_ok[NPOL][3] = { { 0 } };
	const int CLS[] = { 8, 16, 32, 64 }; const int NCLS = 4;
	double gm_ok[NK][4][3] = { { { 0 } } }, gm_n[3] = { 0 }, gm_cols[3] = { 0 };

	for (int u = 0; u < nunits; u++) {
		const int segi = (int)(rnd() % starts.size()), enc = (int)(rnd() % 48);
		const std::string seg = dna.substr(starts[segi], 5000);
		if (fo::same_seq(seg)) continue;
		std::string target, src;
		fo::encode_unit(seg, enc, target, src);
		const int n = (int)target.size();
		std::vector<int> t(n);
		for (int c = 0; c < n; c++) t[c] = code_of(target[c]);
		// ---- segment DP; per (virtual lane, column) maxima
		std::vector<uint16_t> lanemax((size_t)lay.nv * n, 0);
		std::vector<int> cols(n, 0);
		{
			std::vector<int> H(mp, 0), E(mp, 0);
			for (int c = 0; c < n; c++) {
				int diag = 0, F = 0, cm = 0;
				for (int i = 0; i < mp; i++) {
					int hh = diag + (i < m ? sc(q[i], t[c]) : 0);
					hh = std::max(std::max(hh, 0), std::max(E[i], F));
					diag = H[i]; H[i] = hh;
					if (hh > cm) cm = hh;
					uint16_t& lm = lanemax[(size_t)lay.lane_of_row[i] * n + c];
					if (hh > lm) lm = (uint16_t)hh;
					E[i] = std::max(std::max(E[i] - 4, hh - 16), 0);
					F = std::max(std::max(F - 4, hh - 16), 0);
				}
				cols[c] = cm;
			}
		}
		int mx = 0, cut = n;
		for (int c = 0; c < n; c++) { mx = std::max(mx, cols[c]); if (cut == n && cols[c] >= 251) cut = c; }
		for (int c = cut; c < n; c++) cols[c] = 0;
		const int thr = (int)(mx * 0.8);
		std::vector<fo::Cand> cands = fo::pick_candidates(cols, thr);
		// ---- block maxima U_K[v][b], b = (c + v) / K
		std::vector<std::vector<uint16_t>> UK(NK);
		std::vector<int> nb(NK);
		for (int k = 0; k < NK; k++) {
			const int K = KS[k];
			nb[k] = (n + lay.nv + K - 1) / K + 1;
			UK[k].assign((size_t)lay.nv * nb[k], 0);
			for (int v = 0; v < lay.nv; v++) for (int c = 0; c < n; c++) {
				uint16_t& x = UK[k][(size_t)v * nb[k] + (c + v) / K];
				x = std::max(x, lanemax[(size_t)v * n + c]);
			}
		}
		for (const fo::Cand& cd : cands) {
			ncand++;
			float Iden = 0.6f;
			int bestS = 0, it = 0, prevS = 0;
			bool have_best = false;
			while (Iden <= 1) {
				int L = (int)((int)(cd.score + 24) / (9 * Iden - 4) + 1);
				L = cd.pos - L + 1 > 0 ? L : cd.pos + 1;
				const int w0 = cd.pos - L + 1;
				if (have_best) break;                      // the engine skips tries once a best one exists
				const WinRes wr = window_dp(q, m, mp, t.data() + w0, L);
				ntries++; try_hist[it]++;
				full_cells += (double)mp * (L + 2);
				if (it > 0 && prevS > 0) { const int rr = (int)(100.0 * wr.S / prevS); ratio_hist[rr / 5 * 5]++; ratio_sum += (double)wr.S / prevS; ratio_n++; }
				const int prevS_pol = prevS;
				prevS = wr.S;
				// ---- flagged lanes per K with theta = the try's true score
				for (int k = 0; k < NK; k++) {
					const int K = KS[k];
					int vmin = 1 << 30, vmax = -1, nfl = 0, umax = 0;
					std::vector<int> ub(lay.nv);
					for (int v = 0; v < lay.nv; v++) {
						int uu = 0;
						for (int b = (w0 + v) / K; b <= (cd.pos + v) / K; b++) uu = std::max(uu, (int)UK[k][(size_t)v * nb[k] + b]);
						uu = std::min(uu, 5 * L);
						ub[v] = uu; umax = std::max(umax, uu);
						if (uu >= wr.S) { nfl++; vmin = std::min(vmin, v); vmax = std::max(vmax, v); }
					}
					if (k == 2 && umax >= 148) q2poss++;
					flagged_sum[k] += nfl; span_sum[k] += vmax - vmin + 1;
					if (k == 2) span_hist16[vmax - vmin + 1]++;
					for (int mi = 0; mi < NM; mi++) {
						const int M = MS[mi];
						int v0 = std::max(0, vmin - M);
						v0 &= ~1;                                  // bands start on an even virtual lane (lane pair)
						const int r0 = lay.row0[v0], r1 = lay.row0[vmax] + lay.rows[vmax];
						const int utop = v0 > 0 ? ub[v0 - 1] : 0;
						const BandRes br = band_dp(q, m, r0, r1, t.data() + w0, L, utop);
						const bool ok = br.clean && br.S == wr.S && br.ref_end == wr.ref_end && br.read_end == wr.read_end;
						if (br.clean && !ok) { fprintf(stderr, "BUG: clean band result differs: unit %d cand %d/%d try %d K %d M %d: band %d %d %d vs %d %d %d (v %d..%d)\n", u, cd.score, cd.pos, it, K, M, br.S, br.ref_end, br.read_end, wr.S, wr.ref_end, wr.read_end, v0, vmax); }
						if (getenv("BAND_DEBUG") && k == 2 && mi == 4 && !ok) fprintf(stderr, "fail: try %d s %d L %d S %d end %d/%d | band S %d end %d/%d clean %d utop %d lanes %d..%d (flag %d..%d) rows %d..%d\n", it, cd.score, L, wr.S, wr.ref_end, wr.read_end, br.S, br.ref_end, br.read_end, (int)br.clean, utop, v0, vmax, vmin, vmax, r0, r1);
					const int nvb = vmax - v0 + 1;
						int cls = 8; while (cls < nvb) cls *= 2;
						if (!ok || cls > lay.nv) { band_fail[k][mi]++; band_cells[k][mi] += (double)mp * (L + 2) + (double)(r1 - r0) * (L + 2); band_cells_cls[k][mi] += (double)mp * (L + 2) + (double)cls * lay.rp * (L + 2); }
						else { band_cells[k][mi] += (double)(r1 - r0) * (L + 2); band_cells_cls[k][mi] += (double)cls * lay.rp * (L + 2); }
					}
				}
				// ---- guaranteed-margin scheme: for class C (lanes) the smallest provable threshold over all even placements
				for (int k = 0; k < NK; k++) {
					const int K = KS[k];
					std::vector<int> ub(lay.nv);
					for (int v = 0; v < lay.nv; v++) {
						int uu = 0;
						for (int b = (w0 + v) / K; b <= (cd.pos + v) / K; b++) uu = std::max(uu, (int)UK[k][(size_t)v * nb[k] + b]);
						ub[v] = std::min(uu, 5 * L);
					}
					std::vector<int> pre(lay.nv + 1, 0), suf(lay.nv + 2, 0);
					for (int v = 0; v < lay.nv; v++) pre[v + 1] = std::max(pre[v], ub[v]);
					for (int v = lay.nv - 1; v >= 0; v--) suf[v] = std::max(suf[v + 1], ub[v]);
					auto rows_max = [&](int th) { if (5 * L < th) return 0; const int g = (5 * L - th - 12) / 4; return (5 * L - th - 12 >= 4) ? L + g : L; };
					for (int ci = 0; ci < NCLS; ci++) {
						const int C = CLS[ci];
						int best = 1 << 30;
						for (int v0 = 0; v0 + C <= lay.nv; v0 += 2) {
							int th = std::max(pre[v0], suf[v0 + C]) + 1;
							if (v0 > 0) for (int kk = 0; kk < C; kk++) {
								const int Rk = lay.row0[v0 + kk] - lay.row0[v0];
								// smallest theta with rows_max(theta) <= Rk + 1
								int lo = 1, hi = 5 * L + 1;
								while (lo < hi) { const int mid = (lo + hi) / 2; if (rows_max(mid) <= Rk + 1) hi = mid; else lo = mid + 1; }
								th = std::max(th, std::min(ub[v0 + kk] + 1, lo));
							}
							best = std::min(best, th);
						}
						if (wr.S >= best) gm_ok[k][ci][it == 0 ? (wr.S >= cd.score ? 0 : 1) : 2]++;
					}
				}
				// ---- realistic policy (K = 64 bounds): heuristic placement for a target threshold, proof test, fallback
				{
					const int k = NK - 1, K = KS[k];
					std::vector<int> ub(lay.nv);
					for (int v = 0; v < lay.nv; v++) {
						int uu = 0;
						for (int b = (w0 + v) / K; b <= (cd.pos + v) / K; b++) uu = std::max(uu, (int)UK[k][(size_t)v * nb[k] + b]);
						ub[v] = std::min(uu, 5 * L);
					}
					auto rows_max = [&](int th) { if (5 * L < th) return 0; const int g = (5 * L - th - 12) / 4; return (5 * L - th - 12 >= 4) ? L + g : L; };
					auto theta_min = [&](int v0, int C) {
						int th = 1;
						for (int v = 0; v < v0; v++) th = std::max(th, ub[v] + 1);
						for (int v = v0 + C; v < lay.nv; v++) th = std::max(th, ub[v] + 1);
						if (v0 > 0) for (int kk = 0; kk < C; kk++) {
							const int Rk = lay.row0[v0 + kk] - lay.row0[v0];
							int lo = 1, hi = 5 * L + 1;
							while (lo < hi) { const int mid = (lo + hi) / 2; if (rows_max(mid) <= Rk + 1) hi = mid; else lo = mid + 1; }
							th = std::max(th, std::min(ub[v0 + kk] + 1, lo));
						}
						return th;
					};
					// returns the class used (0 = none) and its theta_min for a target threshold
					auto place = [&](int target, int maxC, int* thmin) {
						int vf = -1, vl = -1;
						for (int v = 0; v < lay.nv; v++) if (ub[v] >= target) { if (vf < 0) vf = v; vl = v; }
						if (vf < 0) { vf = vl = lay.lane_of_row[std::min(wr.read_end, mp - 1)]; }
						const int need_rows = rows_max(target);
						int top_row = lay.row0[vf] - need_rows + 1; if (top_row < 0) top_row = 0;
						int v0 = lay.lane_of_row[top_row] & ~1;
						for (int C = 16; C <= maxC; C *= 2) {
							if (vl - v0 + 1 > C) continue;
							int vv = std::min(v0, lay.nv - C); vv &= ~1;
							const int th = theta_min(vv, C);
							if (th <= target) { *thmin = th; return C; }
						}
						return 0;
					};
					const double cols = L + 2;
					const int grp = it == 0 ? (wr.S >= cd.score ? 0 : 1) : 2;
					for (int pol = 0; pol < NPOL; pol++) {
						const int maxC = pol_maxC[pol];
						const double rho = pol_rho[pol];
						const int target = it == 0 ? cd.score : std::max(1, (int)(rho * prevS_pol));
						int thmin = 0;
						const int C = place(target, maxC, &thmin);
						double cost = 0;
						if (C == 0) { cost = lay.nv * cols; pol_direct_full[pol][grp]++; }
						else if (wr.S >= thmin) { cost = C * cols; pol_proven[pol][grp]++; }
						else {
							cost = C * cols;
							// second attempt: target = what a clean band result would be (unknown here: use the true score as the optimistic case)
							int th2 = 0; const int C2 = pol_second[pol] ? place(wr.S, 64, &th2) : 0;
							if (C2 > C && wr.S >= th2) { cost += C2 * cols; pol_second_ok[pol][grp]++; }
							else { cost += lay.nv * cols; pol_fallback[pol][grp]++; }
						}
						pol_cost[pol][grp] += cost;
					}
					pol_full[grp] += lay.nv * cols;
				}
				gm_n[it == 0 ? (wr.S >= cd.score ? 0 : 1) : 2]++;
				gm_cols[it == 0 ? (wr.S >= cd.score ? 0 : 1) : 2] += L + 2;
				if (wr.S >= cd.score) { naccept++; break; }
				if (wr.S > bestS && wr.ref_end == L - 1) { bestS = wr.S; have_best = true; }
				Iden += 0.1f; it++;
			}
		}
		if ((u + 1) % 10 == 0) fprintf(stderr, "unit %d/%d\r", u + 1, nunits);
	}
	printf("units %d candidates %.0f (%.1f/unit) accepted %.1f%% tries %.0f (%.2f/cand) by try: %ld %ld %ld %ld\n", nunits, ncand, ncand / nunits, 100 * naccept / ncand, ntries, ntries / ncand, try_hist[0], try_hist[1], try_hist[2], try_hist[3]);
	printf("full-height cells per try %.0f ; windows with an upper bound >= 148 (Q2 possible): %.2f%%\n", full_cells / ntries, 100 * q2poss / ntries);
	printf("S_t / S_(t-1) mean %.3f ; histogram (pct: count):", ratio_sum / ratio_n);
	for (auto& kv : ratio_hist) printf(" %d:%ld", kv.first, kv.second);
	printf("\nspan histogram K=16 (lanes: tries):");
	long acc = 0; for (auto& kv : span_hist16) { acc += kv.second; if (kv.first <= 16 || kv.first % 8 == 0) printf(" %d:%.1f%%", kv.first, 100.0 * acc / ntries); }
	printf("\n");
	for (int k = 0; k < NK; k++) {
		printf("K=%-3d flagged %.2f lanes, span %.2f lanes |", KS[k], flagged_sum[k] / ntries, span_sum[k] / ntries);
		for (int mi = 0; mi < NM; mi++) printf("  M=%d: cells %.3f cls %.3f fail %.2f%%", MS[mi], band_cells[k][mi] / full_cells, band_cells_cls[k][mi] / full_cells, 100 * band_fail[k][mi] / ntries);
		printf("\n");
	}
	const char* gname[3] = { "try1 accepted", "try1 not accepted", "tries 2-4" };
	for (int pol = 0; pol < NPOL; pol++) {
		double c = 0, f = 0;
		printf("policy maxC=%d rho=%.2f second=%d:", pol_maxC[pol], pol_rho[pol], pol_second[pol]);
		for (int g = 0; g < 3; g++) { c += pol_cost[pol][g]; f += pol_full[g]; printf("  [%s: cost %.3f proven %.0f fb %.0f direct %.0f 2nd %.0f]", gname[g], pol_cost[pol][g] / pol_full[g], pol_proven[pol][g], pol_fallback[pol][g], pol_direct_full[pol][g], pol_second_ok[pol][g]); }
		printf("  TOTAL %.3f\n", c / f);
	}
	for (int g = 0; g < 3; g++) {
		printf("%s: %.0f tries, %.1f columns avg; provable with class (lanes):\n", gname[g], gm_n[g], gm_cols[g] / gm_n[g]);
		for (int k = 0; k < NK; k++) { printf("   K=%-3d", KS[k]); for (int ci = 0; ci < NCLS; ci++) printf("  C=%d: %.1f%%", CLS[ci], 100 * gm_ok[k][ci][g] / gm_n[g]); printf("\n"); }
	}
	return 0;
}
