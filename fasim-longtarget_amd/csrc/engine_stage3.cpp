// fasim-longtarget_amd/csrc/engine_stage3.cpp -- stage 3 of the host engine: window tries (band passes, reverse pass, full-height
// passes), exact replays, finish kernels, and the per-batch driver stage3_range (candidates -> triplex records).
#include "engine.h"


int run_finish(fasim_engine* E, const UnitBatch& B, const std::vector<WindowProb>& W, const std::vector<FwdOut>& fo,
	std::vector<AlignResult>& out, std::vector<uint32_t>& cigars, std::vector<char>& status);
bool align_v2_fits(const fasim_engine* E, const std::vector<WindowProb>& W);
int run_fwd(fasim_engine* E, const UnitBatch& B, const std::vector<WindowProb>& W, std::vector<FwdOut>& fo, bool word, const FwdZones* Z = nullptr);

// a9-a11: ssw_align for a list of windows (forward + reverse on the GPU, 16-bit re-runs, banded traceback)
int run_align(fasim_engine* E, const UnitBatch& B, const std::vector<WindowProb>& W, std::vector<AlignResult>& out,
	std::vector<uint32_t>& cigars, fasim_scan_stats* stats)
{
	const int n = (int)W.size();
	out.assign(n, AlignResult());
	if (!n) return FASIM_OK;
	ProfScope ps(6, "run_align (exact) total");
	std::vector<StripedProb> probs(n);
	for (int k = 0; k < n; k++) {
		probs[k].tbase = (int64_t)W[k].unit * B.tstride; probs[k].t0 = W[k].t0; probs[k].ref_len = W[k].len;
		probs[k].q_len = E->m; probs[k].unit = k; probs[k].aux = 0; probs[k].pad = 0;
	}
	HIPOK(E->ends.ensure(sizeof(AlignEnds) * n));
	int rc = run_striped(E, MODE_ALIGN, false, probs, false, tcv(E), E->m);
	if (rc) return rc;
	std::vector<AlignEnds> ends(n);
	HIPOK(hipMemcpyAsync(ends.data(), E->ends.p, sizeof(AlignEnds) * n, hipMemcpyDeviceToHost, E->st));
	HIPOK(hipStreamSynchronize(E->st));
	std::vector<int> redo;
	for (int k = 0; k < n; k++) if (ends[k].score_fwd == 255) redo.push_back(k);
	std::vector<char> from_sys(n, 0);        // forward result taken from the systolic kernel (16-bit semantics)
	if (!redo.empty()) {
		// bests[0].score == 255 -> the whole alignment is redone with the 16-bit kernels (sswNew.cpp:1473-1477), which have
		// no overflow rule and no signed-compare problem: their result is the textbook one
		std::vector<WindowProb> W2(redo.size());
		for (size_t r = 0; r < redo.size(); r++) W2[r] = W[redo[r]];
		if (align_v2_fits(E, W2)) {
			std::vector<FwdOut> f2;
			rc = run_fwd(E, B, W2, f2, true); if (rc) return rc;
			for (size_t r = 0; r < redo.size(); r++) {
				AlignEnds& e = ends[redo[r]];
				e.score_fwd = f2[r].score; e.ref_end = f2[r].ref_end; e.read_end = f2[r].read_end;
				e.score_rev = f2[r].score; e.ref_begin = 0; e.read_begin = 0;
				from_sys[redo[r]] = 1;
			}
		} else {
			std::vector<StripedProb> wp(redo.size());
			for (size_t r = 0; r < redo.size(); r++) { wp[r] = probs[redo[r]]; wp[r].unit = (int)r; }
			rc = run_striped(E, MODE_ALIGN, true, wp, false, tcv(E), E->m);
			if (rc) return rc;
			std::vector<AlignEnds> we(redo.size());
			HIPOK(hipMemcpyAsync(we.data(), E->ends.p, sizeof(AlignEnds) * redo.size(), hipMemcpyDeviceToHost, E->st));
			HIPOK(hipStreamSynchronize(E->st));
			for (size_t r = 0; r < redo.size(); r++) ends[redo[r]] = we[r];
		}
		if (stats) stats->align_word_reruns += (int64_t)redo.size();
	}
	// banded traceback for every alignment with a positive score
	std::vector<int> bidx;
	std::vector<BandProb> bp;
	for (int k = 0; k < n; k++) {
		const AlignEnds& e = ends[k];
		if (e.score_fwd <= 0 || e.ref_end < 0 || e.ref_begin < 0) continue;     // nothing aligned -> sw_score 0
		const int rl = e.ref_end - e.ref_begin + 1, ql = e.read_end - e.read_begin + 1;
		if (rl <= 0 || ql <= 0) continue;
		BandProb b;
		b.tbase = (int64_t)W[k].unit * B.tstride + W[k].t0 + e.ref_begin;
		b.q_begin = e.read_begin; b.ref_len = rl; b.read_len = ql;
		b.score = e.score_rev < e.score_fwd ? e.score_rev : e.score_fwd;          // sswNew.cpp:1518
		b.scratch_off = 0; b.scratch_cap = 0; b.pad = 0;
		bidx.push_back(k); bp.push_back(b);
	}
	std::vector<BandOut> bo(bp.size());
	std::vector<int> todo(bp.size());
	for (size_t i = 0; i < todo.size(); i++) todo[i] = (int)i;
	std::vector<char> via_finish(bp.size(), 0);
	std::vector<AlignResult> fres;
	if (!E->align_v1 && !bp.empty()) {
		// traceback through the finish kernel with the exact (ref_begin, read_begin, score) supplied (flag 2); only what
		// it cannot hold falls through to k_banded below
		std::vector<WindowProb> W2(bp.size()); std::vector<FwdOut> f2(bp.size());
		for (size_t i = 0; i < bp.size(); i++) {
			const int k = bidx[i]; const AlignEnds& e = ends[k];
			W2[i] = W[k];
			f2[i].score = bp[i].score; f2[i].ref_end = e.ref_end; f2[i].read_end = e.read_end; f2[i].flags = from_sys[k] ? 0 : 2;
			f2[i].ref_begin = e.ref_begin; f2[i].read_begin = e.read_begin;
		}
		std::vector<char> fst;
		rc = run_finish(E, B, W2, f2, fres, cigars, fst); if (rc) return rc;
		std::vector<int> left;
		for (size_t i = 0; i < bp.size(); i++) {
			if (fst[i] == 2) { left.push_back((int)i); continue; }
			via_finish[i] = 1;
			bo[i].status = fst[i] == 0 ? 0 : 1;
		}
		todo.swap(left);
	}
	size_t cap = 8192;
	for (int attempt = 0; attempt < 4 && !todo.empty(); attempt++, cap *= 32) {
		std::vector<BandProb> cur(todo.size());
		// keep the scratch arena bounded: process in slices
		const size_t max_arena = (size_t)6 << 30;
		size_t per_slice = std::max<size_t>(1, max_arena / cap);
		for (size_t s0 = 0; s0 < todo.size(); s0 += per_slice) {
			const size_t cnt = std::min(per_slice, todo.size() - s0);
			for (size_t i = 0; i < cnt; i++) { cur[i] = bp[todo[s0 + i]]; cur[i].scratch_off = (int64_t)(i * cap); cur[i].scratch_cap = (int)cap; }
			HIPOK(E->scratch.ensure(cnt * cap));
			HIPOK(E->bout.ensure(sizeof(BandOut) * cnt));
			rc = upload(E, E->bprobs, cur.data(), sizeof(BandProb) * cnt);
			if (rc) return rc;
			hipError_t he;
			{
				TimedScope ts(E, 6);
				he = launch_banded(tcv(E), E->q2.as<uint8_t>(), E->bprobs.as<BandProb>(), (int)cnt,
					E->scratch.as<uint8_t>(), E->bout.as<BandOut>(), E->st);
			}
			if (he != hipSuccess) return fail(E, FASIM_E_HIP, "banded kernel launch failed: %s", hipGetErrorString(he));
			std::vector<BandOut> tmp(cnt);
			HIPOK(hipMemcpyAsync(tmp.data(), E->bout.p, sizeof(BandOut) * cnt, hipMemcpyDeviceToHost, E->st));
			HIPOK(hipStreamSynchronize(E->st));
			for (size_t i = 0; i < cnt; i++) bo[todo[s0 + i]] = tmp[i];
		}
		std::vector<int> next;
		for (int i : todo) if (bo[i].status == 2) next.push_back(i);
		todo.swap(next);
	}
	if (!todo.empty()) return fail(E, FASIM_E_UNSUPPORTED, "banded traceback of %zu alignments exceeds the scratch limit", todo.size());
	for (size_t i = 0; i < bp.size(); i++) {
		const int k = bidx[i];
		const AlignEnds& e = ends[k];
		AlignResult& r = out[k];
		if (bo[i].status != 0) { r.sw_score = 0; r.failed = 1; continue; }    // NULL from ssw_align -> sw_score 0 (ssw_cpp.cpp:631-633)
		r.sw_score = bp[i].score; r.ref_begin = e.ref_begin; r.ref_end = e.ref_end;
		r.query_begin = e.read_begin; r.query_end = e.read_end;
		if (via_finish[i]) {
			r.cigar_len = fres[i].cigar_len; r.cigar_off = fres[i].cigar_off;
			if (from_sys[k]) { r.ref_begin = fres[i].ref_begin; r.query_begin = fres[i].query_begin; }
			if (fres[i].sw_score <= 0) { r.sw_score = 0; r.failed = 1; }
			continue;
		}
		r.cigar_len = bo[i].cigar_len;
		r.cigar_off = (uint32_t)cigars.size();
		cigars.insert(cigars.end(), bo[i].cigar, bo[i].cigar + bo[i].cigar_len);
	}
	return FASIM_OK;
}

// ---- stage 3 through align.hip ---------------------------------------------------------------------
bool align_v2_fits(const fasim_engine* E, const std::vector<WindowProb>& W)
{
	if (E->align_v1 || !systolic_fits(E->m)) return false;
	for (const WindowProb& w : W) if (w.len > 200 || w.len <= 0) return false;
	return true;
}

// forward pass of every window (k_build_stream + k_align_fwd): score, ref_end, read_end, hazard flag
// word = false: the reference's 8-bit pass (taint-tracking kernel; scores from 251 on only mean "overflow");
// word = true : its 16-bit pass (plain kernel, exact scores up to 980, flags always 0)
int run_fwd(fasim_engine* E, const UnitBatch& B, const std::vector<WindowProb>& W, std::vector<FwdOut>& fo, bool word, const FwdZones* Z)
{
	const int n = (int)W.size();
	fo.resize(n);
	if (!n) return FASIM_OK;
	ProfScope ps(0, "run_fwd total");
	std::vector<FwdProb> probs(n);
	int64_t off = 0;
	for (int k = 0; k < n; k++) {
		probs[k].tbase = (int64_t)W[k].unit * B.tstride + W[k].t0; probs[k].len = W[k].len; probs[k].stream_off = (int32_t)off;
		off += W[k].len + 2;
	}
	if (off > 0x7fff0000ll) return fail(E, FASIM_E_UNSUPPORTED, "window stream of one round exceeds 2 GiB; lower FASIM_SEG_BATCH");
	const int per_task = std::max(8, std::min(64, n / 3072));
	std::vector<int32_t> tasks;
	for (int k = 0; k < n; k += per_task) tasks.push_back(k);
	tasks.push_back(n);
	double tp = now_s();
	int rc = upload_async(E, E->fprobs, probs.data(), sizeof(FwdProb) * n); if (rc) return rc;      // (both vectors outlive the
	// Z: the REVERSE pass (plain kernel, reversed query and windows): leaves lane maxima per zone in E->lane_ub and no FwdOut
	const bool emit = Z != nullptr;
	if (emit && (!word || !E->lane_ub.p || (int)Z->zones.size() != n || (int)Z->slot.size() != n)) return fail(E, FASIM_E_ARG, "reverse pass: bad arguments");
	if (emit) {
		rc = upload_async(E, E->fzones, Z->zones.data(), sizeof(uint32_t) * n); if (rc) return rc;
		rc = upload_async(E, E->fubslot, Z->slot.data(), sizeof(int32_t) * n); if (rc) return rc;
	}
	rc = upload(E, E->ftasks, tasks.data(), sizeof(int32_t) * tasks.size()); if (rc) return rc;          //  synchronisation in here)
	g_prof.add(1, "run_fwd upload", now_s() - tp);
	HIPOK(E->fstream.ensure((size_t)off + 256));
	HIPOK(E->fout.ensure(sizeof(FwdOut) * n));
	GateScope gate(E);
	hipError_t he = launch_build_stream(tcv(E), E->fprobs.as<FwdProb>(), n, E->fstream.as<uint8_t>(), emit ? E->fzones.as<uint32_t>() : nullptr, E->st);
	if (he != hipSuccess) return fail(E, FASIM_E_HIP, "build_stream launch failed: %s", hipGetErrorString(he));
	FwdLaunch L;
	L.stream = E->fstream.as<uint8_t>(); L.probs = E->fprobs.as<FwdProb>(); L.task_first = E->ftasks.as<int32_t>();
	L.ntask = (int)tasks.size() - 1; L.counter = E->counter.as<uint32_t>(); L.qcodes = E->q2.as<uint8_t>(); L.m = E->m;
	L.out = E->fout.as<FwdOut>(); L.word = word ? 1 : 0;
	if (emit) { L.lane_ub = E->lane_ub.as<uint16_t>(); L.ub_slot = E->fubslot.as<int32_t>(); }
	L.boundary = nullptr;
	if (systolic_tiles(E->m) > 1) { HIPOK(E->fboundary.ensure(((size_t)off + 256) * sizeof(uint4))); L.boundary = E->fboundary.as<uint4>(); }
	{ TimedScope ts(E, 2, E->st); he = launch_align_fwd(L, E->st); }
	if (he != hipSuccess) return fail(E, FASIM_E_HIP, "align_fwd launch failed: %s", hipGetErrorString(he));
	tp = now_s();
	HIPOK(hipStreamSynchronize(E->st));
	gate.release();
	g_prof.add(2, "run_fwd kernel wait", now_s() - tp);
	if (emit) return FASIM_OK;
	tp = now_s();
	HIPOK(hipMemcpyAsync(fo.data(), E->fout.p, sizeof(FwdOut) * n, hipMemcpyDeviceToHost, E->st));
	HIPOK(hipStreamSynchronize(E->st));
	g_prof.add(3, "run_fwd D2H", now_s() - tp);
	return FASIM_OK;
}

// Forward pass as ssw_align runs it: the 8-bit pass first; a maximum of 251 or more overflows the reference's 8-bit
// kernel, which then repeats the whole alignment with its 16-bit kernels (sswNew.cpp:1473-1477, no overflow rule,
// unsigned-safe compare) -> second pass with the plain systolic kernel for those windows (flags = 4).  A window
// whose winning cell is tainted (flags & 1) is not trusted either way: the caller replays it exactly.
int run_fwd_both(fasim_engine* E, const UnitBatch& B, const std::vector<WindowProb>& W, std::vector<FwdOut>& fo, int64_t* word_reruns)
{
	int rc = run_fwd(E, B, W, fo, false); if (rc) return rc;
	std::vector<int> ov;
	for (size_t i = 0; i < fo.size(); i++) if (!(fo[i].flags & 1) && fo[i].score >= 255 - BIAS) ov.push_back((int)i);
	if (ov.empty()) return FASIM_OK;
	std::vector<WindowProb> W2(ov.size()); std::vector<FwdOut> f2;
	for (size_t r = 0; r < ov.size(); r++) W2[r] = W[ov[r]];
	rc = run_fwd(E, B, W2, f2, true); if (rc) return rc;
	for (size_t r = 0; r < ov.size(); r++) { fo[ov[r]] = f2[r]; fo[ov[r]].flags = 4; }
	if (word_reruns) *word_reruns += (int64_t)ov.size();
	return FASIM_OK;
}

// Banded forward pass (band.hip).  For every try k with target score target[k] (what the try is expected to reach) k_band_select
// picks a row band and the score theta_min from which the band's result is provably the full-height result; k_align_band runs
// the bands; a try whose band came back below its theta_min gets a second band chosen for the score it did reach (a lower
// bound of the true score).  On return fo[k].flags & 24 marks the tries that still need the full-height kernel.
int run_fwd_band(fasim_engine* E, const UnitBatch& B, const std::vector<WindowProb>& W, const std::vector<int32_t>& target,
	const std::vector<int32_t>* prev, std::vector<FwdOut>& fo, fasim_scan_stats* st)
{
	const int n = (int)W.size();
	fo.resize(n);
	if (!n) return FASIM_OK;
	ProfScope ps(29, "run_fwd_band total");
	const int mask = band_mask(E);
	std::vector<FwdProb> probs(n);
	for (int k = 0; k < n; k++) { probs[k].tbase = (int64_t)W[k].unit * B.tstride + W[k].t0; probs[k].len = W[k].len; probs[k].stream_off = W[k].unit; }      // (stream_off carries the unit here)
	int rc = upload_async(E, E->fprobs, probs.data(), sizeof(FwdProb) * n); if (rc) return rc;
	HIPOK(E->fout.ensure(sizeof(FwdOut) * n));
	HIPOK(E->bcounts.ensure(sizeof(uint32_t) * (BAND_COUNTS + 3 * BAND_MAX_ZONES)));
	HIPOK(E->bdec.ensure(sizeof(int4) * (size_t)n));
	for (int c = 0; c < 3; c++) {
		if (!((mask >> c) & 1)) continue;
		HIPOK(E->blist[c].ensure(sizeof(BandTry) * (size_t)n));
		HIPOK(E->bslots[c].ensure(sizeof(uint16_t) * BAND_SLOT_COLS * (size_t)n));
	}
	const bool second = true;      // (a band that came back below its theta_min is followed by one chosen for the score it reached)
	std::vector<int32_t> idx, tgt2, prev2;
	const bool have_prev = prev && (int)prev->size() == n && E->lane_ub.p;
	for (int pass = 0; pass < (second ? 2 : 1); pass++) {
		int np = n;
		const int32_t* tsrc = target.data();
		const int32_t* psrc = have_prev ? prev->data() : nullptr;
		if (pass == 1) {
			idx.clear(); tgt2.clear();
			// (a try with start-based bounds was given its exact score as the target: a second band would be the same one)
			for (int k = 0; k < n; k++) if (fo[k].flags == 8 && fo[k].score > 0 && !(have_prev && (*prev)[k] >= 0)) { idx.push_back(k); tgt2.push_back(fo[k].score); if (have_prev) prev2.push_back(-1); }
			np = (int)idx.size(); tsrc = tgt2.data(); if (have_prev) psrc = prev2.data();
			if (!np) break;
			rc = upload_async(E, E->bidx, idx.data(), sizeof(int32_t) * np); if (rc) return rc;
		}
		rc = upload_async(E, E->btarget, tsrc, sizeof(int32_t) * np); if (rc) return rc;
		if (psrc) { rc = upload_async(E, E->bprev, psrc, sizeof(int32_t) * np); if (rc) return rc; }
		BandSelLaunch S;
		if (psrc) { S.prev_ub = E->lane_ub.as<uint16_t>(); S.prev = E->bprev.as<int32_t>(); }
		S.probs = E->fprobs.as<FwdProb>(); S.target = E->btarget.as<int32_t>(); S.idx = pass ? E->bidx.as<int32_t>() : nullptr; S.n = np; S.tstride = B.tstride;
		S.ublk = E->ublk.as<uint16_t>(); S.ublk_blocks = E->ublk_blocks; S.m = E->m; S.tcodes = tcv(E);
		for (int c = 0; c < 3; c++) { S.list[c] = E->blist[c].as<BandTry>(); S.slots[c] = E->bslots[c].as<uint16_t>(); }
		S.list_cap = (uint32_t)n; S.counts = E->bcounts.as<uint32_t>(); S.cursors = E->bcounts.as<uint32_t>() + BAND_COUNTS; S.dec = E->bdec.as<int4>();
		S.out = E->fout.as<FwdOut>(); S.class_mask = mask;
		static const bool dbg = getenv("FASIM_BAND_DEBUG") != nullptr;
		S.debug = dbg ? 1 : 0;
		hipError_t he;
		{ TimedScope ts(E, 9); he = launch_band_decide(S, E->st); }
		if (he != hipSuccess) return fail(E, FASIM_E_HIP, "band_decide launch failed: %s", hipGetErrorString(he));
		uint32_t counts[BAND_COUNTS] = { 0 };
		HIPOK(hipMemcpyAsync(counts, E->bcounts.p, sizeof counts, hipMemcpyDeviceToHost, E->st));
		HIPOK(hipStreamSynchronize(E->st));       // (the host vectors uploaded above may go out of scope from here on)
		// the (class, zone) segments of the lists, and the workgroups that will serve them
		uint32_t first[3 * BAND_MAX_ZONES], per_class[3] = { 0, 0, 0 };
		for (int c = 0; c < 3; c++) for (int z = 0; z < BAND_MAX_ZONES; z++) { first[c * BAND_MAX_ZONES + z] = per_class[c]; per_class[c] += counts[c * BAND_MAX_ZONES + z]; }
		if (per_class[0] + per_class[1] + per_class[2]) {
			std::vector<BandZoneTab> tabs[3]; std::vector<BandZoneTab> all;
			size_t toff[3] = { 0, 0, 0 };
			for (int c = 0; c < 3; c++) { if (per_class[c]) tabs[c] = band_plan(E->m, c, counts + c * BAND_MAX_ZONES, first + c * BAND_MAX_ZONES); toff[c] = all.size(); all.insert(all.end(), tabs[c].begin(), tabs[c].end()); }
			HIPOK(hipMemcpyAsync(S.cursors, first, sizeof first, hipMemcpyHostToDevice, E->st));
			rc = upload_async(E, E->btab, all.data(), sizeof(BandZoneTab) * all.size()); if (rc) return rc;
			{ TimedScope ts(E, 9); he = launch_band_emit(S, E->st); }
			if (he != hipSuccess) return fail(E, FASIM_E_HIP, "band_emit launch failed: %s", hipGetErrorString(he));
			HIPOK(hipStreamSynchronize(E->st));       // (`first` and `all` are host stack / heap)
			GateScope gate(E);
			for (int c = 0; c < 3; c++) {
				if (!per_class[c]) continue;
				BandLaunch L;
				L.list = E->blist[c].as<BandTry>(); L.slots = E->bslots[c].as<uint16_t>(); L.tab = E->btab.as<BandZoneTab>() + toff[c]; L.nwg = (int)tabs[c].size(); L.cls = c;
				L.qcodes = E->q2.as<uint8_t>(); L.m = E->m; L.out = E->fout.as<FwdOut>();
				{ TimedScope ts(E, 8, E->st); he = launch_align_band(L, E->st); }
				if (he != hipSuccess) return fail(E, FASIM_E_HIP, "align_band launch failed: %s", hipGetErrorString(he));
				const int64_t cells = (int64_t)counts[BAND_COUNT_COLS + c] * 48 * (8 << c);
				if (st) { st->band_tries += per_class[c]; st->band_cells += cells; st->cells_stage3 += cells; }
			}
			HIPOK(hipStreamSynchronize(E->st));
		} else {
			// nothing banded: every try of the pass is marked for the full-height kernel
			HIPOK(hipMemcpyAsync(S.cursors, first, sizeof first, hipMemcpyHostToDevice, E->st));
			{ TimedScope ts(E, 9); he = launch_band_emit(S, E->st); }
			if (he != hipSuccess) return fail(E, FASIM_E_HIP, "band_emit launch failed: %s", hipGetErrorString(he));
			HIPOK(hipStreamSynchronize(E->st));
		}
		HIPOK(hipMemcpyAsync(fo.data(), E->fout.p, sizeof(FwdOut) * n, hipMemcpyDeviceToHost, E->st));
		HIPOK(hipStreamSynchronize(E->st));
		if (dbg) {
			long proven = 0, unproven = 0, withprev = 0;
			for (int k = 0; k < n; k++) { if (!(fo[k].flags & 24)) proven++; else if (fo[k].flags & 8) unproven++; }
			if (psrc) for (int k = 0; k < np; k++) if (psrc[k] >= 0) withprev++;
			fprintf(stderr, "[band] pass %d: %d tries looked at (%ld with bounds of an earlier pass), classes %u / %u / %u, bound >= 148: %u, no band: %u; after the pass %ld of %d proven, %ld unproven\n",
				pass, np, withprev, per_class[0], per_class[1], per_class[2], counts[BAND_COUNT_HOT], counts[BAND_COUNT_NOBAND], proven, n, unproven);
		}
	}
	if (st) for (int k = 0; k < n; k++) if (!(fo[k].flags & 24)) st->band_proven++;
	return FASIM_OK;
}

// Forward pass of a round of tries.
//   1. band pass: a try whose candidate has start-based bounds from a reverse pass (ru[k] = slot * 4 + zone) gets the band those
//      prove (its exact score is known); at a candidate's first try the block maxima of k_scan bound the window instead and the
//      target is the candidate's own score (an accepted try reaches it); target 0 = no attempt.
//   2. reverse pass (plain full-height kernel on the reversed problem) for the unproven tries without such bounds: leaves the
//      bounds of this try (zone 0) and of the candidate's later tries (zones 1-3) in E->lane_ub; got_ru[k] = 1.
//   3. band pass of those tries with the new bounds.
//   4. whatever is still unproven (scores that can meet the reference's Q2 / overflow behaviour, start lanes too far apart for a
//      band) takes the full-height forward passes (8-bit with taint tracking, 16-bit where the maximum reaches 251).
bool band_ready(const fasim_engine* E, const UnitBatch& B) { return band_mask(E) != 0 && E->ublk_units >= B.nunit && E->ublk_blocks > 0; }
int run_fwd_smart(fasim_engine* E, const UnitBatch& B, const std::vector<WindowProb>& W, const std::vector<int32_t>& target,
	const std::vector<int32_t>* ru, const FwdZones* Z, std::vector<FwdOut>& fo, std::vector<char>* got_ru, fasim_scan_stats& st)
{
	if (got_ru) got_ru->assign(W.size(), 0);
	if (!band_ready(E, B)) {
		for (const WindowProb& w : W) st.cells_stage3 += (int64_t)E->m * w.len;
		return run_fwd_both(E, B, W, fo, &st.align_word_reruns);
	}
	int rc = run_fwd_band(E, B, W, target, ru, fo, &st); if (rc) return rc;
	std::vector<int> rest;
	for (size_t k = 0; k < fo.size(); k++) if (fo[k].flags & 24) rest.push_back((int)k);
	if (rest.empty()) return FASIM_OK;
	if (Z && ru && E->lane_ub.p) {
		std::vector<int> R;
		for (int k : rest) if ((*ru)[(size_t)k] < 0) R.push_back(k);
		if (!R.empty()) {
			std::vector<WindowProb> WR(R.size()); FwdZones ZR; std::vector<int32_t> tR(R.size(), 1 << 30), ruR(R.size());
			ZR.zones.resize(R.size()); ZR.slot.resize(R.size());
			for (size_t r = 0; r < R.size(); r++) {
				WR[r] = W[(size_t)R[r]]; ZR.zones[r] = Z->zones[(size_t)R[r]]; ZR.slot[r] = Z->slot[(size_t)R[r]]; ruR[r] = Z->slot[(size_t)R[r]] * 4;
				st.cells_stage3 += (int64_t)E->m * WR[r].len; st.rev_bound_passes++;
			}
			std::vector<FwdOut> dummy, fR;
			rc = run_fwd(E, B, WR, dummy, true, &ZR); if (rc) return rc;
			rc = run_fwd_band(E, B, WR, tR, &ruR, fR, &st); if (rc) return rc;
			for (size_t r = 0; r < R.size(); r++) { fo[(size_t)R[r]] = fR[r]; if (got_ru) (*got_ru)[(size_t)R[r]] = 1; }
			rest.clear();
			for (size_t k = 0; k < fo.size(); k++) if (fo[k].flags & 24) rest.push_back((int)k);
			if (rest.empty()) return FASIM_OK;
		}
	}
	std::vector<WindowProb> W3(rest.size()); std::vector<FwdOut> f3;
	for (size_t r = 0; r < rest.size(); r++) { W3[r] = W[(size_t)rest[r]]; st.cells_stage3 += (int64_t)E->m * W3[r].len; }
	rc = run_fwd_both(E, B, W3, f3, &st.align_word_reruns); if (rc) return rc;
	if (getenv("FASIM_BAND_DEBUG")) {
		int shown = 0;
		for (size_t r = 0; r < rest.size() && shown < 12; r++) {
			const FwdOut& b = fo[(size_t)rest[r]];
			if (b.flags != 8) continue;
			fprintf(stderr, "[band] unproven: unit %d t0 %d len %d band(score %d ref_end %d read_end %d) full(score %d ref_end %d read_end %d flags %d)\n",
				W3[r].unit, W3[r].t0, W3[r].len, b.score, b.ref_end, b.read_end, f3[r].score, f3[r].ref_end, f3[r].read_end, f3[r].flags);
			shown++;
		}
	}
	for (size_t r = 0; r < rest.size(); r++) fo[(size_t)rest[r]] = f3[r];
	return FASIM_OK;
}

// exact (stripe-faithful) reverse pass for windows whose forward result is exact but whose score (>= 148) would allow
// the signed lazy-F exit in the reverse pass.  Fills fo[k].score = min(forward, reverse), ref_begin, read_begin, flag 2;
// flag 1 is set where the result cannot be used (caller replays the candidate).
int run_rev_exact(fasim_engine* E, const UnitBatch& B, const std::vector<WindowProb>& W, std::vector<FwdOut>& fo,
	const std::vector<int>& idx)
{
	const int n = (int)idx.size();
	if (!n) return FASIM_OK;
	ProfScope ps(5, "run_rev_exact total");
	std::vector<StripedProb> probs(n);
	for (int i = 0; i < n; i++) {
		const int k = idx[i];
		probs[i].tbase = (int64_t)W[k].unit * B.tstride; probs[i].t0 = W[k].t0; probs[i].ref_len = fo[k].ref_end + 1;
		probs[i].q_len = fo[k].read_end + 1; probs[i].unit = i; probs[i].aux = fo[k].score; probs[i].pad = 0;
	}
	HIPOK(E->ends.ensure(sizeof(AlignEnds) * n));
	int rc = run_striped(E, MODE_REV, false, probs, false, tcv(E), E->m); if (rc) return rc;
	std::vector<AlignEnds> ends(n);
	HIPOK(hipMemcpyAsync(ends.data(), E->ends.p, sizeof(AlignEnds) * n, hipMemcpyDeviceToHost, E->st));
	HIPOK(hipStreamSynchronize(E->st));
	for (int i = 0; i < n; i++) {
		FwdOut& f = fo[idx[i]];
		const AlignEnds& e = ends[i];
		if (e.ref_begin < 0 || e.read_begin < 0 || e.score_rev >= 255) { f.flags |= 1; continue; }
		f.score = e.score_rev < f.score ? e.score_rev : f.score;          // sswNew.cpp:1518
		f.ref_begin = e.ref_begin; f.read_begin = e.read_begin; f.flags |= 2;
	}
	return FASIM_OK;
}

// reverse pass + banded traceback (k_finish) of windows whose forward result is known.
// status[k]: 0 = result valid (sw_score 0 when nothing aligned); 1 = the reference's traceback fails (NULL);
//            2 = must be decided by the stripe-faithful path
int run_finish(fasim_engine* E, const UnitBatch& B, const std::vector<WindowProb>& W, const std::vector<FwdOut>& fo,
	std::vector<AlignResult>& out, std::vector<uint32_t>& cigars, std::vector<char>& status)
{
	const int n = (int)W.size();
	out.assign(n, AlignResult()); status.assign(n, 0);
	if (!n) return FASIM_OK;
	ProfScope ps(4, "run_finish total");
	std::vector<FwdProb> probs(n);
	for (int k = 0; k < n; k++) { probs[k].tbase = (int64_t)W[k].unit * B.tstride + W[k].t0; probs[k].len = W[k].len; probs[k].stream_off = 0; }
	const int scratch_cap = 16384;
	int rc = upload_async(E, E->fprobs, probs.data(), sizeof(FwdProb) * n); if (rc) return rc;      // (probs, fo and order outlive the
	rc = upload_async(E, E->fout, fo.data(), sizeof(FwdOut) * n); if (rc) return rc;                 //  first synchronisation below)
	HIPOK(E->aout.ensure(sizeof(AlignOutDev) * n));
	const size_t pool_cap = (size_t)n * 12 + 4096;
	HIPOK(E->cigpool.ensure(pool_cap * sizeof(uint32_t)));
	HIPOK(E->cigcount.ensure(64));
	hipError_t he;
	std::vector<int32_t> order(n);
	{
		// process alignments grouped by score (a proxy for their size): the 64 threads of a wave then run DPs of similar
		// length instead of all waiting for the largest one
		std::vector<int32_t> cnt(1026, 0);
		for (int k = 0; k < n; k++) cnt[std::min(1024, std::max(0, fo[k].score)) + 1]++;
		for (int b = 1; b < 1026; b++) cnt[b] += cnt[b - 1];
		for (int k = 0; k < n; k++) order[cnt[std::min(1024, std::max(0, fo[k].score))]++] = k;
		rc = upload_async(E, E->forder, order.data(), sizeof(int32_t) * n); if (rc) return rc;
	}
	HIPOK(E->scratch.ensure((size_t)((n + 63) / 64) * 64 * 2048));
	{ TimedScope ts(E, 3);
	he = launch_finish(tcv(E), E->q2.as<uint8_t>(), E->fprobs.as<FwdProb>(), E->fout.as<FwdOut>(),
		E->forder.as<int32_t>(), n, E->scratch.as<uint8_t>(), E->aout.as<AlignOutDev>(), E->cigpool.as<uint32_t>(), (uint32_t)pool_cap,
		E->cigcount.as<uint32_t>(), E->st); }
	if (he != hipSuccess) return fail(E, FASIM_E_HIP, "finish launch failed: %s", hipGetErrorString(he));
	std::vector<AlignOutDev> ao(n);
	uint32_t pool_used = 0;
	HIPOK(hipMemcpyAsync(ao.data(), E->aout.p, sizeof(AlignOutDev) * n, hipMemcpyDeviceToHost, E->st));
	HIPOK(hipMemcpyAsync(&pool_used, E->cigcount.p, sizeof pool_used, hipMemcpyDeviceToHost, E->st));
	HIPOK(hipStreamSynchronize(E->st));
	{
		// alignments whose band / direction matrix did not fit the LDS kernel (2.3 % on the bench): the same kernel with four times
		// the LDS per alignment (64 KB per workgroup) first; only what does not fit that either goes to global scratch below
		std::vector<int32_t> mid;
		for (int k = 0; k < n; k++) if (ao[k].status == 2) mid.push_back(k);
		if (!mid.empty()) {
			if (g_prof.on) g_prof.add(30, "finish: alignments sent to the 64 KB LDS pass (count)", 1e-6 * mid.size());
			rc = upload(E, E->unit_ids, mid.data(), sizeof(int32_t) * mid.size()); if (rc) return rc;
			HIPOK(E->scratch2.ensure((size_t)((mid.size() + 63) / 64) * 64 * 16384));
			{ TimedScope ts(E, 3);
			he = launch_finish_mid(tcv(E), E->q2.as<uint8_t>(), E->fprobs.as<FwdProb>(), E->fout.as<FwdOut>(), E->unit_ids.as<int32_t>(), (int)mid.size(),
				E->scratch2.as<uint8_t>(), E->aout.as<AlignOutDev>(), E->cigpool.as<uint32_t>(), (uint32_t)pool_cap, E->cigcount.as<uint32_t>(), E->st); }
			if (he != hipSuccess) return fail(E, FASIM_E_HIP, "finish (64 KB LDS) launch failed: %s", hipGetErrorString(he));
			HIPOK(hipMemcpyAsync(ao.data(), E->aout.p, sizeof(AlignOutDev) * n, hipMemcpyDeviceToHost, E->st));
			HIPOK(hipMemcpyAsync(&pool_used, E->cigcount.p, sizeof pool_used, hipMemcpyDeviceToHost, E->st));
			HIPOK(hipStreamSynchronize(E->st));
		}
	}
	{
		// then the same algorithm on global scratch, first with 16 KB per alignment, then (wide bands after several doublings:
		// gapped alignments) with 1 MB
		const int caps[2] = { scratch_cap, 1 << 20 };
		for (int pass = 0; pass < 2; pass++) {
			std::vector<int32_t> big;
			for (int k = 0; k < n; k++) if (ao[k].status == 2) big.push_back(k);
			if (big.empty()) break;
			if (g_prof.on) g_prof.add(17 + pass, pass ? "finish: alignments sent to the 1 MB pass (count)" : "finish: alignments sent to the 16 KB pass (count)", 1e-6 * big.size());
			rc = upload(E, E->unit_ids, big.data(), sizeof(int32_t) * big.size()); if (rc) return rc;
			// at most 2 GiB of scratch per launch (the 1 MB pass of a pathological batch is cut into slices)
			const size_t per_launch = std::max<size_t>(64, ((size_t)2 << 30) / (size_t)caps[pass]);
			HIPOK(E->scratch2.ensure(std::min(big.size(), per_launch) * (size_t)caps[pass]));
			for (size_t b0 = 0; b0 < big.size(); b0 += per_launch) {
				const size_t cnt = std::min(per_launch, big.size() - b0);
				{ TimedScope ts(E, 6);
				he = launch_finish_big(tcv(E), E->q2.as<uint8_t>(), E->fprobs.as<FwdProb>(), E->fout.as<FwdOut>(),
					E->unit_ids.as<int32_t>() + b0, (int)cnt, E->scratch2.as<uint8_t>(), caps[pass], E->aout.as<AlignOutDev>(),
					E->cigpool.as<uint32_t>(), (uint32_t)pool_cap, E->cigcount.as<uint32_t>(), E->st); }
				if (he != hipSuccess) break;
			}
			if (he != hipSuccess) return fail(E, FASIM_E_HIP, "finish (global scratch) launch failed: %s", hipGetErrorString(he));
			HIPOK(hipMemcpyAsync(ao.data(), E->aout.p, sizeof(AlignOutDev) * n, hipMemcpyDeviceToHost, E->st));
			HIPOK(hipMemcpyAsync(&pool_used, E->cigcount.p, sizeof pool_used, hipMemcpyDeviceToHost, E->st));
			HIPOK(hipStreamSynchronize(E->st));
		}
	}
	if (pool_used > pool_cap) pool_used = (uint32_t)pool_cap;
	const uint32_t pool_base = (uint32_t)cigars.size();
	cigars.resize((size_t)pool_base + pool_used);
	if (pool_used) {
		HIPOK(hipMemcpyAsync(cigars.data() + pool_base, E->cigpool.p, sizeof(uint32_t) * pool_used, hipMemcpyDeviceToHost, E->st));
		HIPOK(hipStreamSynchronize(E->st));
	}
	for (int k = 0; k < n; k++) {
		const AlignOutDev& a = ao[k];
		if (g_prof.on && (a.status == 2 || a.status == 3)) {
			static std::atomic<int> shown(0);
			if (shown.fetch_add(1) < 12) fprintf(stderr, "[finish status %d] unit %d t0 %d score %d fwd(ref_end %d read_end %d flags %d) out(ref_begin %d q_begin %d) win_len %d\n",
				(int)a.status, W[k].unit, W[k].t0, fo[k].score, fo[k].ref_end, fo[k].read_end, fo[k].flags, a.ref_begin, a.query_begin, W[k].len);
		}
		if (g_prof.on && a.status != 0) {
			static const char* nm[5] = { "finish: device status 2 (count)", "finish: device status 4 (count)", "finish: device status 10 (count)", "finish: device status 11 (count)", "finish: device status 1/3 (count)" };
			const int si = a.status == 2 ? 0 : a.status == 4 ? 1 : a.status == 10 ? 2 : a.status == 11 ? 3 : 4;
			g_prof.add(19 + si, nm[si], 1e-6);
		}
		if (a.status == 2 || a.status == 4 || a.status == 10 || a.status == 11) { status[k] = 2; continue; }
		if (a.status == 1 || a.status == 3) { status[k] = 1; continue; }
		AlignResult& r = out[k];
		if (a.sw_score <= 0) { r.sw_score = 0; continue; }
		r.sw_score = a.sw_score; r.ref_begin = a.ref_begin; r.ref_end = a.ref_end; r.query_begin = a.query_begin; r.query_end = a.query_end;
		r.cigar_len = a.cigar_len; r.cigar_off = pool_base + a.cigar_off;
	}
	return FASIM_OK;
}

// ssw_align for a list of windows: systolic forward pass + finish kernel; everything that may hit the reference's
// layout-dependent behaviour is re-run by run_align()
int run_align_v2(fasim_engine* E, const UnitBatch& B, const std::vector<WindowProb>& W, std::vector<AlignResult>& out,
	std::vector<uint32_t>& cigars, fasim_scan_stats* stats)
{
	const int n = (int)W.size();
	if (!n) { out.clear(); return FASIM_OK; }
	if (!align_v2_fits(E, W)) return run_align(E, B, W, out, cigars, stats);
	std::vector<FwdOut> fo;
	int rc = run_fwd_both(E, B, W, fo, stats ? &stats->align_word_reruns : nullptr); if (rc) return rc;
	{
		std::vector<int> rv;
		for (int k = 0; k < n; k++) if (!fo[k].flags && fo[k].score >= 148 && fo[k].score < 255 - BIAS) rv.push_back(k);
		rc = run_rev_exact(E, B, W, fo, rv); if (rc) return rc;
	}
	std::vector<char> status;
	rc = run_finish(E, B, W, fo, out, cigars, status); if (rc) return rc;
	std::vector<int> redo;
	for (int k = 0; k < n; k++) { if (status[k] == 2) redo.push_back(k); else if (status[k] == 1) { out[k].sw_score = 0; out[k].failed = 1; } }
	if (!redo.empty()) {
		std::vector<WindowProb> W2(redo.size());
		for (size_t i = 0; i < redo.size(); i++) W2[i] = W[redo[i]];
		std::vector<AlignResult> r2;
		rc = run_align(E, B, W2, r2, cigars, nullptr); if (rc) return rc;
		for (size_t i = 0; i < redo.size(); i++) out[redo[i]] = r2[i];
		if (stats) stats->exact_replays += (int64_t)redo.size();
	}
	return FASIM_OK;
}


// Stage 3 (candidates, window tries, finish kernels, triplex records) for units [ua, ub) of a scanned batch.  Runs on any
// engine of the device that has the batch's lncRNA set: the target codes are read from the owner's resident buffer.
int stage3_range(fasim_engine* E, BatchCtx& C, int ua, int ub, fasim_scan_stats& st)
{
	int rc = FASIM_OK;
	const UnitBatch& B = C.B;
	const fasim_params& p = *C.p;
	const std::vector<int>& encs = *C.encs;
	const char* dna = C.dna;
	const int64_t step = C.step;
	const int nenc = C.nenc;
	const std::vector<int32_t>& slen = C.slen; const std::vector<int64_t>& sidx = C.sidx;
	const std::vector<int32_t>& hoff = C.hoff; const std::vector<int32_t>& hcnt = C.hcnt; const std::vector<uint32_t>& hits = C.hits;
	double t0;
	{

		// ---- candidates (a7) and the window tries (a8).  fastSIM() decides on sw_score and ref_end only
		//      (fastsim.h:218-235); both are known after the FORWARD pass (the reverse pass returns the same
		//      score: sswNew.cpp:1518 takes the minimum), so up to four forward rounds run first and the reverse
		//      pass + traceback (k_finish) run once, for the chosen try.  Candidates with a try that may hit
		//      the reference's layout-dependent behaviour, or whose traceback fails in the reference (NULL ->
		//      score 0 -> the loop would have continued), are replayed try by try on the stripe-faithful path.
		t0 = now_s();
		struct CandState { int unit; Cand c; AlignResult al, best; FwdOut fsel, fbest; int cut, bestcut; char done, flag, exact, ru_it; };
		std::vector<uint32_t> cigars;
		std::vector<CandState> cs;
		{
			ProfScope ps(7, "pick candidates");
			// contiguous unit ranges on the host threads, concatenated in unit order
			const int nt = std::max(1, std::min(E->host_threads, (ub - ua) / 256));
			std::vector<std::vector<CandState>> part(nt);
			auto work = [&](int ti) {
				CpuScope cpu(27, "CPU seconds: pick candidates");
				std::vector<Cand> tmp;
				const int u0 = ua + (int)((int64_t)(ub - ua) * ti / nt), u1 = ua + (int)((int64_t)(ub - ua) * (ti + 1) / nt);
				for (int u = u0; u < u1; u++) {
					pick_candidates(hits.data() + hoff[u], hcnt[u], tmp);
					for (const Cand& c : tmp) { CandState x; memset(&x.fsel, 0, sizeof x.fsel); memset(&x.fbest, 0, sizeof x.fbest);
						x.unit = u; x.c = c; x.done = 0; x.cut = 0; x.bestcut = 0; x.flag = 0; x.exact = 0; x.ru_it = -1; part[ti].push_back(x); }
				}
			};
			if (nt == 1) work(0);
			else { std::vector<std::thread> th; for (int k = 0; k < nt; k++) th.emplace_back(work, k); for (auto& t : th) t.join(); }
			size_t total = 0;
			for (auto& v : part) total += v.size();
			cs.reserve(total);
			for (auto& v : part) cs.insert(cs.end(), v.begin(), v.end());
		}
		st.candidates += (int64_t)cs.size();
		bool v2 = true;
		{ std::vector<WindowProb> probe(1, WindowProb{ 0, 0, 1 }); v2 = align_v2_fits(E, probe); }
		// lane maxima left by the reverse passes (start-based bounds of a candidate's tries): [candidate][4 zones][lanes]
		const bool zb = v2 && band_mode(E) == 1 && band_ready(E, B) && !cs.empty() &&
			E->lane_ub.ensure((size_t)cs.size() * 4 * 128 * systolic_tiles(E->m) * sizeof(uint16_t)) == hipSuccess;
		if (v2 && !zb) { (void)hipGetLastError(); E->lane_ub.release(); }
		if (v2) {
			for (int it = 0; it < 4; it++) {
				std::vector<WindowProb> W; std::vector<int> who;
				for (size_t k = 0; k < cs.size(); k++) {
					if (cs[k].done) continue;
					int cut;
					if (!window_for_try(it, cs[k].c.score, cs[k].c.pos, &cut)) { cs[k].done = 1; continue; }
					cs[k].cut = cut;
					W.push_back({ cs[k].unit, cs[k].c.pos - cut + 1, cut });
					who.push_back((int)k);
				}
				if (W.empty()) break;
				st.align_calls += (int64_t)W.size();
				// Band targets.  A candidate that has been through a reverse pass has start-based bounds for all of its tries (its
				// exact score is then known to the selection kernel); otherwise the first try aims at the candidate's own score (an
				// accepted try reaches it), and a later try goes straight to the reverse pass, or, without reverse passes
				// (band = 2), aims at 85 % of the previous try's score.
				const int rho = 85;
				std::vector<int32_t> target(W.size()), ru(zb ? W.size() : 0);
				FwdZones Z;
				if (zb) { Z.zones.resize(W.size()); Z.slot.resize(W.size()); }
				for (size_t i = 0; i < who.size(); i++) {
					const CandState& x = cs[who[i]];
					target[i] = it == 0 ? x.c.score : (zb ? 0 : std::max(1, x.fsel.score * rho / 100));
					if (zb) {
						ru[i] = x.ru_it >= 0 ? who[i] * 4 + (it - x.ru_it) : -1;
						if (ru[i] >= 0) target[i] = 1 << 30;
						uint32_t z = 0;
						for (int j = 1; j <= 3 && it + j < 4; j++) { int cut; if (window_for_try(it + j, x.c.score, x.c.pos, &cut) && cut <= 255) z |= (uint32_t)cut << (8 * (j - 1)); }
						Z.zones[i] = z; Z.slot[i] = who[i];
					}
				}
				std::vector<FwdOut> fo; std::vector<char> got_ru;
				rc = run_fwd_smart(E, B, W, target, zb ? &ru : nullptr, zb ? &Z : nullptr, fo, &got_ru, st); if (rc) return rc;
				if (zb) for (size_t i = 0; i < who.size(); i++) if (got_ru[i]) cs[who[i]].ru_it = (char)it;
				std::vector<int> fwd_score(fo.size());
				for (size_t i = 0; i < fo.size(); i++) fwd_score[i] = fo[i].score;
				{
					// score >= 148: the REVERSE pass (its own stripe geometry) could hit Q2 -> exact reverse pass now,
					// so that sw_score = min(forward, reverse) is known before the decision
					std::vector<int> rv;
					// ... but only for the tries whose alignment can still be chosen: a try that is accepted by its forward score,
					// one that would become the best try (alignment ends in the window's last column), or the last try.  Any
					// other try is neither accepted nor remembered, whatever its reverse pass returns.
					for (size_t i = 0; i < fo.size(); i++) {
						if (fo[i].flags || fo[i].score < 148 || fo[i].score >= 255 - BIAS) continue;
						const CandState& x = cs[who[i]];
						if (fo[i].score >= x.c.score || fo[i].ref_end == x.cut - 1 || it == 3) rv.push_back((int)i);
					}
					rc = run_rev_exact(E, B, W, fo, rv); if (rc) return rc;
					st.rev_exact += (int64_t)rv.size();
				}
				for (size_t i = 0; i < who.size(); i++) {
					CandState& x = cs[who[i]];
					const FwdOut& f = fo[i];
					// flag 1: the forward pass may hit Q2 (or the exact reverse pass was unusable); scores >= 251 go through
					// the reference's 16-bit kernels: both are replayed on the stripe-faithful path
					const bool can_be_chosen = f.score >= x.c.score || f.ref_end == x.cut - 1 || it == 3;
					if (!can_be_chosen && !(f.flags & 1)) { x.fsel = f; continue; }     // neither accepted nor remembered
					if ((f.flags & 1) || (!(f.flags & 6) && f.score >= 148)) {
						if (g_prof.on) { if (f.flags & 1) g_prof.add(8 + it, "exact: forward winner tainted / reverse unusable (count)", 1e-6); else g_prof.add(12, "exact: score >= 148 without exact reverse (count)", 1e-6); }
						x.exact = 1; x.done = 1; continue;
					}
					x.fsel = f;                                                                        // "last tried" so far
					if (f.score >= x.c.score) { x.flag = 1; x.done = 1; continue; }                    // fastsim.h:218-221
					if (f.score > x.fbest.score && f.ref_end == x.cut - 1) {                           // :222-235
						x.fbest = f; x.bestcut = x.cut; x.flag = 2;
						// The windows of the later tries are suffixes of this one (same last column, shorter), so their scores
						// cannot exceed this forward score: no later try is accepted (this one was not) and none can replace
						// the best one, which needs a strictly larger score.  The reference still runs them; their results are
						// never used.  (Not when the exact reverse pass lowered this try's score below its forward score.)
						if (f.score == fwd_score[i]) { x.done = 1; st.tries_skipped += 3 - it; }
					}
				}
			}
			// the chosen try of every candidate -> reverse pass + traceback
			std::vector<WindowProb> W; std::vector<FwdOut> fsel; std::vector<int> who;
			for (size_t k = 0; k < cs.size(); k++) {
				CandState& x = cs[k];
				if (x.exact) continue;
				if (x.flag == 2) { x.fsel = x.fbest; x.cut = x.bestcut; }                              // fastsim.h:238-250
				if (x.fsel.score <= 0) { x.al.sw_score = 0; continue; }
				W.push_back({ x.unit, x.c.pos - x.cut + 1, x.cut }); fsel.push_back(x.fsel); who.push_back((int)k);
			}
			std::vector<AlignResult> res; std::vector<char> status;
			rc = run_finish(E, B, W, fsel, res, cigars, status); if (rc) return rc;
			for (size_t i = 0; i < who.size(); i++) {
				CandState& x = cs[who[i]];
				if (status[i] != 0) { if (g_prof.on) { static const char* nm[6] = { "exact: finish status 0", "exact: finish status 1 (count)", "exact: finish status 2 (count)", "exact: finish status 3 (count)", "exact: finish status 4 (count)", "exact: finish status >= 5 (count)" }; const int si = std::min(5, (int)status[i]); g_prof.add(13 + si, nm[si], 1e-6); } x.exact = 1; continue; }
				x.al = res[i];
				st.cells_stage3 += (int64_t)(x.al.ref_end - x.al.ref_begin + 1) * (x.al.query_end - x.al.query_begin + 1);
			}
		}
		// stripe-faithful replay (all candidates when the systolic kernels do not fit the query)
		{
			std::vector<int> ex;
			for (size_t k = 0; k < cs.size(); k++) if (!v2 || cs[k].exact) { ex.push_back((int)k); cs[k].done = 0; cs[k].flag = 0; cs[k].best = AlignResult(); cs[k].al = AlignResult(); }
			if (v2) st.exact_replays += (int64_t)ex.size();
			if (v2) {
				// few candidates (of the order of 10^-5): all four window tries of each in ONE pass over the stripe-faithful
				// path (the windows depend only on the candidate's score and position, fastsim.h:209-211), then the
				// accept / best / last rule of fastsim.h:218-250 over the results in try order
				std::vector<WindowProb> W; std::vector<int> who, cuts;
				for (int k : ex) {
					for (int it = 0; it < 4; it++) {
						int cut;
						if (!window_for_try(it, cs[k].c.score, cs[k].c.pos, &cut)) break;
						W.push_back({ cs[k].unit, cs[k].c.pos - cut + 1, cut }); who.push_back(k); cuts.push_back(cut);
					}
				}
				if (!W.empty()) {
					std::vector<AlignResult> res;
					rc = run_align(E, B, W, res, cigars, nullptr); if (rc) return rc;
					for (size_t i = 0; i < who.size(); i++) {
						CandState& x = cs[who[i]];
						if (x.done) continue;
						x.al = res[i]; x.cut = cuts[i];
						if (x.al.sw_score >= x.c.score) { x.flag = 1; x.done = 1; continue; }
						if (x.al.sw_score > x.best.sw_score && x.al.ref_end == x.cut - 1) { x.best = x.al; x.bestcut = x.cut; x.flag = 2; }
					}
				}
			} else
			for (int it = 0; it < 4 && !ex.empty(); it++) {
				std::vector<WindowProb> W; std::vector<int> who;
				for (int k : ex) {
					if (cs[k].done) continue;
					int cut;
					if (!window_for_try(it, cs[k].c.score, cs[k].c.pos, &cut)) { cs[k].done = 1; continue; }
					cs[k].cut = cut;
					W.push_back({ cs[k].unit, cs[k].c.pos - cut + 1, cut });
					who.push_back(k);
				}
				if (W.empty()) break;
				st.align_calls += (int64_t)W.size(); for (const WindowProb& w : W) st.cells_stage3 += (int64_t)E->m * w.len;
				std::vector<AlignResult> res;
				rc = run_align(E, B, W, res, cigars, nullptr); if (rc) return rc;
				for (size_t i = 0; i < who.size(); i++) {
					CandState& x = cs[who[i]];
					x.al = res[i];
					if (x.al.sw_score >= x.c.score) { x.flag = 1; x.done = 1; continue; }
					if (x.al.sw_score > x.best.sw_score && x.al.ref_end == x.cut - 1) { x.best = x.al; x.bestcut = x.cut; x.flag = 2; }
				}
			}
			for (int k : ex) if (cs[k].flag == 2) { cs[k].al = cs[k].best; cs[k].cut = cs[k].bestcut; }
		}
		st.t_stage3_s += now_s() - t0;

		// ---- host: triplex records per unit (a12-a14), then LongTarget()'s tail filter (a15)
		t0 = now_s();
		std::vector<std::vector<HostTriplex>>& per_unit = C.per_unit;      // slots [ua, ub) belong to this call
		const std::vector<char>& seg_acgtn = C.seg_acgtn;
		{
			std::vector<size_t> first_((size_t)(ub - ua) + 1, 0);
			size_t* first = first_.data() - ua;                                // first[u] for u in [ua, ub]
			for (const CandState& x : cs) first[x.unit + 1]++;
			for (int u = ua; u < ub; u++) first[u + 1] += first[u];
			std::atomic<int> next(ua);
			auto work = [&]() {
				CpuScope cpu(26, "CPU seconds: records (convert_triplex, dedup)");
				std::vector<TriplexNum> mine, kept;
				std::vector<HostTriplex> one;
				for (;;) {
					const int u = next.fetch_add(1);
					if (u >= ub) break;
					if (first[u] == first[u + 1]) continue;
					const int s = u / nenc, enc = encs[u % nenc];
					const char* seg = dna + sidx[s] * step;
					const long dna_start = (long)(sidx[s] * step);
					mine.clear(); kept.clear();
					const bool acgtn = seg_acgtn[s] != 0;
					for (size_t k = first[u]; k < first[u + 1]; k++) {
						CandState& x = cs[k];
						AlignResult al = x.al; const int cut = x.cut;
						if (al.sw_score == 0) continue;                                                    // fastsim.h:253
						al.ref_begin += x.c.pos - cut + 1; al.ref_end += x.c.pos - cut + 1;                // :254-255
						const size_t before = mine.size();
						convert_triplex_num(al, cigars.data() + al.cigar_off, E->rna, seg, slen[s], enc, dna_start, p, mine, acgtn);
						if (mine.size() > before) mine.back().cand = (int)k;
					}
					dedup_top_num(mine, p, kept);
					for (const TriplexNum& tn : kept) {
						// LongTarget()'s tail filter (Fasim-LongTarget.cpp:589-597): what it drops is dropped here already
						if (!(tn.score >= p.scoreMin && tn.identity >= p.minIdentity && tn.tri_score >= p.minStability && tn.nt >= p.cLength)) continue;
						// a surviving record: the same conversion once more, this time with its TFO / TTS strings
						const CandState& x = cs[(size_t)tn.cand];
						AlignResult al = x.al;
						al.ref_begin += x.c.pos - x.cut + 1; al.ref_end += x.c.pos - x.cut + 1;
						one.clear();
						convert_triplex(al, cigars.data() + al.cigar_off, E->rna, seg, slen[s], enc, dna_start, p, one, acgtn, true);
						one[0].seg = (int)sidx[s]; one[0].enc = enc; one[0].cand = tn.cand;
						per_unit[u].push_back(std::move(one[0]));
					}
				}
			};
			int share = E->host_threads;
			if (E->active_workers) { const int act = std::max(1, E->active_workers->load()); share = std::max(share, std::min(32, E->host_threads_share_total / act)); }
			const int nt = std::max(1, std::min(share, ub - ua));
			if (nt == 1) work();
			else { std::vector<std::thread> th; for (int k = 0; k < nt; k++) th.emplace_back(work); for (auto& t : th) t.join(); }
		}
		st.t_host_s += now_s() - t0;
	}
	return rc;
}

