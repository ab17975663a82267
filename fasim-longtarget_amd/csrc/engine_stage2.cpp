// fasim-longtarget_amd/csrc/engine_stage2.cpp -- stages 1 + 2 of the host engine: the fused systolic scan (k_scan), the stripe-faithful
// re-runs of hazard units (whole units or parallel column chunks), raw-target helpers of the single-call entry points.
#include "engine.h"

int run_striped(fasim_engine* E, StripedMode mode, bool word, const std::vector<StripedProb>& probs, bool stage1,
	const uint8_t* tcodes, int max_qlen)
{
	if (probs.empty()) return FASIM_OK;
	int rc = upload(E, E->probs, probs.data(), probs.size() * sizeof(StripedProb));
	if (rc) return rc;
	StripedLaunch L;
	L.tcodes = tcodes; L.qcodes = stage1 ? E->q1.as<uint8_t>() : E->q2.as<uint8_t>();
	L.probs = E->probs.as<StripedProb>(); L.nprob = (int)probs.size(); L.counter = E->counter.as<uint32_t>();
	L.lut = stage1 ? E->lut1 : E->lut2; L.max_qlen = max_qlen;
	L.colmax = E->colmax.as<uint8_t>(); L.max_out = E->max_out.as<int32_t>(); L.ends = E->ends.as<AlignEnds>();
	hipError_t he;
	{
		TimedScope ts(E, mode == MODE_ALIGN || mode == MODE_REV ? 5 : 1);
		he = launch_striped(mode, word, !stage1, L, E->st);
	}
	if (he == hipErrorInvalidValue) return fail(E, FASIM_E_UNSUPPORTED, "query of %d nt does not fit the LDS-resident striped kernel", max_qlen);
	if (he != hipSuccess) return fail(E, FASIM_E_HIP, "striped kernel launch failed: %s", hipGetErrorString(he));
	return FASIM_OK;
}

// stage 1 (a4): exact max per unit.  8-bit first, 16-bit re-run where the byte kernel overflowed.
int run_stage1(fasim_engine* E, const UnitBatch& B, std::vector<int>& score, int64_t* word_reruns)
{
	score.assign(B.nunit, 0);
	if (!B.nunit) return FASIM_OK;
	HIPOK(E->max_out.ensure(sizeof(int32_t) * B.nunit));
	int rc = run_striped(E, MODE_MAX1, false, whole_unit_probs(B, E->m, nullptr), true, E->tcodes.as<uint8_t>(), E->m);
	if (rc) return rc;
	HIPOK(hipMemcpyAsync(score.data(), E->max_out.p, sizeof(int32_t) * B.nunit, hipMemcpyDeviceToHost, E->st));
	HIPOK(hipStreamSynchronize(E->st));
	std::vector<int> redo;
	for (int u = 0; u < B.nunit; u++) if (score[u] >= 255) redo.push_back(u);
	if (!redo.empty()) {
		rc = run_striped(E, MODE_MAX1, true, whole_unit_probs(B, E->m, &redo), true, E->tcodes.as<uint8_t>(), E->m);
		if (rc) return rc;
		std::vector<int> all(B.nunit);
		HIPOK(hipMemcpyAsync(all.data(), E->max_out.p, sizeof(int32_t) * B.nunit, hipMemcpyDeviceToHost, E->st));
		HIPOK(hipStreamSynchronize(E->st));
		for (int u : redo) {
			score[u] = all[u];
			if (score[u] >= 32767) return fail(E, FASIM_E_OVERFLOW, "stage-1 score of unit %d left the 16-bit range", u);
		}
		if (word_reruns) *word_reruns += (int64_t)redo.size();
	}
	return FASIM_OK;
}

// stage 2 (a5/a6): column maxima into E->colmax
int run_stage2(fasim_engine* E, const UnitBatch& B)
{
	if (!B.nunit) return FASIM_OK;
	HIPOK(E->colmax.ensure((size_t)B.nunit * B.tstride));
	HIPOK(E->max_out.ensure(sizeof(int32_t) * B.nunit));
	return run_striped(E, MODE_PRE, false, whole_unit_probs(B, E->m, nullptr), false, E->tcodes.as<uint8_t>(), E->m);
}


void fill_scores(int8_t* sc, bool stage1)
{
	for (int t = 0; t < 5; t++) for (int q = 0; q < 5; q++)
		sc[t * 5 + q] = (int8_t)(stage1 ? ((t == 4 || q == 4) ? -1 : (t == q ? 5 : -4)) : ((t == q && t < 4) ? 5 : -4));
}


// Stripe-faithful re-run of the hazard units (Q2), cut into column chunks that run in PARALLEL (kernels.hip, "chunked hazard
// re-run"; scan.hip, DUMP variant).
//   * Columns before the first step at which k_scan saw a possible taint are exact already: the re-run starts there.
//   * The rest is cut into up to HAZARD_MAX_CHUNKS chunks of about equal cost (k_hazard_plan).  A quick second k_scan pass
//     over the hazard units only (the checkpoint pass) leaves the reference's DP state -- H, and the E of the reference's
//     own recurrence -- at every chunk boundary, as it is when no Q2 deviation is alive; every chunk starts from its
//     checkpoint at once, in one launch.
//   * A chunk's result is the reference's if its start state is.  Chunk 0 starts where nothing has deviated yet.  A group
//     that ends chunk j in exactly the next checkpoint stops: chunk j + 1's own group started from that very state (equal
//     states have equal futures).  Otherwise a deviation is alive, and the group keeps going through chunk j + 1, j + 2, ...
//     until its state meets a checkpoint (deviations live for the length of one alignment) or the unit ends.  Every group
//     writes into a row of its own, so the speculative run of a chunk and the run that came through from the left never
//     touch the same bytes; the host then picks, chunk by chunk from the left, the row of the group that was exact.
//   * The overflow rule (Q1: everything from the first column >= 251 on is zero) needs no history: the first exact group
//     that reports an overflow column ends the unit.
// Latency of the re-run: the checkpoint pass plus one chunk (plus the length of the longest living deviation) instead of
// 5 000 sequential columns of the 16-lane emulation.
int run_hazard_chunked(fasim_engine* E, const UnitBatch& B, const std::vector<int>& hz, const ScanLaunch& Lmain)
{
	constexpr int KC = HAZARD_MAX_CHUNKS;
	const int nh = (int)hz.size();
	const int rows_total = 16 * ((E->m + 15) / 16);
	const bool dbg = getenv("FASIM_DEBUG_HAZARD") != nullptr;
	const int env_target = 200;
	const int target = E->hz_target > 0 ? E->hz_target : env_target;
	const int hot_thr = 144, env_hot_w = 2;
	const int hot_w = E->hz_hot_w > 0 ? E->hz_hot_w : env_hot_w;
	const bool spread = false;
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
	const auto t_begin = now();

	// 1. plan (device), read back for the problem list
	std::vector<int32_t> ids(hz.begin(), hz.end());
	int rc = upload(E, E->unit_ids, ids.data(), sizeof(int32_t) * nh); if (rc) return rc;
	HIPOK(E->hz_plan.ensure(sizeof(int32_t) * (size_t)nh * (KC + 1)));
	hipError_t he = launch_hazard_plan(E->unit_ids.as<int32_t>(), nh, E->unit_len.as<int32_t>(), E->unit_first.as<int32_t>(), E->colmax16.as<uint16_t>(),
		B.tstride, target, hot_thr, hot_w, E->hz_plan.as<int32_t>(), E->st);
	if (he != hipSuccess) return fail(E, FASIM_E_HIP, "hazard plan launch failed: %s", hipGetErrorString(he));
	std::vector<int32_t> plan((size_t)nh * (KC + 1));
	HIPOK(hipMemcpyAsync(plan.data(), E->hz_plan.p, sizeof(int32_t) * plan.size(), hipMemcpyDeviceToHost, E->st));
	HIPOK(hipStreamSynchronize(E->st));

	std::vector<StripedProb> probs;
	std::vector<int32_t> nchunk((size_t)nh, 0), base((size_t)nh, 0);
	for (int k = 0; k < nh; k++) {
		const int32_t* c = &plan[(size_t)k * (KC + 1)];
		const int n = B.unit_len[(size_t)hz[(size_t)k]];
		int K = 0;
		while (K < KC && c[K + 1] >= 0) K++;
		// the shape the kernels index by: strictly increasing boundaries inside the unit, ending at its last column
		bool ok = K >= 1 && c[0] >= 0 && c[K] == n;
		for (int j = 0; j < K && ok; j++) ok = c[j] < c[j + 1];
		if (!ok) return fail(E, FASIM_E_HIP, "hazard re-run: malformed chunk plan for unit %d", hz[(size_t)k]);
		nchunk[(size_t)k] = K; base[(size_t)k] = (int32_t)probs.size();
		for (int j = 0; j < K; j++) {
			StripedProb q;
			q.tbase = (int64_t)hz[(size_t)k] * B.tstride; q.t0 = c[j]; q.ref_len = c[j + 1] - c[j]; q.q_len = E->m; q.unit = k; q.aux = j; q.pad = 0;
			probs.push_back(q);
		}
	}
	const int np = (int)probs.size();
	rc = upload(E, E->hz_base, base.data(), sizeof(int32_t) * nh); if (rc) return rc;

	// 2. checkpoint pass: the state chunk j starts from = the state after column c[j] - 1.  One work item per window of
	// SCAN_SNAP_STEPS columns that holds such a column: it continues from the pipeline snapshot the main pass left at the
	// window's first step (columns less than 64 past a snapshot belong to the window before: see scan.hip)
	std::vector<int32_t> cols((size_t)np, -1);
	std::vector<ScanDumpItem> items;
	const bool windows = E->snap_units >= B.nunit && E->snap_per_unit > 0;
	for (int k = 0; k < nh; k++) {
		const int32_t* c = &plan[(size_t)k * (KC + 1)];
		int cur = -1;
		for (int j = 0; j < nchunk[(size_t)k]; j++) {
			const int X = c[j] - 1;
			if (X < 0) continue;                                   // (chunk 0 of a unit that starts at column 0: the zero state)
			cols[(size_t)base[(size_t)k] + j] = X;
			int win = (!windows || X < SCAN_SNAP_STEPS + 64) ? 0 : (X - 64) / SCAN_SNAP_STEPS;
			if (win > E->snap_per_unit) win = E->snap_per_unit;
			if (win != cur) { items.push_back({ hz[(size_t)k], win * SCAN_SNAP_STEPS, base[(size_t)k] + j, 0 }); cur = win; }
			items.back().count++;
		}
	}
	HIPOK(E->hz_state.ensure((size_t)np * 2 * rows_total * sizeof(uint16_t)));
	if (!items.empty()) {
		rc = upload(E, E->hz_cols, cols.data(), sizeof(int32_t) * np); if (rc) return rc;
		rc = upload(E, E->hz_items, items.data(), sizeof(ScanDumpItem) * items.size()); if (rc) return rc;
		ScanLaunch L = Lmain;
		L.unit_ids = nullptr; L.nwork = (int)items.size(); L.unit_hz = nullptr; L.unit_first = nullptr;
		L.snap = windows ? E->snap.as<uint32_t>() : nullptr; L.snap_per_unit = windows ? E->snap_per_unit : 0;
		L.dump_items = E->hz_items.as<ScanDumpItem>(); L.dump_cols = E->hz_cols.as<int32_t>(); L.dump_state = E->hz_state.as<uint16_t>();
		{ TimedScope ts(E, 1); he = launch_scan(L, E->st); }
		if (he != hipSuccess) return fail(E, FASIM_E_HIP, "scan (checkpoint pass) launch failed: %s", hipGetErrorString(he));
		if (dbg && atoi(getenv("FASIM_DEBUG_HAZARD")) >= 2 && windows) {
			// self-check of the windowed pass: the same checkpoints from a pass that starts every unit at column 0
			const size_t bytes = (size_t)np * 2 * rows_total * sizeof(uint16_t);
			std::vector<uint16_t> got(bytes / 2), want(bytes / 2);
			HIPOK(hipMemcpyAsync(got.data(), E->hz_state.p, bytes, hipMemcpyDeviceToHost, E->st));
			HIPOK(hipStreamSynchronize(E->st));
			std::vector<ScanDumpItem> whole;
			for (const ScanDumpItem& it : items) { if (!whole.empty() && whole.back().unit == it.unit) whole.back().count += it.count; else whole.push_back({ it.unit, 0, it.first, it.count }); }
			rc = upload(E, E->hz_items, whole.data(), sizeof(ScanDumpItem) * whole.size()); if (rc) return rc;
			L.nwork = (int)whole.size(); L.snap = nullptr; L.snap_per_unit = 0;
			he = launch_scan(L, E->st);
			if (he != hipSuccess) return fail(E, FASIM_E_HIP, "scan (checkpoint self-check) launch failed: %s", hipGetErrorString(he));
			HIPOK(hipMemcpyAsync(want.data(), E->hz_state.p, bytes, hipMemcpyDeviceToHost, E->st));
			HIPOK(hipStreamSynchronize(E->st));
			size_t bad = 0, checked = 0;
			for (int x = 0; x < np; x++) {
				if (cols[(size_t)x] < 0) continue;
				for (int r = 0; r < 2 * rows_total; r++) { checked++; if ((got[(size_t)x * 2 * rows_total + r] >> 1) != (want[(size_t)x * 2 * rows_total + r] >> 1)) bad++; }
			}
			fprintf(stderr, "[hazard] checkpoint self-check: %zu of %zu values differ between the windowed and the whole pass (%zu items vs %zu)\n", bad, checked, items.size(), whole.size());
			if (bad) return fail(E, FASIM_E_HIP, "hazard re-run: windowed checkpoint pass disagrees with the whole pass");
		}
	}
	if (dbg) { HIPOK(hipStreamSynchronize(E->st)); fprintf(stderr, "[hazard] %d units, %d chunks; plan + checkpoint pass done at %.2f ms\n", nh, np, ms_since(t_begin)); }

	// 3. all chunks in one launch
	HIPOK(E->hz_rows.ensure((size_t)np * B.tstride));
	HIPOK(E->hz_chunk.ensure(sizeof(int32_t) * 4 * np));
	rc = upload(E, E->probs, probs.data(), probs.size() * sizeof(StripedProb)); if (rc) return rc;
	StripedLaunch SL;
	SL.tcodes = E->tcodes.as<uint8_t>(); SL.qcodes = E->q2.as<uint8_t>(); SL.probs = E->probs.as<StripedProb>(); SL.nprob = np;
	SL.counter = E->counter.as<uint32_t>(); SL.lut = E->lut2; SL.max_qlen = E->m; SL.colmax = nullptr; SL.max_out = nullptr; SL.ends = nullptr;
	SL.state = E->hz_state.as<uint16_t>(); SL.state_rows = rows_total; SL.chunk_cols = E->hz_plan.as<int32_t>(); SL.chunk_base = E->hz_base.as<int32_t>();
	SL.chunk_rows = E->hz_rows.as<uint8_t>(); SL.row_stride = B.tstride; SL.chunk_out = E->hz_chunk.as<int32_t>(); SL.spread = spread;
	{ TimedScope ts(E, 1); he = launch_striped(MODE_PRE, false, true, SL, E->st); }
	if (he == hipErrorInvalidValue) return fail(E, FASIM_E_UNSUPPORTED, "query of %d nt does not fit the LDS-resident striped kernel", E->m);
	if (he != hipSuccess) return fail(E, FASIM_E_HIP, "striped kernel launch failed: %s", hipGetErrorString(he));
	std::vector<int32_t> co((size_t)4 * np);
	HIPOK(hipMemcpyAsync(co.data(), E->hz_chunk.p, sizeof(int32_t) * 4 * np, hipMemcpyDeviceToHost, E->st));
	HIPOK(hipStreamSynchronize(E->st));
	if (dbg) {
		std::vector<int> order((size_t)np); for (int x = 0; x < np; x++) order[(size_t)x] = x;
		std::sort(order.begin(), order.end(), [&](int a, int b) { return co[(size_t)4 * a + 2] > co[(size_t)4 * b + 2]; });
		long sum = 0; int went_on = 0;
		for (int x = 0; x < np; x++) { sum += co[(size_t)4 * x + 2]; if (co[(size_t)4 * x] != probs[(size_t)x].aux) went_on++; }
		fprintf(stderr, "[hazard] chunk launch done at %.2f ms; mean problem time %.3f ms; %d of %d groups went on past their chunk\n", ms_since(t_begin), sum / (double)np / 1e5, went_on, np);
		for (int r = 0; r < std::min(np, 5); r++) {
			const int x = order[(size_t)r]; const StripedProb& q = probs[(size_t)x];
			fprintf(stderr, "[hazard]   slow: unit %d chunk %d..%d of %d, first chunk %d cols: %.3f ms\n", hz[(size_t)q.unit], q.aux, co[(size_t)4 * x], nchunk[(size_t)q.unit], q.ref_len, co[(size_t)4 * x + 2] / 1e5);
		}
	}

	// 4. from the left: the group that started chunk j exact covers chunks j .. last(j); the next exact group starts at last(j) + 1
	std::vector<int32_t> src((size_t)nh * KC, 0), zero_from((size_t)nh, -1);
	{
		int x0 = 0;
		for (int k = 0; k < nh; k++) {
			const int K = nchunk[(size_t)k];
			int j = 0;
			while (j < K) {
				const int last = co[(size_t)4 * (x0 + j)], ovf = co[(size_t)4 * (x0 + j) + 1];
				if (last < j || last >= K) return fail(E, FASIM_E_HIP, "hazard re-run: inconsistent chunk report for unit %d", hz[(size_t)k]);
				for (int r = j; r <= last; r++) src[(size_t)k * KC + r] = j;
				if (ovf >= 0) { zero_from[(size_t)k] = ovf; for (int r = last + 1; r < K; r++) src[(size_t)k * KC + r] = j; break; }
				j = last + 1;
			}
			x0 += K;
		}
	}
	HIPOK(E->hz_src.ensure(sizeof(int32_t) * (size_t)nh * KC));
	rc = upload(E, E->hz_src, src.data(), sizeof(int32_t) * src.size()); if (rc) return rc;
	rc = upload(E, E->hz_zero, zero_from.data(), sizeof(int32_t) * nh); if (rc) return rc;
	he = launch_hazard_merge(E->colmax16.as<uint16_t>(), E->colmax.as<uint8_t>(), E->unit_ids.as<int32_t>(), nh, E->unit_len.as<int32_t>(),
		E->hz_plan.as<int32_t>(), E->hz_base.as<int32_t>(), E->hz_src.as<int32_t>(), E->hz_zero.as<int32_t>(), E->hz_rows.as<uint8_t>(), B.tstride, B.tstride, E->st);
	if (he != hipSuccess) return fail(E, FASIM_E_HIP, "hazard merge launch failed: %s", hipGetErrorString(he));
	HIPOK(hipStreamSynchronize(E->st));
	if (dbg) fprintf(stderr, "[hazard] merged at %.2f ms\n", ms_since(t_begin));
	return FASIM_OK;
}

// returns 1 when the query does not fit the kernel (caller falls back to the striped kernels)
int run_scan_v2(fasim_engine* E, const UnitBatch& B, const std::vector<char>& unit_needs_stage1, ScanOut& out,
	fasim_scan_stats* st)
{
	const int nu = B.nunit;
	HIPOK(E->colmax16.ensure((size_t)nu * B.tstride * sizeof(uint16_t)));
	std::vector<int32_t> ids(nu), sep;
	for (int u = 0; u < nu; u++) { ids[u] = u; if (unit_needs_stage1[u]) sep.push_back(u); }
	std::vector<int32_t> s1in(nu, -1);
	int rc = upload(E, E->stage1_in, s1in.data(), sizeof(int32_t) * nu); if (rc) return rc;
	ScanLaunch L;
	L.tcodes = E->tcodes.as<uint8_t>(); L.unit_len = E->unit_len.as<int32_t>(); L.tstride = B.tstride;
	L.counter = E->counter.as<uint32_t>(); L.m = E->m; L.colmax16 = E->colmax16.as<uint16_t>();
	L.boundary = nullptr; L.unit_hz = nullptr;
	L.coarse = 0;
	if (systolic_fits(E->m) && systolic_tiles(E->m) > 1) {
		HIPOK(E->boundary.ensure((size_t)nu * B.tstride * sizeof(uint2)));
		L.boundary = E->boundary.as<uint2>();
	}
	hipError_t he;
	if (!sep.empty()) {
		// units whose segment holds N (or every unit, when the query has letters outside ACGT): the stage-1
		// alphabet differs (Q4), so the exact stage-1 maximum needs its own pass
		rc = upload(E, E->unit_ids, sep.data(), sizeof(int32_t) * sep.size()); if (rc) return rc;
		L.unit_ids = E->unit_ids.as<int32_t>(); L.nwork = (int)sep.size(); L.qcodes = E->q1.as<uint8_t>(); fill_scores(L.score, true);
		{ TimedScope ts(E, 0, E->st); he = launch_scan(L, E->st); }
		if (he == hipErrorInvalidValue) return 1;
		if (he != hipSuccess) return fail(E, FASIM_E_HIP, "scan (stage-1 pass) launch failed: %s", hipGetErrorString(he));
		he = launch_max16(E->colmax16.as<uint16_t>(), E->unit_ids.as<int32_t>(), (int)sep.size(), E->unit_len.as<int32_t>(),
			B.tstride, E->stage1_in.as<int32_t>(), E->st);
		if (he != hipSuccess) return fail(E, FASIM_E_HIP, "max16 launch failed: %s", hipGetErrorString(he));
		HIPOK(hipStreamSynchronize(E->st));
		if (st) { st->stage1_word_reruns += (int64_t)sep.size(); for (int u : sep) st->cells_stage1 += (int64_t)E->m * B.unit_len[u]; }
	}
	rc = upload(E, E->unit_ids, ids.data(), sizeof(int32_t) * nu); if (rc) return rc;
	L.unit_ids = E->unit_ids.as<int32_t>(); L.nwork = nu; L.qcodes = E->q2.as<uint8_t>(); fill_scores(L.score, false);
	HIPOK(E->unit_hz.ensure(sizeof(int32_t) * nu));
	HIPOK(hipMemsetAsync(E->unit_hz.p, 0, sizeof(int32_t) * nu, E->st));
	L.unit_hz = E->unit_hz.as<int32_t>();
	HIPOK(E->unit_first.ensure(sizeof(int32_t) * nu));
	HIPOK(hipMemsetAsync(E->unit_first.p, 0x7f, sizeof(int32_t) * nu, E->st));      // 0x7f7f7f7f = "no taint arose"
	L.unit_first = E->unit_first.as<int32_t>();
	// pipeline snapshots for the chunked hazard re-run (single-tile queries): [unit][snapshot][dwords][64 lanes]
	E->snap_units = 0; E->snap_per_unit = 0;
	if (hazard_chunks_enabled(E) && systolic_fits(E->m) && systolic_tiles(E->m) == 1 && hazard_snapshots_enabled(E)) {
		const int spu = (B.tstride + 127) / SCAN_SNAP_STEPS;
		if (spu > 0) {
			// (an optimisation only: when the device has no room for it, the checkpoint pass starts at column 0)
			if (E->snap.ensure((size_t)nu * spu * systolic_snap_dwords(E->m) * 64 * sizeof(uint32_t)) == hipSuccess) {
				L.snap = E->snap.as<uint32_t>(); L.snap_per_unit = spu; E->snap_units = nu; E->snap_per_unit = spu;
			} else (void)hipGetLastError();
		}
	}
	// block maxima for the banded stage 3 (only the main pass, whose scoring is the stage-2/3 one, leaves them)
	E->ublk_units = 0; E->ublk_blocks = 0;
	if (band_mask(E)) {
		const int nb = scan_ublk_blocks(B.tstride);
		if (E->ublk.ensure((size_t)nu * systolic_tiles(E->m) * nb * 64 * sizeof(uint16_t)) == hipSuccess) {
			L.ublk = E->ublk.as<uint16_t>(); L.ublk_blocks = nb; E->ublk_units = nu; E->ublk_blocks = nb;
		} else (void)hipGetLastError();
	}
	{
		GateScope gate(E);
		{ TimedScope ts(E, 0, E->st); he = launch_scan(L, E->st); }
		if (he == hipErrorInvalidValue) return 1;
		if (he != hipSuccess) return fail(E, FASIM_E_HIP, "scan launch failed: %s", hipGetErrorString(he));
		HIPOK(hipStreamSynchronize(E->st));
	}

	HIPOK(E->hit_off.ensure(sizeof(int32_t) * nu)); HIPOK(E->hit_cnt.ensure(sizeof(int32_t) * nu));
	HIPOK(E->thr.ensure(sizeof(int32_t) * nu)); HIPOK(E->hits_total.ensure(64));
	HIPOK(E->stage1.ensure(sizeof(int32_t) * nu)); HIPOK(E->flags.ensure(sizeof(int32_t) * nu));
	out.stage1.resize(nu); out.thr.resize(nu); out.hit_off.resize(nu); out.hit_cnt.resize(nu); out.flags.resize(nu);
	size_t hits_cap = std::max<size_t>(E->hits.cap / 4, (size_t)nu * 128);
	for (;;) {
		HIPOK(E->hits.ensure(hits_cap * sizeof(uint32_t)));
		{ TimedScope ts(E, 4);
		he = launch_scan_post(E->colmax16.as<uint16_t>(), E->unit_ids.as<int32_t>(), nu, E->unit_len.as<int32_t>(), B.tstride,
			E->stage1_in.as<int32_t>(), E->hits.as<uint32_t>(), (uint32_t)hits_cap, E->hits_total.as<uint32_t>(),
			E->hit_off.as<int32_t>(), E->hit_cnt.as<int32_t>(), E->thr.as<int32_t>(), E->stage1.as<int32_t>(), E->flags.as<int32_t>(),
			E->unit_hz.as<int32_t>(), E->st); }
		if (he != hipSuccess) return fail(E, FASIM_E_HIP, "scan_post launch failed: %s", hipGetErrorString(he));
		uint32_t total = 0;
		HIPOK(hipMemcpyAsync(&total, E->hits_total.p, sizeof total, hipMemcpyDeviceToHost, E->st));
		HIPOK(hipStreamSynchronize(E->st));
		if (total <= hits_cap) { out.hits.resize(total); break; }
		hits_cap = (size_t)total + 1024;
	}
	HIPOK(hipMemcpyAsync(out.hit_off.data(), E->hit_off.p, sizeof(int32_t) * nu, hipMemcpyDeviceToHost, E->st));
	HIPOK(hipMemcpyAsync(out.hit_cnt.data(), E->hit_cnt.p, sizeof(int32_t) * nu, hipMemcpyDeviceToHost, E->st));
	HIPOK(hipMemcpyAsync(out.thr.data(), E->thr.p, sizeof(int32_t) * nu, hipMemcpyDeviceToHost, E->st));
	HIPOK(hipMemcpyAsync(out.stage1.data(), E->stage1.p, sizeof(int32_t) * nu, hipMemcpyDeviceToHost, E->st));
	HIPOK(hipMemcpyAsync(out.flags.data(), E->flags.p, sizeof(int32_t) * nu, hipMemcpyDeviceToHost, E->st));
	if (!out.hits.empty()) HIPOK(hipMemcpyAsync(out.hits.data(), E->hits.p, sizeof(uint32_t) * out.hits.size(), hipMemcpyDeviceToHost, E->st));
	HIPOK(hipStreamSynchronize(E->st));

	// hazard units: the signed lazy-F exit (Q2) may have fired in the reference -> stripe-faithful re-run
	std::vector<int> hz, sat;
	for (int u = 0; u < nu; u++) {
		if (out.flags[u] & 4) sat.push_back(u);
		if (out.flags[u] & 2) { if (st) st->stage2_overflow_units++; }
		if (out.flags[u] & 5) hz.push_back(u);
	}
	if (!sat.empty()) {
		// a score of 16383 or more saturated the doubled 16-bit lanes of k_scan: exact stage-1 score from the 16-bit
		// stripe-faithful kernel (as calc_score_once's word pass, stats.h:918), column maxima from the hazard path below
		HIPOK(E->max_out.ensure(sizeof(int32_t) * nu));
		rc = run_striped(E, MODE_MAX1, true, whole_unit_probs(B, E->m, &sat), true, E->tcodes.as<uint8_t>(), E->m); if (rc) return rc;
		std::vector<int32_t> all(nu);
		HIPOK(hipMemcpyAsync(all.data(), E->max_out.p, sizeof(int32_t) * nu, hipMemcpyDeviceToHost, E->st));
		HIPOK(hipStreamSynchronize(E->st));
		for (int u : sat) {
			if (all[u] >= 32767) return fail(E, FASIM_E_OVERFLOW, "stage-1 score of unit %d left the 16-bit range", u);
			out.stage1[u] = all[u]; out.thr[u] = (int32_t)((double)all[u] * 0.8);
		}
		rc = upload(E, E->stage1, out.stage1.data(), sizeof(int32_t) * nu); if (rc) return rc;
		if (st) st->stage1_word_reruns += (int64_t)sat.size();
	}
	if (!hz.empty()) {
		if (st) st->hazard_units += (int64_t)hz.size();
		HIPOK(E->colmax.ensure((size_t)nu * B.tstride));
		HIPOK(E->max_out.ensure(sizeof(int32_t) * nu));
		const bool chunked = hazard_chunks_enabled(E);
		// (a query of more than one tile of 128 virtual lanes, > 3 072 nt, keeps the whole-unit re-run: the checkpoint pass does not
		//  hand the restarted F chain from tile to tile)
		if (chunked && systolic_tiles(E->m) == 1) { rc = run_hazard_chunked(E, B, hz, L); if (rc) return rc; }
		else { rc = run_striped(E, MODE_PRE, false, whole_unit_probs(B, E->m, &hz), false, E->tcodes.as<uint8_t>(), E->m); if (rc) return rc; }
		std::vector<int32_t> hzids(hz.begin(), hz.end());
		rc = upload(E, E->unit_ids, hzids.data(), sizeof(int32_t) * hzids.size()); if (rc) return rc;
		std::vector<int32_t> off2(nu), cnt2(nu);
		std::vector<uint32_t> hits2;
		size_t cap2 = std::max<size_t>(E->hits2.cap / 4, hz.size() * 256);
		for (;;) {
			HIPOK(E->hits2.ensure(cap2 * sizeof(uint32_t)));
			{ TimedScope ts(E, 4);
			he = launch_hits(E->colmax.as<uint8_t>(), E->unit_ids.as<int32_t>(), E->unit_len.as<int32_t>(), E->stage1.as<int32_t>(),
				(int)hz.size(), B.tstride, E->hits2.as<uint32_t>(), (uint32_t)cap2, E->hits_total.as<uint32_t>(),
				E->hit_off.as<int32_t>(), E->hit_cnt.as<int32_t>(), E->thr.as<int32_t>(), E->st); }
			if (he != hipSuccess) return fail(E, FASIM_E_HIP, "hits launch failed: %s", hipGetErrorString(he));
			uint32_t total = 0;
			HIPOK(hipMemcpyAsync(&total, E->hits_total.p, sizeof total, hipMemcpyDeviceToHost, E->st));
			HIPOK(hipStreamSynchronize(E->st));
			if (total <= cap2) { hits2.resize(total); break; }
			cap2 = (size_t)total + 1024;
		}
		HIPOK(hipMemcpyAsync(off2.data(), E->hit_off.p, sizeof(int32_t) * nu, hipMemcpyDeviceToHost, E->st));
		HIPOK(hipMemcpyAsync(cnt2.data(), E->hit_cnt.p, sizeof(int32_t) * nu, hipMemcpyDeviceToHost, E->st));
		if (!hits2.empty()) HIPOK(hipMemcpyAsync(hits2.data(), E->hits2.p, sizeof(uint32_t) * hits2.size(), hipMemcpyDeviceToHost, E->st));
		HIPOK(hipStreamSynchronize(E->st));
		// splice: hazard units point into an appended region of the hit array
		const size_t base = out.hits.size();
		out.hits.insert(out.hits.end(), hits2.begin(), hits2.end());
		for (int u : hz) { out.hit_off[u] = (int32_t)(base + off2[u]); out.hit_cnt[u] = cnt2[u]; }
	}
	return FASIM_OK;
}


// raw targets (letters) -> a UnitBatch whose codes use the given alphabet
int load_raw_targets(fasim_engine* E, const char* targets, const int64_t* offsets, const int32_t* lens, int nprob,
	bool stage1, UnitBatch& B)
{
	int maxlen = 1;
	for (int i = 0; i < nprob; i++) { if (lens[i] <= 0) return fail(E, FASIM_E_ARG, "empty target %d", i); maxlen = std::max(maxlen, lens[i]); }
	B.nunit = nprob; B.tstride = (maxlen + 15) & ~15; B.unit_len.assign(lens, lens + nprob);
	std::vector<uint8_t> codes((size_t)nprob * B.tstride, CODE_N);
	for (int i = 0; i < nprob; i++)
		for (int c = 0; c < lens[i]; c++) codes[(size_t)i * B.tstride + c] = stage1 ? code1(targets[offsets[i] + c]) : code2(targets[offsets[i] + c]);
	return upload(E, E->tcodes, codes.data(), codes.size());
}

int need_query(fasim_engine* E)
{
	if (!E) return fail(nullptr, FASIM_E_ARG, "null engine");
	if (E->m <= 0) return fail(E, FASIM_E_ARG, "no query set: call fasim_set_query first");
	return FASIM_OK;
}

