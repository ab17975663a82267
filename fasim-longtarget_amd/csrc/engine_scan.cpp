// fasim-longtarget_amd/csrc/engine_scan.cpp -- batches and workers: scan_batch (segments -> units -> stages 1+2 of one batch), the
// forward sweep of -F, result packing, and scan_core (work queue of (lncRNA, batch) items over the worker engines).
#include "engine.h"
#include <memory>

// ---- row f3: forward sweep of classic SIM ---------------------------------------------------------------------
// addnode() (sim.h:99-148) over the events of one unit in row-major order: a known start point is updated (strictly larger
// score moves the end point; the bounding box grows), a new one is appended or, with K nodes present, overwrites the first
// node of lowest score whatever its own score is.
// Forward sweep + node list (k_sim_forward: the list is kept and replayed inside the wave, sim.hip) for units
// [first, first + nrun) of a resident code buffer.  min_scores[u] belongs to unit first + u.  lists[u] receives the node list.
// Units run in slices of <= 1024 (64 row segments of one unit's length each = 5 MB of scratch per unit); *ready (if given) is
// the number of leading units whose lists are complete, so that the host half can start on a slice while the next one runs.
// `lists` must have been sized to nrun by the caller.
int sim_forward_units(fasim_engine* E, const uint8_t* tcodes_dev, int tstride, const int32_t* unit_len_dev, const int32_t* unit_len_host,
	int first, int nrun, const int64_t* min_scores, std::atomic<int>* ready, std::vector<std::vector<fasim_sim_node>>& lists)
{
	static_assert(sizeof(SimNodeDev) == sizeof(fasim_sim_node) && SIM_K == FASIM_SIM_K, "node layout");
	if (nrun <= 0) return FASIM_OK;
	int maxlen = 1;
	for (int u = 0; u < nrun; u++) maxlen = std::max(maxlen, unit_len_host[first + u]);
	if (E->m > 65534 || maxlen > 65534) return fail(E, FASIM_E_UNSUPPORTED, "the SIM forward sweep holds start points in 16 bits: query %d / target %d nt is too long", E->m, maxlen);
	const int64_t row_stride = (maxlen + 2 + 15) & ~15;
	const uint32_t cap = (uint32_t)((maxlen + 15) & ~15);
	ProfScope fwd_wall(13, "-F forward sweep (k_sim_forward slices), wall");
	DevBuf& d_min = E->sim_min; DevBuf& d_row = E->sim_row; DevBuf& d_ev = E->sim_ev; DevBuf& d_cnt = E->sim_cnt; DevBuf& d_nodes = E->sim_nodes;
	int rc = upload(E, d_min, min_scores, sizeof(int64_t) * nrun); if (rc) return rc;
	// Units per launch = waves in flight.  The node-list replay is a serial chain per wave, so the chip wants all of its 8192 wave
	// slots filled (4 units per slice and CU left 3 of 4 SIMD slots idle: 1.7 s per 1024 units, the same for 4096); what limits the
	// slice is the event scratch (64 row segments of one unit's length = 5 MB per unit): a third of the free HBM at most.
	int per_slice = 1024;
	{
		size_t free_b = 0, total_b = 0;
		const size_t unit_b = (size_t)64 * cap * sizeof(SimEvent) + (size_t)2 * row_stride * sizeof(uint64_t);
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
			// (all the workers may be doing this at once: a fixed share of 96 GB each, whatever hipMemGetInfo says at the moment)
			const size_t have = std::min<size_t>(((size_t)96 << 30) / (size_t)std::max(1, E->scan_workers), E->sim_ev.cap + free_b / 3);
			per_slice = (int)std::max<size_t>(256, std::min<size_t>(8192, have / unit_b));
		}
		const int nslices = (nrun + per_slice - 1) / per_slice;
		per_slice = (nrun + nslices - 1) / nslices;           // even slices
	}
	std::vector<fasim_sim_node> hn((size_t)per_slice * FASIM_SIM_K);
	std::vector<int32_t> hc((size_t)per_slice);
	for (int u0 = 0; u0 < nrun; u0 += per_slice) {
		const int cnt = std::min(per_slice, nrun - u0);
		HIPOK(d_ev.ensure((size_t)cnt * 64 * cap * sizeof(SimEvent)));
		HIPOK(d_row.ensure((size_t)cnt * 2 * row_stride * sizeof(uint64_t)));
		HIPOK(d_cnt.ensure(sizeof(int32_t) * cnt));
		HIPOK(d_nodes.ensure(sizeof(SimNodeDev) * (size_t)cnt * SIM_K));
		SimFwdArgs a;
		a.tcodes = tcodes_dev + (size_t)(first + u0) * tstride; a.unit_len = unit_len_dev + first + u0; a.tstride = tstride;
		a.qcodes = E->qsim.as<uint8_t>(); a.m = E->m; a.min_score = d_min.as<int64_t>() + u0;
		a.rowbuf = d_row.as<uint64_t>(); a.row_stride = row_stride;
		a.events = d_ev.as<SimEvent>(); a.event_cap = cap; a.nodes = d_nodes.as<SimNodeDev>(); a.node_count = d_cnt.as<int32_t>();
		hipError_t he;
		{ TimedScope ts(E, 7); he = launch_sim_forward(a, cnt, E->st); }
		if (he != hipSuccess) return fail(E, FASIM_E_HIP, "sim_forward launch failed: %s", hipGetErrorString(he));
		HIPOK(hipMemcpyAsync(hc.data(), d_cnt.p, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, E->st));
		HIPOK(hipMemcpyAsync(hn.data(), d_nodes.p, sizeof(fasim_sim_node) * (size_t)cnt * FASIM_SIM_K, hipMemcpyDeviceToHost, E->st));
		HIPOK(hipStreamSynchronize(E->st));
		for (int k = 0; k < cnt; k++) {
			if (hc[(size_t)k] < 0 || hc[(size_t)k] > FASIM_SIM_K) return fail(E, FASIM_E_HIP, "sim_forward: bad node count");
			lists[(size_t)(u0 + k)].assign(hn.begin() + (size_t)k * FASIM_SIM_K, hn.begin() + (size_t)k * FASIM_SIM_K + hc[(size_t)k]);
		}
		if (ready) ready->store(u0 + cnt, std::memory_order_release);
	}
	return FASIM_OK;
}

// n independent pieces of host work on up to `nthreads` threads
template <typename F> static void parallel_units(int n, int nthreads, F&& fn)
{
	nthreads = std::max(1, std::min(nthreads, n));
	if (nthreads == 1) { for (int k = 0; k < n; k++) fn(k); return; }
	std::atomic<int> next(0);
	std::vector<std::thread> th;
	for (int t = 0; t < nthreads; t++) th.emplace_back([&]() { for (;;) { const int k = next.fetch_add(1); if (k >= n) break; fn(k); } });
	for (auto& t : th) t.join();
}

// ---- row f3: the K rounds of classic SIM after the forward sweep, in lock step over the units [first, first + cnt) of a resident
// code buffer.  Per round the host threads take each unit's best node, trace its alignment back and build the record
// (SimUnit::next_round, host_sim.cpp: ~0.6 % of the reference's time in this phase); the re-sweeps of the influenced rectangles
// (sim.h:884-1141, the other 99 %) are one launch of k_sim_resweep for all units.  The device keeps per unit the DP state per
// column / row and the used pairs; the node lists travel both ways each round (3.6 KB per unit).
// FASIM_SIM_RESWEEP=host keeps the re-sweeps on the host threads (A/B); =check runs both and compares every round.
int sim_resweep_rounds(fasim_engine* E, const uint8_t* tcodes_dev, int tstride, const int32_t* unit_len_dev, const int32_t* unit_len_host,
	int first, int cnt, SimUnit* const* units, int nthreads)
{
	if (cnt <= 0) return FASIM_OK;
	const char* mode_env = getenv("FASIM_SIM_RESWEEP");      // read per call: the tests switch it
	const int mode = !mode_env ? 0 : !strcmp(mode_env, "host") ? 1 : !strcmp(mode_env, "check") ? 2 : 0;
	if (mode == 1) {
		parallel_units(cnt, nthreads, [&](int k) { bool sweep; int box[4]; while (units[k]->next_round(&sweep, box, nullptr)) if (sweep) units[k]->resweep_host(); });
		return FASIM_OK;
	}
	int maxlen = 1;
	for (int u = 0; u < cnt; u++) maxlen = std::max(maxlen, unit_len_host[first + u]);
	const int M = E->m;
	const int64_t col_stride = (maxlen + 2 + 15) & ~15, row_stride = (M + 2 + 15) & ~15;
	const size_t used_per_unit = (size_t)SIM_K * (size_t)(M + 2) * sizeof(uint16_t);
	const size_t usedc_per_unit = (size_t)SIM_K * (size_t)col_stride * sizeof(uint16_t);
	const size_t per_unit = used_per_unit + usedc_per_unit + (size_t)(2 * col_stride + 2 * row_stride) * sizeof(uint64_t);
	// units advancing together: as many as a quarter of the free HBM holds state for (0.9 MB per unit for H19 x 5 kb) -- every
	// slice has its own tail of heavy units, so fewer slices are better
	size_t state_budget = (size_t)4 << 30;
	{
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
			state_budget = std::max(state_budget, std::min<size_t>(((size_t)96 << 30) / (size_t)std::max(1, E->scan_workers), (E->sim_used.cap + E->sim_usedc.cap + E->sim_col.cap + E->sim_rowst.cap + free_b) / 4));
	}
	const int slice = (int)std::max<size_t>(1, std::min<size_t>((size_t)cnt, state_budget / per_unit));
	std::vector<SimRoundReq> req((size_t)slice);
	std::vector<uint32_t> pairs;
	std::vector<int32_t> act;
	std::vector<std::vector<std::pair<int, int>>> np((size_t)slice);
	// node lists both ways every launch (3.6 KB per unit): pinned, or the copies are staged at a few GB/s
	const size_t hn_bytes = sizeof(fasim_sim_node) * (size_t)slice * FASIM_SIM_K;
	if (hn_bytes > E->pin_sim_cap) {
		if (E->pin_sim) { (void)hipHostFree(E->pin_sim); E->pin_sim = nullptr; E->pin_sim_cap = 0; }
		HIPOK(hipHostMalloc(&E->pin_sim, hn_bytes, hipHostMallocDefault));
		E->pin_sim_cap = hn_bytes;
	}
	fasim_sim_node* const hn = (fasim_sim_node*)E->pin_sim;
	std::vector<SimRoundReq> rq;
	std::vector<SimRoundOut> ho;
	std::vector<char> pend((size_t)slice);
	// A unit's share of a launch: a time slice (50 ms while more than 1 024 units are active, 60 ms for the tail; the kernel reads
	// the clock every eighth line), or a number of 64-cell steps when FASIM_SIM_BUDGET is set (tests).  A unit that needs more (a
	// re-sweep of most of the matrix) carries on in the next launch: the average round of H19 x 5 kb takes 10 k steps, the heaviest
	// unit of a 500 kb record 12 M steps in all.
	// ticks of the device's constant-rate clock (wall_clock64 in the kernel) per millisecond: 100 MHz on MI300 / MI355X
	int wall_khz = 0;
	if (hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, E->device) != hipSuccess || wall_khz <= 0) wall_khz = 100000;
	const char* budget_s = getenv("FASIM_SIM_BUDGET");       // (tests: a small budget exercises suspend / resume)
	const int budget_env = budget_s && atoi(budget_s) > 0 ? atoi(budget_s) : 0;
	static const bool debug = getenv("FASIM_SIM_DEBUG") != nullptr;
	long launches = 0, unit_launches = 0;
	int contributed = 0;                                     // this batch's share of E->sim_in_flight
	struct Leave { fasim_engine* E; int* c; ~Leave() { if (E->sim_in_flight && *c) E->sim_in_flight->fetch_sub(*c); } } leave{ E, &contributed };
	if (debug) { HIPOK(E->sim_debug.ensure(128)); HIPOK(hipMemsetAsync(E->sim_debug.p, 0, 128, E->st)); }
	ProfScope rounds_wall(31, "-F rounds (re-sweep launches + host half), wall");
	for (int s0 = 0; s0 < cnt; s0 += slice) {
		const int n = std::min(slice, cnt - s0);
		HIPOK(E->sim_used.ensure(used_per_unit * n)); HIPOK(E->sim_rounds.ensure(sizeof(int32_t) * n));
		HIPOK(E->sim_col.ensure((size_t)n * 2 * col_stride * sizeof(uint64_t))); HIPOK(E->sim_rowst.ensure((size_t)n * 2 * row_stride * sizeof(uint64_t)));
		HIPOK(E->sim_out.ensure(sizeof(SimRoundOut) * n)); HIPOK(E->sim_cnt.ensure(sizeof(int32_t) * n));
		HIPOK(E->sim_nodes_in.ensure(sizeof(SimNodeDev) * (size_t)n * SIM_K)); HIPOK(E->sim_nodes_out.ensure(sizeof(SimNodeDev) * (size_t)n * SIM_K));
		HIPOK(E->sim_nodes.ensure(sizeof(SimNodeDev) * (size_t)n * SIM_K)); HIPOK(E->sim_req.ensure(sizeof(SimRoundReq) * n));
		HIPOK(hipMemsetAsync(E->sim_used.p, 0, used_per_unit * n, E->st));
		HIPOK(E->sim_usedc.ensure(usedc_per_unit * n));
		HIPOK(hipMemsetAsync(E->sim_usedc.p, 0, usedc_per_unit * n, E->st));
		HIPOK(hipMemsetAsync(E->sim_rounds.p, 0, sizeof(int32_t) * n, E->st));
		HIPOK(E->sim_state.ensure(sizeof(SimSweepState) * n));
		std::fill(pend.begin(), pend.end(), 0);
		for (;;) {
			double t0 = now_s();
			parallel_units(n, nthreads, [&](int k) {
				if (pend[(size_t)k]) { req[(size_t)k].active = 2; return; }
				bool sweep = false; int box[4] = { 0, 0, 0, 0 };
				SimUnit& U = *units[s0 + k];
				SimRoundReq r; memset(&r, 0, sizeof r);
				if (U.next_round(&sweep, box, &np[(size_t)k]) && sweep) {
					r.active = 1; r.m1 = box[0]; r.mm = box[1]; r.n1 = box[2]; r.nn = box[3]; r.floor_score = (int32_t)U.floor_score;
				} else np[(size_t)k].clear();
				req[(size_t)k] = r;
			});
			g_prof.add(30, "-F rounds: host half (best node, traceback, record), wall", now_s() - t0);
			pairs.clear(); act.clear(); rq.clear();
			int active = 0;
			for (int k = 0; k < n; k++) {
				SimRoundReq& r = req[(size_t)k];
				if (!r.active) continue;
				r.unit = k;
				if (r.active == 1) {
					r.pairs_first = (int32_t)pairs.size(); r.pairs_count = (int32_t)np[(size_t)k].size();
					for (const auto& pr : np[(size_t)k]) pairs.push_back(((uint32_t)pr.first << 16) | (uint32_t)pr.second);
					const std::vector<fasim_sim_node>& nl = units[s0 + k]->nodes;
					r.node_count = (int32_t)nl.size();
					std::copy(nl.begin(), nl.end(), hn + (size_t)active * FASIM_SIM_K);       // slot = position among the active units
				}                                                                                 // (active == 2: its list waits on the device)
				active++; act.push_back(k); rq.push_back(r);
			}
			if (!active) { if (E->sim_in_flight && contributed) { E->sim_in_flight->fetch_sub(contributed); contributed = 0; } break; }
			launches++; unit_launches += active;
			// units in flight over all the workers' batches: the LDS variant holds a CU per unit, so it is for the moments when the
			// whole scan is down to its tail
			int everywhere = active;
			if (E->sim_in_flight) everywhere = E->sim_in_flight->fetch_add(active - contributed) + active - contributed;
			contributed = active;
			// per launch only the active units' requests and node lists cross PCIe (a finished or suspended unit costs nothing)
			int rc = upload_async(E, E->sim_req, rq.data(), sizeof(SimRoundReq) * active); if (rc) return rc;
			rc = upload_async(E, E->sim_pairs, pairs.data(), sizeof(uint32_t) * pairs.size()); if (rc) return rc;
			HIPOK(hipMemcpyAsync(E->sim_nodes_in.p, hn, sizeof(fasim_sim_node) * (size_t)active * FASIM_SIM_K, hipMemcpyHostToDevice, E->st));
			SimResweepArgs a;
			a.tcodes = tcodes_dev + (size_t)(first + s0) * tstride; a.unit_len = unit_len_dev + first + s0; a.tstride = tstride;
			a.qcodes = E->qsim.as<uint8_t>(); a.m = M;
			a.req = E->sim_req.as<SimRoundReq>(); a.pairs = E->sim_pairs.as<uint32_t>();
			a.nodes_in = E->sim_nodes_in.as<SimNodeDev>(); a.nodes_out = E->sim_nodes_out.as<SimNodeDev>(); a.out = E->sim_out.as<SimRoundOut>();
			a.used = E->sim_used.as<uint16_t>(); a.usedc = E->sim_usedc.as<uint16_t>(); a.used_cnt = E->sim_rounds.as<int32_t>();
			a.colS = E->sim_col.as<uint64_t>(); a.colG = a.colS + (size_t)n * col_stride;
			a.rowS = E->sim_rowst.as<uint64_t>(); a.rowG = a.rowS + (size_t)n * row_stride;
			a.col_stride = col_stride; a.row_stride = row_stride;
			a.nodes = E->sim_nodes.as<SimNodeDev>(); a.node_count = E->sim_cnt.as<int32_t>();
			a.state = E->sim_state.as<SimSweepState>(); a.budget = budget_env ? budget_env : (1 << 20); a.slice_ticks = budget_env ? ((int64_t)1 << 40) : (int64_t)(everywhere > 1024 ? 50 : 60) * wall_khz; a.debug = debug ? E->sim_debug.as<uint64_t>() : nullptr;
			const double tl0 = now_s();
			hipError_t he;
			{ TimedScope ts(E, 7); he = launch_sim_resweep(a, active, everywhere <= 256 /* one unit per CU: all at once */, E->st); }
			if (he != hipSuccess) return fail(E, FASIM_E_HIP, "sim_resweep launch failed: %s", hipGetErrorString(he));
			ho.resize((size_t)active);
			HIPOK(hipMemcpyAsync(ho.data(), E->sim_out.p, sizeof(SimRoundOut) * active, hipMemcpyDeviceToHost, E->st));
			HIPOK(hipMemcpyAsync(hn, E->sim_nodes_out.p, sizeof(fasim_sim_node) * (size_t)active * FASIM_SIM_K, hipMemcpyDeviceToHost, E->st));
			HIPOK(hipStreamSynchronize(E->st));
			int suspended = 0;
			for (int s = 0; s < active; s++) {
				if (ho[(size_t)s].node_count < 0 || ho[(size_t)s].node_count > FASIM_SIM_K) return fail(E, FASIM_E_HIP, "sim_resweep: bad node count");
				pend[(size_t)act[(size_t)s]] = ho[(size_t)s].pending != 0;
				suspended += ho[(size_t)s].pending != 0;
			}
			if (debug) fprintf(stderr, "[fasim sim] launch %ld: %d active, %d suspended, %.1f ms\n", launches, active, suspended, 1e3 * (now_s() - tl0));
			if (mode == 2) {
				std::atomic<int> bad(-1);
				parallel_units(active, nthreads, [&](int s) {
					if (ho[(size_t)s].pending) return;
					SimUnit& U = *units[s0 + act[(size_t)s]];
					const fasim_sim_node* dn = hn + (size_t)s * FASIM_SIM_K;
					U.resweep_host();
					if ((int)U.nodes.size() != ho[(size_t)s].node_count || U.floor_score != ho[(size_t)s].floor_score || memcmp(U.nodes.data(), dn, sizeof(fasim_sim_node) * U.nodes.size())) bad.store(act[(size_t)s]);
				});
				if (bad.load() >= 0) return fail(E, FASIM_E_HIP, "sim_resweep self-check: unit %d differs from the host re-sweep", first + s0 + bad.load());
				continue;
			}
			for (int s = 0; s < active; s++) {
				if (ho[(size_t)s].pending) continue;
				const fasim_sim_node* dn = hn + (size_t)s * FASIM_SIM_K;
				units[s0 + act[(size_t)s]]->nodes.assign(dn, dn + ho[(size_t)s].node_count);
				units[s0 + act[(size_t)s]]->floor_score = ho[(size_t)s].floor_score;
			}
		}
	}
	if (debug) {
		uint64_t d[8];
		HIPOK(hipMemcpy(d, E->sim_debug.p, sizeof d, hipMemcpyDeviceToHost));
		fprintf(stderr, "[fasim sim] %d units, %ld launches (%ld unit-launches); backward %llu steps %.3f s, forward %llu steps %.3f s of which %llu events %.3f s (wave-seconds); replay passes %llu, outranking events %llu\n",
			cnt, launches, unit_launches, (unsigned long long)d[0], 1e-8 * (double)d[1], (unsigned long long)d[2], 1e-8 * (double)d[3], (unsigned long long)d[4], 1e-8 * (double)d[5], (unsigned long long)d[6], (unsigned long long)d[7]);
	}
	return FASIM_OK;
}


int scan_batch(fasim_engine* E, const char* dna, int64_t dna_len, const uint8_t* dna_dev, int64_t shard_lo, int64_t b0, int64_t b1,
	const fasim_params& p, const std::vector<int>& encs, int tstride, BatchCtx& C, fasim_scan_stats& st)
{
	int rc = FASIM_OK;
	const int64_t step = p.cutLength - p.overlapLength;
	const int nenc = (int)encs.size();
	C.B = UnitBatch(); C.tstride = tstride; C.nenc = nenc; C.nseg = 0; C.step = step; C.dna = dna; C.p = &p; C.encs = &encs; C.stage3_done = false;
	C.per_unit.clear();
	{
		// segments of this batch that are not skipped by same_seq()
		std::vector<int32_t>& sstart = C.sstart; std::vector<int32_t>& slen = C.slen; std::vector<int64_t>& sidx = C.sidx;
		sstart.clear(); slen.clear(); sidx.clear();
		if (!dna_dev) {
			// Streaming ingest: the record is in host memory only.  The slice this batch needs goes through the worker's
			// pinned staging buffer and its own stream; with ~10 batches in flight the copy of one batch overlaps the kernels
			// of the others, and HBM holds 10 slices of ~2.5 MB instead of the whole record.
			const int64_t lo = b0 * step, hi = std::min<int64_t>(dna_len, (b1 - 1) * step + p.cutLength);
			const size_t bytes = (size_t)(hi - lo);
			if (bytes > E->pin_cap) {
				if (E->pin_dna) { (void)hipHostFree(E->pin_dna); E->pin_dna = nullptr; E->pin_cap = 0; }
				HIPOK(hipHostMalloc(&E->pin_dna, bytes + bytes / 8, hipHostMallocDefault));
				E->pin_cap = bytes + bytes / 8;
			}
			memcpy(E->pin_dna, dna + lo, bytes);
			HIPOK(E->dna.ensure(bytes));
			HIPOK(hipMemcpyAsync(E->dna.p, E->pin_dna, bytes, hipMemcpyHostToDevice, E->st));
			dna_dev = E->dna.as<uint8_t>(); shard_lo = lo;
		}
		for (int64_t s = b0; s < b1; s++) {
			const int64_t pos = s * step;
			const int len = (int)std::min<int64_t>(p.cutLength, dna_len - pos);
			st.segments++;
			if (same_seq(dna + pos, len)) { st.segments_skipped++; continue; }
			sstart.push_back((int32_t)(pos - shard_lo)); slen.push_back(len); sidx.push_back(s);
			st.logical_cells += (int64_t)E->m * len * nenc;
		}
		const int nseg = (int)sidx.size();
		if (!nseg) return FASIM_OK;
		C.nseg = nseg;
		UnitBatch& B = C.B; B.nunit = nseg * nenc; B.tstride = tstride; B.unit_len.resize(B.nunit);
		for (int s = 0; s < nseg; s++) for (int k = 0; k < nenc; k++) B.unit_len[s * nenc + k] = slen[s];
		st.units += B.nunit;
		// executed DP cells: the fused k_scan pass serves stage 1 AND stage 2, so it is counted once (as stage 2); stage 1 is
		// counted only where it really is a pass of its own (units with N / non-ACGT queries, the striped fallback)
		for (int s = 0; s < nseg; s++) st.cells_stage2 += (int64_t)E->m * slen[s] * nenc;
		rc = upload(E, E->seg_start, sstart.data(), sizeof(int32_t) * nseg); if (rc) return rc;
		rc = upload(E, E->seg_len, slen.data(), sizeof(int32_t) * nseg); if (rc) return rc;
		rc = upload(E, E->unit_len, B.unit_len.data(), sizeof(int32_t) * B.nunit); if (rc) return rc;
		HIPOK(E->tcodes.ensure((size_t)B.nunit * tstride));
		hipError_t he;
		{ TimedScope ts(E, 4);
		he = launch_encode(dna_dev, E->seg_start.as<int32_t>(), E->seg_len.as<int32_t>(), nseg,
			E->enc_ids.as<int32_t>(), nenc, E->enc_lut.as<uint8_t>(), E->tcodes.as<uint8_t>(), tstride, E->st); }
		if (he != hipSuccess) return fail(E, FASIM_E_HIP, "encode launch failed: %s", hipGetErrorString(he));

		// ---- stages 1+2: fused systolic scan (scan.hip); stripe-faithful kernels for hazard units, for
		//      queries beyond 3072 rows, or when FASIM_SCAN_V1=1
		double t0 = now_s();
		std::vector<int32_t>& hoff = C.hoff; std::vector<int32_t>& hcnt = C.hcnt; std::vector<int32_t>& thr = C.thr;
		std::vector<uint32_t>& hits = C.hits;
		hoff.clear(); hcnt.clear(); thr.clear(); hits.clear();
		bool done_v2 = false;
		if (!E->scan_v1) {
			std::vector<char> need1(B.nunit, E->query_acgt ? 0 : 1);
			if (E->query_acgt) {
				for (int s = 0; s < nseg; s++) {
					const char* sg = dna + sidx[s] * step; bool clean = true;
					for (int i = 0; i < slen[s]; i++) { const char c = sg[i]; if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T')) { clean = false; break; } }
					if (!clean) for (int k = 0; k < nenc; k++) need1[s * nenc + k] = 1;
				}
			}
			ScanOut so;
			rc = run_scan_v2(E, B, need1, so, &st);
			if (rc < 0) return rc;
			if (rc == 0) { hoff.swap(so.hit_off); hcnt.swap(so.hit_cnt); thr.swap(so.thr); hits.swap(so.hits); done_v2 = true; }
		}
		st.t_stage2_s += now_s() - t0;
		if (!done_v2) {
		// ---- stage 1
		t0 = now_s();
		std::vector<int> s1;
		for (int u = 0; u < B.nunit; u++) st.cells_stage1 += (int64_t)E->m * B.unit_len[u];
		rc = run_stage1(E, B, s1, &st.stage1_word_reruns); if (rc) return rc;
		st.t_stage1_s += now_s() - t0;

		// ---- stage 2 + hits
		t0 = now_s();
		rc = run_stage2(E, B); if (rc) return rc;
		rc = upload(E, E->stage1, s1.data(), sizeof(int32_t) * B.nunit); if (rc) return rc;
		HIPOK(E->hit_off.ensure(sizeof(int32_t) * B.nunit)); HIPOK(E->hit_cnt.ensure(sizeof(int32_t) * B.nunit));
		HIPOK(E->thr.ensure(sizeof(int32_t) * B.nunit)); HIPOK(E->hits_total.ensure(64));
		hoff.resize(B.nunit); hcnt.resize(B.nunit); thr.resize(B.nunit);
		std::vector<int32_t> pre_max(B.nunit);
		size_t hits_cap = std::max<size_t>(E->hits.cap / 4, (size_t)B.nunit * 128);
		for (;;) {
			HIPOK(E->hits.ensure(hits_cap * sizeof(uint32_t)));
			{ TimedScope ts(E, 4);
			he = launch_hits(E->colmax.as<uint8_t>(), nullptr, E->unit_len.as<int32_t>(), E->stage1.as<int32_t>(), B.nunit, tstride,
				E->hits.as<uint32_t>(), (uint32_t)hits_cap, E->hits_total.as<uint32_t>(), E->hit_off.as<int32_t>(),
				E->hit_cnt.as<int32_t>(), E->thr.as<int32_t>(), E->st); }
			if (he != hipSuccess) return fail(E, FASIM_E_HIP, "hits launch failed: %s", hipGetErrorString(he));
			uint32_t total = 0;
			HIPOK(hipMemcpyAsync(&total, E->hits_total.p, sizeof total, hipMemcpyDeviceToHost, E->st));
			HIPOK(hipStreamSynchronize(E->st));
			if (total <= hits_cap) { hits.resize(total); break; }
			hits_cap = (size_t)total + 1024;
		}
		HIPOK(hipMemcpyAsync(hoff.data(), E->hit_off.p, sizeof(int32_t) * B.nunit, hipMemcpyDeviceToHost, E->st));
		HIPOK(hipMemcpyAsync(hcnt.data(), E->hit_cnt.p, sizeof(int32_t) * B.nunit, hipMemcpyDeviceToHost, E->st));
		HIPOK(hipMemcpyAsync(thr.data(), E->thr.p, sizeof(int32_t) * B.nunit, hipMemcpyDeviceToHost, E->st));
		HIPOK(hipMemcpyAsync(pre_max.data(), E->max_out.p, sizeof(int32_t) * B.nunit, hipMemcpyDeviceToHost, E->st));
		if (!hits.empty()) HIPOK(hipMemcpyAsync(hits.data(), E->hits.p, sizeof(uint32_t) * hits.size(), hipMemcpyDeviceToHost, E->st));
		HIPOK(hipStreamSynchronize(E->st));
		for (int u = 0; u < B.nunit; u++) if (pre_max[u] == 255) st.stage2_overflow_units++;
		st.t_stage2_s += now_s() - t0;
		}

		if (p.classicSim) {
			// ---- -F: classic SIM instead of fastSIM (Fasim-LongTarget.cpp:420-426): the forward sweep of every unit on the GPU
			//      (k_sim_forward), then K rounds in lock step: traceback + triplex record on the host threads (host_sim.cpp), re-sweeps on the GPU
			t0 = now_s();
			std::vector<int64_t> mins((size_t)B.nunit);
			for (int u = 0; u < B.nunit; u++) mins[(size_t)u] = thr[(size_t)u];
			std::vector<std::vector<fasim_sim_node>> lists((size_t)B.nunit);
			std::vector<std::vector<HostTriplex>>& per_unit = C.per_unit;
			per_unit.assign((size_t)B.nunit, std::vector<HostTriplex>());
			rc = sim_forward_units(E, E->tcodes.as<uint8_t>(), tstride, E->unit_len.as<int32_t>(), B.unit_len.data(), 0, B.nunit, mins.data(), nullptr, lists);
			if (rc) return rc;
			std::vector<std::unique_ptr<SimUnit>> units((size_t)B.nunit);
			parallel_units(B.nunit, E->sim_threads, [&](int u) {
				std::string target, src;
				const int s = u / nenc, enc = encs[(size_t)(u % nenc)];
				encode_unit_host(dna + sidx[(size_t)s] * step, slen[(size_t)s], enc, target, src);
				units[(size_t)u].reset(new SimUnit(E->rna, target, src, (long)(sidx[(size_t)s] * step), thr[(size_t)u], enc, p, std::move(lists[(size_t)u])));
			});
			std::vector<SimUnit*> up((size_t)B.nunit);
			for (int u = 0; u < B.nunit; u++) up[(size_t)u] = units[(size_t)u].get();
			rc = sim_resweep_rounds(E, E->tcodes.as<uint8_t>(), tstride, E->unit_len.as<int32_t>(), B.unit_len.data(), 0, B.nunit, up.data(), E->sim_threads);
			if (rc) return rc;
			for (int u = 0; u < B.nunit; u++) {
				const int s = u / nenc, enc = encs[(size_t)(u % nenc)];
				per_unit[(size_t)u] = std::move(units[(size_t)u]->out);
				for (HostTriplex& t : per_unit[(size_t)u]) { t.seg = (int)sidx[(size_t)s]; t.enc = enc; }
			}
			st.t_stage3_s += now_s() - t0;
			C.stage3_done = true;
			return FASIM_OK;
		}
		// the scan phase ends here: stage 3 runs per unit range (stage3_range), on this engine or on helpers
		C.per_unit.assign((size_t)B.nunit, std::vector<HostTriplex>());
		C.seg_acgtn.resize((size_t)nseg);
		for (int s = 0; s < nseg; s++) C.seg_acgtn[(size_t)s] = only_acgtn(dna + sidx[(size_t)s] * step, slen[(size_t)s]) ? 1 : 0;
	}
	return FASIM_OK;
}


// LongTarget()'s tail filter (Fasim-LongTarget.cpp:589-597) over the units of a finished batch, in canonical order
static void collect_batch(BatchCtx& C, std::vector<HostTriplex>& all)
{
	const fasim_params& p = *C.p;
	for (auto& unit : C.per_unit)
		for (HostTriplex& t : unit)
			if (t.score >= p.scoreMin && t.identity >= p.minIdentity && t.tri_score >= p.minStability && t.nt >= p.cLength)
				all.push_back(std::move(t));
}

// pack the records of one query into the C result
int pack_result(fasim_engine* E, std::vector<HostTriplex>& all, const fasim_scan_stats& st, fasim_result** out)
{
	fasim_result* R = (fasim_result*)calloc(1, sizeof(fasim_result));
	if (!R) return fail(E, FASIM_E_NOMEM, "out of memory");
	size_t pool = 0;
	for (const HostTriplex& t : all) pool += t.tfo.size() + t.tts.size() + 2;
	R->count = (int64_t)all.size();
	R->recs = (fasim_triplex*)calloc(std::max<size_t>(1, all.size()), sizeof(fasim_triplex));
	R->pool = (char*)calloc(std::max<size_t>(1, pool), 1);
	if (!R->recs || !R->pool) { fasim_result_free(R); return fail(E, FASIM_E_NOMEM, "out of memory"); }
	R->pool_len = (int64_t)pool;
	size_t off = 0;
	for (size_t i = 0; i < all.size(); i++) {
		const HostTriplex& t = all[i];
		fasim_triplex& r = R->recs[i];
		r.stari = t.stari; r.endi = t.endi; r.starj = t.starj; r.endj = t.endj; r.strand = t.strand; r.reverse = t.reverse;
		r.rule = t.rule; r.nt = t.nt; r.score = t.score; r.identity = t.identity; r.tri_score = t.tri_score; r.seg = t.seg; r.enc = t.enc;
		r.tfo_off = (int64_t)off; memcpy(R->pool + off, t.tfo.c_str(), t.tfo.size() + 1); off += t.tfo.size() + 1;
		r.tts_off = (int64_t)off; memcpy(R->pool + off, t.tts.c_str(), t.tts.size() + 1); off += t.tts.size() + 1;
	}
	R->stats = st;
	*out = R;
	return FASIM_OK;
}

// The same from the batches' lists as they are (one query's batches in canonical order): record and pool positions of every
// batch follow from a prefix sum, so the batches are copied side by side on `threads` host threads.
static int pack_result_parts(fasim_engine* E, const std::vector<const std::vector<HostTriplex>*>& parts, const fasim_scan_stats& st, int threads,
	fasim_result** out)
{
	fasim_result* R = (fasim_result*)calloc(1, sizeof(fasim_result));
	if (!R) return fail(E, FASIM_E_NOMEM, "out of memory");
	const size_t np = parts.size();
	std::vector<size_t> rbase(np + 1, 0), pbase(np + 1, 0);
	for (size_t k = 0; k < np; k++) {
		size_t pool = 0;
		for (const HostTriplex& t : *parts[k]) pool += t.tfo.size() + t.tts.size() + 2;
		rbase[k + 1] = rbase[k] + parts[k]->size(); pbase[k + 1] = pbase[k] + pool;
	}
	const size_t count = rbase[np], pool = pbase[np];
	R->count = (int64_t)count;
	R->recs = (fasim_triplex*)malloc(std::max<size_t>(1, count) * sizeof(fasim_triplex));
	R->pool = (char*)malloc(std::max<size_t>(1, pool));
	if (!R->recs || !R->pool) { fasim_result_free(R); return fail(E, FASIM_E_NOMEM, "out of memory"); }
	R->pool_len = (int64_t)pool;
	if (!pool) R->pool[0] = 0;
	std::atomic<size_t> next(0);
	auto work = [&]() {
		for (;;) {
			const size_t k = next.fetch_add(1);
			if (k >= np) break;
			size_t off = pbase[k];
			fasim_triplex* dst = R->recs + rbase[k];
			for (const HostTriplex& t : *parts[k]) {
				fasim_triplex r;
				memset(&r, 0, sizeof r);
				r.stari = t.stari; r.endi = t.endi; r.starj = t.starj; r.endj = t.endj; r.strand = t.strand; r.reverse = t.reverse;
				r.rule = t.rule; r.nt = t.nt; r.score = t.score; r.identity = t.identity; r.tri_score = t.tri_score; r.seg = t.seg; r.enc = t.enc;
				r.tfo_off = (int64_t)off; memcpy(R->pool + off, t.tfo.c_str(), t.tfo.size() + 1); off += t.tfo.size() + 1;
				r.tts_off = (int64_t)off; memcpy(R->pool + off, t.tts.c_str(), t.tts.size() + 1); off += t.tts.size() + 1;
				*dst++ = r;
			}
		}
	};
	const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, threads), count > 20000 ? np : 1));
	if (nt == 1) work();
	else { std::vector<std::thread> th; for (int k = 0; k < nt; k++) th.emplace_back(work); for (auto& t : th) t.join(); }
	R->stats = st;
	*out = R;
	return FASIM_OK;
}

static void add_stats(fasim_scan_stats& st, const fasim_scan_stats& x)
{
	st.segments += x.segments; st.segments_skipped += x.segments_skipped; st.units += x.units; st.candidates += x.candidates;
	st.align_calls += x.align_calls; st.align_word_reruns += x.align_word_reruns; st.stage2_overflow_units += x.stage2_overflow_units;
	st.stage1_word_reruns += x.stage1_word_reruns; st.logical_cells += x.logical_cells; st.t_stage1_s += x.t_stage1_s;
	st.t_stage2_s += x.t_stage2_s; st.t_stage3_s += x.t_stage3_s; st.t_host_s += x.t_host_s; st.cells_stage1 += x.cells_stage1;
	st.cells_stage2 += x.cells_stage2; st.cells_stage3 += x.cells_stage3; st.hazard_units += x.hazard_units; st.rev_exact += x.rev_exact;
	st.exact_replays += x.exact_replays; st.tries_skipped += x.tries_skipped;
	st.band_tries += x.band_tries; st.band_proven += x.band_proven; st.band_cells += x.band_cells; st.rev_bound_passes += x.rev_bound_passes;
	for (int k = 0; k < FASIM_KERNEL_FAMILIES; k++) { st.kernel_ms[k] += x.kernel_ms[k]; st.kernel_launches[k] += x.kernel_launches[k]; }
}

// The body of fasim_scan / fasim_scan_queries: every (query, batch of segments) pair is one work item; the worker engines
// take items from one queue, so the tail of one query's scan overlaps the head of the next (no ramp-up / drain per query).
// nq == 0: the engine's current query.
int scan_core(fasim_engine* E, const char* const* rnas, const int32_t* rna_lens, int nq, const char* dna, int64_t dna_len,
	int64_t seg_first, int64_t seg_count, const fasim_params* pp, fasim_result** outs)
{
	const bool resident = (dna == nullptr);
	if (resident) {
		if (E->dna_host.empty()) return fail(E, FASIM_E_ARG, "no resident DNA: call fasim_load_dna first");
		dna = E->dna_host.data(); dna_len = (int64_t)E->dna_host.size();
	}
	if (!dna || dna_len <= 0 || !pp || !outs) return fail(E, FASIM_E_ARG, "bad arguments");
	const fasim_params p = *pp;
	if (p.cutLength <= 0 || p.cutLength - p.overlapLength <= 0) return fail(E, FASIM_E_ARG, "cutLength/overlapLength invalid");
	if (dna_len > 0x7fffffffll) return fail(E, FASIM_E_ARG, "one record is limited to 2^31-1 nt (the reference's int positions)");
	HIPOK(hipSetDevice(E->device));
	const double t_begin = now_s();
	AffinityScope numa(E->device, E->opt_numa != 0);
	{ const char* pe = getenv("FASIM_PROFILE"); g_prof.on = pe && atoi(pe) != 0; g_prof.reset(); }

	std::vector<std::string> queries;
	if (nq <= 0) queries.push_back(E->rna);
	else for (int q = 0; q < nq; q++) {
		if (!rnas || !rna_lens || !rnas[q] || rna_lens[q] <= 0) return fail(E, FASIM_E_ARG, "empty query %d", q);
		queries.emplace_back(rnas[q], rnas[q] + rna_lens[q]);
	}
	const int nquery = (int)queries.size();
	for (int q = 0; q < nquery; q++) outs[q] = nullptr;

	const int64_t nseg_all = fasim_segment_count(dna_len, &p);
	if (seg_first < 0) seg_first = 0;
	if (seg_count < 0 || seg_first + seg_count > nseg_all) seg_count = std::max<int64_t>(0, nseg_all - seg_first);
	const int64_t step = p.cutLength - p.overlapLength;
	const std::vector<int> encs = enabled_encodings(p);
	const int nenc = (int)encs.size();

	std::vector<std::vector<HostTriplex>> all(nquery);
	bool packed = false;
	std::vector<fasim_scan_stats> qst(nquery);
	for (auto& x : qst) memset(&x, 0, sizeof x);
	if (seg_count > 0 && nenc > 0) {
		// the shard's DNA stays resident for the whole scan (all queries)
		const int64_t shard_lo = seg_first * step;
		const int64_t shard_hi = std::min<int64_t>(dna_len, (seg_first + seg_count - 1) * step + p.cutLength);
		// resident record: the kernels read it in place; host buffer: every batch streams its own slice (scan_batch)
		const uint8_t* dna_dev = resident ? E->dna_res.as<uint8_t>() + shard_lo : nullptr;
		(void)shard_hi;
		const int tstride = (p.cutLength + 15) & ~15;
		// Batches of ~384 segments x 48 encodings; several batches are in flight at once on worker engines (own HIP
		// stream + buffers + host thread), so the latency-bound kernels (stripe-faithful re-runs, tracebacks) and the
		// host-side work of one batch overlap the VALU-bound kernels of another.
		// (384 rather than 512: a 50 Mb record then gives 27 batches for the 10 workers instead of exactly two rounds of ten, which
		//  made all workers finish their last batch together: tools/sweep_sched.py, profiles/r02_sched_sweep.txt)
		int64_t seg_batch = std::max<int64_t>(1, std::min<int64_t>(384, ((int64_t)8 << 30) / ((int64_t)4 * nenc * tstride)));
		const char* envb = getenv("FASIM_SEG_BATCH");
		if (envb) seg_batch = std::max(1, atoi(envb));
		if (E->opt_seg_batch > 0) seg_batch = E->opt_seg_batch;
		int nworkers = 10;
		const char* envw = getenv("FASIM_WORKERS");
		if (envw) nworkers = std::max(1, std::min(16, atoi(envw)));
		if (E->opt_workers > 0) nworkers = std::min(16, E->opt_workers);
		// option taper = t (percent): the last t % of the segments go in half-size batches, so that the workers do not all finish
		// their last batch at the same moment (shorter drain at the end of a scan)
		int taper_pct = E->opt_taper >= 0 ? E->opt_taper : 0;
		// Batch size fitted to the record (single-lncRNA scans of >= 128 segments per worker; an explicit seg_batch switches it off): the segments are cut so that every worker gets R whole rounds of batches of at most 512 segments (R = the
		// fewest rounds that allow it), and the last quarter of the record goes in half-size batches.  The workers then neither
		// idle through a partial last round nor finish their last full-size batch all at once (the drain of a scan is the
		// stage 3 of its last batches on an otherwise idle GPU): 50 Mb = 10 204 segments -> 15 batches of 511 + 10 of 255,
		// 2.41 s against 2.61 s with fixed batches of 384 (profiles/r02_ab_batch_shape.txt: the optimum sits exactly where
		// the batches tile the ten workers, 448 and 576 are both slower than 512; other record sizes: r02_ab_sizes.txt).
		// A batch of several lncRNAs is one stream of items, lncRNA after lncRNA, and keeps fixed batches of 384.
		if (!envb && E->opt_seg_batch <= 0 && nquery == 1 && seg_count >= (int64_t)128 * nworkers) {
			const int64_t target = 512;
			const int64_t rounds = std::max<int64_t>(1, (seg_count + target * (int64_t)nworkers - 1) / (target * (int64_t)nworkers));
			seg_batch = std::max<int64_t>(1, std::min<int64_t>((seg_count + rounds * nworkers - 1) / (rounds * nworkers), ((int64_t)8 << 30) / ((int64_t)4 * nenc * tstride)));
			if (E->opt_taper < 0) taper_pct = 25;
		}
		if (p.classicSim && !envb && E->opt_seg_batch <= 0) {
			// -F: a batch's rounds advance launch by launch and end with the tail of its heaviest unit, so several
			// batches should be in flight -- but each batch's host half runs on its worker's share of the cores, so not too many
			// either: about eight batches per record, 16 ... 128 segments (768 ... 6 144 units) each.  Measured with step-count launches
			// (profiles/r03_simF_500kb.txt): 500 kb 36 s as one batch, 25.6 s as four to seven; 2 Mb 49.8 s as eight batches of 52
			// segments, 97 s as twenty of 21 (the final time-sliced build: 18.2 / 17.8 s and 27.1 s).
			seg_batch = std::max<int64_t>(16, std::min<int64_t>(128, (seg_count + 7) / 8));
			taper_pct = 0;
		}
		std::vector<std::pair<int64_t, int64_t>> chunks;
		{
			int64_t b0 = seg_first; const int64_t b_end = seg_first + seg_count;
			const int64_t taper_from = b_end - seg_count * taper_pct / 100;
			while (b0 < b_end) {
				int64_t len = (taper_pct > 0 && b0 >= taper_from) ? std::max<int64_t>(1, seg_batch / 2) : seg_batch;
				len = std::min(len, b_end - b0);
				chunks.push_back({ b0, b0 + len });
				b0 += len;
			}
		}
		struct Item { int q; int64_t b0, b1; };
		std::vector<Item> items;
		items.reserve(chunks.size() * (size_t)nquery);
		for (int q = 0; q < nquery; q++) for (const auto& c : chunks) items.push_back({ q, c.first, c.second });
		nworkers = (int)std::min<size_t>((size_t)nworkers, items.size());
		// worker 0 is this engine; the others are lazily created engines on the same device
		while ((int)E->workers.size() < nworkers - 1) {
			fasim_engine* w = nullptr;
			int rc = fasim_engine_create(E->device, &w); if (rc) return fail(E, rc, "cannot create worker engine: %s", fasim_last_error(nullptr));
			E->workers.push_back(w);
		}
		std::atomic<int> active_workers(nworkers), sim_active(0);
		std::vector<fasim_engine*> ws(1, E);
		for (int k = 0; k < nworkers - 1; k++) ws.push_back(E->workers[k]);
		{
			const char* envg = getenv("FASIM_HEAVY_GATE");      // heavy kernels in flight at once (0 = no gate)
			E->own_gate.cap = E->opt_gate >= 0 ? E->opt_gate : (envg ? atoi(envg) : 4);
		}
		for (fasim_engine* w : ws) {
			w->gate = (ws.size() > 1 && E->own_gate.cap > 0) ? &E->own_gate : nullptr;
			w->scan_v1 = E->scan_v1; w->align_v1 = E->align_v1;
			w->hz_chunks = E->hz_chunks; w->hz_snap = E->hz_snap; w->hz_target = E->hz_target; w->hz_hot_w = E->hz_hot_w; w->opt_band = E->opt_band;
			w->host_threads = std::max(1, E->host_threads_total / nworkers);
			w->host_threads_share_total = E->host_threads_total; w->active_workers = &active_workers; w->sim_in_flight = &sim_active; w->scan_workers = (int)ws.size();
			{
				// -F: the finish half of classic SIM is ~40 ms of host work per unit and nothing else needs the cores meanwhile
				const int all = E->host_threads_explicit ? E->host_threads_total : usable_cores();
				// (the batches' host halves rarely coincide: a quarter of the workers share the cores; 5 Mb: 38.4 -> 35.2 s, all cores each: 36.1 s)
				w->sim_threads = std::max(1, all / (int)std::max<size_t>(1, std::min<size_t>((size_t)nworkers, items.size()) / 4));
			}
			HIPOK(hipSetDevice(E->device));
			int rc = upload(w, w->enc_ids, encs.data(), sizeof(int) * nenc); if (rc) return rc;
			drain_timed(w);
			for (int k = 0; k < FASIM_KERNEL_FAMILIES; k++) { w->kernel_ms[k] = 0; w->kernel_launches[k] = 0; }
		}
		std::vector<std::vector<HostTriplex>> per_item(items.size());
		std::vector<fasim_scan_stats> ist(items.size());
		for (auto& x : ist) memset(&x, 0, sizeof x);
		std::vector<double> it0(items.size(), 0.0), it1(items.size(), 0.0);
		std::vector<int> wrc(ws.size(), FASIM_OK);
		std::atomic<size_t> next(0);
		// A worker owns an item (one lncRNA x one batch of segments) from its scan phase (stages 1+2) through stage 3.
		// (A cooperative tail -- the stage 3 of the last batches cut into sub-tasks that idle workers steal -- was built in round 2
		//  and measured 0.1-0.2 s per 50 Mb scan slower than without it, profiles/r02_ab_trees.txt; it was removed in round 3.)
		auto run = [&](size_t wi) {
			CpuScope cpu(28, "CPU seconds: worker threads themselves (HIP calls, lists, decisions)");
			(void)hipSetDevice(E->device);
			fasim_engine* w = ws[wi];
			for (;;) {
				const size_t c = next.fetch_add(1);
				if (c >= items.size()) { active_workers.fetch_sub(1); break; }
				const Item& itx = items[c];
				const std::string& rq = queries[(size_t)itx.q];
				it0[c] = now_s();
				BatchCtx ctx;
				int r = FASIM_OK;
				if (w->rna != rq) {          // the worker switches to this item's lncRNA (3 x m bytes H2D)
					r = fasim_set_query(w, rq.data(), (int)rq.size());
					if (r && w != E) w->err = std::string("worker set_query failed: ") + w->err;
				}
				if (!r) r = scan_batch(w, dna, dna_len, dna_dev, shard_lo, itx.b0, itx.b1, p, encs, tstride, ctx, ist[c]);
				if (!r && ctx.B.nunit > 0 && !ctx.stage3_done) r = stage3_range(w, ctx, 0, ctx.B.nunit, ist[c]);
				(void)hipStreamSynchronize(w->st);
				drain_timed(w);
				for (int k = 0; k < FASIM_KERNEL_FAMILIES; k++) { ist[c].kernel_ms[k] = w->kernel_ms[k]; ist[c].kernel_launches[k] = w->kernel_launches[k]; w->kernel_ms[k] = 0; w->kernel_launches[k] = 0; }
				if (!r) collect_batch(ctx, per_item[c]);
				it1[c] = now_s();
				if (r) wrc[wi] = r;
			}
			(void)hipStreamSynchronize(w->st);
		};
		if (g_prof.on) fprintf(stderr, "[fasim prof] scan head (setup before the workers start)  %.3f s\n", now_s() - t_begin);
		const double t_workers = now_s();
		if (ws.size() == 1) run(0);
		else { std::vector<std::thread> th; for (size_t wi = 0; wi < ws.size(); wi++) th.emplace_back(run, wi); for (auto& t : th) t.join(); }
		for (fasim_engine* w : ws) { w->active_workers = nullptr; w->sim_in_flight = nullptr; w->scan_workers = 1; }
		if (g_prof.on) fprintf(stderr, "[fasim prof] scan workers                                  %.3f s\n", now_s() - t_workers);
		for (size_t wi = 0; wi < ws.size(); wi++) if (wrc[wi]) { if (ws[wi] != E) E->err = ws[wi]->err; return wrc[wi]; }
		// a multi-query call leaves the engine on its LAST query (documented in fasim_hip.h)
		if (nq > 0 && E->rna != queries.back()) { int rc = fasim_set_query(E, queries.back().data(), (int)queries.back().size()); if (rc) return rc; }
		const double t_merge = now_s();
		std::vector<double> q0(nquery, 1e300), q1(nquery, 0.0);
		std::vector<std::vector<const std::vector<HostTriplex>*>> parts((size_t)nquery);
		for (size_t c = 0; c < items.size(); c++) {
			const int q = items[c].q;
			parts[(size_t)q].push_back(&per_item[c]);
			add_stats(qst[(size_t)q], ist[c]);
			q0[q] = std::min(q0[q], it0[c]); q1[q] = std::max(q1[q], it1[c]);
		}
		// per query: wall clock from the start of its first batch to the end of its last one (neighbouring queries overlap)
		for (int q = 0; q < nquery; q++) qst[(size_t)q].t_total_s = nquery == 1 ? 0.0 : std::max(0.0, q1[q] - q0[q]);
		if (nquery == 1) qst[0].t_total_s = now_s() - t_begin;
		// the records go straight from the batches' lists into the C result (the batches side by side on the host threads);
		// the lists themselves (half a million strings for a 50 Mb record) are freed behind the caller's back
		for (int q = 0; q < nquery; q++) {
			const int rc = pack_result_parts(E, parts[(size_t)q], qst[(size_t)q], E->host_threads_total, &outs[q]);
			if (rc) { for (int k = 0; k < q; k++) { fasim_result_free(outs[k]); outs[k] = nullptr; } return rc; }
		}
		if (nquery == 1) outs[0]->stats.t_total_s = now_s() - t_begin;
		if (g_prof.on) fprintf(stderr, "[fasim prof] scan tail: packing the records                        %.3f s\n", now_s() - t_merge);
		// Free the batches' lists (half a million strings for a 50 Mb record) here, side by side on the host threads, while the
		// GPU is idle: left to a background thread the unmapping runs into the first kernels of the caller's next scan and
		// stretches them by half (the driver's MMU notifier stalls the queues while the address space changes: tools/iso_probe.py,
		// profiles/r02_ab_reaper.txt).
		{
			std::atomic<size_t> nextf(0);
			auto freer = [&]() { for (;;) { const size_t c = nextf.fetch_add(1); if (c >= per_item.size()) break; std::vector<HostTriplex>().swap(per_item[c]); } };
			const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, E->host_threads_total), per_item.size()));
			if (nt == 1) freer();
			else { std::vector<std::thread> th; for (int k = 0; k < nt; k++) th.emplace_back(freer); for (auto& t : th) t.join(); }
		}
		if (g_prof.on) fprintf(stderr, "[fasim prof] scan tail: packing + freeing the batches' lists        %.3f s\n", now_s() - t_merge);
		packed = true;
	}

	if (!packed) {
		// (nothing to scan: empty results)
		if (nquery == 1) qst[0].t_total_s = now_s() - t_begin;
		for (int q = 0; q < nquery; q++) {
			const int rc = pack_result(E, all[(size_t)q], qst[(size_t)q], &outs[q]);
			if (rc) { for (int k = 0; k < q; k++) { fasim_result_free(outs[k]); outs[k] = nullptr; } return rc; }
		}
		if (nquery == 1) outs[0]->stats.t_total_s = now_s() - t_begin;
	}
	if (g_prof.on) {
		static long seen = 0; const long now_r = g_dev_reallocs.load();
		fprintf(stderr, "[fasim prof] device buffer (re)allocations during this scan: %ld\n", now_r - seen); seen = now_r;
		double tot = now_s() - t_begin;
		fprintf(stderr, "[fasim prof] total %.3f s  stage2 %.3f  stage3 %.3f  host %.3f\n", tot, qst[0].t_stage2_s, qst[0].t_stage3_s, qst[0].t_host_s);
		g_prof.dump();
	}
	return FASIM_OK;
}

