// fasim-longtarget_amd/csrc/engine.cpp -- the C-ABI of libfasim_hip.so (include/fasim_hip.h); the engine behind it lives in
// engine.h, engine_stage2.cpp, engine_stage3.cpp and engine_scan.cpp.
#include "engine.h"

// =====================================================================================================
// C-ABI
// =====================================================================================================
extern "C" {

void fasim_params_default(fasim_params* p)
{
	p->rule = 0; p->cutLength = 5000; p->strand = 0; p->overlapLength = 100; p->ntMin = 20; p->ntMax = 100000;
	p->scoreMin = 0.0f; p->minIdentity = 60.0f; p->minStability = 1.0f; p->penaltyT = -1000; p->penaltyC = 0;
	p->cDistance = 15; p->cLength = 50; p->classicSim = 0;
}

const char* fasim_last_error(const fasim_engine* e) { return e ? e->err.c_str() : g_last_error.c_str(); }

int fasim_engine_create(int device, fasim_engine** out) { return fasim_engine_create_ex(device, 0, out); }

int fasim_engine_create_ex(int device, int32_t flags, fasim_engine** out)
{
	if (!out) return fail(nullptr, FASIM_E_ARG, "null out pointer");
	*out = nullptr;
	const bool tune = !(flags & FASIM_CREATE_NO_PROCESS_TUNING);
	// fasim_scan keeps ~10 batches in flight on as many streams; the HIP runtime multiplexes streams onto 4 hardware
	// queues unless told otherwise, which serialises unrelated batches.  Only effective before the runtime initialises
	// (a host application that touches HIP earlier should export GPU_MAX_HW_QUEUES itself).
	if (tune) setenv("GPU_MAX_HW_QUEUES", "8", 0);
	// Host allocator: every batch builds and drops tables of a few MB on its host threads.  glibc hands such blocks back to the
	// kernel (munmap / heap trimming), and while the address space of a process with live HIP queues changes, the driver's MMU
	// notifier holds those queues up (a 120 ms kernel that overlaps the freeing of half a million strings takes 190-250 ms:
	// tools/iso_probe.py).  Keeping freed blocks of up to 32 MB in the heap removes most of that: 2.42 -> 2.38 s per 50 Mb scan
	// (profiles/r02_ab_malloc.txt).  Process-wide settings; FASIM_MALLOPT=0 leaves the allocator alone.
	if (tune) {
		static const bool once = [] {
			const char* e = getenv("FASIM_MALLOPT");
			if (e && atoi(e) == 0) return false;
			(void)mallopt(M_MMAP_THRESHOLD, 32 * 1024 * 1024);
			(void)mallopt(M_TRIM_THRESHOLD, 0x7fffffff);
			(void)mallopt(M_TOP_PAD, 256 * 1024 * 1024);
			return true;
		}();
		(void)once;
	}
	int count = 0;
	hipError_t he = hipGetDeviceCount(&count);
	if (he != hipSuccess || count <= 0) return fail(nullptr, FASIM_E_NODEVICE, "no HIP device available (%s); this library has no CPU fallback", hipGetErrorString(he));
	if (device < 0 || device >= count) return fail(nullptr, FASIM_E_NODEVICE, "device %d out of range (%d devices)", device, count);
	{
		// Host threads SLEEP in hipStreamSynchronize instead of polling: with ten batches in flight ten threads otherwise spin
		// through a scan -- 12 of the 25 CPU-seconds a 50 Mb scan costs -- on cores the host side of the batches needs (an 8-GPU
		// node gives each rank a fraction of its cores).  The flag only takes effect before the device's context exists; a host
		// application that touches HIP first (PyTorch) has to set it itself (bench.py does).  An event created with
		// hipEventBlockingSync does not have this effect on ROCm 7.2.  FASIM_BLOCKING_SYNC=0 leaves the polling wait.
		const char* e = getenv("FASIM_BLOCKING_SYNC");
		if (tune && !(e && atoi(e) == 0)) { (void)hipSetDevice(device); (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync); (void)hipGetLastError(); }
	}
	fasim_engine* E = new fasim_engine();
	E->device = device;
	he = hipSetDevice(device);
	// one stream per engine (CU-masked and prioritised second streams for the two VALU-bound kernels were measured in round 2 and
	// lost: DESIGN.md section 4)
	if (he == hipSuccess) he = hipStreamCreate(&E->st);
	if (he != hipSuccess) { int rc = fail(nullptr, FASIM_E_NODEVICE, "cannot initialise device %d: %s", device, hipGetErrorString(he)); delete E; return rc; }
	E->lut1 = make_lut(true); E->lut2 = make_lut(false);
	std::vector<uint8_t> lut(48 * 256);
	build_enc_lut(lut.data());
	if (upload(E, E->enc_lut, lut.data(), lut.size()) || E->counter.ensure(64) != hipSuccess) { delete E; return FASIM_E_HIP; }
	unsigned hc = std::thread::hardware_concurrency();
	const char* env = getenv("FASIM_HOST_THREADS");
	// host side of a batch (CIGAR -> triplex record for every candidate alignment, ~1 us each): short bursts, shared by
	// the batches in flight; 3/8 of the cores (96 on the 256-thread GPU hosts) keeps the burst of a batch below ~50 ms
	E->host_threads = env ? std::max(1, atoi(env)) : (int)std::min(96u, std::max(1u, hc * 3 / 8));
	E->host_threads_total = E->host_threads;
	E->host_threads_explicit = env != nullptr;
	const char* v1 = getenv("FASIM_SCAN_V1");
	E->scan_v1 = v1 && atoi(v1) != 0;
	const char* a1 = getenv("FASIM_ALIGN_V1");
	E->align_v1 = a1 && atoi(a1) != 0;
	*out = E;
	return FASIM_OK;
}

void fasim_engine_destroy(fasim_engine* e)
{
	if (!e) return;
	for (fasim_engine* w : e->workers) fasim_engine_destroy(w);
	e->workers.clear();
	(void)hipSetDevice(e->device);
	DevBuf* bufs[] = { &e->q1, &e->q2, &e->enc_lut, &e->counter, &e->dna, &e->seg_start, &e->seg_len, &e->enc_ids, &e->tcodes,
		&e->colmax, &e->probs, &e->max_out, &e->unit_len, &e->stage1, &e->hits, &e->hits_total, &e->hit_off, &e->hit_cnt, &e->thr,
		&e->ends, &e->bprobs, &e->bout, &e->scratch, &e->dna_res, &e->colmax16, &e->unit_ids, &e->flags, &e->stage1_in, &e->hits2,
		&e->fprobs, &e->ftasks, &e->fstream, &e->fout, &e->aout, &e->cigpool, &e->cigcount, &e->forder, &e->scratch2, &e->boundary, &e->fboundary, &e->unit_hz,
		&e->unit_first, &e->hz_cols, &e->hz_plan, &e->hz_base, &e->hz_items, &e->snap, &e->hz_state, &e->hz_rows, &e->hz_chunk, &e->hz_src, &e->hz_zero,
		&e->qsim, &e->sim_min, &e->sim_row, &e->sim_ev, &e->sim_cnt, &e->sim_nodes,
		&e->ublk, &e->btarget, &e->bidx, &e->bcounts, &e->blist[0], &e->blist[1], &e->blist[2], &e->bslots[0], &e->bslots[1], &e->bslots[2],
		&e->bprev, &e->lane_ub, &e->fzones, &e->fubslot, &e->bdec, &e->btab };
	for (auto& t : e->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
	for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
	for (DevBuf* b : bufs) b->release();
	if (e->pin_dna) (void)hipHostFree(e->pin_dna);
	if (e->pin_sim) (void)hipHostFree(e->pin_sim);
	if (e->st) (void)hipStreamDestroy(e->st);
	delete e;
}

int fasim_set_option(fasim_engine* E, const char* key, int32_t value)
{
	if (!E || !key) return fail(E, FASIM_E_ARG, "null argument");
	if (!strcmp(key, "workers")) E->opt_workers = value > 0 ? value : 0;
	else if (!strcmp(key, "seg_batch")) E->opt_seg_batch = value > 0 ? value : 0;
	else if (!strcmp(key, "taper")) E->opt_taper = value;          // percent of the segments scanned in half-size batches at the end (-1: default)
	else if (!strcmp(key, "heavy_gate")) E->opt_gate = value;      // k_scan / k_align_fwd launches in flight at once (0: no gate, -1: default)
	else if (!strcmp(key, "hazard_chunks")) E->hz_chunks = value;          // 0: whole-unit stripe-faithful re-run; 1: column chunks (default)
	else if (!strcmp(key, "hazard_snapshots")) E->hz_snap = value;        // 0: the checkpoint pass starts every unit at column 0
	else if (!strcmp(key, "hazard_chunk_cols")) E->hz_target = value > 0 ? std::max(64, value) : 0;
	else if (!strcmp(key, "hazard_hot_weight")) E->hz_hot_w = value > 0 ? std::min(32, value) : 0;
	else if (!strcmp(key, "host_threads")) { if (value > 0) { E->host_threads = value; E->host_threads_total = value; E->host_threads_explicit = true; } }   // host side of the batches (all workers together)
	else if (!strcmp(key, "numa_affinity")) E->opt_numa = value != 0;
	else if (!strcmp(key, "band")) E->opt_band = value;                   // banded stage-3 forward pass: 0 off, 1 on (-1: default / FASIM_BAND)
	else return fail(E, FASIM_E_ARG, "unknown option %s", key);
	return FASIM_OK;
}

int fasim_set_query(fasim_engine* E, const char* rna, int32_t len)
{
	if (!E) return fail(nullptr, FASIM_E_ARG, "null engine");
	if (!rna || len <= 0) return fail(E, FASIM_E_ARG, "empty query");
	HIPOK(hipSetDevice(E->device));
	E->rna.assign(rna, rna + len);
	E->m = len;
	std::vector<uint8_t> c1(len), c2(len);
	E->query_acgt = true;
	for (int i = 0; i < len; i++) {
		c1[i] = code1(rna[i]); c2[i] = code2(rna[i]);
		const char c = rna[i];
		if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'a' || c == 'c' || c == 'g' || c == 't')) E->query_acgt = false;
	}
	std::vector<uint8_t> cs(len);
	for (int i = 0; i < len; i++) cs[i] = sim_code(rna[i]);
	int rc = upload(E, E->q1, c1.data(), len);
	if (!rc) rc = upload(E, E->q2, c2.data(), len);
	if (!rc) rc = upload(E, E->qsim, cs.data(), len);
	if (rc) return rc;
	HIPOK(hipStreamSynchronize(E->st));
	return FASIM_OK;
}

int fasim_pre_align_batch(fasim_engine* E, const char* targets, const int64_t* offsets, const int32_t* lens,
	int32_t nprob, int32_t* out_cols, int32_t* out_stage1)
{
	int rc = need_query(E); if (rc) return rc;
	if (!targets || !offsets || !lens || nprob <= 0) return fail(E, FASIM_E_ARG, "bad batch arguments");
	HIPOK(hipSetDevice(E->device));
	UnitBatch B;
	if (out_stage1) {
		rc = load_raw_targets(E, targets, offsets, lens, nprob, true, B); if (rc) return rc;
		std::vector<int> sc;
		rc = run_stage1(E, B, sc, nullptr); if (rc) return rc;
		memcpy(out_stage1, sc.data(), sizeof(int32_t) * nprob);
	}
	if (out_cols) {
		rc = load_raw_targets(E, targets, offsets, lens, nprob, false, B); if (rc) return rc;
		rc = run_stage2(E, B); if (rc) return rc;
		std::vector<uint8_t> cm((size_t)nprob * B.tstride);
		HIPOK(hipMemcpyAsync(cm.data(), E->colmax.p, cm.size(), hipMemcpyDeviceToHost, E->st));
		HIPOK(hipStreamSynchronize(E->st));
		for (int i = 0; i < nprob; i++)
			for (int c = 0; c < lens[i]; c++) out_cols[offsets[i] + c] = cm[(size_t)i * B.tstride + c];
	}
	return FASIM_OK;
}

int fasim_calc_score_once(fasim_engine* E, const char* target, int32_t n, int32_t* score)
{
	if (!target || !score) return fail(E, FASIM_E_ARG, "null argument");
	const int64_t off = 0;
	return fasim_pre_align_batch(E, target, &off, &n, 1, nullptr, score);
}

int fasim_ssw_pre_align(fasim_engine* E, const char* target, int32_t n, int32_t* out_cols)
{
	if (!target || !out_cols) return fail(E, FASIM_E_ARG, "null argument");
	const int64_t off = 0;
	return fasim_pre_align_batch(E, target, &off, &n, 1, out_cols, nullptr);
}

int fasim_ssw_colmax_word(fasim_engine* E, const char* target, int32_t n, int32_t* out_cols)
{
	int rc = need_query(E); if (rc) return rc;
	if (!target || !out_cols || n <= 0) return fail(E, FASIM_E_ARG, "bad arguments");
	HIPOK(hipSetDevice(E->device));
	UnitBatch B;
	const int64_t off = 0;
	rc = load_raw_targets(E, target, &off, &n, 1, false, B); if (rc) return rc;
	HIPOK(E->colmax16.ensure((size_t)B.tstride * sizeof(uint16_t)));
	HIPOK(E->max_out.ensure(sizeof(int32_t)));
	const std::vector<StripedProb> probs = whole_unit_probs(B, E->m, nullptr);
	rc = upload(E, E->probs, probs.data(), probs.size() * sizeof(StripedProb)); if (rc) return rc;
	StripedLaunch L;
	L.tcodes = E->tcodes.as<uint8_t>(); L.qcodes = E->q2.as<uint8_t>(); L.probs = E->probs.as<StripedProb>(); L.nprob = 1;
	L.counter = E->counter.as<uint32_t>(); L.lut = E->lut2; L.max_qlen = E->m; L.colmax = nullptr;
	L.colmax_w = E->colmax16.as<uint16_t>(); L.max_out = E->max_out.as<int32_t>(); L.ends = nullptr;
	const hipError_t he = launch_striped(MODE_PRE, true, false, L, E->st);
	if (he == hipErrorInvalidValue) return fail(E, FASIM_E_UNSUPPORTED, "query of %d nt does not fit the LDS-resident striped kernel", E->m);
	if (he != hipSuccess) return fail(E, FASIM_E_HIP, "striped kernel launch failed: %s", hipGetErrorString(he));
	std::vector<uint16_t> cm((size_t)n);
	HIPOK(hipMemcpyAsync(cm.data(), E->colmax16.p, sizeof(uint16_t) * (size_t)n, hipMemcpyDeviceToHost, E->st));
	HIPOK(hipStreamSynchronize(E->st));
	for (int c = 0; c < n; c++) out_cols[c] = cm[(size_t)c];
	return FASIM_OK;
}

int fasim_sim_finish_unit(const char* rna, int32_t m, const char* seg, int32_t n, int32_t enc, int64_t dna_start, int64_t min_score,
	const fasim_params* p, const fasim_sim_node* nodes, int32_t nnodes, fasim_result** out)
{
	if (!rna || m <= 0 || !seg || n <= 0 || enc < 0 || enc >= 48 || !p || (nnodes > 0 && !nodes) || nnodes < 0 || nnodes > FASIM_SIM_K || !out)
		return fail(nullptr, FASIM_E_ARG, "bad arguments");
	std::string target, src;
	encode_unit_host(seg, n, enc, target, src);
	std::vector<fasim_sim_node> list(nodes, nodes + nnodes);
	std::vector<HostTriplex> recs;
	sim_finish_unit(std::string(rna, rna + m), target, src, (long)dna_start, (long)min_score, enc, *p, list, recs);
	for (HostTriplex& t : recs) t.enc = enc;
	fasim_scan_stats st; memset(&st, 0, sizeof st);
	return pack_result(nullptr, recs, st, out);
}

int fasim_sim_forward_batch(fasim_engine* E, const char* targets, const int64_t* offsets, const int32_t* lens, int32_t nprob,
	const int64_t* min_scores, fasim_sim_node* nodes, int32_t* counts)
{
	int rc = need_query(E); if (rc) return rc;
	if (!targets || !offsets || !lens || !min_scores || !nodes || !counts || nprob <= 0) return fail(E, FASIM_E_ARG, "bad arguments");
	HIPOK(hipSetDevice(E->device));
	int maxlen = 1;
	for (int k = 0; k < nprob; k++) {
		if (lens[k] <= 0) return fail(E, FASIM_E_ARG, "empty target %d", k);
		maxlen = std::max(maxlen, lens[k]);
	}
	const int tstride = (maxlen + 15) & ~15;
	std::vector<uint8_t> tc((size_t)nprob * tstride, 4);
	for (int k = 0; k < nprob; k++) for (int c = 0; c < lens[k]; c++) tc[(size_t)k * tstride + c] = sim_code(targets[offsets[k] + c]);
	rc = upload(E, E->tcodes, tc.data(), tc.size()); if (rc) return rc;
	rc = upload(E, E->unit_len, lens, sizeof(int32_t) * nprob); if (rc) return rc;
	std::vector<std::vector<fasim_sim_node>> lists((size_t)nprob);
	rc = sim_forward_units(E, E->tcodes.as<uint8_t>(), tstride, E->unit_len.as<int32_t>(), lens, 0, nprob, min_scores, nullptr, lists);
	if (rc) return rc;
	for (int k = 0; k < nprob; k++) {
		counts[k] = (int32_t)lists[(size_t)k].size();
		for (size_t x = 0; x < lists[(size_t)k].size(); x++) nodes[(size_t)k * FASIM_SIM_K + x] = lists[(size_t)k][x];
	}
	drain_timed(E);
	return FASIM_OK;
}

int fasim_selfcheck_records(uint64_t seed, int32_t n, int32_t* mismatches)
{
	if (!mismatches || n <= 0) return fail(nullptr, FASIM_E_ARG, "bad arguments");
	uint64_t x = seed ? seed : 1;
	auto rnd = [&]() { x += 0x9E3779B97F4A7C15ull; uint64_t z = x; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
	const char letters[] = "ACGTACGTACGTNacgt";
	std::string seg(5000, 'A'), rna(3000, 'A');
	for (char& c : seg) c = letters[rnd() % (sizeof letters - 1)];
	for (char& c : rna) c = "ACGU"[rnd() % 4];
	fasim_params p; fasim_params_default(&p);
	int bad = 0;
	const bool acgtn = false;
	std::vector<HostTriplex> hs; std::vector<TriplexNum> ns;
	for (int k = 0; k < n; k++) {
		// a random alignment: a few runs of M / I / D inside the segment and the lncRNA
		AlignResult al; memset(&al, 0, sizeof al);
		uint32_t cig[16]; int nc = 1 + (int)(rnd() % 6), ref = 0, qry = 0;
		for (int c = 0; c < nc; c++) { const uint32_t op = c % 2 == 0 ? 0u : 1u + (uint32_t)(rnd() % 2), len = 1 + (uint32_t)(rnd() % (op == 0 ? 40 : 3)); cig[c] = (len << 4) | op; if (op != 1) ref += (int)len; if (op != 2) qry += (int)len; }
		al.cigar_off = 0; al.cigar_len = nc; al.sw_score = 20 + (int)(rnd() % 200);
		al.ref_begin = (int)(rnd() % (uint64_t)(5000 - ref)); al.ref_end = al.ref_begin + ref - 1;
		al.query_begin = (int)(rnd() % (uint64_t)(3000 - qry)); al.query_end = al.query_begin + qry - 1;
		const int enc = (int)(rnd() % 48);
		std::vector<HostTriplex> a, b; std::vector<TriplexNum> c;
		convert_triplex(al, cig, rna, seg.data(), 5000, enc, 1000, p, a, acgtn, true);
		convert_triplex(al, cig, rna, seg.data(), 5000, enc, 1000, p, b, acgtn, false);
		convert_triplex_num(al, cig, rna, seg.data(), 5000, enc, 1000, p, c, acgtn);
		if (a.size() != b.size() || a.size() != c.size()) { bad++; continue; }
		if (a.empty()) continue;
		const HostTriplex& s1 = a[0]; const HostTriplex& s2 = b[0]; const TriplexNum& s3 = c[0];
		auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
		if (s1.stari != s3.stari || s1.endi != s3.endi || s1.starj != s3.starj || s1.endj != s3.endj || s1.nt != s3.nt || bits(s1.score) != bits(s3.score) ||
			bits(s1.identity) != bits(s3.identity) || bits(s1.tri_score) != bits(s3.tri_score) || bits(s2.identity) != bits(s1.identity) || bits(s2.tri_score) != bits(s1.tri_score)) bad++;
		// the same record, squeezed into a small coordinate range so that the dedup sees ties, containments and equal scores
		HostTriplex h = s2; h.stari = 1 + (int)(rnd() % 6); h.endi = h.stari + (int)(rnd() % 6); h.starj = 100 + (int)(rnd() % 6); h.endj = h.starj + (int)(rnd() % 6);
		h.score = (float)(50 + rnd() % 4); h.cand = k;
		TriplexNum t; t.stari = h.stari; t.endi = h.endi; t.starj = h.starj; t.endj = h.endj; t.nt = h.nt; t.cand = k; t.score = h.score; t.identity = h.identity; t.tri_score = h.tri_score;
		hs.push_back(h); ns.push_back(t);
		if (hs.size() == 40 || k == n - 1) {
			std::vector<HostTriplex> oh; std::vector<TriplexNum> on;
			dedup_top(hs, p, oh); dedup_top_num(ns, p, on);
			if (oh.size() != on.size()) bad++;
			else for (size_t i = 0; i < oh.size(); i++) if (oh[i].cand != on[i].cand) bad++;
			hs.clear(); ns.clear();
		}
	}
	*mismatches = bad;
	return FASIM_OK;
}

int fasim_pick_candidates(const int32_t* cols, int32_t n, int32_t threshold, int32_t* out_score, int32_t* out_pos,
	int32_t cap, int32_t* count)
{
	if (!cols || !count || n < 0) return fail(nullptr, FASIM_E_ARG, "bad arguments");
	std::vector<uint32_t> hits;
	for (int c = 0; c < n; c++) if (cols[c] > threshold) hits.push_back(((uint32_t)c << 8) | (uint32_t)(cols[c] & 0xff));
	std::vector<Cand> cands;
	pick_candidates(hits.data(), (int)hits.size(), cands);
	*count = (int)cands.size();
	for (int i = 0; i < (int)cands.size() && i < cap; i++) { if (out_score) out_score[i] = cands[i].score; if (out_pos) out_pos[i] = cands[i].pos; }
	return FASIM_OK;
}

int fasim_align_batch(fasim_engine* E, const char* windows, const int64_t* offsets, const int32_t* lens, int32_t nprob,
	fasim_alignment* out)
{
	int rc = need_query(E); if (rc) return rc;
	if (!windows || !offsets || !lens || !out || nprob <= 0) return fail(E, FASIM_E_ARG, "bad batch arguments");
	HIPOK(hipSetDevice(E->device));
	UnitBatch B;
	rc = load_raw_targets(E, windows, offsets, lens, nprob, false, B); if (rc) return rc;
	std::vector<WindowProb> W(nprob);
	for (int i = 0; i < nprob; i++) { W[i].unit = i; W[i].t0 = 0; W[i].len = lens[i]; }
	std::vector<AlignResult> res;
	std::vector<uint32_t> cigars;
	rc = run_align_v2(E, B, W, res, cigars, nullptr); if (rc) return rc;
	for (int i = 0; i < nprob; i++) {
		out[i].sw_score = res[i].sw_score; out[i].ref_begin = res[i].ref_begin; out[i].ref_end = res[i].ref_end;
		out[i].query_begin = res[i].query_begin; out[i].query_end = res[i].query_end;
		// (the device tracebacks hold at most 62 runs and fail loudly beyond that; never hand back a truncated CIGAR)
		if (res[i].cigar_len > 256) return fail(E, FASIM_E_UNSUPPORTED, "alignment %d has %d CIGAR runs; fasim_alignment holds 256", i, res[i].cigar_len);
		out[i].cigar_len = res[i].cigar_len;
		if (out[i].cigar_len) memcpy(out[i].cigar, cigars.data() + res[i].cigar_off, sizeof(uint32_t) * out[i].cigar_len);
		if (res[i].failed) { out[i].sw_score = 0; out[i].cigar_len = -1; }     // the reference returns NULL here
	}
	return FASIM_OK;
}

int fasim_ssw_align(fasim_engine* E, const char* window, int32_t n, fasim_alignment* out)
{
	const int64_t off = 0;
	return fasim_align_batch(E, window, &off, &n, 1, out);
}

int fasim_encode_unit(const char* seg, int32_t n, int32_t enc, char* target, char* src)
{
	if (!seg || n < 0 || enc < 0 || enc >= 48 || !target || !src) return fail(nullptr, FASIM_E_ARG, "bad arguments");
	std::string t, s;
	encode_unit_host(seg, n, enc, t, s);
	memcpy(target, t.data(), n);
	memset(src, 0, n);
	memcpy(src, s.data(), s.size());
	return FASIM_OK;
}

int fasim_load_dna(fasim_engine* E, const char* dna, int64_t dna_len)
{
	if (!E) return fail(nullptr, FASIM_E_ARG, "null engine");
	if (!dna || dna_len <= 0) return fail(E, FASIM_E_ARG, "empty DNA");
	if (dna_len > 0x7fffffffll) return fail(E, FASIM_E_ARG, "one record is limited to 2^31-1 nt");
	HIPOK(hipSetDevice(E->device));
	E->dna_host.assign(dna, dna + dna_len);
	int rc = upload(E, E->dna_res, dna, (size_t)dna_len);
	if (rc) { E->dna_host.clear(); return rc; }
	return FASIM_OK;
}

int64_t fasim_segment_count(int64_t dna_len, const fasim_params* p)
{
	// cutSequence (fastsim.h:71-90): pos += cut - overlap while pos < size
	if (dna_len <= 0 || !p || p->cutLength - p->overlapLength <= 0) return 0;
	const int64_t step = p->cutLength - p->overlapLength;
	return (dna_len + step - 1) / step;
}

void fasim_result_free(fasim_result* r)
{
	if (!r) return;
	free(r->recs); free(r->pool); free(r);
}

void fasim_free(void* p) { free(p); }

void fasim_synth_dna(char* out, int64_t n, uint64_t seed)
{
	// splitmix64, 2 bits per base, 32 bases per word (tools/synth.py random_dna)
	uint64_t s = seed;
	for (int64_t i = 0; i < n; i += 32) {
		s += 0x9E3779B97F4A7C15ull;
		uint64_t z = s;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		z ^= z >> 31;
		for (int k = 0; k < 32 && i + k < n; k++) out[i + k] = "ACGT"[(z >> (2 * k)) & 3];
	}
}

int fasim_scan(fasim_engine* E, const char* dna, int64_t dna_len, int64_t seg_first, int64_t seg_count,
	const fasim_params* pp, fasim_result** out)
{
	int rc = need_query(E); if (rc) return rc;
	if (!out) return fail(E, FASIM_E_ARG, "bad arguments");
	return scan_core(E, nullptr, nullptr, 0, dna, dna_len, seg_first, seg_count, pp, out);
}

int fasim_scan_queries(fasim_engine* E, const char* const* rnas, const int32_t* rna_lens, int32_t nq, const char* dna,
	int64_t dna_len, int64_t seg_first, int64_t seg_count, const fasim_params* pp, fasim_result** outs)
{
	if (!E) return fail(nullptr, FASIM_E_ARG, "null engine");
	if (nq <= 0 || !rnas || !rna_lens || !outs) return fail(E, FASIM_E_ARG, "bad arguments");
	return scan_core(E, rnas, rna_lens, nq, dna, dna_len, seg_first, seg_count, pp, outs);
}

// in-place variant for a gather that already placed every shard's records and pool at their final positions
int fasim_rebase_offsets(fasim_triplex* recs, int64_t count, int64_t delta)
{
	if (count < 0 || (count > 0 && !recs)) return fail(nullptr, FASIM_E_ARG, "bad arguments");
	for (int64_t i = 0; i < count; i++) { recs[i].tfo_off += delta; recs[i].tts_off += delta; }
	return FASIM_OK;
}

// ---- the exchange step's host half: shard results -> one result, in the order given ------------------------------
int fasim_merge_results(const fasim_triplex* const* recs, const int64_t* counts, const char* const* pools,
	const int64_t* pool_lens, int32_t nparts, fasim_result** out)
{
	if (nparts < 0 || !out || (nparts > 0 && (!recs || !counts || !pools || !pool_lens))) return fail(nullptr, FASIM_E_ARG, "bad arguments");
	int64_t total = 0, pool_total = 0;
	std::vector<int64_t> rbase((size_t)nparts), pbase((size_t)nparts);
	for (int k = 0; k < nparts; k++) {
		if (counts[k] < 0 || pool_lens[k] < 0 || (counts[k] > 0 && (!recs[k] || !pools[k]))) return fail(nullptr, FASIM_E_ARG, "bad part %d", k);
		rbase[(size_t)k] = total; pbase[(size_t)k] = pool_total;
		total += counts[k]; pool_total += pool_lens[k];
	}
	fasim_result* R = (fasim_result*)calloc(1, sizeof(fasim_result));
	if (!R) return fail(nullptr, FASIM_E_NOMEM, "out of memory");
	R->recs = (fasim_triplex*)malloc(std::max<size_t>(1, (size_t)total) * sizeof(fasim_triplex));
	R->pool = (char*)malloc(std::max<size_t>(1, (size_t)pool_total));
	if (!R->recs || !R->pool) { fasim_result_free(R); return fail(nullptr, FASIM_E_NOMEM, "out of memory"); }
	R->count = total; R->pool_len = pool_total;
	// one pass per part: copy the records with their pool offsets rebased, copy the pool; parts are independent, so
	// large merges run one host thread per part
	auto one = [&](int k) {
		fasim_triplex* dst = R->recs + rbase[(size_t)k];
		const fasim_triplex* src = recs[k];
		const int64_t pb = pbase[(size_t)k];
		for (int64_t i = 0; i < counts[k]; i++) { fasim_triplex t = src[i]; t.tfo_off += pb; t.tts_off += pb; dst[i] = t; }
		if (pool_lens[k]) memcpy(R->pool + pb, pools[k], (size_t)pool_lens[k]);
	};
	if (nparts > 1 && total + pool_total / 64 > 100000) {
		std::vector<std::thread> th;
		for (int k = 0; k < nparts; k++) th.emplace_back(one, k);
		for (auto& t : th) t.join();
	} else for (int k = 0; k < nparts; k++) one(k);
	*out = R;
	return FASIM_OK;
}

static int records_to_list(const fasim_triplex* recs, int64_t count, const char* pool, int64_t pool_len, const fasim_params* p,
	int64_t start_genome, int32_t flags, std::vector<HostTriplex>& list)
{
	list.resize((size_t)count);
	for (int64_t i = 0; i < count; i++) {
		const fasim_triplex& r = recs[i];
		if (pool && (r.tfo_off < 0 || r.tfo_off >= pool_len || r.tts_off < 0 || r.tts_off >= pool_len)) return fail(nullptr, FASIM_E_ARG, "record %lld points outside the pool", (long long)i);
		HostTriplex& t = list[(size_t)i];
		t.stari = r.stari; t.endi = r.endi; t.starj = r.starj; t.endj = r.endj; t.strand = r.strand; t.reverse = r.reverse;
		t.rule = r.rule; t.nt = r.nt; t.score = r.score; t.identity = r.identity; t.tri_score = r.tri_score; t.seg = r.seg; t.enc = r.enc;
		t.tfo = pool ? pool + r.tfo_off : ""; t.tts = pool ? pool + r.tts_off : "";
		// records of a later FASTA record in the reference's accumulating reader carry their own genome start (main(),
		// Fasim-LongTarget.cpp:141-149 patches each record's triplexes with startGenomeTmp[i])
		if (r.genome_shift != 0) { t.genomestart = (long)r.starj + (long)start_genome + r.genome_shift - 1; t.genomeend = (long)r.endj + (long)start_genome + r.genome_shift - 1; t.genome_set = true; }
		if (t.nt > p->cLength && (t.stari + t.endi) / 2 - p->cDistance < 0 && !(flags & FASIM_TAIL_CLAMP_CLUSTER))
			return fail(nullptr, FASIM_E_UNSUPPORTED, "a triplex mid-point lies within -ds of the query start: the reference's clustering does not terminate for this input (FASIM_TAIL_CLAMP_CLUSTER / fasim --clamp-cluster gives a defined result)");
	}
	return FASIM_OK;
}

static int text_out(const std::string& s, char** text, int64_t* text_len)
{
	char* buf = (char*)malloc(s.size() + 1);
	if (!buf) return fail(nullptr, FASIM_E_NOMEM, "out of memory");
	memcpy(buf, s.data(), s.size()); buf[s.size()] = 0;
	*text = buf; *text_len = (int64_t)s.size();
	return FASIM_OK;
}

int fasim_tfosorted_ex(const fasim_triplex* recs, int64_t count, const char* pool, int64_t pool_len, const char* chr,
	int64_t start_genome, const fasim_params* p, int32_t flags, char** text, int64_t* text_len)
{
	if ((count > 0 && (!recs || !pool)) || !chr || !p || !text || !text_len || count < 0) return fail(nullptr, FASIM_E_ARG, "bad arguments");
	std::vector<HostTriplex> list;
	const int rc = records_to_list(recs, count, pool, pool_len, p, start_genome, flags, list);
	if (rc != FASIM_OK) return rc;
	return text_out(tfosorted_text(list, chr, (long)start_genome, *p), text, text_len);
}

int fasim_tfosorted(const fasim_triplex* recs, int64_t count, const char* pool, int64_t pool_len, const char* chr,
	int64_t start_genome, const fasim_params* p, char** text, int64_t* text_len)
{
	return fasim_tfosorted_ex(recs, count, pool, pool_len, chr, start_genome, p, 0, text, text_len);
}

int fasim_tfoclass_ex(const fasim_triplex* recs, int64_t count, int32_t level, const char* chr, int64_t start_genome,
	int64_t dna_len, const char* rna_name, const fasim_params* p, int32_t flags, char** text, int64_t* text_len)
{
	if ((count > 0 && !recs) || !chr || !rna_name || !p || !text || !text_len || count < 0 || level < 1 || level > 5)
		return fail(nullptr, FASIM_E_ARG, "bad arguments");
	std::vector<HostTriplex> list;
	const int rc = records_to_list(recs, count, nullptr, 0, p, start_genome, flags, list);
	if (rc != FASIM_OK) return rc;
	cluster_triplex(p->cDistance, p->cLength, list);
	return text_out(tfoclass_text(list, level, chr, (long)start_genome, (long)dna_len, rna_name, *p), text, text_len);
}

int fasim_tfoclass(const fasim_triplex* recs, int64_t count, int32_t level, const char* chr, int64_t start_genome,
	int64_t dna_len, const char* rna_name, const fasim_params* p, char** text, int64_t* text_len)
{
	return fasim_tfoclass_ex(recs, count, level, chr, start_genome, dna_len, rna_name, p, 0, text, text_len);
}

int fasim_tail_outputs(const fasim_triplex* recs, int64_t count, const char* pool, int64_t pool_len, const char* chr,
	int64_t start_genome, int64_t dna_len, const char* rna_name, const fasim_params* p, int32_t flags,
	char** tfosorted, int64_t* tfosorted_len, char** class1, int64_t* class1_len, char** class2, int64_t* class2_len)
{
	if ((count > 0 && (!recs || !pool)) || !chr || !rna_name || !p || count < 0 || !tfosorted || !tfosorted_len || !class1 || !class1_len ||
		!class2 || !class2_len) return fail(nullptr, FASIM_E_ARG, "bad arguments");
	*tfosorted = *class1 = *class2 = nullptr;
	std::vector<HostTriplex> list;
	int rc = records_to_list(recs, count, pool, pool_len, p, start_genome, flags, list);
	if (rc != FASIM_OK) return rc;
	// one clustering serves the three files (tfosorted_text clusters and orders the list; print_cluster only reads classes)
	rc = text_out(tfosorted_text(list, chr, (long)start_genome, *p), tfosorted, tfosorted_len);
	if (rc == FASIM_OK) rc = text_out(tfoclass_text(list, 1, chr, (long)start_genome, (long)dna_len, rna_name, *p), class1, class1_len);
	if (rc == FASIM_OK) rc = text_out(tfoclass_text(list, 2, chr, (long)start_genome, (long)dna_len, rna_name, *p), class2, class2_len);
	if (rc != FASIM_OK) { free(*tfosorted); free(*class1); free(*class2); *tfosorted = *class1 = *class2 = nullptr; }
	return rc;
}

void fasim_upper_case(char* seq, int64_t n)
{
	if (!seq) return;
	for (int64_t i = 0; i < n; i++) { const char c = seq[i]; if (c >= 'a' && c <= 'z') seq[i] = (char)(c - 32); }
}

} // extern "C"

