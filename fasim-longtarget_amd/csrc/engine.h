// fasim-longtarget_amd/csrc/engine.h -- internals of the host engine shared by its translation units (the C-ABI is
// include/fasim_hip.h; nothing here is exported).
//
// Data layout in HBM (one engine = one GPU):
//   dna        uint8[shard]                      the DNA shard, resident for the whole scan
//   tcodes     uint8[nunit][tstride]             target codes of every (segment x encoding) unit of the batch
//   colmax     uint8[nunit][tstride]             stage-2 column maxima (8-bit, as the reference's maxColumn)
//   q1/q2      uint8[m]                          query codes under the stage-1 / stage-2 alphabets
// Everything the kernels read is sized once per batch and reused; only small records cross PCIe.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <sched.h>
#include <malloc.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <string>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/fasim_hip.h"
#include "device_types.h"
#include "host_post.h"
#include "kernels.h"

using namespace fasim;



inline thread_local std::string g_last_error;     // per thread: the CLI formats and writes outputs on background threads

// device (re)allocations since the process started: hipFree / hipMalloc synchronise the whole device, so a buffer that grows in
// the middle of a scan stalls every batch in flight (FASIM_PROFILE=1 prints the count per scan)
inline std::atomic<long> g_dev_reallocs{ 0 };
struct DevBuf {
	void* p = nullptr; size_t cap = 0;
	hipError_t ensure(size_t bytes) {
		if (bytes <= cap) return hipSuccess;
		g_dev_reallocs.fetch_add(1);
		if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
		size_t want = bytes + bytes / 4 + 256;
		hipError_t e = hipMalloc(&p, want);
		if (e != hipSuccess) { p = nullptr; return e; }
		cap = want;
		return hipSuccess;
	}
	void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
	template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// FASIM_PROFILE=1: wall-clock accumulators of the host phases, printed to stderr at the end of fasim_scan
struct HostProf {
	static constexpr int N = 32;
	double t[N] = { 0 }; const char* name[N] = { nullptr };
	bool on = false;
	std::mutex mu;
	void add(int i, const char* nm, double dt) { if (on) { std::lock_guard<std::mutex> g(mu); t[i] += dt; name[i] = nm; } }
	void dump() { if (!on) return; for (int i = 0; i < N; i++) if (name[i]) fprintf(stderr, "[fasim prof] %-60s %12.6f\n", name[i], t[i]); }
	void reset() { for (int i = 0; i < N; i++) { t[i] = 0; name[i] = nullptr; } }
};
inline HostProf g_prof;
static inline double thread_cpu_s() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
struct CpuScope { int i; const char* nm; double t0; CpuScope(int i_, const char* n) : i(i_), nm(n), t0(g_prof.on ? thread_cpu_s() : 0.0) {} ~CpuScope() { if (g_prof.on) g_prof.add(i, nm, thread_cpu_s() - t0); } };
struct ProfScope { int i; const char* nm; double t0; ProfScope(int i_, const char* n) : i(i_), nm(n), t0(now_s()) {} ~ProfScope() { g_prof.add(i, nm, now_s() - t0); } };


struct fasim_engine {
	int device = 0;
	hipStream_t st = nullptr;
	std::string err;
	std::string rna;
	int m = 0;
	int snap_units = 0, snap_per_unit = 0;      // pipeline snapshots of the last main scan pass (units covered, snapshots per unit)
	ScoreLut lut1, lut2;
	DevBuf q1, q2, enc_lut, counter, dna, seg_start, seg_len, enc_ids, tcodes, colmax, probs, max_out, unit_len,
		stage1, hits, hits_total, hit_off, hit_cnt, thr, ends, bprobs, bout, scratch, colmax16, unit_ids, flags, stage1_in, hits2,
		fprobs, ftasks, fstream, fout, aout, cigpool, cigcount, forder, scratch2, boundary, fboundary, unit_hz,
		unit_first, hz_cols, hz_plan, hz_base, hz_items, snap, hz_state, hz_rows, hz_chunk, hz_src, hz_zero,   // chunked hazard re-run
		qsim, sim_min, sim_row, sim_ev, sim_cnt, sim_nodes,      // -F: query codes of the SIM alphabet, thresholds, strip row buffer, events, counters
		sim_req, sim_pairs, sim_used, sim_rounds, sim_col, sim_rowst, sim_state, sim_usedc, sim_debug, sim_nodes_in, sim_nodes_out, sim_out;   // -F re-sweeps: per-round requests, used pairs, DP state per column / row
	bool align_v1 = false;        // FASIM_ALIGN_V1=1: force the stripe-faithful kernels for stage 3
	std::vector<fasim_engine*> workers;   // extra engines on the same device: batches in flight concurrently
	// Gate for the two GPU-filling kernels (k_scan, k_align_fwd).  Without it the workers fall into lock step: all of them
	// launch a heavy kernel at once, the kernels share the GPU and end together, and then nothing heavy runs while all
	// workers do their latency-bound tail kernels and host work.  With at most `cap` heavy kernels in flight each one
	// runs at full speed and the workers stay staggered.
	struct HeavyGate { std::mutex m; std::condition_variable cv; int in_flight = 0; int cap = 3; };
	HeavyGate own_gate;
	HeavyGate* gate = nullptr;            // shared by the workers of one fasim_scan (points at the parent's own_gate)
	int host_threads_total = 1;
	int host_threads_share_total = 1;            // (workers) the scan's total, for the share of a worker near the end of a scan
	bool host_threads_explicit = false;          // FASIM_HOST_THREADS / option host_threads given: -F keeps to it too
	int scan_workers = 1;                        // worker engines of the scan in progress (they share the HBM)
	std::atomic<int>* sim_in_flight = nullptr;      // (set for the duration of a scan) -F: units in re-sweep launches right now, over all workers
	std::atomic<int>* active_workers = nullptr;  // (set for the duration of a scan) workers that still have batches: the host threads of
	                                             // those that have run out go to the bursts of the others
	int sim_threads = 1;                         // -F: host threads of this worker for the finish half (all cores shared by the batches in flight)
	int opt_workers = 0, opt_seg_batch = 0;      // fasim_set_option overrides (0 = default / environment)
	int opt_taper = -1, opt_gate = -1;           // (-1 = default / environment)
	int hz_chunks = -1, hz_snap = -1, hz_target = 0, hz_hot_w = 0;   // chunked hazard re-run: on/off, snapshots on/off (-1 = default / environment), chunk cost target, hot-column weight (0 = default)
	bool query_acgt = true;       // query holds only A,C,G,T: stage-1 and stage-2 scoring coincide on N-free segments
	bool scan_v1 = false;         // FASIM_SCAN_V1=1: force the stripe-faithful kernels for stages 1 and 2
	int host_threads = 1;
	// resident DNA record (fasim_load_dna)
	std::string dna_host;
	DevBuf dna_res;
	// streaming ingest (fasim_scan with a host buffer): pinned staging buffer of this worker's current batch slice
	void* pin_dna = nullptr; size_t pin_cap = 0;
	void* pin_sim = nullptr; size_t pin_sim_cap = 0;     // -F: node lists of a slice, both ways every launch
	// HIP-event timing of kernel launches on `st`
	struct Timed { hipEvent_t a, b; int family; };
	std::vector<Timed> timed;
	std::vector<hipEvent_t> ev_pool;
	double kernel_ms[FASIM_KERNEL_FAMILIES] = { 0 };
	int64_t kernel_launches[FASIM_KERNEL_FAMILIES] = { 0 };
	// banded stage 3 (band.hip): block maxima left by the last main k_scan pass of this engine, lists and column streams of the
	// tries selected per band class
	DevBuf ublk, btarget, bidx, bcounts, blist[3], bslots[3], bprev, lane_ub, fzones, fubslot, bdec, btab;
	int ublk_units = 0, ublk_blocks = 0;         // units covered by `ublk` (0: none), blocks per (unit, tile)
	int opt_band = -1;                           // option "band": 0 off, 1 on (-1 = default / environment FASIM_BAND)
	int opt_numa = 1;                            // option "numa_affinity": pin the scan's host threads to the GPU's NUMA node (no-op on one node)
};



inline int fail(fasim_engine* e, int code, const char* fmt, ...)
{
	char buf[1024];
	va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
	g_last_error = buf;
	if (e) e->err = buf;
	return code;
}

#define HIPOK(call) do { hipError_t _e = (call); if (_e != hipSuccess) return fail(E, FASIM_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); } while (0)

// stage-2/3 alphabet (ssw_cpp.cpp:13-26): A,a,U,u -> 0 ; C,c -> 1 ; G,g -> 2 ; T,t -> 3 ; else 4
inline uint8_t code2(char c) { switch (c) { case 'A': case 'a': case 'U': case 'u': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; } }
// SIM (-F) alphabet: the score table of sim.h:464-468 knows ACGT only; every other letter is a mismatch with everything
inline uint8_t sim_code(char c) { switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; } }
// stage-1 alphabet (stats.h:201-228, 306-334): U == T, everything outside ACGTU is N
inline uint8_t code1(char c) { switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': case 'U': case 'u': return 3; default: return 4; } }

inline ScoreLut make_lut(bool stage1)
{
	// row t, entry q (4 bits): score(t,q) + BIAS ; entry 5 = pad row = score 0
	ScoreLut L;
	for (int t = 0; t < 5; t++) {
		uint32_t w = 0;
		for (int q = 0; q < 5; q++) {
			int s;
			if (stage1) s = (t == 4 || q == 4) ? -1 : (t == q ? 5 : -4);      // npam: N row all -1 (stats.h:227-228)
			else s = (t == q && t < 4) ? 5 : -4;                              // ssw_cpp.cpp:28-53
			w |= (uint32_t)(s + BIAS) << (4 * q);
		}
		w |= (uint32_t)BIAS << 20;
		L.row[t] = w;
	}
	return L;
}

// ---- HIP-event timing ----------------------------------------------------------------------------
inline hipEvent_t get_event(fasim_engine* E)
{
	if (!E->ev_pool.empty()) { hipEvent_t e = E->ev_pool.back(); E->ev_pool.pop_back(); return e; }
	hipEvent_t e = nullptr;
	if (hipEventCreate(&e) != hipSuccess) return nullptr;
	return e;
}
struct TimedScope {
	fasim_engine* E; hipEvent_t a = nullptr, b = nullptr; int family;
	hipStream_t s;
	TimedScope(fasim_engine* e, int fam, hipStream_t stream = nullptr) : E(e), family(fam), s(stream ? stream : e->st) { a = get_event(E); b = get_event(E); if (a) (void)hipEventRecord(a, s); }
	~TimedScope() { if (a && b) { (void)hipEventRecord(b, s); E->timed.push_back({ a, b, family }); } }
};
// call after a stream synchronisation
inline void drain_timed(fasim_engine* E)
{
	for (auto& t : E->timed) {
		float ms = 0.0f;
		if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) { E->kernel_ms[t.family] += ms; E->kernel_launches[t.family]++; }
		E->ev_pool.push_back(t.a); E->ev_pool.push_back(t.b);
	}
	E->timed.clear();
}

// ---- a batch of units whose target codes are resident on the device ------------------------------
struct UnitBatch {
	int nunit = 0;
	int tstride = 0;
	std::vector<int> unit_len;      // columns per unit
};

// H2D copy without the trailing synchronisation: the caller keeps `src` alive until its next stream synchronisation
inline int upload_async(fasim_engine* E, DevBuf& b, const void* src, size_t bytes)
{
	HIPOK(b.ensure(bytes ? bytes : 1));
	if (bytes) HIPOK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, E->st));
	return FASIM_OK;
}

inline int upload(fasim_engine* E, DevBuf& b, const void* src, size_t bytes)
{
	HIPOK(b.ensure(bytes ? bytes : 1));
	// sources are short-lived pageable host vectors: make the copy complete before returning
	if (bytes) { HIPOK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, E->st)); HIPOK(hipStreamSynchronize(E->st)); }
	return FASIM_OK;
}

inline std::vector<StripedProb> whole_unit_probs(const UnitBatch& B, int m, const std::vector<int>* subset)
{
	std::vector<StripedProb> v;
	const int n = subset ? (int)subset->size() : B.nunit;
	v.reserve(n);
	for (int k = 0; k < n; k++) {
		const int u = subset ? (*subset)[k] : k;
		StripedProb p; p.tbase = (int64_t)u * B.tstride; p.t0 = 0; p.ref_len = B.unit_len[u]; p.q_len = m; p.unit = u; p.aux = 0; p.pad = 0;
		v.push_back(p);
	}
	return v;
}


// ---- stages 1+2 through the fused systolic kernel (scan.hip) -----------------------------------------
struct ScanOut {
	std::vector<int32_t> stage1, thr, hit_off, hit_cnt, flags;
	std::vector<uint32_t> hits;
};


struct GateScope {
	fasim_engine::HeavyGate* g;
	explicit GateScope(fasim_engine* E) : g(E->gate)
	{
		if (!g) return;
		std::unique_lock<std::mutex> lk(g->m);
		g->cv.wait(lk, [&] { return g->in_flight < g->cap; });
		g->in_flight++;
	}
	void release()
	{
		if (!g) return;
		{ std::lock_guard<std::mutex> lk(g->m); g->in_flight--; }
		g->cv.notify_one();
		g = nullptr;
	}
	~GateScope() { release(); }
};

// cores this process may really use: scheduler affinity, capped by the cgroup CPU quota when there is one
// (std::thread::hardware_concurrency() reports the whole host on a shared GPU node)
inline int usable_cores()
{
	int n = (int)std::max(1u, std::thread::hardware_concurrency());
	cpu_set_t set;
	if (sched_getaffinity(0, sizeof set, &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0) n = std::min(n, a); }
	if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
		char q[64] = { 0 }; long long period = 0;
		if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) { const long long c = atoll(q) / period; if (c >= 1) n = (int)std::min<long long>(n, c); }
		fclose(f);
	}
	return std::max(1, n);
}

// Banded stage 3 (band.hip): classes usable for the current query; 0 = off (FASIM_BAND=0 / option band = 0, stripe-faithful
// modes, queries the band kernel does not hold)
// option band / FASIM_BAND: 0 off, 1 on (default), 2 = bands from k_scan's block maxima only, no reverse passes (for measurements)
inline int band_mode(const fasim_engine* E)
{
	static const int env = [] { const char* e = getenv("FASIM_BAND"); return e ? atoi(e) : 1; }();
	return E->opt_band >= 0 ? E->opt_band : env;
}
inline int band_mask(const fasim_engine* E)
{
	if (!band_mode(E) || E->align_v1 || E->scan_v1) return 0;
	return band_classes(E->m);
}

// CPUs of the NUMA node the GPU hangs on (local_cpulist of its PCI device), intersected with what this thread may use.  false:
// unknown, or no restriction (single-node machine): nothing to pin.
inline bool gpu_local_cpus(int device, cpu_set_t* out)
{
	char bus[64] = { 0 };
	if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return false; }
	for (char* c = bus; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
	char path[160]; snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bus);
	FILE* f = fopen(path, "r");
	if (!f) return false;
	char list[1024] = { 0 };
	const bool ok = fgets(list, sizeof list, f) != nullptr;
	fclose(f);
	if (!ok) return false;
	cpu_set_t local; CPU_ZERO(&local);
	for (const char* p = list; *p && *p != '\n'; ) {
		char* e = nullptr;
		const long a = strtol(p, &e, 10);
		if (e == p) break;
		long b = a; p = e;
		if (*p == '-') { b = strtol(p + 1, &e, 10); p = e; }
		for (long k = a; k <= b && k < CPU_SETSIZE; k++) if (k >= 0) CPU_SET((int)k, &local);
		if (*p == ',') p++;
	}
	cpu_set_t allowed;
	if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
	CPU_AND(out, &local, &allowed);
	const int n = CPU_COUNT(out);
	return n > 0 && n < CPU_COUNT(&allowed);
}
// Pins the calling thread (and the worker / host threads it starts, which inherit the mask) to the GPU's NUMA node for the
// duration of a scan: on an 8-GPU node every rank / every --devices engine then keeps its host side next to its own GPU
// instead of wandering over both sockets.  Option numa_affinity = 0 leaves the affinity alone.
struct AffinityScope {
	cpu_set_t saved; bool active = false;
	AffinityScope(int device, bool enabled) {
		cpu_set_t local;
		if (!enabled || sched_getaffinity(0, sizeof saved, &saved) != 0 || !gpu_local_cpus(device, &local)) return;
		active = sched_setaffinity(0, sizeof local, &local) == 0;
	}
	~AffinityScope() { if (active) (void)sched_setaffinity(0, sizeof saved, &saved); }
};

// FASIM_HAZARD_CHUNKS=0: whole-unit re-run of the hazard units (the round-1 path); FASIM_HAZARD_SNAP=0: the checkpoint pass
// runs every hazard unit from column 0 instead of from the main pass's pipeline snapshots (both for measurements)
// (options hazard_chunks / hazard_snapshots override the environment)
inline bool hazard_chunks_enabled(const fasim_engine* E) { static const bool v = [] { const char* e = getenv("FASIM_HAZARD_CHUNKS"); return e ? atoi(e) != 0 : true; }(); return E->hz_chunks >= 0 ? E->hz_chunks != 0 : v; }
// Snapshots are OFF by default since round 3: they cost 67 KB of HBM writes per unit (2.3 x the algorithmic traffic of k_scan, 1.4 GB
// per batch in flight) for the 0.8 % of the units that become hazard units, and buy 7 ms of a batch's latency that ten batches in flight
// hide anyway (2.18 vs 2.21 s per 50 Mb step, inside the run-to-run noise: profiles/r03_ab_snapshots.txt).
inline bool hazard_snapshots_enabled(const fasim_engine* E) { static const bool v = [] { const char* e = getenv("FASIM_HAZARD_SNAP"); return e ? atoi(e) != 0 : false; }(); return E->hz_snap >= 0 ? E->hz_snap != 0 : v; }


// ---- the batched body of LongTarget() ---------------------------------------------------------------
// One batch of segments [b0, b1) on one worker engine (own stream and buffers): encode, scan, candidates, window
// alignments, triplex records.  Several batches run concurrently on different workers (fasim_scan below).
// What one batch leaves after its scan phase (stages 1+2) and what its stage 3 needs: host-side hit lists and segment
// tables, plus a pointer to the target codes that stay resident on the owner engine.  Stage 3 is separable by unit range
// (stage3_range), so near the end of a scan a batch publishes its stage 3 as sub-tasks that idle workers take over.
struct BatchCtx {
	UnitBatch B;
	int tstride = 0, nenc = 0, nseg = 0;
	int64_t step = 0;
	std::vector<int32_t> sstart, slen; std::vector<int64_t> sidx;
	std::vector<int32_t> hoff, hcnt, thr; std::vector<uint32_t> hits;
	std::vector<char> seg_acgtn;
	const char* dna = nullptr; const fasim_params* p = nullptr; const std::vector<int>* encs = nullptr;
	std::vector<std::vector<HostTriplex>> per_unit;     // [unit]: records of the unit after fastSIM's own filter
	bool stage3_done = false;                           // -F: the whole batch was finished in the scan phase
};

// ---- internals shared by the engine's translation units (engine.cpp: C-ABI; engine_stage2.cpp: stages 1+2; engine_stage3.cpp:
//      stage 3; engine_scan.cpp: batches, workers, result packing) ------------------------------------------------------------
struct WindowProb { int unit, t0, len; };
inline const uint8_t* tcv(const fasim_engine* E) { return E->tcodes.as<uint8_t>(); }
// input of the reverse pass (band.hip, align.hip): per window the lengths of the candidate's next three tries (zone tags of the
// reversed stream) and the candidate's slot in E->lane_ub
struct FwdZones { std::vector<uint32_t> zones; std::vector<int32_t> slot; };

int run_striped(fasim_engine* E, StripedMode mode, bool word, const std::vector<StripedProb>& probs, bool stage1, const uint8_t* tcodes, int max_qlen);
int run_stage1(fasim_engine* E, const UnitBatch& B, std::vector<int>& score, int64_t* word_reruns);
int run_stage2(fasim_engine* E, const UnitBatch& B);
int run_scan_v2(fasim_engine* E, const UnitBatch& B, const std::vector<char>& unit_needs_stage1, ScanOut& out, fasim_scan_stats* st);
int load_raw_targets(fasim_engine* E, const char* targets, const int64_t* offsets, const int32_t* lens, int nprob, bool stage1, UnitBatch& B);
int need_query(fasim_engine* E);
int run_align(fasim_engine* E, const UnitBatch& B, const std::vector<WindowProb>& W, std::vector<AlignResult>& out, std::vector<uint32_t>& cigars, fasim_scan_stats* stats);
int run_align_v2(fasim_engine* E, const UnitBatch& B, const std::vector<WindowProb>& W, std::vector<AlignResult>& out, std::vector<uint32_t>& cigars, fasim_scan_stats* stats);
int stage3_range(fasim_engine* E, BatchCtx& C, int ua, int ub, fasim_scan_stats& st);
int scan_batch(fasim_engine* E, const char* dna, int64_t dna_len, const uint8_t* dna_dev, int64_t shard_lo, int64_t b0, int64_t b1,
	const fasim_params& p, const std::vector<int>& encs, int tstride, BatchCtx& C, fasim_scan_stats& st);
int sim_forward_units(fasim_engine* E, const uint8_t* tcodes_dev, int tstride, const int32_t* unit_len_dev, const int32_t* unit_len_host,
	int first, int nunit, const int64_t* mins, std::atomic<int>* ready, std::vector<std::vector<fasim_sim_node>>& lists);
int sim_resweep_rounds(fasim_engine* E, const uint8_t* tcodes_dev, int tstride, const int32_t* unit_len_dev, const int32_t* unit_len_host,
	int first, int cnt, SimUnit* const* units, int nthreads);
int pack_result(fasim_engine* E, std::vector<HostTriplex>& all, const fasim_scan_stats& st, fasim_result** out);
int scan_core(fasim_engine* E, const char* const* rnas, const int32_t* rna_lens, int nq, const char* dna, int64_t dna_len,
	int64_t seg_first, int64_t seg_count, const fasim_params* pp, fasim_result** outs);
