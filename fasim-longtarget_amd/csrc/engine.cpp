// fasim-longtarget_amd/csrc/engine.cpp -- host engine + C-ABI (include/fasim_hip.h) of libfasim_hip.so.
//
// Data layout in HBM (one engine = one GPU):
//   dna        uint8[shard]                      the DNA shard, resident for the whole scan
//   tcodes     uint8[nunit][tstride]             target codes of every (segment x encoding) unit of the batch
//   colmax     uint8[nunit][tstride]             stage-2 column maxima (8-bit, as the reference's maxColumn)
//   q1/q2      uint8[m]                          query codes under the stage-1 / stage-2 alphabets
// Everything the kernels read is sized once per batch and reused; only small records cross PCIe.
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

namespace {

thread_local std::string g_last_error;     // per thread: the CLI formats and writes outputs on background threads

// device (re)allocations since the process started: hipFree / hipMalloc synchronise the whole device, so a buffer that grows in
// the middle of a scan stalls every batch in flight (FASIM_PROFILE=1 prints the count per scan)
static std::atomic<long> g_dev_reallocs{ 0 };
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

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

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
HostProf g_prof;
static inline double thread_cpu_s() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
struct CpuScope { int i; const char* nm; double t0; CpuScope(int i_, const char* n) : i(i_), nm(n), t0(g_prof.on ? thread_cpu_s() : 0.0) {} ~CpuScope() { if (g_prof.on) g_prof.add(i, nm, thread_cpu_s() - t0); } };
struct ProfScope { int i; const char* nm; double t0; ProfScope(int i_, const char* n) : i(i_), nm(n), t0(now_s()) {} ~ProfScope() { g_prof.add(i, nm, now_s() - t0); } };

} // namespace

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
		qsim, sim_min, sim_row, sim_ev, sim_cnt, sim_nodes;      // -F: query codes of the SIM alphabet, thresholds, strip row buffer, events, counters
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

namespace {

int fail(fasim_engine* e, int code, const char* fmt, ...)
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

ScoreLut make_lut(bool stage1)
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
hipEvent_t get_event(fasim_engine* E)
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
void drain_timed(fasim_engine* E)
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
int upload_async(fasim_engine* E, DevBuf& b, const void* src, size_t bytes)
{
	HIPOK(b.ensure(bytes ? bytes : 1));
	if (bytes) HIPOK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, E->st));
	return FASIM_OK;
}

int upload(fasim_engine* E, DevBuf& b, const void* src, size_t bytes)
{
	HIPOK(b.ensure(bytes ? bytes : 1));
	// sources are short-lived pageable host vectors: make the copy complete before returning
	if (bytes) { HIPOK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, E->st)); HIPOK(hipStreamSynchronize(E->st)); }
	return FASIM_OK;
}

std::vector<StripedProb> whole_unit_probs(const UnitBatch& B, int m, const std::vector<int>* subset)
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

// ---- stages 1+2 through the fused systolic kernel (scan.hip) -----------------------------------------
struct ScanOut {
	std::vector<int32_t> stage1, thr, hit_off, hit_cnt, flags;
	std::vector<uint32_t> hits;
};

void fill_scores(int8_t* sc, bool stage1)
{
	for (int t = 0; t < 5; t++) for (int q = 0; q < 5; q++)
		sc[t * 5 + q] = (int8_t)(stage1 ? ((t == 4 || q == 4) ? -1 : (t == q ? 5 : -4)) : ((t == q && t < 4) ? 5 : -4));
}

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
static int usable_cores()
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
static int band_mode(const fasim_engine* E)
{
	static const int env = [] { const char* e = getenv("FASIM_BAND"); return e ? atoi(e) : 1; }();
	return E->opt_band >= 0 ? E->opt_band : env;
}
static int band_mask(const fasim_engine* E)
{
	if (!band_mode(E) || E->align_v1 || E->scan_v1) return 0;
	return band_classes(E->m);
}

// CPUs of the NUMA node the GPU hangs on (local_cpulist of its PCI device), intersected with what this thread may use.  false:
// unknown, or no restriction (single-node machine): nothing to pin.
static bool gpu_local_cpus(int device, cpu_set_t* out)
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
static bool hazard_chunks_enabled(const fasim_engine* E) { static const bool v = [] { const char* e = getenv("FASIM_HAZARD_CHUNKS"); return e ? atoi(e) != 0 : true; }(); return E->hz_chunks >= 0 ? E->hz_chunks != 0 : v; }
// Snapshots are OFF by default since round 3: they cost 67 KB of HBM writes per unit (2.3 x the algorithmic traffic of k_scan, 1.4 GB
// per batch in flight) for the 0.8 % of the units that become hazard units, and buy 7 ms of a batch's latency that ten batches in flight
// hide anyway (2.18 vs 2.21 s per 50 Mb step, inside the run-to-run noise: profiles/r03_ab_snapshots.txt).
static bool hazard_snapshots_enabled(const fasim_engine* E) { static const bool v = [] { const char* e = getenv("FASIM_HAZARD_SNAP"); return e ? atoi(e) != 0 : false; }(); return E->hz_snap >= 0 ? E->hz_snap != 0 : v; }

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

struct WindowProb { int unit, t0, len; };
static inline const uint8_t* tcv(const fasim_engine* E) { return E->tcodes.as<uint8_t>(); }

int run_finish(fasim_engine* E, const UnitBatch& B, const std::vector<WindowProb>& W, const std::vector<FwdOut>& fo,
	std::vector<AlignResult>& out, std::vector<uint32_t>& cigars, std::vector<char>& status);
bool align_v2_fits(const fasim_engine* E, const std::vector<WindowProb>& W);
// input of the reverse pass (band.hip, align.hip): per window the lengths of the candidate's next three tries (zone tags of the
// reversed stream) and the candidate's slot in E->lane_ub
struct FwdZones { std::vector<uint32_t> zones; std::vector<int32_t> slot; };
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
		// alignments whose band / direction matrix did not fit the LDS kernel: same algorithm on global scratch, first
		// with 16 KB per alignment, then (wide bands after several doublings: gapped alignments) with 1 MB
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

} // namespace

// =====================================================================================================
// C-ABI
// =====================================================================================================
static int pack_result(fasim_engine* E, std::vector<HostTriplex>& all, const fasim_scan_stats& st, fasim_result** out);

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

// ---- row f3: forward sweep of classic SIM ---------------------------------------------------------------------
// addnode() (sim.h:99-148) over the events of one unit in row-major order: a known start point is updated (strictly larger
// score moves the end point; the bounding box grows), a new one is appended or, with K nodes present, overwrites the first
// node of lowest score whatever its own score is.
// Forward sweep + node list (k_sim_forward: the list is kept and replayed inside the wave, sim.hip) for units
// [first, first + nrun) of a resident code buffer.  min_scores[u] belongs to unit first + u.  lists[u] receives the node list.
// Units run in slices of <= 1024 (64 row segments of one unit's length each = 5 MB of scratch per unit); *ready (if given) is
// the number of leading units whose lists are complete, so that the host half can start on a slice while the next one runs.
// `lists` must have been sized to nrun by the caller.
static int sim_forward_units(fasim_engine* E, const uint8_t* tcodes_dev, int tstride, const int32_t* unit_len_dev, const int32_t* unit_len_host,
	int first, int nrun, const int64_t* min_scores, std::atomic<int>* ready, std::vector<std::vector<fasim_sim_node>>& lists)
{
	static_assert(sizeof(SimNodeDev) == sizeof(fasim_sim_node) && SIM_K == FASIM_SIM_K, "node layout");
	if (nrun <= 0) return FASIM_OK;
	int maxlen = 1;
	for (int u = 0; u < nrun; u++) maxlen = std::max(maxlen, unit_len_host[first + u]);
	if (E->m > 65535 || maxlen > 65535) return fail(E, FASIM_E_UNSUPPORTED, "the SIM forward sweep holds start points in 16 bits: query %d / target %d nt is too long", E->m, maxlen);
	const int64_t row_stride = (maxlen + 2 + 15) & ~15;
	const uint32_t cap = (uint32_t)((maxlen + 15) & ~15);
	DevBuf& d_min = E->sim_min; DevBuf& d_row = E->sim_row; DevBuf& d_ev = E->sim_ev; DevBuf& d_cnt = E->sim_cnt; DevBuf& d_nodes = E->sim_nodes;
	int rc = upload(E, d_min, min_scores, sizeof(int64_t) * nrun); if (rc) return rc;
	const int per_slice = 1024;         // waves in flight: the sweep of one unit takes ~1.5 s of one wave, the chip holds thousands
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
			//      (k_sim_forward + node-list replay), traceback / re-sweeps / triplex records on the host threads (host_sim.cpp)
			t0 = now_s();
			std::vector<int64_t> mins((size_t)B.nunit);
			for (int u = 0; u < B.nunit; u++) mins[(size_t)u] = thr[(size_t)u];
			std::vector<std::vector<fasim_sim_node>> lists((size_t)B.nunit);
			std::vector<std::vector<HostTriplex>>& per_unit = C.per_unit;
			per_unit.assign((size_t)B.nunit, std::vector<HostTriplex>());
			// the host threads finish the units of a slice while the GPU sweeps the next one
			std::atomic<int> next(0), ready(0);
			std::atomic<bool> abort(false);
			auto work = [&]() {
				std::string target, src;
				for (;;) {
					const int u = next.fetch_add(1);
					if (u >= B.nunit) break;
					while (u >= ready.load(std::memory_order_acquire) && !abort.load()) std::this_thread::sleep_for(std::chrono::microseconds(200));
					if (abort.load()) break;
					const int s = u / nenc, enc = encs[(size_t)(u % nenc)];
					encode_unit_host(dna + sidx[(size_t)s] * step, slen[(size_t)s], enc, target, src);
					sim_finish_unit(E->rna, target, src, (long)(sidx[(size_t)s] * step), thr[(size_t)u], enc, p, lists[(size_t)u], per_unit[(size_t)u]);
					for (HostTriplex& t : per_unit[(size_t)u]) { t.seg = (int)sidx[(size_t)s]; t.enc = enc; }
				}
			};
			const int nt = std::max(1, std::min(E->sim_threads, B.nunit));
			std::vector<std::thread> th;
			for (int k = 0; k < nt; k++) th.emplace_back(work);
			rc = sim_forward_units(E, E->tcodes.as<uint8_t>(), tstride, E->unit_len.as<int32_t>(), B.unit_len.data(), 0, B.nunit, mins.data(), &ready, lists);
			if (rc) abort.store(true);
			for (auto& t : th) t.join();
			if (rc) return rc;
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
static int pack_result(fasim_engine* E, std::vector<HostTriplex>& all, const fasim_scan_stats& st, fasim_result** out)
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
static int scan_core(fasim_engine* E, const char* const* rnas, const int32_t* rna_lens, int nq, const char* dna, int64_t dna_len,
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
		std::atomic<int> active_workers(nworkers);
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
			w->host_threads_share_total = E->host_threads_total; w->active_workers = &active_workers;
			{
				// -F: the finish half of classic SIM is ~40 ms of host work per unit and nothing else needs the cores meanwhile
				const int all = E->host_threads_explicit ? E->host_threads_total : usable_cores();
				w->sim_threads = std::max(1, all / (int)std::max<size_t>(1, std::min<size_t>((size_t)nworkers, items.size())));
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
		for (fasim_engine* w : ws) w->active_workers = nullptr;
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
		if (r.genome_shift != 0) { t.genomestart = (long)r.starj + (long)start_genome + r.genome_shift - 1; t.genomeend = (long)r.endj + (long)start_genome + r.genome_shift - 1; }
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
