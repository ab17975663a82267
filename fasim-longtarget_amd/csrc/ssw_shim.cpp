// fasim-longtarget_amd/csrc/ssw_shim.cpp -- the reference's ssw.h ABI (include/ssw.h) on top of the HIP engine.
//
// ssw_init / init_destroy / ssw_pre_align / ssw_align / align_destroy with the signatures, the s_align layout and the
// calloc/free ownership of /root/reference/ssw.h:48-58,78-142, so that the reference's unchanged C++ wrapper
// (ssw_cpp.cpp) links against libfasim_hip.so instead of sswNew.cpp.  Every call is one small job on a process-wide
// engine (device $FASIM_DEVICE, default 0); calls are serialised.  There is no CPU path in here: without a HIP device the
// calls fail the way the reference's do (NULL + a message on stderr).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fasim_hip.h"
#include "../../include/ssw.h"

struct _profile {
	const int8_t* read;          // borrowed (sswNew.cpp:1290)
	const int8_t* mat;           // borrowed (sswNew.cpp:1291)
	int32_t readLen, n;
	int8_t score_size;
	uint8_t bias;
	char* letters;               // owned: the read as letters, the form the engine takes its query in
};

namespace {

std::mutex g_mu;
fasim_engine* g_engine = nullptr;
std::string g_query;             // query currently set on g_engine

// 0..3 -> ACGT, everything else -> N: the inverse of the wrapper's base translation (ssw_cpp.cpp:13-26)
inline char letter_of(int8_t code) { return (code >= 0 && code < 4) ? "ACGT"[code] : 'N'; }

fasim_engine* engine_locked()
{
	if (g_engine) return g_engine;
	const char* d = getenv("FASIM_DEVICE");
	// (a drop-in inside somebody else's process: no mallopt, no device flags, no environment changes)
	if (fasim_engine_create_ex(d ? atoi(d) : 0, FASIM_CREATE_NO_PROCESS_TUNING, &g_engine) != FASIM_OK) {
		fprintf(stderr, "ssw (fasim HIP shim): %s\n", fasim_last_error(nullptr));
		g_engine = nullptr;
	}
	return g_engine;
}

// the one scoring the engine implements = the Aligner's (ssw_cpp.cpp:28-53, 238-250)
bool scoring_supported(const s_profile* p, uint8_t gapO, uint8_t gapE)
{
	if (!p || !p->mat || p->n != 5 || gapO != 16 || gapE != 4) return false;
	for (int i = 0; i < 5; i++) for (int j = 0; j < 5; j++)
		if (p->mat[i * 5 + j] != ((i == j && i < 4) ? 5 : -4)) return false;
	return true;
}

bool set_query_locked(fasim_engine* e, const s_profile* p)
{
	if (g_query.size() == (size_t)p->readLen && memcmp(g_query.data(), p->letters, (size_t)p->readLen) == 0) return true;
	if (fasim_set_query(e, p->letters, p->readLen) != FASIM_OK) { fprintf(stderr, "ssw (fasim HIP shim): %s\n", fasim_last_error(e)); g_query.clear(); return false; }
	g_query.assign(p->letters, p->letters + p->readLen);
	return true;
}

std::string ref_letters(const int8_t* ref, int32_t refLen)
{
	std::string t((size_t)refLen, 'N');
	for (int32_t i = 0; i < refLen; i++) t[(size_t)i] = letter_of(ref[i]);
	return t;
}

} // namespace

extern "C" {

// ssw.h:27 / sswNew.cpp:140: op code of a CIGAR letter ('M' 0, 'I' 1, 'D' 2, 'N' 3, 'S' 4, 'H' 5, 'P' 6, '=' 7, 'X' 8)
const uint8_t encoded_ops[128] = {
	0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
	0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
	0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
	0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 7 /* = */, 0, 0,
	0, 0, 0, 0, 2 /* D */, 0, 0, 0, 5 /* H */, 1 /* I */, 0, 0, 0, 0 /* M */, 3 /* N */, 0,
	6 /* P */, 0, 0, 4 /* S */, 0, 0, 0, 0, 8 /* X */, 0, 0, 0, 0, 0, 0, 0,
	0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
	0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
};

s_profile* ssw_init(const int8_t* read, const int32_t readLen, const int8_t* mat, const int32_t n, const int8_t score_size)
{
	s_profile* p = (s_profile*)calloc(1, sizeof(struct _profile));
	if (!p) return nullptr;
	p->read = read; p->mat = mat; p->readLen = readLen; p->n = n; p->score_size = score_size;
	if (mat && n > 0 && (score_size == 0 || score_size == 2)) {
		int bias = 0;                                  // sswNew.cpp:1283-1286
		for (int i = 0; i < n * n; i++) if (mat[i] < bias) bias = mat[i];
		p->bias = (uint8_t)abs(bias);
	}
	p->letters = (char*)malloc(readLen > 0 ? (size_t)readLen : 1);
	if (!p->letters) { free(p); return nullptr; }
	for (int32_t i = 0; i < readLen; i++) p->letters[i] = letter_of(read[i]);
	return p;
}

void init_destroy(s_profile* p)
{
	if (!p) return;
	free(p->letters);
	free(p);
}

int* ssw_pre_align(const s_profile* prof, const int8_t* ref, int32_t refLen, const uint8_t weight_gapO, const uint8_t weight_gapE,
	const uint8_t, const uint16_t, const int32_t, const int32_t, int)
{
	if (!prof || !ref || refLen <= 0 || prof->readLen <= 0) { fprintf(stderr, "ssw (fasim HIP shim): ssw_pre_align: empty input\n"); return nullptr; }
	if (!scoring_supported(prof, weight_gapO, weight_gapE)) {
		fprintf(stderr, "ssw (fasim HIP shim): only the Aligner's scoring is supported (5x5 matrix +5/-4, gap open 16, extension 4)\n");
		return nullptr;
	}
	int* cols = (int*)calloc((size_t)refLen, sizeof(int));        // the caller free()s it (ssw_cpp.cpp:440)
	if (!cols) return nullptr;
	std::lock_guard<std::mutex> lk(g_mu);
	fasim_engine* e = engine_locked();
	if (!e || !set_query_locked(e, prof)) { free(cols); return nullptr; }
	const std::string t = ref_letters(ref, refLen);
	if (fasim_ssw_pre_align(e, t.data(), refLen, cols) != FASIM_OK) {
		fprintf(stderr, "ssw (fasim HIP shim): %s\n", fasim_last_error(e));
		free(cols); return nullptr;
	}
	return cols;
}

s_align* ssw_align(const s_profile* prof, const int8_t* ref, int32_t refLen, const uint8_t weight_gapO, const uint8_t weight_gapE,
	const uint8_t flag, const uint16_t filters, const int32_t filterd, const int32_t maskLen)
{
	if (!prof || prof->readLen <= 0) { fprintf(stderr, "Please call the function ssw_init before ssw_align.\n"); return nullptr; }
	if (prof->score_size != 2) {
		// score_size 0 / 1 select only the 8-bit / only the 16-bit kernel in the reference (sswNew.cpp:1282-1289); the wrapper
		// always passes 2 (ssw_cpp.cpp:405,615) and that is the behaviour the engine implements
		fprintf(stderr, "Please set 2 to the score_size parameter of the function ssw_init, otherwise the alignment results will be incorrect.\n");
		return nullptr;
	}
	if (!ref || refLen <= 0) { fprintf(stderr, "ssw (fasim HIP shim): ssw_align: empty reference\n"); return nullptr; }
	if (!scoring_supported(prof, weight_gapO, weight_gapE)) {
		fprintf(stderr, "ssw (fasim HIP shim): only the Aligner's scoring is supported (5x5 matrix +5/-4, gap open 16, extension 4)\n");
		return nullptr;
	}
	if (maskLen < 1) fprintf(stderr, "When maskLen < 15, the function ssw_align doesn't return 2nd best alignment information.\n");   // sswNew.cpp:1467
	s_align* r = (s_align*)calloc(1, sizeof(s_align));
	if (!r) return nullptr;
	r->ref_begin1 = -1; r->read_begin1 = -1;

	std::lock_guard<std::mutex> lk(g_mu);
	fasim_engine* e = engine_locked();
	if (!e || !set_query_locked(e, prof)) { free(r); return nullptr; }
	const std::string t = ref_letters(ref, refLen);
	fasim_alignment al;
	if (fasim_ssw_align(e, t.data(), refLen, &al) != FASIM_OK) { fprintf(stderr, "ssw (fasim HIP shim): %s\n", fasim_last_error(e)); free(r); return nullptr; }
	if (al.cigar_len < 0) { free(r); return nullptr; }              // banded_sw found no path (sswNew.cpp:1535-1538)

	r->score1 = (uint16_t)al.sw_score;
	if (al.sw_score <= 0) {
		// nothing aligned: the reference's end_ref / end_read initial values (sswNew.cpp:500-501); what it does next reads
		// ref[-1] (undefined behaviour), so no begin position and no cigar are reported
		r->ref_end1 = -1; r->read_end1 = prof->readLen - 1; r->score2 = 0; r->ref_end2 = maskLen >= 1 ? 0 : -1;
		return r;
	}
	r->ref_end1 = al.ref_end; r->read_end1 = al.query_end;

	// sub-optimal score: the largest column maximum of the forward pass outside [ref_end1 - maskLen, ref_end1 + maskLen]
	// (sswNew.cpp:641-665); below 251 these are the 8-bit kernel's maxima (= ssw_pre_align's, no column reaches the
	// overflow cut), from 251 on the whole alignment ran on the 16-bit kernels (sswNew.cpp:1473-1477)
	r->score2 = 0; r->ref_end2 = maskLen >= 1 ? 0 : -1;
	if (maskLen >= 1) {
		std::vector<int32_t> cols((size_t)refLen, 0);
		const int rc = al.sw_score < 255 - 4 ? fasim_ssw_pre_align(e, t.data(), refLen, cols.data())
		                                     : fasim_ssw_colmax_word(e, t.data(), refLen, cols.data());
		if (rc != FASIM_OK) { fprintf(stderr, "ssw (fasim HIP shim): %s\n", fasim_last_error(e)); free(r); return nullptr; }
		int best = 0, at = 0;
		int edge = (al.ref_end - maskLen) > 0 ? (al.ref_end - maskLen) : 0;
		for (int i = 0; i < edge; i++) if (cols[(size_t)i] > best) { best = cols[(size_t)i]; at = i; }
		edge = (al.ref_end + maskLen) > refLen ? refLen : (al.ref_end + maskLen);
		for (int i = edge + 1; i < refLen; i++) if (cols[(size_t)i] > best) { best = cols[(size_t)i]; at = i; }
		r->score2 = (uint16_t)best; r->ref_end2 = at;
	}

	// flag semantics of sswNew.cpp:1504, 1524 (the engine always computes begin positions and cigar; they are reported
	// only where the reference would have computed them)
	if (flag == 0 || (flag == 2 && r->score1 < filters)) return r;
	r->ref_begin1 = al.ref_begin; r->read_begin1 = al.query_begin;
	if ((7 & flag) == 0 || ((2 & flag) != 0 && r->score1 < filters) ||
		((4 & flag) != 0 && (r->ref_end1 - r->ref_begin1 > filterd || r->read_end1 - r->read_begin1 > filterd))) return r;
	r->cigarLen = al.cigar_len;
	r->cigar = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(al.cigar_len > 0 ? al.cigar_len : 1));
	if (!r->cigar) { free(r); return nullptr; }
	memcpy(r->cigar, al.cigar, sizeof(uint32_t) * (size_t)al.cigar_len);
	return r;
}

void align_destroy(s_align* a)
{
	if (!a) return;
	free(a->cigar);
	free(a);
}

} // extern "C"
