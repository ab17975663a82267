/* include/ssw.h -- the reference's own native ABI for this path, exported by libfasim_hip.so.
 *
 * These five symbols are the `extern "C"` interface of the reference's SSW library that its C++ wrapper binds
 * (ssw_cpp.cpp:350-368, 406-426, 616-640).  Signatures, the s_align layout and the ownership rules are those of
 * /root/reference/ssw.h (declarations written out here, nothing else of that file is reproduced):
 *
 *   ssw_init       ssw.h:78    returns a calloc'd profile that BORROWS `read` and `mat` (sswNew.cpp:1274-1295)
 *   init_destroy   ssw.h:83    frees the profile
 *   ssw_align      ssw.h:118   returns a calloc'd s_align (cigar malloc'd) or NULL + a message on stderr
 *                              (sswNew.cpp:1478-1492, 1535-1538); callers treat NULL as score 0 (ssw_cpp.cpp:627-633)
 *   ssw_pre_align  ssw.h:128   returns a calloc'd int[refLen] the CALLER free()s (ssw_cpp.cpp:440)
 *   align_destroy  ssw.h:142   frees cigar and the struct
 *
 * so the reference's unchanged ssw_cpp.cpp links against libfasim_hip.so in place of sswNew.cpp
 * (oracle/Makefile target `shim_probe`; tests/test_gpu_parity.py::test_ssw_h_shim_runs_reference_wrapper).
 *
 * Behind the symbols every call is a small job on the HIP engine (fasim_hip.h): a validation and migration path,
 * not the fast one -- the batched entry points of fasim_hip.h are.  The engine implements exactly the scoring the
 * reference's Aligner uses (ssw_cpp.cpp:28-53, 238-250): a 5x5 matrix with +5 on the diagonal of codes 0..3 and -4
 * elsewhere, gap open 16, gap extension 4.  Any other matrix or gap pair is refused: NULL + a message on stderr,
 * the reference's own error convention.  The device is HIP device $FASIM_DEVICE (default 0); there is no CPU fallback.
 */
#ifndef FASIM_SSW_ABI_H
#define FASIM_SSW_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct _profile;                       /* opaque, as in the reference (ssw.h:30-32) */
typedef struct _profile s_profile;

typedef struct {                       /* ssw.h:48-58 */
	uint16_t score1;                   /* best score = min(forward, reverse) (sswNew.cpp:1518)                          */
	uint16_t score2;                   /* best column maximum outside +-maskLen of ref_end1 (sswNew.cpp:641-665)         */
	int32_t ref_begin1;                /* -1 when not computed                                                         */
	int32_t ref_end1;
	int32_t read_begin1;               /* -1 when not computed                                                         */
	int32_t read_end1;
	int32_t ref_end2;
	uint32_t* cigar;                   /* BAM encoding: len << 4 | op, op 0=M 1=I 2=D; NULL when not computed           */
	int32_t cigarLen;
} s_align;

s_profile* ssw_init(const int8_t* read, const int32_t readLen, const int8_t* mat, const int32_t n, const int8_t score_size);
void init_destroy(s_profile* p);
s_align* ssw_align(const s_profile* prof, const int8_t* ref, int32_t refLen, const uint8_t weight_gapO,
                   const uint8_t weight_gapE, const uint8_t flag, const uint16_t filters, const int32_t filterd,
                   const int32_t maskLen);
int* ssw_pre_align(const s_profile* prof, const int8_t* ref, int32_t refLen, const uint8_t weight_gapO,
                   const uint8_t weight_gapE, const uint8_t flag, const uint16_t filters, const int32_t filterd,
                   const int32_t maskLen, int threshold);
void align_destroy(s_align* a);

/* ssw.h:27 -- the op table behind to_cigar_int(); the wrapper references the symbol (ssw_cpp.cpp:108-207) */
extern const uint8_t encoded_ops[];

#ifdef __cplusplus
}
#endif
#endif
