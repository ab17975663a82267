// oracle/longtarget_binding.cpp -- TEST INFRASTRUCTURE: the reference-side binding shown in INTEGRATION.md section 2,
// compiled.  oracle/Makefile (target `ref`) links it with the reference's UNCHANGED driver sources
// (Fasim-LongTarget.cpp, ssw_cpp.cpp, sswNew.cpp, compiled where they lie) into oracle/_ref/fasim_ref_hipbind: the
// reference's own main(), readDna(), printResult(), cluster_triplex() and print_cluster() on top of THIS LongTarget(),
// which hands the segment x encoding loops to libfasim_hip.so.  The linker is told to keep the first definition of
// LongTarget (this one); tests/test_gpu_parity.py::test_reference_driver_with_longtarget_binding runs it on the GPU box.
#include "fastsim.h"      // the reference's header: struct para, struct triplex (compiled with -I/root/reference)

// >>> INTEGRATION.md section 2: begin
#include "fasim_hip.h"

// replaces the body of LongTarget() (Fasim-LongTarget.cpp:379-598)
void LongTarget(struct para &paraList, string rnaSequence, string dnaSequence,
                vector<struct triplex> &sort_triplex_list)
{
    static fasim_engine *eng = NULL;
    if (!eng && fasim_engine_create(0, &eng) != FASIM_OK) { fprintf(stderr, "%s\n", fasim_last_error(NULL)); exit(1); }
    fasim_set_query(eng, rnaSequence.data(), (int32_t)rnaSequence.size());

    fasim_params p; fasim_params_default(&p);
    p.rule = paraList.rule;            p.cutLength = paraList.cutLength;   p.strand = paraList.strand;
    p.overlapLength = paraList.overlapLength; p.ntMin = paraList.ntMin;    p.ntMax = paraList.ntMax;
    p.scoreMin = paraList.scoreMin;    p.minIdentity = paraList.minIdentity; p.minStability = paraList.minStability;
    p.penaltyT = paraList.penaltyT;    p.penaltyC = paraList.penaltyC;
    p.cDistance = paraList.cDistance;  p.cLength = paraList.cLength;   p.classicSim = paraList.doFastSim ? 0 : 1;   // -F

    fasim_result *r = NULL;
    if (fasim_scan(eng, dnaSequence.data(), (int64_t)dnaSequence.size(), 0, -1, &p, &r) != FASIM_OK) {
        fprintf(stderr, "%s\n", fasim_last_error(eng)); exit(1);
    }
    for (int64_t i = 0; i < r->count; i++) {
        const fasim_triplex &t = r->recs[i];
        sort_triplex_list.push_back(triplex(t.stari, t.endi, t.starj, t.endj, t.strand, t.reverse, t.rule, t.nt,
            t.score, t.identity, t.tri_score, r->pool + t.tfo_off, r->pool + t.tts_off, 0, 0, 0, 0, 0, 0, ""));
    }
    fasim_result_free(r);
}
// <<< INTEGRATION.md section 2: end
