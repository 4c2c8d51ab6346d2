/*
 * tme_adapter.h -- the binding of libx265hip's ThreadedME producer into the reference encoder.
 *
 * x265's decoupled motion estimation (--threaded-me) has a producer / consumer contract: ThreadedME::findJob calls Analysis::deriveMVsForCTU(ctu, geom, frame) per CTU
 * (encoder/threadedme.cpp:207-261), which fills that CTU's MEData records in slice->m_ctuMV (encoder/analysis.cpp:248-306); Search::predInterSearch consumes them.
 * tme_adapter.cpp defines Analysis::deriveMVsForCTU: the first call for a picture hands the WHOLE picture to x265hip_tme_picture (include/x265hip_ctx.h) and copies the
 * table it returns into slice->m_ctuMV; the calls for the picture's other CTUs find their records there.  The encoder's own body of that member stays available under the
 * name deriveMVsForCTU_cpu (a maintainer renames it in analysis.cpp / analysis.h; a build without source changes compiles analysis.cpp with
 * -DderiveMVsForCTU=deriveMVsForCTU_cpu, oracle/Makefile target tmegpu) and is what runs when the adapter is not loaded.
 *
 * No reference header is patched and no private member is reached from outside: everything the adapter needs beyond public data (Analysis::calculateQpforCuSize is
 * protected) is used from inside the member function it defines.
 */
#ifndef X265HIP_TME_ADAPTER_H
#define X265HIP_TME_ADAPTER_H

#ifdef __cplusplus
extern "C" {
#endif
/* dlopen libx265hip_<depth>.so and bind the producer; 0 on success.  Before x265_encoder_open.  Without it (or after tme_adapter_enable(0)) deriveMVsForCTU runs the encoder's own body. */
int  x265hip_tme_adapter_load(const char* libraryPath, int device);
void x265hip_tme_adapter_enable(int on);
void x265hip_tme_adapter_close(void);          /* after x265_encoder_close: destroys the producer and its context */

typedef struct x265hip_tme_adapter_stats
{
    int pictures, weightedRefs;
    int bands;                     /* producer calls: bands of CTU rows (== pictures with one frame thread; more with frame threads, where a picture's rows become ready as its references' rows are final) */
    double producerSeconds;        /* inside x265hip_tme_picture                                                        */
    double producerSecondsWarm; int callsWarm;      /* the same without the first four calls (every kernel's first launch loads its code object) */
    double adapterSeconds;         /* the whole per-picture call: harvest + producer + write-back                       */
    double createSeconds;          /* creating the producer, once (inside adapterSeconds and sections[0..1])                */
    double sections[4];            /* [0] job set-up, [1] wall time up to the producer call (set-up + harvest, spread over the ThreadedME workers), [2] CTUs harvested by workers
                                      other than the one that opened the job, [3] write-back */
} x265hip_tme_adapter_stats;
void x265hip_tme_adapter_get_stats(x265hip_tme_adapter_stats* out);
#ifdef __cplusplus
}
#endif

#endif
