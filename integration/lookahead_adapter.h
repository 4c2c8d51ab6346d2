/*
 * lookahead_adapter.h -- the binding of libx265hip's lookahead producer into the reference encoder.
 *
 * lookahead_adapter.cpp defines LookaheadTLD::lowresIntraEstimate (encoder/slicetype.cpp:755-864), CostEstimateGroup::estimateFrameCost (:4366-4463) and
 * CostEstimateGroup::finishBatch (:4271-4278: the queued estimates of a batch go up together as x265hip_la_estimate_batch calls) and Lookahead::estimateCUPropagate
 * (:3850-3956: a cuTree propagation step = one x265hip_la_cutree_propagate call): the intra
 * estimate of a picture entering the lookahead and every (p0, b, p1) frame-cost estimate become one x265hip_la_intra / x265hip_la_estimate call (include/x265hip_ctx.h);
 * slice-type decision, scene cuts, cuTree, VBV look-ahead -- everything that consumes costEst / lowresCosts / lowresMvs / rowSatds / intraMbs -- is the encoder's own code
 * reading the same arrays.  The encoder's bodies stay available under the names lowresIntraEstimate_cpu / estimateFrameCost_cpu (a maintainer renames the two members;
 * oracle/Makefile target e2e2 does it at the object level without touching a source file) and run when the adapter is not loaded, for --hme (not offered by the
 * producer) and for pictures beyond the producer's limits.
 */
#ifndef X265HIP_LOOKAHEAD_ADAPTER_H
#define X265HIP_LOOKAHEAD_ADAPTER_H
#ifdef __cplusplus
extern "C" {
#endif
int  x265hip_la_adapter_load(const char* libraryPath, int device);      /* 0 on success; before x265_encoder_open */
void x265hip_la_adapter_enable(int on);
void x265hip_la_adapter_close(void);                                      /* after x265_encoder_close */
typedef struct x265hip_la_adapter_stats
{
    int intraPictures, estimates, cpuEstimates /* fell through to the encoder's own body */, weighted;
    int launches;                              /* device launches the estimates went up in (concurrent callers share one); filled by x265hip_la_adapter_close */
    int batches, batchCalls;                   /* CostEstimateGroup::finishBatch calls taken whole; x265hip_la_estimate_batch calls they became (waves) */
    double intraSeconds, estimateSeconds;      /* whole calls, harvest and write-back included */
    double producerSeconds;                    /* inside x265hip_la_intra / x265hip_la_estimate */
    int cutreeSteps; double cutreeSeconds;     /* Lookahead::estimateCUPropagate calls that went through x265hip_la_cutree_propagate */
} x265hip_la_adapter_stats;
void x265hip_la_adapter_get_stats(x265hip_la_adapter_stats* out);
#ifdef __cplusplus
}
#endif
#endif
