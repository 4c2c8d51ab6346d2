/*
 * filter_adapter.h -- the binding of libx265hip's in-loop filter producer into the reference encoder.
 *
 * filter_adapter.cpp defines FrameFilter::processRow (encoder/framefilter.cpp:576-664), Deblock::deblockCTU (common/deblock.cpp:37-72) and SAO::calcSaoStatsCTU
 * (encoder/sao.cpp:729-905).  The reference filters a picture row by row behind its encoder; nothing in the encode of a picture reads its filtered samples, so the binding
 * lets the rows pass and filters the whole picture when its last row arrives: one x265hip_ff_picture call (include/x265hip_ctx.h) deblocks the picture and collects the SAO
 * statistics of every CTU, then the encoder's own row loop runs over all rows with deblockCTU a no-op and calcSaoStatsCTU a table look-up -- the SAO decision (rdoSaoUnitCu
 * with the encoder's entropy coder), the SAO itself, border extension, PSNR / SSIM / hashes and the row flags are the encoder's own code.  The encoder's bodies stay
 * available under the names processRow_cpu / deblockCTU_cpu / calcSaoStatsCTU_cpu (a maintainer renames the three members; oracle/Makefile target e2e2 does it at the
 * object level without touching a source file) and run when the adapter is not loaded, and for what the producer does not offer: 4:0:0 and picture sizes that are not
 * multiples of 8.
 *
 * Frame threads (the encoder's default): the next pictures wait for the rows this picture's filters finish (Frame::m_reconRowFlag, set by processPostRow), so a picture cannot
 * wait for its last row.  There the binding works in BANDS of CTU rows: the rows pass until X265FF_BAND_ROWS of them (default 4) are waiting or the picture's last row arrives,
 * then one x265hip_ff_picture call with desc.ctuRowFirst / ctuRowCount deblocks those rows (their top edge changes the last lines of the row above) and takes their statistics,
 * and the encoder's own row loop runs over the band's rows.  With --slices a band stays inside its slice (a slice's rows are a chain of their own; the slices of a picture and the
 * pictures in flight interleave on the one producer).  The fourth member the binding defines for that, FrameFilter::ParallelFilter::processTasks, is the entry the ROW
 * ENCODERS use to start a row's deblocking early (frameencoder.cpp:2067-2076): in band mode that call returns at once -- the row is filtered with its band (the call from
 * processRow itself keeps the encoder's body, xff_processTasks_cpu).
 */
#ifndef X265HIP_FILTER_ADAPTER_H
#define X265HIP_FILTER_ADAPTER_H
#ifdef __cplusplus
extern "C" {
#endif
int  x265hip_ff_adapter_load(const char* libraryPath, int device);      /* 0 on success; before x265_encoder_open */
void x265hip_ff_adapter_enable(int on);
void x265hip_ff_adapter_close(void);                                      /* after x265_encoder_close */
typedef struct x265hip_ff_adapter_stats
{
    int pictures, cpuPictures /* filtered by the encoder's own body */;
    long long deblockSkipped, statsServed;     /* deblockCTU calls that found the picture deblocked, calcSaoStatsCTU calls answered from the table */
    double gatherSeconds, producerSeconds /* inside x265hip_ff_picture */, replaySeconds /* the encoder's row loop behind it */;
    int bands;                                 /* producer calls; == pictures with one frame thread (whole pictures), more under frame threads (bands of CTU rows) */
} x265hip_ff_adapter_stats;
void x265hip_ff_adapter_get_stats(x265hip_ff_adapter_stats* out);
#ifdef __cplusplus
}
#endif
#endif
