/*
 * ref_tme_gpu.cpp -- TEST INFRASTRUCTURE ONLY: the driver of the end-to-end ThreadedME check.
 *
 * The reference encoder (all of source/common + source/encoder, compiled where it lies) run with --threaded-me on a synthetic clip, linked with the binding
 * integration/tme_adapter.cpp (Analysis::deriveMVsForCTU -> one x265hip_tme_picture call per picture).  With X265TMEGPU=0 the adapter is not loaded and every call runs
 * the encoder's own body: the same binary, CPU producer.  The bitstream of the two runs must be identical (tests/test_e2e_tme_gpu.py); the printed frames-per-second are
 * the end-to-end figures of DESIGN section 6 and bench.py's e2e_fps.
 *
 * Built a second time as x265e2e_<depth> (-DWITH_LA_ADAPTER, make -C oracle e2e2) with integration/lookahead_adapter.cpp linked in as well: X265LAGPU=1 makes the GPU the
 * producer of the lookahead's intra costs and frame-cost estimates (x265hip_la_intra / x265hip_la_estimate), X265TME=0 runs without --threaded-me (lookahead seam alone);
 * and with integration/filter_adapter.cpp: X265FFGPU=1 makes the GPU deblock every picture and collect its SAO statistics (x265hip_ff_picture).
 *
 * usage: x265tmegpu_<depth> <libx265hip.so> <width> <height> <frames> <preset> <out.hevc> [option=value ...]
 */
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "x265.h"
#include "common.h"
#include "../integration/tme_adapter.h"
#ifdef WITH_LA_ADAPTER
#include "../integration/lookahead_adapter.h"
#include "../integration/filter_adapter.h"
#endif

using namespace X265_NS;

static void synth(std::vector<pixel>& y, std::vector<pixel>& u, std::vector<pixel>& v, int w, int h, int f, int cw, int ch)      /* cw x ch: the chroma planes (the clip's format) */
{   /* the clip of ref_tme.cpp: textured picture in (not purely translational) motion + deterministic noise */
    uint32_t s = 4242u + 733u * (uint32_t)f;
    const int pm = (1 << X265_DEPTH) - 1;
    /* read ONCE: the loop below used to ask per pixel -- 68 ms of the main thread per 1080p picture inside the timed encode, and a getenv that runs while a pool thread loads the
       HIP libraries (their initialisers call setenv) can fault (seen with the CPU mock producer, which loads libx265hip on the first picture) */
    static const bool fade = getenv("X265TME_FADE") != NULL;
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++)
        {
            const int x = i + 5 * f + ((j >> 5) & 1) * f, yy = j + 3 * f;
            const int t = (((x * x) / 9 + yy * 7 + (x * yy) / 13 + ((x >> 3) ^ (yy >> 3)) * 11) & 255) * (pm + 1) / 256;
            s = s * 1664525u + 1013904223u;
            int val = t + (int)((s >> 24) & 7) - 3;
            if (fade) val = val * (16 - 2 * (f < 6 ? f : 6)) / 16 + 4 * f * (pm + 1) / 256;      /* a fade: weighted prediction gets something to do */
            y[(size_t)j * w + i] = (pixel)(val < 0 ? 0 : val > pm ? pm : val);
        }
    for (int j = 0; j < ch; j++)
        for (int i = 0; i < cw; i++)
        {
            u[(size_t)j * cw + i] = (pixel)((((i + f) * 3 + j) & 127) * (pm + 1) / 256 + (pm + 1) / 4);
            v[(size_t)j * cw + i] = (pixel)((((j + 2 * f) * 5 + i) & 127) * (pm + 1) / 256 + (pm + 1) / 4);
        }
}

int main(int argc, char** argv)
{
    if (argc < 7) { fprintf(stderr, "usage: %s libx265hip.so width height frames preset out.hevc [option=value ...]\n", argv[0]); return 2; }
    const int useGpu = getenv("X265TMEGPU") ? atoi(getenv("X265TMEGPU")) : 1;
    if (useGpu && x265hip_tme_adapter_load(argv[1], getenv("X265TME_DEVICE") ? atoi(getenv("X265TME_DEVICE")) : 0)) return 2;
    int useLa = 0;
#ifdef WITH_LA_ADAPTER
    useLa = getenv("X265LAGPU") ? atoi(getenv("X265LAGPU")) : 0;
    if (useLa && x265hip_la_adapter_load(argv[1], getenv("X265TME_DEVICE") ? atoi(getenv("X265TME_DEVICE")) : 0)) return 2;
    const int useFf = getenv("X265FFGPU") ? atoi(getenv("X265FFGPU")) : 0;
    if (useFf && x265hip_ff_adapter_load(argv[1], getenv("X265TME_DEVICE") ? atoi(getenv("X265TME_DEVICE")) : 0)) return 2;
#endif
    const int w = atoi(argv[2]), h = atoi(argv[3]), frames = atoi(argv[4]);
    x265_param* p = x265_param_alloc();
    if (x265_param_default_preset(p, argv[5], NULL) < 0) { fprintf(stderr, "bad preset\n"); return 2; }
    /* X265_CSP=i422 / i444: the clip's (and the encode's) chroma format; default 4:2:0 */
    const int csp = getenv("X265_CSP") ? (!strcmp(getenv("X265_CSP"), "i422") ? X265_CSP_I422 : !strcmp(getenv("X265_CSP"), "i444") ? X265_CSP_I444 : X265_CSP_I420) : X265_CSP_I420;
    const int cw = csp == X265_CSP_I444 ? w : w / 2, ch = csp == X265_CSP_I420 ? h / 2 : h;
    p->sourceWidth = w; p->sourceHeight = h; p->fpsNum = 25; p->fpsDenom = 1; p->internalCsp = csp;
    p->totalFrames = frames; p->logLevel = X265_LOG_WARNING; p->bRepeatHeaders = 1;
    if (!getenv("X265_CLI_THREADING"))
    {   /* the seams are checked with one frame thread and no WPP (complete reference pictures); X265_CLI_THREADING=1 leaves the threading the x265 CLI would use by
           default (frame threads, WPP, all cores): BASELINE.md's C1 "x265 CLI on host CPU" figure */
        p->frameNumThreads = 1; p->bEnableWavefront = 0; p->lookaheadSlices = 0;
        x265_param_parse(p, "pools", "32");
    }
    const int tmeOn = getenv("X265TME") ? atoi(getenv("X265TME")) : 1;
    if (tmeOn) x265_param_parse(p, "threaded-me", "1");
    for (int i = 7; i < argc; i++)
    {
        char* eq = strchr(argv[i], '=');
        if (eq) *eq = 0;
        if (x265_param_parse(p, argv[i], eq ? eq + 1 : NULL) < 0) { fprintf(stderr, "bad option %s\n", argv[i]); return 2; }
    }
    /* X265_QUALITY=1: the encoder's own quality accounting (x265_stats: global PSNR / SSIM, bitrate) printed beside the fps -- a run with --threaded-me and a run without it write
       different bitstreams, so their speeds are only comparable next to what each spent and what each kept.  The accounting reads the reconstructed pictures, it does not change
       a decision: the bitstream is the one written without it */
    const bool quality = getenv("X265_QUALITY") && atoi(getenv("X265_QUALITY"));
    if (quality) { p->bEnablePsnr = 1; p->bEnableSsim = 1; p->logLevel = X265_LOG_INFO; }      /* (below log level info the encoder switches both off again: encoder.cpp:4349-4354) */
    const bool frameStats = getenv("X265_FRAME_STATS") && atoi(getenv("X265_FRAME_STATS"));      /* diagnosis: the encoder's own per-frame clocks (x265_frame_stats, csv-log-level 2) summed over the clip */
    if (frameStats) p->csvLogLevel = 2;
    FILE* out = fopen(argv[6], "wb");
    if (!out) { fprintf(stderr, "cannot write %s\n", argv[6]); return 2; }
    x265_encoder* enc = x265_encoder_open(p);
    if (!enc) { fprintf(stderr, "encoder_open failed\n"); return 2; }
    x265_picture* pic = x265_picture_alloc();
    x265_picture_init(p, pic);
    std::vector<pixel> Y((size_t)w * h), U((size_t)cw * ch), V((size_t)cw * ch);
    pic->planes[0] = Y.data(); pic->planes[1] = U.data(); pic->planes[2] = V.data();
    pic->stride[0] = w * (int)sizeof(pixel); pic->stride[1] = pic->stride[2] = cw * (int)sizeof(pixel);
    pic->bitDepth = X265_DEPTH; pic->colorSpace = csp;
    x265_nal* nal; uint32_t nnal;
    size_t bytes = 0;
    x265_picture* picOut = frameStats ? x265_picture_alloc() : NULL;
    double fsTmeWaitMs = 0, fsTmeMs = 0, fsCtuMs = 0, fsStallMs = 0, fsWallMs = 0, fsWpp = 0, fsRefWaitMs = 0; int fsN = 0;
    auto take = [&](int got) { if (got > 0 && picOut) { const x265_frame_stats& f = picOut->frameData; fsTmeWaitMs += f.tmeWaitTime / 1000.0; fsTmeMs += f.tmeTime / 1000.0; fsCtuMs += f.totalCTUTime; fsStallMs += f.stallTime; fsWallMs += f.wallTime; fsRefWaitMs += f.refWaitWallTime; fsWpp += f.avgWPP; fsN++; } };
    const auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < frames; f++)
    {
        synth(Y, U, V, w, h, f, cw, ch);
        pic->pts = f;
        const int r = x265_encoder_encode(enc, &nal, &nnal, pic, picOut);
        if (r < 0) { fprintf(stderr, "encode failed\n"); return 2; }
        take(r);
        for (uint32_t i = 0; i < nnal; i++) { fwrite(nal[i].payload, 1, nal[i].sizeBytes, out); bytes += nal[i].sizeBytes; }
    }
    for (;;)
    {
        const int r = x265_encoder_encode(enc, &nal, &nnal, NULL, picOut);
        if (r <= 0) break;
        take(r);
        for (uint32_t i = 0; i < nnal; i++) { fwrite(nal[i].payload, 1, nal[i].sizeBytes, out); bytes += nal[i].sizeBytes; }
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    x265_param* live = x265_param_alloc();
    x265_encoder_parameters(enc, live);
    const int tme = live->bThreadedME, frameThreads = live->frameNumThreads, wpp = live->bEnableWavefront;
    x265_param_free(live);
    char qual[300] = "";
    if (quality)
    {
        x265_stats st;
        x265_encoder_get_stats(enc, &st, sizeof(st));
        const double n = st.encodedPictureCount ? st.encodedPictureCount : 1;      /* (x265_stats::globalPsnrY / U / V are sums over the pictures: encoder.cpp:3101-3103) */
        snprintf(qual, sizeof(qual), "\"quality\": {\"kbps\": %.2f, \"psnr_y\": %.4f, \"psnr_u\": %.4f, \"psnr_v\": %.4f, \"psnr_global\": %.4f, \"ssim\": %.6f, \"pictures\": %u}, ",
                 st.bitrate, st.globalPsnrY / n, st.globalPsnrU / n, st.globalPsnrV / n, st.globalPsnr, st.globalSsim, st.encodedPictureCount);
    }
    x265_encoder_close(enc); x265_picture_free(pic); x265_param_free(p);
    fclose(out);
    x265hip_tme_adapter_stats s;
    x265hip_tme_adapter_get_stats(&s);
    x265hip_tme_adapter_close();
    char la[1400] = "";
#ifdef WITH_LA_ADAPTER
    {
        x265hip_la_adapter_stats ls;
        x265hip_la_adapter_close();
        x265hip_la_adapter_get_stats(&ls);
        snprintf(la, sizeof(la), "\"lookahead_producer\": \"%s\", \"la_intra_pictures\": %d, \"la_estimates\": %d, \"la_launches\": %d, \"la_batches\": %d, \"la_batch_calls\": %d, \"la_cpu_estimates\": %d, \"la_weighted\": %d, \"la_intra_seconds\": %.3f, \"la_estimate_seconds\": %.3f, \"la_producer_seconds\": %.3f, \"la_cutree_steps\": %d, \"la_cutree_seconds\": %.3f, ",
                 useLa ? "gpu" : "cpu", ls.intraPictures, ls.estimates, ls.launches, ls.batches, ls.batchCalls, ls.cpuEstimates, ls.weighted, ls.intraSeconds, ls.estimateSeconds, ls.producerSeconds,
                 ls.cutreeSteps, ls.cutreeSeconds);
        x265hip_ff_adapter_stats fs;
        x265hip_ff_adapter_close();
        x265hip_ff_adapter_get_stats(&fs);
        const size_t at = strlen(la);
        snprintf(la + at, sizeof(la) - at, "\"filter_producer\": \"%s\", \"ff_pictures\": %d, \"ff_bands\": %d, \"ff_cpu_pictures\": %d, \"ff_deblock_calls_skipped\": %lld, \"ff_stats_served\": %lld, \"ff_gather_seconds\": %.3f, \"ff_producer_seconds\": %.3f, \"ff_replay_seconds\": %.3f, ",
                 useFf ? "gpu" : "cpu", fs.pictures, fs.bands, fs.cpuPictures, fs.deblockSkipped, fs.statsServed, fs.gatherSeconds, fs.producerSeconds, fs.replaySeconds);
    }
#endif
    if (frameStats && fsN)
        fprintf(stderr, "frame stats over %d pictures (ms per picture): wall %.1f, all rows' reference wait %.1f, CTU worker time %.1f, stall (no worker) %.1f, ThreadedME tasks %.1f, rows blocked on ThreadedME %.1f, avg WPP %.1f\n",
                fsN, fsWallMs / fsN, fsRefWaitMs / fsN, fsCtuMs / fsN, fsStallMs / fsN, fsTmeMs / fsN, fsTmeWaitMs / fsN, fsWpp / fsN);
    char fs[400] = "";
    if (frameStats && fsN)
        snprintf(fs, sizeof(fs), "\"frame_stats_ms_per_picture\": {\"pictures\": %d, \"wall\": %.1f, \"ctu_worker_time\": %.1f, \"threaded_me_tasks\": %.1f, \"rows_blocked_on_threaded_me\": %.1f, \"reference_wait\": %.1f, \"avg_wpp\": %.2f}, ",
                 fsN, fsWallMs / fsN, fsCtuMs / fsN, fsTmeMs / fsN, fsTmeWaitMs / fsN, fsRefWaitMs / fsN, fsWpp / fsN);
    printf("{%s%s%s\"producer\": \"%s\", \"weighted_refs\": %d, \"frames\": %d, \"seconds\": %.3f, \"fps\": %.3f, \"bytes\": %zu, \"threaded_me\": %d, \"gpu_pictures\": %d, \"gpu_bands\": %d, \"frame_threads\": %d, \"wpp\": %d, \"gpu_seconds\": %.3f, \"gpu_seconds_warm\": %.4f, \"gpu_calls_warm\": %d, \"adapter_seconds\": %.3f, \"adapter_create_seconds\": %.3f, \"adapter_sections\": [%.3f, %.3f, %.3f, %.3f]}\n",
           la, fs, qual, useGpu ? "gpu" : "cpu", s.weightedRefs, frames, secs, frames / secs, bytes, tme, s.pictures, s.bands, frameThreads, wpp, s.producerSeconds, s.producerSecondsWarm, s.callsWarm, s.adapterSeconds, s.createSeconds, s.sections[0], s.sections[1], s.sections[2], s.sections[3]);
    return 0;
}
