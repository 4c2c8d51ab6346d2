/*
 * ref_encode.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * The reference ENCODER (every source/common + source/encoder file, compiled where it lies by oracle/Makefile, no asm) driven
 * through its public API (x265.h) on a synthetic 4:2:0 clip, with the EncoderPrimitives table either as the reference
 * builds it (mode "c") or with libx265hip's overwrite pass applied on top (mode "hip"):
 *
 *     x265_setup_primitives(param);                               // primitives.cpp:336-376 (C table; this build has no asm)
 *     x265hip_setup_primitives(&primitives, X265_DEPTH, 0);        // the drop-in pass of INTEGRATION.md section 2
 *     setupAliasPrimitives(primitives);                           // primitives.cpp:367
 *     x265_encoder_open(param);                                   // its own setup call is a no-op (one-time guard :338)
 *
 * Both modes must produce the SAME BITSTREAM: that is the drop-in claim, checked end to end through the untouched
 * RDO / entropy coder.  (The per-slot path stages every block over PCIe -- it is a parity vehicle, not the fast path.)
 *
 * usage: x265enc_<depth> c|hip <libx265hip.so or -> <width> <height> <frames> <preset> <out.hevc> [option=value ...]   (x265_param_parse names)
 */
#include "x265.h"
#include "common.h"
#include "primitives.h"
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace X265_NS;
namespace X265_NS { void enableLowpassDCTPrimitives(EncoderPrimitives& p); }     /* primitives.cpp:77-88 */

static void synth(std::vector<pixel>& y, std::vector<pixel>& u, std::vector<pixel>& v, int w, int h, int f)
{   // moving smooth texture + deterministic noise (fixed LCG), so that inter and intra tools both have work
    uint32_t s = 12345u + 977u * (uint32_t)f;
    const int pm = (1 << X265_DEPTH) - 1;
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++)
        {
            int x = i + 3 * f, yy = j + 2 * f;
            int t = ((x * x / 7 + yy * 5 + (x * yy) / 11) & 255) * (pm + 1) / 256;
            s = s * 1664525u + 1013904223u;
            int nz = (int)((s >> 24) & 7) - 3;
            int val = t + nz; y[j * w + i] = (pixel)(val < 0 ? 0 : val > pm ? pm : val);
        }
    for (int j = 0; j < h / 2; j++)
        for (int i = 0; i < w / 2; i++)
        {
            u[j * (w / 2) + i] = (pixel)((((i + f) * 3 + j) & 127) * (pm + 1) / 256 + (pm + 1) / 4);
            v[j * (w / 2) + i] = (pixel)((((j + 2 * f) * 5 + i) & 127) * (pm + 1) / 256 + (pm + 1) / 4);
        }
}

int main(int argc, char** argv)
{
    if (argc < 8) { fprintf(stderr, "usage: %s c|hip lib width height frames preset out.hevc\n", argv[0]); return 2; }
    const bool hip = !strcmp(argv[1], "hip");
    const int w = atoi(argv[3]), h = atoi(argv[4]), frames = atoi(argv[5]);
    x265_param* p = x265_param_alloc();
    if (x265_param_default_preset(p, argv[6], NULL) < 0) { fprintf(stderr, "bad preset\n"); return 2; }
    p->sourceWidth = w; p->sourceHeight = h; p->fpsNum = 25; p->fpsDenom = 1; p->internalCsp = X265_CSP_I420;
    p->totalFrames = frames; p->logLevel = X265_LOG_WARNING; p->bRepeatHeaders = 1;
    p->frameNumThreads = 1; p->bEnableWavefront = 0; p->lookaheadSlices = 0;
    x265_param_parse(p, "pools", "none");
    x265_param_parse(p, "hash", "1");                 // decoded-picture MD5 SEI: the reconstruction is part of the bitstream
    for (int i = 8; i < argc; i++)
    {
        char* eq = strchr(argv[i], '=');
        if (eq) *eq = 0;
        if (x265_param_parse(p, argv[i], eq ? eq + 1 : NULL) < 0) { fprintf(stderr, "bad option %s\n", argv[i]); return 2; }
    }

    x265_setup_primitives(p);
    if (hip)
    {
        void* lib = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
        if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
        typedef int (*abi_t)(size_t, int);
        typedef int (*setup_t)(void*, int, uint32_t);
        typedef const char* (*err_t)(void);
        abi_t chk = (abi_t)dlsym(lib, "x265hip_abi_check"); setup_t set = (setup_t)dlsym(lib, "x265hip_setup_primitives"); err_t err = (err_t)dlsym(lib, "x265hip_last_error");
        static EncoderPrimitives cTable;
        memcpy(&cTable, &primitives, sizeof(cTable));
        if (!chk || !set || chk(sizeof(EncoderPrimitives), X265_DEPTH) != 0 || set(&primitives, X265_DEPTH, 0) != 0)
        { fprintf(stderr, "x265hip setup failed: %s\n", err ? err() : "?"); return 2; }
        if (const char* r = getenv("X265ENC_HIP_RANGE"))
        {   /* bisection aid: only slots whose byte offset lies in [lo, hi) keep the HIP pointer */
            long lo = 0, hi = 0; sscanf(r, "%ld:%ld", &lo, &hi);
            void** t = (void**)&primitives; void** c = (void**)&cTable;
            for (size_t i = 0; i < sizeof(EncoderPrimitives) / 8; i++)
                if ((long)(i * 8) < lo || (long)(i * 8) >= hi) t[i] = c[i];
        }
        setupAliasPrimitives(primitives);
        if (p->bLowPassDct) enableLowpassDCTPrimitives(primitives);          /* the step after the alias pass (primitives.cpp:369-372) */
    }
    x265_encoder* enc = x265_encoder_open(p);
    if (!enc) { fprintf(stderr, "encoder_open failed\n"); return 2; }
    FILE* out = fopen(argv[7], "wb");
    if (!out) { fprintf(stderr, "cannot write %s\n", argv[7]); return 2; }
    x265_picture* pic = x265_picture_alloc();
    x265_picture_init(p, pic);
    std::vector<pixel> Y((size_t)w * h), U((size_t)w * h / 4), V((size_t)w * h / 4);
    pic->planes[0] = Y.data(); pic->planes[1] = U.data(); pic->planes[2] = V.data();
    pic->stride[0] = w * (int)sizeof(pixel); pic->stride[1] = pic->stride[2] = (w / 2) * (int)sizeof(pixel);
    pic->bitDepth = X265_DEPTH; pic->colorSpace = X265_CSP_I420;
    /* X265ENC_DUMP=<file>: every output picture as  int32 poc, double ssim, psnrY, psnrU, psnrV  then the source Y and the reconstructed Y, U, V planes
     * (16-bit samples, tightly packed) -- the pin of the frame-level SSIM / SSD restatements against the encoder's own per-frame statistics */
    FILE* dump = getenv("X265ENC_DUMP") ? fopen(getenv("X265ENC_DUMP"), "wb") : NULL;
    x265_picture* rec = x265_picture_alloc();
    auto dumpPic = [&](int got)
    {
        if (!dump || got <= 0) return;
        const int32_t poc = rec->poc;
        const double st[4] = { rec->frameData.ssim, rec->frameData.psnrY, rec->frameData.psnrU, rec->frameData.psnrV };
        fwrite(&poc, 4, 1, dump); fwrite(st, 8, 4, dump);
        std::vector<pixel> sy((size_t)w * h), su((size_t)w * h / 4), sv((size_t)w * h / 4);
        synth(sy, su, sv, w, h, poc);
        std::vector<uint16_t> line;
        auto put = [&](const pixel* src, intptr_t stride, int pw, int ph)
        {
            line.resize(pw);
            for (int y = 0; y < ph; y++) { for (int x = 0; x < pw; x++) line[x] = src[y * stride + x]; fwrite(line.data(), 2, pw, dump); }
        };
        put(sy.data(), w, w, h); put(su.data(), w / 2, w / 2, h / 2); put(sv.data(), w / 2, w / 2, h / 2);
        for (int c = 0; c < 3; c++)
            put((const pixel*)rec->planes[c], rec->stride[c] / (int)sizeof(pixel), c ? w / 2 : w, c ? h / 2 : h);
    };
    size_t bytes = 0;
    const auto t0 = std::chrono::steady_clock::now();
    x265_nal* nal; uint32_t nnal;
    for (int f = 0; f < frames; f++)
    {
        synth(Y, U, V, w, h, f);
        pic->pts = f;
        const int got = x265_encoder_encode(enc, &nal, &nnal, pic, dump ? rec : NULL);
        if (got < 0) { fprintf(stderr, "encode failed\n"); return 2; }
        dumpPic(got);
        for (uint32_t i = 0; i < nnal; i++) { fwrite(nal[i].payload, 1, nal[i].sizeBytes, out); bytes += nal[i].sizeBytes; }
    }
    for (int got; (got = x265_encoder_encode(enc, &nal, &nnal, NULL, dump ? rec : NULL)) > 0;)
    {
        dumpPic(got);
        for (uint32_t i = 0; i < nnal; i++) { fwrite(nal[i].payload, 1, nal[i].sizeBytes, out); bytes += nal[i].sizeBytes; }
    }
    if (dump) fclose(dump);
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fclose(out);
    x265_encoder_close(enc); x265_picture_free(pic); x265_param_free(p);
    printf("{\"mode\": \"%s\", \"frames\": %d, \"bytes\": %zu, \"seconds\": %.3f, \"fps\": %.3f}\n", argv[1], frames, bytes, sec, frames / sec);
    return 0;
}
