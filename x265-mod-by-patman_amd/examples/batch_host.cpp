// batch_host.cpp -- a C++ host driving the frame-batched path through include/x265hip_ctx.h only (no Python, no torch):
//   batch_host <lib.so> <width> <height> <frames> <method> <subme> <merange> <qp> <in.raw> <out.bin>
// in.raw : frames x (source picture, reference picture), each width x height samples of the library's pixel type, tightly packed
// out.bin: for the levels 64, 32, 16, 8 the x265hip_me_result arrays, then the numSig array and the coefficients of the 32x32 TUs
// The library is loaded with dlopen the way a plugin host would; tests/test_ctx_gpu.py runs this program and compares with the Python pipeline.
#include "../../include/x265hip_ctx.h"
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define SYM(name) decltype(&name) p_##name = (decltype(&name))dlsym(lib, #name); if (!p_##name) { fprintf(stderr, "missing %s\n", #name); return 2; }
#define CHECK(call) do { int rc_ = (call); if (rc_ != X265HIP_OK) { fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, p_x265hip_last_error()); return 1; } } while (0)

int main(int argc, char** argv)
{
    if (argc < 11) { fprintf(stderr, "usage: %s lib width height frames method subme merange qp in.raw out.bin\n", argv[0]); return 2; }
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    SYM(x265hip_last_error) SYM(x265hip_bit_depth) SYM(x265hip_ctx_create) SYM(x265hip_ctx_destroy) SYM(x265hip_ctx_sync) SYM(x265hip_batch_create)
    SYM(x265hip_batch_destroy) SYM(x265hip_batch_upload_plane) SYM(x265hip_batch_step) SYM(x265hip_batch_read_results) SYM(x265hip_batch_read_coeffs)
    SYM(x265hip_batch_task_count) SYM(x265hip_batch_tu_count)
    x265hip_batch_desc d = {};
    d.width = atoi(argv[2]); d.height = atoi(argv[3]); d.frames = atoi(argv[4]); d.margin = 96; d.method = atoi(argv[5]); d.subme = atoi(argv[6]);
    d.merange = atoi(argv[7]); d.qp = atoi(argv[8]); d.tuLog2 = 5; d.recon = 0; d.usePlanes = 1;
    const size_t es = p_x265hip_bit_depth() == 8 ? 1 : 2, pic = (size_t)d.width * d.height * es;
    std::vector<char> in(pic * 2 * d.frames);
    FILE* f = fopen(argv[9], "rb");
    if (!f || fread(in.data(), 1, in.size(), f) != in.size()) { fprintf(stderr, "cannot read %s\n", argv[9]); return 2; }
    fclose(f);
    x265hip_ctx* ctx = nullptr; x265hip_batch* b = nullptr;
    CHECK(p_x265hip_ctx_create(0, &ctx));
    CHECK(p_x265hip_batch_create(ctx, &d, &b));
    for (int k = 0; k < d.frames; k++)
    {
        CHECK(p_x265hip_batch_upload_plane(b, 0, k, in.data() + (2 * k) * pic, d.width));
        CHECK(p_x265hip_batch_upload_plane(b, 1, k, in.data() + (2 * k + 1) * pic, d.width));
    }
    CHECK(p_x265hip_batch_step(b));
    CHECK(p_x265hip_batch_step(b));                       // a second pass over the resident planes gives the same bytes
    CHECK(p_x265hip_ctx_sync(ctx));
    FILE* o = fopen(argv[10], "wb");
    if (!o) { fprintf(stderr, "cannot write %s\n", argv[10]); return 2; }
    const int levels[4] = { 64, 32, 16, 8 };
    for (int lv : levels)
    {
        std::vector<x265hip_me_result> r((size_t)p_x265hip_batch_task_count(&d, lv));
        CHECK(p_x265hip_batch_read_results(b, lv, r.data()));
        fwrite(r.data(), sizeof(x265hip_me_result), r.size(), o);
    }
    const size_t ntu = (size_t)p_x265hip_batch_tu_count(&d);
    std::vector<uint32_t> ns(ntu); std::vector<int16_t> co(ntu << 10);
    CHECK(p_x265hip_batch_read_coeffs(b, co.data(), ns.data()));
    fwrite(ns.data(), 4, ns.size(), o); fwrite(co.data(), 2, co.size(), o);
    fclose(o);
    p_x265hip_batch_destroy(b); p_x265hip_ctx_destroy(ctx);
    printf("ok %zu TUs\n", ntu);
    return 0;
}
