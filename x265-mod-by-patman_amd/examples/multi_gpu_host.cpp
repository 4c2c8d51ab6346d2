// multi_gpu_host.cpp -- the frame-batched path on several GPUs of one node from a C++ host (include/x265hip_ctx.h through dlopen; no Python, no torch, no HIP headers):
// one worker per device -- a thread (default) or a forked process (--procs) -- each with its own x265hip_ctx and x265hip_batch, its own pictures (independent frames /
// GOP segments, SURVEY 8(e): no data-path exchange between the devices), all workers released together, the job timed from that release to the last worker's sync.
//
//   multi_gpu_host <libx265hip_N.so> --devices 0,1,2,... [--width 3840 --height 2176 --frames 8 --steps 20 --warmup 3 --inner 1
//                   --method 3 --subme 3 --merange 57 --qp 28 --refs 1 --rect 0 --streams 2 --same-frames --input in.raw --procs]
//
// Worker k of n takes the pictures with global indices k * frames .. k * frames + frames - 1 (with --same-frames every worker takes pictures 0 .. frames - 1: the digests
// of the workers' results must then agree -- the cross-device parity check of tests/test_multi_gpu_host_gpu.py).  Prints ONE JSON line: aggregate Mpixels/s over all
// devices (pixels of the padded pictures, as bench.py counts them), per-device milliseconds per step and an FNV-1a digest of every worker's MV records and coefficients.
// Exit codes: 0 ok, 2 usage / library, 3 a device could not be opened (X265HIP_EDEVICE: the message names it), 1 any other failure.
#include "../../include/x265hip_ctx.h"
#include <dlfcn.h>
#include <sys/wait.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {
struct Api
{
    const char* (*last_error)();
    int (*bit_depth)();
    int (*ctx_create)(int, x265hip_ctx**);
    void (*ctx_destroy)(x265hip_ctx*);
    int (*ctx_sync)(x265hip_ctx*);
    int (*batch_create)(x265hip_ctx*, const x265hip_batch_desc*, x265hip_batch**);
    void (*batch_destroy)(x265hip_batch*);
    int (*batch_upload_plane)(x265hip_batch*, int, int, const void*, intptr_t);
    int (*batch_step)(x265hip_batch*);
    int (*batch_read_results)(x265hip_batch*, int, x265hip_me_result*);
    int (*batch_read_coeffs)(x265hip_batch*, int16_t*, uint32_t*);
    int (*batch_task_count)(const x265hip_batch_desc*, int);
    int (*batch_tu_count)(const x265hip_batch_desc*);
};
bool load(const char* path, Api& a, void*& lib)
{
    lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "multi_gpu_host: dlopen %s: %s\n", path, dlerror()); return false; }
#define SYM(field, name) *(void**)&a.field = dlsym(lib, name); if (!a.field) { fprintf(stderr, "multi_gpu_host: %s lacks %s\n", path, name); return false; }
    SYM(last_error, "x265hip_last_error") SYM(bit_depth, "x265hip_bit_depth") SYM(ctx_create, "x265hip_ctx_create") SYM(ctx_destroy, "x265hip_ctx_destroy") SYM(ctx_sync, "x265hip_ctx_sync")
    SYM(batch_create, "x265hip_batch_create") SYM(batch_destroy, "x265hip_batch_destroy") SYM(batch_upload_plane, "x265hip_batch_upload_plane") SYM(batch_step, "x265hip_batch_step")
    SYM(batch_read_results, "x265hip_batch_read_results") SYM(batch_read_coeffs, "x265hip_batch_read_coeffs") SYM(batch_task_count, "x265hip_batch_task_count") SYM(batch_tu_count, "x265hip_batch_tu_count")
#undef SYM
    return true;
}

struct Options
{
    std::string lib;
    std::vector<int> devices;
    int width = 3840, height = 2176, frames = 8, steps = 20, warmup = 3, inner = 1, method = 3, subme = 3, merange = 57, qp = 28, refs = 1, rect = 0, streams = 2;
    bool sameFrames = false, procs = false;
    std::string input;          // --input in.raw: frames x (source, reference) pictures, tightly packed (the layout batch_host reads); every worker then takes THESE pictures
};
// "0,1,2" or "0-3" or a mix; every entry a non-negative integer
bool parse_devices(const char* s, std::vector<int>& out)
{
    out.clear();
    const char* p = s;
    while (*p)
    {
        char* e;
        const long a = strtol(p, &e, 10);
        if (e == p || a < 0 || a > 4095) return false;
        long b = a;
        if (*e == '-') { const char* q = e + 1; b = strtol(q, &e, 10); if (e == q || b < a || b > 4095) return false; }
        for (long v = a; v <= b; v++) out.push_back((int)v);
        if (*e == ',') e++; else if (*e) return false;
        p = e;
    }
    return !out.empty();
}
bool parse(int argc, char** argv, Options& o)
{
    if (argc < 2) return false;
    o.lib = argv[1];
    for (int i = 2; i < argc; i++)
    {
        const std::string k = argv[i];
        if (k == "--same-frames") { o.sameFrames = true; continue; }
        if (k == "--procs") { o.procs = true; continue; }
        if (i + 1 >= argc) { fprintf(stderr, "multi_gpu_host: %s needs a value\n", k.c_str()); return false; }
        const char* v = argv[++i];
        if (k == "--input") { o.input = v; o.sameFrames = true; continue; }
        if (k == "--devices") { if (!parse_devices(v, o.devices)) { fprintf(stderr, "multi_gpu_host: bad device list '%s' (e.g. 0,1,2 or 0-7)\n", v); return false; } continue; }
        int* t = k == "--width" ? &o.width : k == "--height" ? &o.height : k == "--frames" ? &o.frames : k == "--steps" ? &o.steps : k == "--warmup" ? &o.warmup : k == "--inner" ? &o.inner :
                 k == "--method" ? &o.method : k == "--subme" ? &o.subme : k == "--merange" ? &o.merange : k == "--qp" ? &o.qp : k == "--refs" ? &o.refs : k == "--rect" ? &o.rect :
                 k == "--streams" ? &o.streams : nullptr;
        if (!t) { fprintf(stderr, "multi_gpu_host: unknown option %s\n", k.c_str()); return false; }
        char* e; const long n = strtol(v, &e, 10);
        if (*e || e == v) { fprintf(stderr, "multi_gpu_host: %s takes an integer, not '%s'\n", k.c_str(), v); return false; }
        *t = (int)n;
    }
    if (o.devices.empty()) { fprintf(stderr, "multi_gpu_host: --devices is required\n"); return false; }
    if (o.frames < 1 || o.steps < 1 || o.warmup < 0 || o.inner < 1 || o.width < 64 || o.height < 64 || (o.width & 63) || (o.height & 63) || o.refs < 1)
    { fprintf(stderr, "multi_gpu_host: width / height are multiples of 64, frames / steps / inner / refs >= 1\n"); return false; }
    return true;
}

// ---- synthetic pictures (integer arithmetic only: the same bytes on every host): a multi-octave value-noise texture in global full-pel motion + per-picture noise ----
inline uint32_t hash3(uint32_t a, uint32_t b, uint32_t c) { uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du; h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15; return h; }
inline int lattice(int x, int y, int cell, uint32_t seed)
{   // bilinear interpolation of a hash lattice with spacing `cell`: 0 .. 65535
    const int cx = x / cell, cy = y / cell, fx = x - cx * cell, fy = y - cy * cell;
    const int v00 = (int)(hash3((uint32_t)cx, (uint32_t)cy, seed) >> 16), v10 = (int)(hash3((uint32_t)cx + 1, (uint32_t)cy, seed) >> 16);
    const int v01 = (int)(hash3((uint32_t)cx, (uint32_t)cy + 1, seed) >> 16), v11 = (int)(hash3((uint32_t)cx + 1, (uint32_t)cy + 1, seed) >> 16);
    const int64_t top = (int64_t)v00 * (cell - fx) + (int64_t)v10 * fx, bot = (int64_t)v01 * (cell - fx) + (int64_t)v11 * fx;
    return (int)((top * (cell - fy) + bot * fy) / ((int64_t)cell * cell));
}
template<class P> void make_pair(int w, int h, int depth, int index, std::vector<P>& cur, std::vector<P>& ref)
{
    const uint32_t seed = 0x5EED0000u + (uint32_t)index;
    const int pm = (1 << depth) - 1, dx = (int)(hash3(seed, 1, 2) % 29u) - 14, dy = (int)(hash3(seed, 3, 4) % 29u) - 14, off = 64;
    cur.resize((size_t)w * h); ref.resize((size_t)w * h);
    auto tex = [&](int x, int y) {
        x += off; y += off;
        const int64_t v = 40 * (int64_t)lattice(x, y, 64, seed) + 25 * (int64_t)lattice(x, y, 32, seed + 1) + 18 * (int64_t)lattice(x, y, 16, seed + 2) + 12 * (int64_t)lattice(x, y, 8, seed + 3) +
                          5 * (int64_t)lattice(x, y, 3, seed + 4);
        return (int)(v * pm / (100 * 65535ll));
    };
    const int amp = 2 << (depth - 8);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            const uint32_t n = hash3((uint32_t)x, (uint32_t)y, seed + 9);
            const int c = tex(x, y) + (int)(n % (uint32_t)(2 * amp + 1)) - amp, r = tex(x - dx, y - dy) + (int)((n >> 12) % (uint32_t)(2 * amp + 1)) - amp;      // ref(x + dx, y + dy) = cur(x, y)
            cur[(size_t)y * w + x] = (P)(c < 0 ? 0 : c > pm ? pm : c);
            ref[(size_t)y * w + x] = (P)(r < 0 ? 0 : r > pm ? pm : r);
        }
}

uint64_t fnv(uint64_t h, const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } return h; }

struct Result { int rc = 0; double msPerStep = 0, seconds = 0; uint64_t digest = 0; std::string error; };

// everything one device does; `ready` / `go`: the worker reports that its batch is warm, waits to be released, runs the timed steps, reports again
template<class Ready, class Go>
void run_device(const Api& api, const Options& o, int worker, Ready ready, Go go, Result& out)
{
    const int dev = o.devices[(size_t)worker];
    auto fail = [&](const char* what, int rc) { out.rc = rc == X265HIP_EDEVICE ? 3 : 1; out.error = std::string(what) + " on device " + std::to_string(dev) + ": " + api.last_error(); };
    x265hip_ctx* ctx = nullptr; x265hip_batch* b = nullptr;
    int rc = api.ctx_create(dev, &ctx);
    if (rc) { fail("x265hip_ctx_create", rc); ready(); go(); ready(); return; }
    x265hip_batch_desc d = {};
    d.width = o.width; d.height = o.height; d.frames = o.frames; d.margin = 96; d.qp = o.qp; d.merange = o.merange; d.method = o.method; d.subme = o.subme; d.tuLog2 = 5; d.usePlanes = 1;
    d.refs = o.refs; d.rect = o.rect; d.streams = o.streams;
    if ((rc = api.batch_create(ctx, &d, &b))) { fail("x265hip_batch_create", rc); api.ctx_destroy(ctx); ready(); go(); ready(); return; }
    const int depth = api.bit_depth();
    auto upload = [&]() -> int {
        if (!o.input.empty())
        {
            const size_t pic = (size_t)o.width * o.height * (depth == 8 ? 1 : 2);
            std::vector<char> in(pic * 2 * (size_t)o.frames);
            FILE* f = fopen(o.input.c_str(), "rb");
            if (!f || fread(in.data(), 1, in.size(), f) != in.size()) { if (f) fclose(f); fprintf(stderr, "multi_gpu_host: cannot read %zu bytes of %s\n", in.size(), o.input.c_str()); return X265HIP_EARG; }
            fclose(f);
            for (int k = 0; k < o.frames; k++)
            {
                int r = api.batch_upload_plane(b, 0, k, in.data() + (size_t)(2 * k) * pic, o.width);
                for (int q = 0; !r && q < o.refs; q++) r = api.batch_upload_plane(b, 1 + q, k, in.data() + (size_t)(2 * k + 1) * pic, o.width);
                if (r) return r;
            }
            return api.ctx_sync(ctx);                       // the copies are asynchronous: `in` must outlive them
        }
        for (int k = 0; k < o.frames; k++)
        {
            const int index = (o.sameFrames ? 0 : worker * o.frames) + k;
            int r = 0;
            if (depth == 8)
            {
                std::vector<uint8_t> c, f; make_pair(o.width, o.height, depth, index, c, f);
                if ((r = api.batch_upload_plane(b, 0, k, c.data(), o.width))) return r;
                for (int q = 0; q < o.refs; q++) if ((r = api.batch_upload_plane(b, 1 + q, k, f.data(), o.width))) return r;
                if ((r = api.ctx_sync(ctx))) return r;      // the copies are asynchronous: the pictures must outlive them
            }
            else
            {
                std::vector<uint16_t> c, f; make_pair(o.width, o.height, depth, index, c, f);
                if ((r = api.batch_upload_plane(b, 0, k, c.data(), o.width))) return r;
                for (int q = 0; q < o.refs; q++) if ((r = api.batch_upload_plane(b, 1 + q, k, f.data(), o.width))) return r;
                if ((r = api.ctx_sync(ctx))) return r;
            }
        }
        return 0;
    };
    if ((rc = upload())) fail("x265hip_batch_upload_plane", rc);
    for (int i = 0; !out.rc && i < o.warmup * o.inner; i++) if ((rc = api.batch_step(b))) fail("x265hip_batch_step", rc);
    if (!out.rc && (rc = api.ctx_sync(ctx))) fail("x265hip_ctx_sync", rc);
    ready(); go();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; !out.rc && i < o.steps * o.inner; i++) if ((rc = api.batch_step(b))) fail("x265hip_batch_step", rc);
    if (!out.rc && (rc = api.ctx_sync(ctx))) fail("x265hip_ctx_sync", rc);
    out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    out.msPerStep = out.seconds * 1e3 / o.steps;
    ready();
    if (!out.rc)
    {   // digest of what the last step left: MV records of the four levels, numSig, coefficients
        uint64_t h = 1469598103934665603ull;
        const int levels[4] = { 64, 32, 16, 8 };
        for (int lv : levels)
        {
            std::vector<x265hip_me_result> r((size_t)api.batch_task_count(&d, lv));
            if ((rc = api.batch_read_results(b, lv, r.data()))) { fail("x265hip_batch_read_results", rc); break; }
            h = fnv(h, r.data(), r.size() * sizeof(x265hip_me_result));
        }
        if (!out.rc)
        {
            const size_t ntu = (size_t)api.batch_tu_count(&d);
            std::vector<uint32_t> ns(ntu); std::vector<int16_t> co(ntu << 10);
            if ((rc = api.batch_read_coeffs(b, co.data(), ns.data()))) fail("x265hip_batch_read_coeffs", rc);
            else { h = fnv(h, ns.data(), ns.size() * 4); h = fnv(h, co.data(), co.size() * 2); }
        }
        out.digest = h;
    }
    api.batch_destroy(b); api.ctx_destroy(ctx);
}

void report(const Options& o, const std::vector<Result>& res, double jobSeconds)
{
    int rc = 0;
    for (const Result& r : res) if (r.rc) { fprintf(stderr, "multi_gpu_host: %s\n", r.error.c_str()); rc = rc == 3 ? 3 : r.rc; }
    if (rc) exit(rc);
    const double px = (double)o.width * o.height * o.frames * o.inner;
    const double mpx = (double)res.size() * px * o.steps / jobSeconds / 1e6;
    printf("{\"metric\": \"Mpixels/s ME+DCT+quant\", \"host\": \"c++ (%s per device)\", \"n_gpus\": %zu, \"devices\": [", o.procs ? "process" : "thread", res.size());
    for (size_t i = 0; i < o.devices.size(); i++) printf("%s%d", i ? ", " : "", o.devices[i]);
    printf("], \"value\": %.1f, \"unit\": \"Mpixels/s\", \"steps\": %d, \"warmup\": %d, \"inner\": %d, \"frames_per_device\": %d, \"width\": %d, \"height\": %d, \"ms_per_step\": %.4f, \"scaling\": \"weak\", "
           "\"per_device_ms_per_step\": [", mpx, o.steps, o.warmup, o.inner, o.frames, o.width, o.height, jobSeconds * 1e3 / o.steps);
    for (size_t i = 0; i < res.size(); i++) printf("%s%.4f", i ? ", " : "", res[i].msPerStep);
    printf("], \"digests\": [");
    for (size_t i = 0; i < res.size(); i++) printf("%s\"%016llx\"", i ? ", " : "", (unsigned long long)res[i].digest);
    printf("], \"same_frames\": %s}\n", o.sameFrames ? "true" : "false");
}

// ---- one thread per device ----
int run_threads(const Api& api, const Options& o)
{
    const int n = (int)o.devices.size();
    std::vector<Result> res((size_t)n);
    std::mutex mu; std::condition_variable cv; int arrived = 0, generation = 0; bool released = false;
    auto barrier = [&]() {          // all workers + the main thread
        std::unique_lock<std::mutex> lk(mu);
        const int gen = generation;
        if (++arrived == n + 1) { arrived = 0; generation++; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    };
    (void)released;
    std::vector<std::thread> th;
    for (int k = 0; k < n; k++)
        th.emplace_back([&, k] { run_device(api, o, k, barrier, barrier, res[(size_t)k]); });
    barrier();                                              // every batch is warm
    const auto t0 = std::chrono::steady_clock::now();
    barrier();                                              // go
    barrier();                                              // every device is done
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (auto& t : th) t.join();
    report(o, res, secs);
    return 0;
}

// ---- one forked process per device (a process per GPU is how a frame-parallel encoder farm runs; the library is loaded AFTER the fork) ----
int run_procs(const Options& o)
{
    const int n = (int)o.devices.size();
    struct Child { pid_t pid; int toChild[2], fromChild[2]; };
    std::vector<Child> ch((size_t)n);
    for (int k = 0; k < n; k++)
    {
        if (pipe(ch[(size_t)k].toChild) || pipe(ch[(size_t)k].fromChild)) { perror("pipe"); return 1; }
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); return 1; }
        if (pid == 0)
        {
            close(ch[(size_t)k].toChild[1]); close(ch[(size_t)k].fromChild[0]);
            const int in = ch[(size_t)k].toChild[0], outfd = ch[(size_t)k].fromChild[1];
            Api api{}; void* lib = nullptr;
            Result r;
            if (!load(o.lib.c_str(), api, lib)) { r.rc = 2; r.error = "library"; }
            auto ready = [&] { char c = 'R'; if (write(outfd, &c, 1) != 1) _exit(1); };
            auto go = [&] { char c; if (read(in, &c, 1) != 1) _exit(1); };
            if (!r.rc) run_device(api, o, k, ready, go, r);
            else { ready(); go(); ready(); }
            char msg[512];
            int len = snprintf(msg, sizeof(msg), "%d %.9f %.9f %llu %s", r.rc, r.msPerStep, r.seconds, (unsigned long long)r.digest, r.error.c_str());
            if (len < 0) len = 0;
            if (len > (int)sizeof(msg) - 1) len = (int)sizeof(msg) - 1;           // snprintf returns the untruncated length: a long error text is cut, not read past
            if (write(outfd, msg, (size_t)len + 1) != len + 1) _exit(1);
            _exit(0);
        }
        ch[(size_t)k].pid = pid;
        close(ch[(size_t)k].toChild[0]); close(ch[(size_t)k].fromChild[1]);
    }
    auto wait_all = [&] { for (auto& c : ch) { char b; if (read(c.fromChild[0], &b, 1) != 1) { fprintf(stderr, "multi_gpu_host: a worker process died\n"); exit(1); } } };
    wait_all();
    const auto t0 = std::chrono::steady_clock::now();
    for (auto& c : ch) { char g = 'G'; if (write(c.toChild[1], &g, 1) != 1) return 1; }
    wait_all();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<Result> res((size_t)n);
    for (int k = 0; k < n; k++)
    {
        char msg[512]; size_t got = 0;
        while (got < sizeof(msg) - 1) { const ssize_t r = read(ch[(size_t)k].fromChild[0], msg + got, sizeof(msg) - 1 - got); if (r <= 0) break; got += (size_t)r; if (msg[got - 1] == 0) break; }
        msg[got] = 0;
        unsigned long long dg = 0; int used = 0;
        if (sscanf(msg, "%d %lf %lf %llu %n", &res[(size_t)k].rc, &res[(size_t)k].msPerStep, &res[(size_t)k].seconds, &dg, &used) < 4) { res[(size_t)k].rc = 1; res[(size_t)k].error = "worker process sent no result"; }
        else { res[(size_t)k].digest = dg; res[(size_t)k].error = msg + used; }
        int st; waitpid(ch[(size_t)k].pid, &st, 0);
    }
    report(o, res, secs);
    return 0;
}
}

int main(int argc, char** argv)
{
    Options o;
    if (!parse(argc, argv, o))
    {
        fprintf(stderr, "usage: %s libx265hip_N.so --devices 0,1,... [--width W --height H --frames F --steps K --warmup W --inner I --method M --subme S --merange R --qp Q --refs N --rect 0|1 --streams S --same-frames --input in.raw --procs]\n", argv[0]);
        return 2;
    }
    if (o.procs) return run_procs(o);
    Api api{}; void* lib = nullptr;
    if (!load(o.lib.c_str(), api, lib)) return 2;
    return run_threads(api, o);
}
