// stream_launcher.cpp -- BASELINE configs[3]: N independent encodes, one per GPU of a node, started together; aggregate frames per second when the last one ends.
//
//   stream_launcher --devices 0,1,...,7 [--env NAME] -- <encoder command ...>
//
// One copy of the command per listed device.  In every argument `{dev}` is replaced by the device number and `{k}` by the stream's index (output files, logs); the
// device also goes into the environment as NAME (default X265TME_DEVICE, what the end-to-end driver oracle/_ref/x265e2e_<depth> and any host of the three adapters
// hands to x265hip_*_adapter_load) and as X265HIP_DEVICE.  Every stream's stdout is captured; the last line that holds "frames": F and "seconds": S (the driver's JSON
// line) gives its frame count.  The launcher's own clock runs from the first fork to the last exit:
//   aggregate_fps = sum of frames / wall seconds           (what a node delivers on N streams)
// and the streams' own fps are listed beside it.  Pure POSIX: no GPU library is loaded here.
// Exit codes: 0 every stream exited 0; 2 usage; 1 a stream failed (its status and the tail of its output are printed).
#include <sys/wait.h>
#include <unistd.h>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {
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
std::string subst(std::string s, const char* key, int value)
{
    const std::string v = std::to_string(value);
    for (size_t at = s.find(key); at != std::string::npos; at = s.find(key, at + v.size())) s.replace(at, strlen(key), v);
    return s;
}
// the number behind "name": in the LAST line of `text` that has it
bool last_number(const std::string& text, const char* name, double& out)
{
    const std::string key = std::string("\"") + name + "\":";
    const size_t at = text.rfind(key);
    if (at == std::string::npos) return false;
    char* e; const char* p = text.c_str() + at + key.size();
    out = strtod(p, &e);
    return e != p;
}
}

int main(int argc, char** argv)
{
    std::vector<int> devices; std::string envName = "X265TME_DEVICE";
    int i = 1;
    for (; i < argc; i++)
    {
        const std::string k = argv[i];
        if (k == "--") { i++; break; }
        if (k == "--devices" && i + 1 < argc) { if (!parse_devices(argv[++i], devices)) { fprintf(stderr, "stream_launcher: bad device list '%s' (e.g. 0,1,2 or 0-7)\n", argv[i]); return 2; } }
        else if (k == "--env" && i + 1 < argc) envName = argv[++i];
        else { fprintf(stderr, "stream_launcher: unknown option %s\n", k.c_str()); return 2; }
    }
    if (devices.empty() || i >= argc) { fprintf(stderr, "usage: %s --devices 0,1,... [--env NAME] -- command [args with {dev} / {k}]\n", argv[0]); return 2; }
    struct Stream { pid_t pid; int fd; int dev; std::string out; int status; bool done; double seconds; };
    std::vector<Stream> st(devices.size());
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t k = 0; k < devices.size(); k++)
    {
        int p[2];
        if (pipe(p)) { perror("pipe"); return 1; }
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); return 1; }
        if (pid == 0)
        {
            close(p[0]); dup2(p[1], 1); close(p[1]);
            const std::string d = std::to_string(devices[k]);
            setenv(envName.c_str(), d.c_str(), 1); setenv("X265HIP_DEVICE", d.c_str(), 1);
            std::vector<std::string> args;
            for (int a = i; a < argc; a++) args.push_back(subst(subst(argv[a], "{dev}", devices[k]), "{k}", (int)k));
            std::vector<char*> av;
            for (auto& s : args) av.push_back(&s[0]);
            av.push_back(nullptr);
            execvp(av[0], av.data());
            fprintf(stderr, "stream_launcher: cannot run %s: %s\n", av[0], strerror(errno));
            _exit(127);
        }
        close(p[1]);
        st[k] = Stream{ pid, p[0], devices[k], std::string(), 0, false, 0 };
    }
    // drain the streams' output one after the other (each pipe is read to its end; a stream that fills its pipe only waits for its turn -- the encodes print one line)
    for (auto& s : st)
    {
        char buf[4096]; ssize_t r;
        while ((r = read(s.fd, buf, sizeof(buf))) > 0) s.out.append(buf, (size_t)r);
        close(s.fd);
        waitpid(s.pid, &s.status, 0);
        s.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int failed = 0; double frames = 0;
    std::string list;
    for (size_t k = 0; k < st.size(); k++)
    {
        const Stream& s = st[k];
        const bool ok = WIFEXITED(s.status) && WEXITSTATUS(s.status) == 0;
        double f = 0, secs = 0, fps = 0;
        const bool parsed = last_number(s.out, "frames", f) && last_number(s.out, "seconds", secs);
        (void)last_number(s.out, "fps", fps);
        if (!ok || !parsed)
        {
            failed++;
            const std::string tail = s.out.size() > 600 ? s.out.substr(s.out.size() - 600) : s.out;
            fprintf(stderr, "stream_launcher: stream %zu (device %d) %s (status %d)%s\n%s\n", k, s.dev, ok ? "printed no \"frames\" / \"seconds\"" : "failed",
                    WIFEXITED(s.status) ? WEXITSTATUS(s.status) : -1, "", tail.c_str());
            continue;
        }
        frames += f;
        char item[160];
        snprintf(item, sizeof(item), "%s{\"device\": %d, \"frames\": %.0f, \"seconds\": %.3f, \"fps\": %.3f}", list.empty() ? "" : ", ", s.dev, f, secs, fps > 0 ? fps : f / secs);
        list += item;
    }
    if (failed) return 1;
    printf("{\"streams\": %zu, \"frames\": %.0f, \"wall_seconds\": %.6f, \"aggregate_fps\": %.3f, \"per_stream\": [%s]}\n", st.size(), frames, wall, frames / wall, list.c_str());
    return 0;
}
