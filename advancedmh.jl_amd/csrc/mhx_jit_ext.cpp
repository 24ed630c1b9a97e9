// mhx_jit_ext.cpp -- see mhx_jit_ext.h
#include "mhx_jit_ext.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include <errno.h>
#include <fcntl.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

extern char** environ;

namespace {

struct ext_compiler {
    std::string path, root, id;
};

ext_compiler find_compiler()
{
    ext_compiler c;
    std::vector<std::string> cand;
    if (const char* e = getenv("MHX_JIT_CLANG")) {
        if (!*e || !strcmp(e, "0")) return c;
        cand.push_back(e);
    } else {
        if (const char* r = getenv("ROCM_PATH")) if (*r) cand.push_back(std::string(r) + "/lib/llvm/bin/clang++");
        cand.push_back("/opt/rocm/lib/llvm/bin/clang++");
    }
    for (const std::string& p : cand) {
        struct stat st;
        if (stat(p.c_str(), &st) != 0 || !S_ISREG(st.st_mode) || access(p.c_str(), X_OK) != 0) continue;
        c.path = p;
        const std::string tail = "/lib/llvm/bin/clang++";
        if (p.size() > tail.size() && p.compare(p.size() - tail.size(), tail.size(), tail) == 0) c.root = p.substr(0, p.size() - tail.size());
        char b[96];
        snprintf(b, sizeof b, ":%lld:%lld", (long long)st.st_size, (long long)st.st_mtime);
        c.id = "clang++:" + p + b;
        return c;
    }
    return c;
}

const ext_compiler& compiler()
{
    static std::once_flag once;
    static ext_compiler c;
    std::call_once(once, [] { c = find_compiler(); });
    return c;
}

bool write_file(const std::string& path, const char* data, size_t n)
{
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(data, 1, n, f) == n;
    return fclose(f) == 0 && ok;
}

bool read_file(const std::string& path, std::string* out, size_t limit)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0 && out->size() < limit) out->append(buf, n);
    fclose(f);
    return true;
}

}  // namespace

const std::string& mhx_jit_ext_identity() { return compiler().id; }

bool mhx_jit_ext_compile(const std::string& source, const char* const* hdr_src, const char* const* hdr_name, int nhdr,
                         const std::vector<std::string>& opts, std::vector<char>* code, std::string* log)
{
    const ext_compiler& c = compiler();
    if (c.path.empty()) { if (log) *log = "no offline compiler"; return false; }
    const char* tmp = getenv("TMPDIR");
    std::string dir = std::string(tmp && *tmp ? tmp : "/tmp") + "/mhx-jit-XXXXXX";
    std::vector<char> dbuf(dir.begin(), dir.end());
    dbuf.push_back('\0');
    if (!mkdtemp(dbuf.data())) { if (log) *log = std::string("mkdtemp: ") + strerror(errno); return false; }
    dir = dbuf.data();
    std::vector<std::string> files;
    bool ok = true;
    for (int i = 0; i < nhdr && ok; ++i) {
        files.push_back(dir + "/" + hdr_name[i]);
        ok = write_file(files.back(), hdr_src[i], strlen(hdr_src[i]));
    }
    const std::string src = dir + "/mhx_jit.hip", obj = dir + "/mhx_jit.hsaco", out = dir + "/log.txt";
    files.push_back(src); files.push_back(obj); files.push_back(out);
    ok = ok && write_file(src, source.data(), source.size());
    int status = -1;
    if (ok) {
        // MHX_JIT_BUILD: what __HIPCC_RTC__ tells the headers under hiprtc -- a run-time module, no namespace around the engine
        std::vector<std::string> args = {c.path, "-x", "hip", "--offload-device-only", "--no-gpu-bundle-output", "-DMHX_JIT_BUILD=1", "-I" + dir};
        if (!c.root.empty()) args.push_back("--rocm-path=" + c.root);
        for (const std::string& o : opts) args.push_back(o);
        args.push_back("-o"); args.push_back(obj); args.push_back(src);
        std::vector<char*> argv;
        for (std::string& a : args) argv.push_back(&a[0]);
        argv.push_back(nullptr);
        posix_spawn_file_actions_t fa;
        posix_spawn_file_actions_init(&fa);
        posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
        posix_spawn_file_actions_addopen(&fa, 1, out.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
        posix_spawn_file_actions_adddup2(&fa, 1, 2);
        pid_t pid = 0;
        const int rc = posix_spawn(&pid, c.path.c_str(), &fa, nullptr, argv.data(), environ);
        posix_spawn_file_actions_destroy(&fa);
        if (rc != 0) { ok = false; if (log) *log = std::string("posix_spawn: ") + strerror(rc); }
        else {
            int w;
            while ((w = waitpid(pid, &status, 0)) < 0 && errno == EINTR) {}
            // (a host that ignores SIGCHLD reaps the child itself: ECHILD -- the object file then says how it went)
            ok = w < 0 ? errno == ECHILD : (WIFEXITED(status) && WEXITSTATUS(status) == 0);
        }
    }
    if (ok) {
        std::string bin;
        ok = read_file(obj, &bin, (size_t)1 << 30) && bin.size() > 16 && memcmp(bin.data(), "\x7f" "ELF", 4) == 0;
        if (ok) code->assign(bin.begin(), bin.end());
    }
    if (!ok && log && log->empty()) { (void)read_file(out, log, 4000); if (log->empty()) *log = "the offline compiler failed"; }
    for (const std::string& f : files) (void)unlink(f.c_str());
    (void)rmdir(dir.c_str());
    return ok;
}
