// rt_jit.hip — per-scene instances compiled at run time (the generalisation of RT_BOX_SIGNATURES).
//
// The reference gets its specialisation from Taichi's JIT: `ti.static(range(len(OBJECTS)))` unrolls the object loop and
// picks each object's shape function at compile time (src/scene.py:44-56).  Here the ahead-of-time library carries
// general instances plus one listed rotation signature; for any OTHER scene of <= 8 analytic shapes the complete-path
// kernels are compiled on demand by `hipcc --genco` from the same sources (rt_jit_tu.hip: object count, shape types,
// rotation classes and the culling decision as compile-time constants; ~2 s), cached as a code object under
// $RTPBR_JIT_CACHE / $XDG_CACHE_HOME/rtpbr / ~/.cache/rtpbr keyed by those constants and a hash of the sources, loaded
// with hipModuleLoadData and launched with hipModuleLaunchKernel.  Results are bit-identical to the ahead-of-time
// instances (same arithmetic).  If hipcc or the sources are not available the library keeps using the ahead-of-time
// instance (option "jit": -1 = when no ahead-of-time specialisation serves the scene, 0 = never, 1 = always with that
// fallback, 2 = always and an error otherwise).
//
// Trust: a code object is executed inside the caller's GPU context, so the cache must not be writable by anybody
// else.  The cache directory is created with mode 0700 and is only used when lstat() shows a real directory owned by
// this user with no group/other write bit; a code object is read through one O_NOFOLLOW descriptor that fstat() shows
// to be a regular file of this user (no check-then-open race).  When $HOME is unusable the fallback is the PER-USER
// /tmp/rtpbr-cache-<uid>, held to the same checks; if no candidate passes, run-time compilation is off and the
// ahead-of-time kernels serve the scene.
//
// Concurrency: compilation (~2 s) runs OUTSIDE the process-wide lock behind a per-key "building" marker — other keys,
// contexts and threads proceed; threads that want the same key wait for it.  Only deterministic failures (hipcc ran
// and rejected the unit, sources missing) are remembered; transient ones (fork, disk, a stale object that was
// removed) are retried by the next call.  Loaded modules are reference-counted by the contexts that use them and
// the least recently used unpinned ones are unloaded beyond RTPBR_JIT_MAX_MODULES (default 64); baked code objects on
// disk are pruned, oldest first, beyond RTPBR_JIT_CACHE_MAX files (default 512).
#include <dirent.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "rt_ctx.hpp"

using namespace rt;

extern "C" const char* rtpbr_last_error(void);

namespace {
// process-wide registry: contexts on the same device share code objects
enum { E_BUILDING = 0, E_READY = 1, E_FAILED = 2 };
struct Entry {
    int state = E_BUILDING;
    RtJitModule* mod = nullptr;
    std::string error;        // E_FAILED: the (deterministic) reason
};
std::mutex g_mu;
std::condition_variable g_cv;
std::map<std::string, Entry> g_modules;
unsigned long long g_tick = 0;   // LRU clock

std::string lib_dir() {
    Dl_info info;
    if (!dladdr((void*)&rt_jit_acquire, &info) || !info.dli_fname) return ".";
    std::string p = info.dli_fname;
    size_t k = p.rfind('/');
    return k == std::string::npos ? "." : p.substr(0, k);
}

bool read_file(const std::string& path, std::vector<char>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    bool ok = n >= 0 && fread(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    return ok;
}

// FNV-1a over the sources the translation unit is made of: a changed source invalidates the cache
bool source_hash(const std::string& dir, uint64_t* h) {
    const char* files[] = {"rt_jit_tu.hip", "rt_trace.hpp", "rt_persistent.hpp", "rt_split.hpp", "rt_chain.hpp", "rt_device.hpp", "rt_types.hpp", "rt_math.hpp", "../../include/rtpbr.h"};
    uint64_t x = 1469598103934665603ull;
    std::vector<char> buf;
    for (const char* f : files) {
        if (!read_file(dir + "/" + f, buf)) return false;
        for (char c : buf) x = (x ^ (unsigned char)c) * 1099511628211ull;
    }
    // ... and the command line compile() builds is part of the product too: bump when its flags change
    for (const char* c = "flags-r5c"; *c; c++) x = (x ^ (unsigned char)*c) * 1099511628211ull;
    *h = x;
    return true;
}

// A code object from the cache: ONE descriptor, opened without following a symlink, that fstat() shows to be a regular
// file owned by this user and not writable by group or others — what is checked is what is read.
// (or_root: a file of the CATALOG that ships with the library may also belong to root — whoever installed the package)
bool read_owned_file(const std::string& path, std::vector<char>& out, bool or_root = false) {
    const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat st;
    bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && (st.st_uid == getuid() || (or_root && st.st_uid == 0)) && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0 && st.st_size > 0;
    if (ok) {
        out.resize((size_t)st.st_size);
        size_t got = 0;
        while (got < out.size()) {
            const ssize_t k = read(fd, out.data() + got, out.size() - got);
            if (k <= 0) break;
            got += (size_t)k;
        }
        ok = got == out.size();
    }
    close(fd);
    return ok;
}

// mkdir -p; the LEAF is created 0700.  Then the leaf must be a real directory (not a symlink) of this user that nobody
// else can write to — a directory somebody else prepared (or can write into) is never used.
bool secure_dir(const std::string& d) {
    for (size_t i = 1; i <= d.size(); i++)
        if (i == d.size() || d[i] == '/') (void)mkdir(d.substr(0, i).c_str(), i == d.size() ? 0700 : 0755);
    struct stat st;
    if (lstat(d.c_str(), &st) != 0) return false;
    return S_ISDIR(st.st_mode) && st.st_uid == getuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0 && access(d.c_str(), W_OK | X_OK) == 0;
}

// The first candidate that passes secure_dir(): $RTPBR_JIT_CACHE (the only candidate when set), $XDG_CACHE_HOME/rtpbr,
// ~/.cache/rtpbr, then the per-user /tmp/rtpbr-cache-<uid> (read-only or missing home).  Empty = run-time compilation off.
std::string cache_dir() {
    std::vector<std::string> cand;
    if (const char* e = getenv("RTPBR_JIT_CACHE")) {
        cand.push_back(e);
    } else {
        if (const char* x = getenv("XDG_CACHE_HOME")) cand.push_back(std::string(x) + "/rtpbr");
        else if (const char* h = getenv("HOME")) cand.push_back(std::string(h) + "/.cache/rtpbr");
        char alt[64];
        snprintf(alt, sizeof alt, "/tmp/rtpbr-cache-%d", (int)getuid());
        cand.push_back(alt);
    }
    for (const std::string& d : cand)
        if (!d.empty() && secure_dir(d)) return d;
    return "";
}

// keep at most RTPBR_JIT_CACHE_MAX (default 512) code objects in the cache: a baked instance exists per (scene, config),
// an animation bakes one per frame.  Oldest (mtime) first; files of other processes' builds in flight (.tmp.) are left alone.
void prune_cache(const std::string& dir) {
    long cap = 512;
    if (const char* e = getenv("RTPBR_JIT_CACHE_MAX")) cap = atol(e);
    if (cap < 1) cap = 1;
    DIR* d = opendir(dir.c_str());
    if (!d) return;
    std::vector<std::pair<long long, std::string>> files;
    while (dirent* e = readdir(d)) {
        const std::string n = e->d_name;
        if (n.size() < 7 || n.compare(n.size() - 6, 6, ".hsaco") != 0) continue;
        struct stat st;
        if (lstat((dir + "/" + n).c_str(), &st) == 0 && S_ISREG(st.st_mode))
            files.push_back({(long long)st.st_mtim.tv_sec * 1000000000LL + st.st_mtim.tv_nsec, n});
    }
    closedir(d);
    if ((long)files.size() <= cap) return;
    std::sort(files.begin(), files.end());
    for (size_t i = 0; i + (size_t)cap < files.size(); i++) unlink((dir + "/" + files[i].second).c_str());
}

std::string hipcc_path() {
    if (const char* e = getenv("HIPCC")) return e;
    if (access("/opt/rocm/bin/hipcc", X_OK) == 0) return "/opt/rocm/bin/hipcc";
    return "hipcc";
}

// fork/exec, no shell: the arguments are ours, but paths come from the environment
int run(const std::vector<std::string>& argv, const std::string& log) {
    pid_t pid = fork();
    if (pid < 0) return -1;
    if (pid == 0) {
        FILE* f = fopen(log.c_str(), "w");
        if (f) {
            dup2(fileno(f), 1);
            dup2(fileno(f), 2);
        }
        std::vector<char*> a;
        for (const std::string& s : argv) a.push_back(const_cast<char*>(s.c_str()));
        a.push_back(nullptr);
        execvp(a[0], a.data());
        _exit(127);
    }
    int st = 0;
    for (;;) {
        if (waitpid(pid, &st, 0) >= 0) break;
        if (errno == EINTR) continue;
        // ECHILD: the host ignores SIGCHLD (or reaps children itself), the exit status is gone — the caller decides by
        // whether the output file exists
        return errno == ECHILD ? -2 : -1;
    }
    return WIFEXITED(st) ? WEXITSTATUS(st) : -1;
}
}  // namespace

// where the catalog lives: $RTPBR_JIT_CATALOG, else <library directory>/../data/jit
std::string rt_jit_catalog_dir() {
    if (const char* e = getenv("RTPBR_JIT_CATALOG")) return e;
    return lib_dir() + "/../data/jit";
}

static uint64_t baked_hash(const RtJitKey& key) {
    if (!key.baked) return 0;
    uint64_t th = 1469598103934665603ull;
    for (int i = 0; i < key.n_obj * 16; i++) th = (th ^ key.table[i]) * 1099511628211ull;
    for (size_t k = 0; k < sizeof(rtpbr_config) / 4; k++) th = (th ^ key.cfg_words[k]) * 1099511628211ull;
    for (int k = 0; k < 4; k++) th = (th ^ key.extra[k]) * 1099511628211ull;
    for (int k = 0; k < 8; k++) th = (th ^ (unsigned)key.ints[k]) * 1099511628211ull;
    if (key.baked == 2)
        for (int k = 0; k < 21; k++) th = (th ^ key.cam_words[k]) * 1099511628211ull + 2;
    return th;
}

// Compile (or fetch from the cache) the code object of `key`; path of the .hsaco in *out.  Needs no device.
// *deterministic (optional) is set when a failure would repeat on every call (hipcc rejected the unit, sources missing).
int rt_jit_build(const RtJitKey& key, std::string* out, bool* deterministic) {
    if (deterministic) *deterministic = false;
    const std::string dir = lib_dir();
    uint64_t sh = 0;
    if (!source_hash(dir, &sh)) {
        if (deterministic) *deterministic = true;
        return rt_fail(RTPBR_ESTATE, "run-time compilation: kernel sources not found next to the library (%s)", dir.c_str());
    }
    uint64_t th = baked_hash(key);
    // RTPBR_JIT_EXTRA_FLAGS: extra compiler arguments (space separated) for experiments and instrumented builds
    // (e.g. -DRT_DEBUG_PHASE); part of the cache key
    std::vector<std::string> extra;
    if (const char* e = getenv("RTPBR_JIT_EXTRA_FLAGS")) {
        std::string cur;
        for (const char* q = e;; q++) {
            if (*q == ' ' || *q == 0) {
                if (!cur.empty()) extra.push_back(cur);
                cur.clear();
                if (!*q) break;
            } else
                cur += *q;
        }
        for (const std::string& f : extra)
            for (char ch : f) th = (th ^ (unsigned char)ch) * 1099511628211ull + 1;
    }
    char name[256];
    snprintf(name, sizeof name, "k%d_n%d_t%llx_s%x_c%d_w%d_f%d%s%s_b%016llx_%016llx", key.kind, key.n_obj, (unsigned long long)key.types, key.sig,
             key.cull, key.waves, key.form, key.fast ? "_fast" : "", key.dense ? "_dense" : "", (unsigned long long)th, (unsigned long long)sh);
    const std::string cdir = cache_dir();
    const std::string path = cdir + "/" + name + ".hsaco";
    if (!cdir.empty()) {
        std::vector<char> probe;
        if (read_owned_file(path, probe)) {      // a cache hit must pass the same ownership test the loader applies
            (void)utimensat(AT_FDCWD, path.c_str(), nullptr, 0);      // pruning is by mtime: make it follow use
            *out = path;
            return RTPBR_OK;
        }
    }
    {
        // the CATALOG shipped next to the library (raytracingpbr_amd/data/jit, read-only): code objects that
        // __graft_entry__.build() / `python -m raytracingpbr_amd.prebuild` compiled ahead of time for the BASELINE scenes — same
        // key, same name (the hash of the kernel sources is part of it: a stale catalog is never picked up).  A target without
        // hipcc runs the scene-specialised kernels from here; nothing is ever written to it at run time.
        const std::string shipped = rt_jit_catalog_dir() + "/" + name + ".hsaco";
        std::vector<char> probe;
        if (read_owned_file(shipped, probe, true)) {
            *out = shipped;
            return RTPBR_OK;
        }
    }
    if (cdir.empty())
        return rt_fail(RTPBR_ESTATE, "run-time compilation is off: no cache directory that is owned by this user and closed to others "
                                     "(tried $RTPBR_JIT_CACHE, else $XDG_CACHE_HOME/rtpbr or ~/.cache/rtpbr, then /tmp/rtpbr-cache-<uid>)%s", "");
    static std::atomic<unsigned> g_build_seq{0};
    char tmp[64];
    snprintf(tmp, sizeof tmp, ".tmp.%d.%u", (int)getpid(), g_build_seq.fetch_add(1u));
    // everything this build writes carries the pid (ranks of one job build the same key at the same time) and a
    // process-wide sequence number (two threads of one process building it for two devices: the registry's BUILDING
    // marker is per device)
    const std::string tpath = path + tmp, log = tpath + ".log";
    std::string table_def;
    const std::string tfile = tpath + ".table.hpp";
    if (key.baked) {
        FILE* f = fopen(tfile.c_str(), "w");
        if (!f) return rt_fail(RTPBR_ESTATE, "cannot write %s", tfile.c_str());
        fprintf(f, "// generated by rt_jit.hip: the scene's march table (ObjM blocks) as bit patterns\n");
        fprintf(f, "static constexpr uint32_t RT_JIT_TABLE_BITS[%d][16] = {\n", key.n_obj);
        for (int i = 0; i < key.n_obj; i++) {
            fprintf(f, "    {");
            for (int k = 0; k < 16; k++) {
                uint32_t w = key.table[i * 16 + k];
                if (key.fast && k >= 3 && k < 12) {
                    // tolerance flavour: a matrix entry within 2^-20 of 0 / +-1 IS 0 / +-1 (to_local's rot2 then drops the arithmetic)
                    float v;
                    memcpy(&v, &w, 4);
                    const float a = v < 0 ? -v : v;
                    if (a < 9.5367431640625e-07f) v = 0.0f;
                    else if (a > 1.0f - 9.5367431640625e-07f && a < 1.0f + 9.5367431640625e-07f) v = v < 0 ? -1.0f : 1.0f;
                    memcpy(&w, &v, 4);
                }
                fprintf(f, "0x%08xu%s", w, k < 15 ? ", " : "");
            }
            fprintf(f, "},\n");
        }
        fprintf(f, "};\n");
        // the render configuration (every knob of rtpbr_config except seed and frame, which stay launch arguments) and the
        // constants derived from it: branches on the variant knobs fold away, thresholds become literals
        fprintf(f, "struct RtJitCfgWords { uint32_t w[%d]; };\n", (int)(sizeof(rtpbr_config) / 4));
        fprintf(f, "static constexpr RtJitCfgWords RT_JIT_CFG_WORDS = {{");
        for (size_t k = 0; k < sizeof(rtpbr_config) / 4; k++) fprintf(f, "0x%08xu%s", key.cfg_words[k], k + 1 < sizeof(rtpbr_config) / 4 ? ", " : "");
        fprintf(f, "}};\n");
        if (key.baked == 2) {
            fprintf(f, "struct RtJitCamWords { uint32_t w[21]; };\nstatic constexpr RtJitCamWords RT_JIT_CAM_WORDS = {{");
            for (int k = 0; k < 21; k++) fprintf(f, "0x%08xu%s", key.cam_words[k], k < 20 ? ", " : "");
            fprintf(f, "}};\n#define RT_JIT_BAKE_CAM(Q) (Q).cam = __builtin_bit_cast(rt::CamFrame, RT_JIT_CAM_WORDS);\n");
        } else {
            fprintf(f, "#define RT_JIT_BAKE_CAM(Q)\n");
        }
        fprintf(f, "#define RT_JIT_BAKE_PARAMS(Q) do { RT_JIT_BAKE_CAM(Q) rtpbr_config b_ = __builtin_bit_cast(rtpbr_config, RT_JIT_CFG_WORDS); "
                   "b_.seed = (Q).cfg.seed; b_.frame = (Q).cfg.frame; (Q).cfg = b_; (Q).n_obj = %d; "
                   "(Q).box_lazy = %d; (Q).box_four_rho = __builtin_bit_cast(float, 0x%08xu); (Q).box_rho2m = __builtin_bit_cast(float, 0x%08xu); "
                   "(Q).box_4rho2m = __builtin_bit_cast(float, 0x%08xu); "
                   "(Q).tile_w = %d; (Q).tile_h = %d; (Q).ntx = %d; (Q).nty = %d; (Q).world = %d; (Q).shade_lanes = %d; (Q).swap_lanes = %d; (Q).mlp_mfma = %d; "
                   "} while (0)\n",
                key.n_obj, key.extra[0], key.extra[1], key.extra[2], key.extra[3], key.ints[0], key.ints[1], key.ints[2], key.ints[3],
                key.ints[4], key.ints[5], key.ints[6], key.ints[7]);
        fclose(f);
        table_def = "-DRT_JIT_TABLE_FILE=\"" + tfile + "\"";
    }
    char d[7][64];
    snprintf(d[6], 64, "-DRT_JIT_FORM=%d", key.form);
    snprintf(d[0], 64, "-DRT_JIT_KIND=%d", key.kind);
    snprintf(d[1], 64, "-DRT_JIT_NOBJ=%d", key.n_obj);
    snprintf(d[2], 64, "-DRT_JIT_TYPES=0x%llxull", (unsigned long long)key.types);
    snprintf(d[3], 64, "-DRT_JIT_SIG=0x%xu", key.sig);
    snprintf(d[4], 64, "-DRT_JIT_CULL=%d", key.cull);
    snprintf(d[5], 64, "-DRT_JIT_WAVES=%d", key.waves);
    // the flags of raytracingpbr_amd/build.py: same code generation as the ahead-of-time library
    std::vector<std::string> argv = {hipcc_path(), "--offload-arch=gfx950", "--genco", "-O3", "-std=c++17", "-ffp-contract=off",
                                     "-fno-slp-vectorize", "-mllvm", "-amdgpu-use-amdgpu-trackers=1", "-Wno-unused-value",
                                     d[0], d[1], d[2], d[3], d[4], d[5], d[6], dir + "/rt_jit_tu.hip", "-o", tpath};
    if (key.fast) {
        // the tolerance flavour (rt_math.hpp RT_FAST_MATH): contraction allowed, v_rcp-based divide, hardware transcendentals
        argv[5] = "-ffp-contract=fast";
        // (-munsafe-fp-atomics: the f32 adds into image_buffer as hardware atomics, not compare-and-swap loops)
        argv.insert(argv.begin() + 10, {"-DRT_FAST_MATH=1", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-munsafe-fp-atomics"});
        // all-box scenes (kind 1 = KIND_BOXES): the root of a box distance is of the distance's own magnitude (|sqrt(s2) / 2 - rho|, rho
        // small), so the bare v_sqrt_f32 (<= 1 ulp) does without the correction step the sphere scenes need (rt_math.hpp sqrt_fast_):
        // Cornell headline 4974 -> 5366 Msamples/s, whole-frame L2 against the exact kernels 6.69e-4 -> 6.71e-4
        if (key.kind == 1) argv.insert(argv.begin() + 10, "-DRT_FAST_HW_SQRT=1");
    }
    if (key.dense) argv.insert(argv.begin() + 10, "-DRT_STAGE_DENSE=1");
    if (key.baked) argv.insert(argv.begin() + 10, table_def);
    argv.insert(argv.begin() + 10, extra.begin(), extra.end());
    const int rc = run(argv, log);
    struct stat ost;
    const bool have_out = stat(tpath.c_str(), &ost) == 0 && ost.st_size > 0;
    // rc -2: the exit status was lost (SIGCHLD ignored by the host): a complete output file decides
    if (!((rc == 0 || rc == -2) && have_out)) {
        // hipcc ran and said no: the same call fails the same way next time — unless its input vanished under it
        const bool table_there = !key.baked || access(tfile.c_str(), R_OK) == 0;
        unlink(tpath.c_str());
        if (key.baked) unlink(tfile.c_str());
        if (deterministic) *deterministic = rc > 0 && table_there;
        return rt_fail(RTPBR_ESTATE, "run-time compilation failed (hipcc log: %s)", log.c_str());
    }
    (void)chmod(tpath.c_str(), 0600);
    if (rename(tpath.c_str(), path.c_str()) != 0) {
        unlink(tpath.c_str());
        return rt_fail(RTPBR_ESTATE, "cannot move the compiled code object into the cache (%s)", path.c_str());
    }
    unlink(log.c_str());
    if (key.baked) unlink(tfile.c_str());
    prune_cache(cdir);
    *out = path;
    return RTPBR_OK;
}

static void unload_module(RtJitModule* m) {
    // kernels of a context that has moved on to another instance may still be in flight; the caller's current device
    // is restored (rtpbr_sample goes on to launch on ITS device after rt_jit_acquire)
    int cur = -1;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    if (hipSetDevice(m->device) == hipSuccess) (void)hipDeviceSynchronize();
    if (m->module) (void)hipModuleUnload(m->module);
    if (have_cur) (void)hipSetDevice(cur);
    delete m;
}

// keep at most RTPBR_JIT_MAX_MODULES loaded instances (g_mu held): take the least recently used unpinned ones out of the
// registry; the caller unloads them AFTER releasing the lock (a device-wide synchronisation must not stall every other
// context's acquire / release)
static void evict_modules(std::vector<RtJitModule*>& victims) {
    long cap = 64;
    if (const char* e = getenv("RTPBR_JIT_MAX_MODULES")) cap = atol(e);
    if (cap < 1) cap = 1;
    for (;;) {
        long n = 0;
        auto victim = g_modules.end();
        for (auto it = g_modules.begin(); it != g_modules.end(); ++it) {
            if (it->second.state != E_READY) continue;
            n++;
            if (it->second.mod->pins == 0 && (victim == g_modules.end() || it->second.mod->last_use < victim->second.mod->last_use)) victim = it;
        }
        if (n <= cap || victim == g_modules.end()) return;
        victims.push_back(victim->second.mod);
        g_modules.erase(victim);
    }
}

// The instance of `key` on c's device, PINNED (rt_jit_release when the context stops using it).
int rt_jit_acquire(rtpbr_ctx* c, const RtJitKey& key, RtJitModule** out) {
    char id[224];
    snprintf(id, sizeof id, "d%d_k%d_n%d_t%llx_s%x_c%d_w%d_f%d_p%d_e%d_b%llx", c->device, key.kind, key.n_obj, (unsigned long long)key.types, key.sig,
             key.cull, key.waves, key.form, key.fast, key.dense, (unsigned long long)baked_hash(key));
    if (const char* e = getenv("RTPBR_JIT_EXTRA_FLAGS")) snprintf(id + strlen(id), sizeof id - strlen(id), "_x%zx", std::hash<std::string>()(e));
    {
        std::unique_lock<std::mutex> lock(g_mu);
        for (;;) {
            auto it = g_modules.find(id);
            if (it == g_modules.end()) break;                       // ours to build
            if (it->second.state == E_READY) {
                it->second.mod->pins++;
                it->second.mod->last_use = ++g_tick;
                *out = it->second.mod;
                return RTPBR_OK;
            }
            if (it->second.state == E_FAILED)
                return rt_fail(RTPBR_ESTATE, "run-time compilation failed earlier for this scene: %s", it->second.error.c_str());
            g_cv.wait(lock);                                         // another thread is building this key
        }
        g_modules[id] = Entry{};                                     // E_BUILDING
    }
    // ---- compile and load WITHOUT the registry lock: other keys, contexts and threads proceed
    bool settled = false;
    auto settle = [&](RtJitModule* m, bool remember_failure, const char* why) {
        std::vector<RtJitModule*> victims;
        {
            std::lock_guard<std::mutex> lock(g_mu);
            if (m) {
                Entry& e = g_modules[id];
                e.state = E_READY;
                e.mod = m;
                m->pins = 1;
                m->last_use = ++g_tick;
                evict_modules(victims);
            } else if (remember_failure) {
                Entry& e = g_modules[id];
                e.state = E_FAILED;
                e.error = why ? why : "";
            } else {
                g_modules.erase(id);                                 // transient: the next call tries again
            }
            settled = true;
            g_cv.notify_all();
        }
        for (RtJitModule* v : victims) unload_module(v);
    };
    // whatever leaves this function without settling (std::bad_alloc from the string / vector work below) must not leave
    // the BUILDING marker behind: every later acquire of the key would wait for it forever
    struct Guard {
        decltype(settle)& fn;
        bool& done;
        ~Guard() {
            if (!done) fn(nullptr, false, nullptr);
        }
    } guard{settle, settled};
    std::string path;
    bool deterministic = false;
    if (int r = rt_jit_build(key, &path, &deterministic)) {
        settle(nullptr, deterministic, rtpbr_last_error());
        return r;
    }
    std::vector<char> image;
    const bool from_catalog = path.compare(0, rt_jit_catalog_dir().size() + 1, rt_jit_catalog_dir() + "/") == 0;
    if (!read_owned_file(path, image, from_catalog)) {
        settle(nullptr, false, nullptr);
        return rt_fail(RTPBR_ESTATE, "cannot read %s (or it is not a private file of this user)", path.c_str());
    }
    hipError_t e = hipSetDevice(c->device);
    RtJitModule* m = new RtJitModule();
    m->device = c->device;
    if (e == hipSuccess) e = hipModuleLoadData(&m->module, image.data());
    if (key.form != 1) {
        if (e == hipSuccess) e = hipModuleGetFunction(&m->trace, m->module, "rt_jit_trace");
        if (e == hipSuccess) e = hipModuleGetFunction(&m->primary, m->module, "rt_jit_primary");
        if (e == hipSuccess) e = hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&m->trace_blocks_per_cu, m->trace, 256, 0);
    }
    if (key.form != 0) {
        if (e == hipSuccess) e = hipModuleGetFunction(&m->persistent_pool, m->module, "rt_jit_persistent_pool");
        if (e == hipSuccess) e = hipModuleGetFunction(&m->persistent_steps, m->module, "rt_jit_persistent_steps");
        if (e == hipSuccess) e = hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&m->persistent_blocks_per_cu, m->persistent_pool, 256, 0);
        if (e == hipSuccess) e = hipModuleGetFunction(&m->chain_steps, m->module, "rt_jit_chain_steps");
        if (e == hipSuccess) e = hipModuleGetFunction(&m->src_gen, m->module, "rt_jit_src_gen");
        if (e == hipSuccess) e = hipModuleGetFunction(&m->src_march, m->module, "rt_jit_src_march");
        if (e == hipSuccess) e = hipModuleGetFunction(&m->src_shade, m->module, "rt_jit_src_shade");
        if (e == hipSuccess) e = hipModuleGetFunction(&m->src_shade_gen, m->module, "rt_jit_src_shade_gen");
        if (e == hipSuccess) e = hipModuleGetFunction(&m->src_shade_gen_count, m->module, "rt_jit_src_shade_gen_count");
        if (e == hipSuccess) e = hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&m->march_blocks_per_cu, m->src_march, 256, 0);
    }
    if (e != hipSuccess) {
        if (m->module) (void)hipModuleUnload(m->module);
        delete m;
        if (!from_catalog) unlink(path.c_str());     // a stale / truncated code object: the next call recompiles (the catalog is read-only)
        settle(nullptr, false, nullptr);
        return rt_fail_hip("loading the run-time compiled code object", e);
    }
    m->path = path;
    settle(m, false, nullptr);
    *out = m;
    return RTPBR_OK;
}

void rt_jit_release(RtJitModule* m) {
    if (!m) return;
    std::lock_guard<std::mutex> lock(g_mu);
    if (m->pins > 0) m->pins--;
}

int rt_jit_launch(hipFunction_t f, const Params& P, unsigned grid, hipStream_t st) {
    Params copy = P;                                  // the launch reads the argument during the call only
    void* args[] = {&copy};
    RT_HIP_TRY(hipModuleLaunchKernel(f, grid, 1, 1, 256, 1, 1, 0, st, args, nullptr));
    return RTPBR_OK;
}

int rt_jit_launch_steps(hipFunction_t f, const Params& P, int steps, unsigned grid, hipStream_t st) {
    Params copy = P;
    void* args[] = {&copy, &steps};
    RT_HIP_TRY(hipModuleLaunchKernel(f, grid, 1, 1, 256, 1, 1, 0, st, args, nullptr));
    return RTPBR_OK;
}

// test hook (no device needed): compile the code object of a key and return its path
extern "C" int rtpbr_test_jit_build(int kind, int n_obj, unsigned long long types, unsigned sig, int cull, int waves, char* path_out, size_t cap) {
    RtJitKey k{};
    k.kind = kind, k.n_obj = n_obj, k.types = types, k.sig = sig, k.cull = cull, k.waves = waves;
    k.form = 2;                                       // all four kernels: "does it build" covers both forms
    std::string p;
    if (int r = rt_jit_build(k, &p, nullptr)) return r;
    if (path_out && cap) snprintf(path_out, cap, "%s", p.c_str());
    return RTPBR_OK;
}
