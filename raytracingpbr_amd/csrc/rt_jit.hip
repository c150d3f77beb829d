// rt_jit.hip — per-scene instances compiled at run time (the generalisation of RT_BOX_SIGNATURES).
//
// The reference gets its specialisation from Taichi's JIT: `ti.static(range(len(OBJECTS)))` unrolls the object loop and
// picks each object's shape function at compile time (src/scene.py:44-56).  Here the ahead-of-time library carries
// general instances plus one listed rotation signature; for any OTHER scene of <= 8 analytic shapes the complete-path
// kernels are compiled on demand by `hipcc --genco` from the same sources (rt_jit_tu.hip: object count, shape types,
// rotation classes and the culling decision as compile-time constants; ~2 s), cached as a code object under
// $RTPBR_JIT_CACHE / $XDG_CACHE_HOME/rtpbr / ~/.cache/rtpbr keyed by those constants and a hash of the sources, loaded
// with hipModuleLoadData and launched with hipModuleLaunchKernel.  Results are bit-identical to the ahead-of-time
// instances (same arithmetic).  If hipcc or the sources are not available the library silently keeps using the
// ahead-of-time instance (option "jit" = 1 turns that into an error, 0 disables run-time compilation).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "rt_ctx.hpp"

using namespace rt;

namespace {
std::mutex g_mu;
std::map<std::string, RtJitModule*> g_modules;   // process-wide: contexts on the same device share code objects

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
    const char* files[] = {"rt_jit_tu.hip", "rt_trace.hpp", "rt_device.hpp", "rt_types.hpp", "rt_math.hpp", "../../include/rtpbr.h"};
    uint64_t x = 1469598103934665603ull;
    std::vector<char> buf;
    for (const char* f : files) {
        if (!read_file(dir + "/" + f, buf)) return false;
        for (char c : buf) x = (x ^ (unsigned char)c) * 1099511628211ull;
    }
    *h = x;
    return true;
}

std::string cache_dir() {
    if (const char* e = getenv("RTPBR_JIT_CACHE")) return e;
    if (const char* e = getenv("XDG_CACHE_HOME")) return std::string(e) + "/rtpbr";
    if (const char* e = getenv("HOME")) return std::string(e) + "/.cache/rtpbr";
    return "/tmp/rtpbr-cache";
}

void mkdirs(const std::string& d) {
    for (size_t i = 1; i <= d.size(); i++)
        if (i == d.size() || d[i] == '/') mkdir(d.substr(0, i).c_str(), 0755);
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
    if (waitpid(pid, &st, 0) < 0) return -1;
    return WIFEXITED(st) ? WEXITSTATUS(st) : -1;
}
}  // namespace

// Compile (or fetch from the cache) the code object of `key`; path of the .hsaco in *out.  Needs no device.
int rt_jit_build(const RtJitKey& key, std::string* out) {
    const std::string dir = lib_dir();
    uint64_t sh = 0;
    if (!source_hash(dir, &sh)) return rt_fail(RTPBR_ESTATE, "run-time compilation: kernel sources not found next to the library (%s)", dir.c_str());
    uint64_t th = 0;
    if (key.baked) {
        th = 1469598103934665603ull;
        for (int i = 0; i < key.n_obj * 16; i++) th = (th ^ key.table[i]) * 1099511628211ull;
        for (size_t k = 0; k < sizeof(rtpbr_config) / 4; k++) th = (th ^ key.cfg_words[k]) * 1099511628211ull;
        for (int k = 0; k < 4; k++) th = (th ^ key.extra[k]) * 1099511628211ull;
        for (int k = 0; k < 7; k++) th = (th ^ (unsigned)key.ints[k]) * 1099511628211ull;
    }
    char name[256];
    snprintf(name, sizeof name, "k%d_n%d_t%llx_s%x_c%d_w%d_b%016llx_%016llx", key.kind, key.n_obj, (unsigned long long)key.types, key.sig,
             key.cull, key.waves, (unsigned long long)th, (unsigned long long)sh);
    std::string cdir = cache_dir();
    std::string path = cdir + "/" + name + ".hsaco";
    if (access(path.c_str(), R_OK) == 0) {
        *out = path;
        return RTPBR_OK;
    }
    mkdirs(cdir);
    if (access(cdir.c_str(), W_OK) != 0 && !getenv("RTPBR_JIT_CACHE")) {
        // read-only home directory: fall back to a per-user directory under /tmp rather than lose the specialisation
        char alt[64];
        snprintf(alt, sizeof alt, "/tmp/rtpbr-cache-%d", (int)getuid());
        cdir = alt;
        path = cdir + "/" + name + ".hsaco";
        if (access(path.c_str(), R_OK) == 0) {
            *out = path;
            return RTPBR_OK;
        }
        mkdirs(cdir);
    }
    char tmp[64];
    snprintf(tmp, sizeof tmp, ".tmp.%d", (int)getpid());
    // everything this process writes carries its pid: ranks of one job build the same key at the same time
    const std::string tpath = path + tmp, log = tpath + ".log";
    std::string table_def;
    if (key.baked) {
        const std::string tfile = tpath + ".table.hpp";
        FILE* f = fopen(tfile.c_str(), "w");
        if (!f) return rt_fail(RTPBR_ESTATE, "cannot write %s", tfile.c_str());
        fprintf(f, "// generated by rt_jit.hip: the scene's march table (ObjM blocks) as bit patterns\n");
        fprintf(f, "static constexpr uint32_t RT_JIT_TABLE_BITS[%d][16] = {\n", key.n_obj);
        for (int i = 0; i < key.n_obj; i++) {
            fprintf(f, "    {");
            for (int k = 0; k < 16; k++) fprintf(f, "0x%08xu%s", key.table[i * 16 + k], k < 15 ? ", " : "");
            fprintf(f, "},\n");
        }
        fprintf(f, "};\n");
        // the render configuration (every knob of rtpbr_config except seed and frame, which stay launch arguments) and the
        // constants derived from it: branches on the variant knobs fold away, thresholds become literals
        fprintf(f, "struct RtJitCfgWords { uint32_t w[%d]; };\n", (int)(sizeof(rtpbr_config) / 4));
        fprintf(f, "static constexpr RtJitCfgWords RT_JIT_CFG_WORDS = {{");
        for (size_t k = 0; k < sizeof(rtpbr_config) / 4; k++) fprintf(f, "0x%08xu%s", key.cfg_words[k], k + 1 < sizeof(rtpbr_config) / 4 ? ", " : "");
        fprintf(f, "}};\n");
        fprintf(f, "#define RT_JIT_BAKE_PARAMS(Q) do { rtpbr_config b_ = __builtin_bit_cast(rtpbr_config, RT_JIT_CFG_WORDS); "
                   "b_.seed = (Q).cfg.seed; b_.frame = (Q).cfg.frame; (Q).cfg = b_; (Q).n_obj = %d; "
                   "(Q).box_lazy = %d; (Q).box_four_rho = __builtin_bit_cast(float, 0x%08xu); (Q).box_rho2m = __builtin_bit_cast(float, 0x%08xu); "
                   "(Q).box_4rho2m = __builtin_bit_cast(float, 0x%08xu); "
                   "(Q).tile_w = %d; (Q).tile_h = %d; (Q).ntx = %d; (Q).nty = %d; (Q).world = %d; (Q).shade_lanes = %d; (Q).swap_lanes = %d; "
                   "} while (0)\n",
                key.n_obj, key.extra[0], key.extra[1], key.extra[2], key.extra[3], key.ints[0], key.ints[1], key.ints[2], key.ints[3],
                key.ints[4], key.ints[5], key.ints[6]);
        fclose(f);
        table_def = "-DRT_JIT_TABLE_FILE=\"" + tfile + "\"";
    }
    char d[6][64];
    snprintf(d[0], 64, "-DRT_JIT_KIND=%d", key.kind);
    snprintf(d[1], 64, "-DRT_JIT_NOBJ=%d", key.n_obj);
    snprintf(d[2], 64, "-DRT_JIT_TYPES=0x%llxull", (unsigned long long)key.types);
    snprintf(d[3], 64, "-DRT_JIT_SIG=0x%xu", key.sig);
    snprintf(d[4], 64, "-DRT_JIT_CULL=%d", key.cull);
    snprintf(d[5], 64, "-DRT_JIT_WAVES=%d", key.waves);
    // the flags of raytracingpbr_amd/build.py: same code generation as the ahead-of-time library
    std::vector<std::string> argv = {hipcc_path(), "--offload-arch=gfx950", "--genco", "-O3", "-std=c++17", "-ffp-contract=off",
                                     "-fno-slp-vectorize", "-mllvm", "-amdgpu-use-amdgpu-trackers=1", "-Wno-unused-value",
                                     d[0], d[1], d[2], d[3], d[4], d[5], dir + "/rt_jit_tu.hip", "-o", tpath};
    if (key.baked) argv.insert(argv.begin() + 10, table_def);
    const int rc = run(argv, log);
    if (rc != 0 || access(tpath.c_str(), R_OK) != 0) {
        unlink(tpath.c_str());
        return rt_fail(RTPBR_ESTATE, "run-time compilation failed (hipcc log: %s)", log.c_str());
    }
    if (rename(tpath.c_str(), path.c_str()) != 0) {
        unlink(tpath.c_str());
        return rt_fail(RTPBR_ESTATE, "cannot move the compiled code object into the cache (%s)", path.c_str());
    }
    unlink(log.c_str());
    if (key.baked) unlink((tpath + ".table.hpp").c_str());
    *out = path;
    return RTPBR_OK;
}

int rt_jit_acquire(rtpbr_ctx* c, const RtJitKey& key, RtJitModule** out) {
    uint64_t th = 0;
    if (key.baked) {
        th = 1469598103934665603ull;
        for (int i = 0; i < key.n_obj * 16; i++) th = (th ^ key.table[i]) * 1099511628211ull;
        for (size_t k = 0; k < sizeof(rtpbr_config) / 4; k++) th = (th ^ key.cfg_words[k]) * 1099511628211ull;
        for (int k = 0; k < 4; k++) th = (th ^ key.extra[k]) * 1099511628211ull;
        for (int k = 0; k < 7; k++) th = (th ^ (unsigned)key.ints[k]) * 1099511628211ull;
    }
    char id[224];
    snprintf(id, sizeof id, "d%d_k%d_n%d_t%llx_s%x_c%d_w%d_b%llx", c->device, key.kind, key.n_obj, (unsigned long long)key.types, key.sig,
             key.cull, key.waves, (unsigned long long)th);
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_modules.find(id);
    if (it != g_modules.end()) {
        *out = it->second;
        return it->second ? RTPBR_OK : rt_fail(RTPBR_ESTATE, "run-time compilation failed earlier for this scene");
    }
    g_modules[id] = nullptr;                         // a failure is remembered: no recompilation storm
    std::string path;
    if (int r = rt_jit_build(key, &path)) return r;
    std::vector<char> image;
    if (!read_file(path, image) || image.empty()) return rt_fail(RTPBR_ESTATE, "cannot read %s", path.c_str());
    RT_HIP_TRY(hipSetDevice(c->device));
    RtJitModule* m = new RtJitModule();
    hipError_t e = hipModuleLoadData(&m->module, image.data());
    if (e == hipSuccess) e = hipModuleGetFunction(&m->trace, m->module, "rt_jit_trace");
    if (e == hipSuccess) e = hipModuleGetFunction(&m->primary, m->module, "rt_jit_primary");
    if (e == hipSuccess) e = hipModuleGetFunction(&m->persistent_pool, m->module, "rt_jit_persistent_pool");
    if (e == hipSuccess) e = hipModuleGetFunction(&m->persistent_steps, m->module, "rt_jit_persistent_steps");
    if (e == hipSuccess) e = hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&m->trace_blocks_per_cu, m->trace, 256, 0);
    if (e == hipSuccess) e = hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&m->persistent_blocks_per_cu, m->persistent_pool, 256, 0);
    if (e != hipSuccess) {
        delete m;
        unlink(path.c_str());                        // a stale / foreign code object: recompile next time
        return rt_fail_hip("loading the run-time compiled code object", e);
    }
    m->path = path;
    g_modules[id] = m;
    *out = m;
    return RTPBR_OK;
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
    std::string p;
    if (int r = rt_jit_build(k, &p)) return r;
    if (path_out && cap) snprintf(path_out, cap, "%s", p.c_str());
    return RTPBR_OK;
}
